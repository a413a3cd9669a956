"""The task lists of the single-launch dense root (csrc/rootplan.cpp -> csrc/rootkernel.hip.h) on the CPU: every list the kernel draws
from must be a subsequence of ONE topological order of the tile DAG of the left-looking LDL^T (that is what rules out a cycle of waiting
workgroups), and together the tasks must apply every tile column to every tile exactly once, in ascending order."""
import ctypes as C

import numpy as np
import pytest

import pips_ipmpp_amd as pa

lib = pa.capi.lib
lib.pips_root_plan_build.restype = C.c_int


def build(ntc, workers=-1, qmin=-1, urgent=-1, chain_slots=-1):
    n, nc, ms = C.c_longlong(), C.c_longlong(), C.c_double()
    rc = lib.pips_root_plan_build(ntc, workers, qmin, urgent, chain_slots, None, C.c_longlong(0), C.byref(n), C.byref(nc), C.byref(ms))
    assert rc == 0
    out = np.zeros(4 * (n.value + nc.value), dtype=np.int32)
    rc = lib.pips_root_plan_build(ntc, workers, qmin, urgent, chain_slots, out.ctypes.data_as(C.POINTER(C.c_int)), C.c_longlong(out.size),
                                  C.byref(n), C.byref(nc), C.byref(ms))
    assert rc == 0
    t = out.reshape(-1, 4)
    return t[:n.value], t[n.value:], ms.value


def replay(ntc, lists):
    """Run the lists like the launch does - each taken in order, a task starts only when what it waits for has finished - with as many
    workers as it takes; returns the number of rounds.  Fails if the heads of all lists wait for each other."""
    prog = np.zeros((ntc, ntc), dtype=int)
    rowdone = np.zeros(ntc, dtype=int)
    dready = np.zeros(ntc, dtype=bool)
    trsm = np.zeros((ntc, ntc), dtype=bool)
    pos = [0] * len(lists)
    done, total, rounds = 0, sum(len(l) for l in lists), 0
    while done < total:
        rounds += 1
        started = []
        for li, l in enumerate(lists):
            while pos[li] < len(l):          # in order: the head must be able to run before the next one is looked at
                kind, i, j, pad = (int(v) for v in l[pos[li]])
                k0, k1 = pad & 0xffff, pad >> 16
                if kind == 2:
                    ok = prog[j, j] == j and not dready[j] and i == j
                elif kind == 1:
                    ok = dready[j] and prog[i, j] == j and not trsm[i, j] and i > j
                else:
                    ok = k0 < k1 <= j and rowdone[i] >= k1 and rowdone[j] >= k1 and prog[i, j] == k0
                if not ok:
                    break
                started.append((kind, i, j, k1))
                pos[li] += 1
                # (the replay finishes a task at once: the lists must work for ANY timing, this is the most eager one)
                if kind == 2:
                    dready[j] = True
                elif kind == 1:
                    trsm[i, j] = True
                    rowdone[i] = j + 1
                else:
                    prog[i, j] = k1
        assert started, f"the heads of all lists wait for each other at {pos} of {[len(l) for l in lists]}"
        done += len(started)
    assert dready.all()
    for i in range(ntc):
        for j in range(i):
            assert trsm[i, j] and prog[i, j] == j
        assert prog[i, i] == i
    return rounds


@pytest.mark.parametrize("ntc", [1, 2, 3, 5, 16, 40])
@pytest.mark.parametrize("variant", ["default", "one_list", "few_workers"])
def test_lists_are_consistent_with_the_tile_dag(ntc, variant):
    kw = {"default": {}, "one_list": dict(chain_slots=0), "few_workers": dict(workers=3, qmin=2)}[variant]
    bulk, chain, makespan = build(ntc, **kw)
    assert len(bulk) + len(chain) >= ntc * (ntc + 1) // 2
    if variant == "one_list":
        assert len(chain) == 0
    if variant == "default" and ntc >= 2:
        assert len(chain) >= 2 * ntc - 1     # DIAG (j), TRSM (j + 1, j), the completing update of the next diagonal tile
    replay(ntc, [bulk, chain])
    assert makespan > 0


def test_far_tiles_are_updated_deeply():
    """What the schedule is for: tiles far right of the chain take their columns in few, deep pieces (the update kernel's efficient regime)."""
    bulk, chain, _ = build(125)
    upd = bulk[bulk[:, 0] == 0]
    depth = (upd[:, 3] >> 16) - (upd[:, 3] & 0xffff)
    assert depth.mean() > 10
    cu = chain[chain[:, 0] == 0]
    assert depth.sum() + ((cu[:, 3] >> 16) - (cu[:, 3] & 0xffff)).sum() == sum(j * (125 - j) for j in range(125))
