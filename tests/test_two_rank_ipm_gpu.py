"""The IPM harness on two ranks (vector reductions with the replicated root part counted once, link rows of A x
and x0 rows of A^T y summed, the Schur complement and b0 reduced inside factorize / solveCompressed): two processes share
device 0 and reduce through an ExternalComm (gloo, staged through host memory), each owns half of the blocks.  Every rank must
return what the one-process run returns: status, iteration count, objective, the per-iterate history, x0 and its blocks' x."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pips_ipmpp_amd as pa
from tests.test_ipm_gpu import build_lp

pytestmark = pytest.mark.gpu

SHAPE = (2027, 4, 120, 60, 8, 6, 0.08)   # seed, N, n_i, my_i, n0, myl, rho


def _local(blocks, c, b, mine, n_i, my_i, n0, myl):
    cl = np.concatenate([c[:n0]] + [c[n0 + k * n_i:n0 + (k + 1) * n_i] for k in mine])
    bl = np.concatenate([b[:myl]] + [b[myl + k * my_i:myl + (k + 1) * my_i] for k in mine])
    return [blocks[k] for k in mine], cl, bl


def _free_mask(c, b, A, n0):
    """A tenth of the variables that are positive at the optimum (HiGHS) are declared free: the optimum stays (inactive bounds)."""
    from scipy.optimize import linprog
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    cand = np.nonzero(ref.x > 0.1)[0]
    free = np.random.default_rng(3).choice(cand, size=max(4, len(cand) // 10), replace=False)
    mask = np.ones(A.shape[1])
    mask[free] = 0.0
    return mask


def _worker(rank, world, port, out, with_free=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed, N, n_i, my_i, n0, myl, rho = SHAPE
    blocks, F0, c, b, A = build_lp(seed, N, n_i, my_i, n0, myl, rho)
    mine = np.nonzero(pa.map_children_to_ranks(N, world) == rank)[0]
    calls = []

    def allreduce(ptr, n):
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()
        calls.append(n)

    comm = pa.ExternalComm(allreduce)
    lb, lc, lbv = _local(blocks, c, b, mine, n_i, my_i, n0, myl)
    ipm = pa.IpmSolver(n0, myl, lb, F0, lc, lbv, comm=comm, rank=rank, n_ranks=world)
    if with_free:
        mask = _free_mask(c, b, A, n0)
        ipm.set_free_variables(np.concatenate([mask[:n0]] + [mask[n0 + k * n_i:n0 + (k + 1) * n_i] for k in mine]))
    res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    x, y = ipm.solution()
    np.savez(os.path.join(out, f"rank{rank}.npz"), res=np.array([res[k] for k in ("status", "iterations", "objective", "dual_objective", "mu", "rnorm", "dnorm")]),
             trace=ipm.trace(), x=x, y=y, mine=mine, ncalls=len(calls), stats=np.array(list(ipm.stats().values())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("with_free", [False, True], ids=["bounded", "free_variables"])
def test_two_rank_ipm_matches_one_rank(tmp_path, with_free):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (13 if not with_free else 17)
    mp.start_processes(_worker, args=(world, port, str(tmp_path), with_free), nprocs=world, join=True, start_method="spawn")
    seed, N, n_i, my_i, n0, myl, rho = SHAPE
    blocks, F0, c, b, A = build_lp(seed, N, n_i, my_i, n0, myl, rho)
    one = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    if with_free:
        one.set_free_variables(_free_mask(c, b, A, n0))
    r1 = one.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    x1, y1 = one.solution()
    t1 = one.trace()
    assert r1["status"] == 0
    got = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    # every rank reports the same scalars (they steer the iteration: any difference would desynchronise the collectives)
    assert np.array_equal(got[0]["res"], got[1]["res"]) and np.array_equal(got[0]["trace"], got[1]["trace"])
    # the replicated root parts agree (to rounding: each rank factorises the reduced Schur complement on its own)
    assert np.allclose(got[0]["x"][:n0], got[1]["x"][:n0], rtol=1e-10, atol=1e-12) and np.allclose(got[0]["y"][:myl], got[1]["y"][:myl], rtol=1e-10, atol=1e-12)
    seen = []
    for g in got:
        status, its, obj = int(g["res"][0]), int(g["res"][1]), g["res"][2]
        assert status == 0 and abs(its - r1["iterations"]) <= 1
        assert abs(obj - r1["objective"]) <= 1e-9 * abs(r1["objective"])
        assert g["res"][6] == r1["dnorm"]
        n_cmp = min(g["trace"].shape[0], t1.shape[0]) - 4
        assert n_cmp >= 8 and np.allclose(g["trace"][:n_cmp, [0, 2, 3]], t1[:n_cmp, [0, 2, 3]], rtol=1e-6, atol=0)
        assert np.abs(g["x"][:n0] - x1[:n0]).max() <= 1e-6 * max(1.0, np.abs(x1[:n0]).max())
        for i, k in enumerate(g["mine"]):
            xs = g["x"][n0 + i * n_i:n0 + (i + 1) * n_i]
            assert np.abs(xs - x1[n0 + k * n_i:n0 + (k + 1) * n_i]).max() <= 1e-6 * max(1.0, np.abs(x1).max())
            seen.append(int(k))
        assert g["ncalls"] > 50
    assert sorted(seen) == list(range(N))


# ---- the general problem class on two ranks: root equality / inequality rows, block inequality rows, linking rows of both kinds
GEN = dict(seed=411, nb=5, n0=7, ni=26, mA=8, mC=5, mBL=3, mDL=3)


def _gen_blocks():
    from tests.general_lp_gen import random_block_lp
    g = GEN
    return random_block_lp(g["seed"], g["nb"], g["n0"], g["ni"], g["mA"], g["mC"], g["mBL"], g["mDL"], free_fraction=0.0)


def _general_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = _gen_blocks()
    N = len(blocks) - 1
    mine = np.nonzero(pa.map_children_to_ranks(N, world) == rank)[0]

    def allreduce(ptr, n):
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()

    comm = pa.ExternalComm(allreduce)
    ipm = pa.GeneralIpmSolver([blocks[0]] + [blocks[1 + k] for k in mine], comm=comm, rank=rank, n_ranks=world)
    # f-2 on two ranks: replicated rows of J x and J^T [y; z] are summed over the ranks (DistributedMatrix.C:224-326)
    rng = np.random.default_rng(9)
    xg = rng.standard_normal(GEN["n0"] + N * GEN["ni"])
    xl = np.concatenate([xg[:GEN["n0"]]] + [xg[GEN["n0"] + k * GEN["ni"]:GEN["n0"] + (k + 1) * GEN["ni"]] for k in mine])
    jx = ipm.mult(xl)
    res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    itr = ipm.iterate()
    np.savez(os.path.join(out, f"grank{rank}.npz"), res=np.array([res[k] for k in ("status", "iterations", "objective", "dual_objective", "mu", "rnorm", "dnorm")]),
             trace=ipm.trace(), mine=mine, jx=jx, pairs=ipm.n_pairs, **{k: v for k, v in itr.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_general_ipm_matches_one_rank(tmp_path):
    import scipy.sparse as sp
    from oracle import ipm_oracle as io
    world = 2
    port = 29500 + (os.getpid() % 2000) + 23
    mp.start_processes(_general_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    blocks = _gen_blocks()
    d = io.assemble(blocks)
    N, n0, ni = len(blocks) - 1, GEN["n0"], GEN["ni"]
    one = pa.GeneralIpmSolver(blocks)
    r1 = one.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    i1, t1 = one.iterate(), one.trace()
    assert r1["status"] == 0
    got = [np.load(os.path.join(str(tmp_path), f"grank{r}.npz")) for r in range(world)]
    assert np.array_equal(got[0]["res"], got[1]["res"]) and np.array_equal(got[0]["trace"], got[1]["trace"])
    rng = np.random.default_rng(9)
    xg = rng.standard_normal(n0 + N * ni)
    J = sp.vstack([d["A"], d["C"]], format="csr")
    want = J @ xg
    my0, myl, mz0, mzl = blocks[0]["mA"], blocks[0]["mBL"], blocks[0]["mC"], blocks[0]["mDL"]
    my_tot = d["A"].shape[0]
    for g in got:
        assert int(g["pairs"]) == one.n_pairs
        status, its, obj = int(g["res"][0]), int(g["res"][1]), g["res"][2]
        assert status == 0 and abs(its - r1["iterations"]) <= 1
        assert abs(obj - r1["objective"]) <= 1e-8 * max(1.0, abs(r1["objective"]))
        # same path; the summation order of the two ranks differs from the single rank's, and the last iterations amplify that
        n_cmp = min(g["trace"].shape[0], t1.shape[0]) - 6
        assert n_cmp >= 5 and np.allclose(g["trace"][:n_cmp, [0, 2, 3]], t1[:n_cmp, [0, 2, 3]], rtol=1e-4, atol=1e-9)
        # replicated parts of the iterate equal the one-rank run
        assert np.abs(g["x"][:n0] - i1["x"][:n0]).max() <= 1e-6 * max(1.0, np.abs(i1["x"]).max())
        assert np.abs(g["z"][:mz0 + mzl] - i1["z"][:mz0 + mzl]).max() <= 1e-3 * max(1.0, np.abs(i1["z"]).max())   # multipliers of degenerate rows are large and loosely determined
        # J x: replicated rows (root and linking rows of both kinds) carry the global sums on every rank
        assert np.abs(g["jx"][:my0 + myl] - want[:my0 + myl]).max() <= 1e-12 * max(1.0, np.abs(want).max())
        assert np.abs(g["jx"][g["y"].shape[0]:g["y"].shape[0] + mz0 + mzl] - want[my_tot:my_tot + mz0 + mzl]).max() <= 1e-12 * max(1.0, np.abs(want).max())
        for i, k in enumerate(g["mine"]):
            xs = g["x"][n0 + i * ni:n0 + (i + 1) * ni]
            assert np.abs(xs - i1["x"][n0 + k * ni:n0 + (k + 1) * ni]).max() <= 1e-6 * max(1.0, np.abs(i1["x"]).max())
