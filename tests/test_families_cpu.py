"""The time-coupled chain is ONE fixed problem that shards (BASELINE configs[3]: 2048 blocks, Schur dimension 8000): its Schur dimension
does not depend on how many ranks hold it, a rank generates its own block range only, and what it generates is what any other
partition generates for those blocks - the way the reference maps a fixed tree onto ranks (Readers/Distributed/DistributedTree.C:62-89)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import pips_ipmpp_amd as pa
import families

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return a.nrows == b.nrows and a.ncols == b.ncols and np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.colidx, b.colidx) and np.array_equal(a.val, b.val)


def test_configs3_chain_has_schur_dimension_8000_on_any_number_of_ranks():
    ch = families.config3_chain(n_i=50000)
    assert ch.G == 2048 and ch.S == 8000 and ch.n0 == 95 and ch.myl == 7905
    per_pair = np.diff(ch.row0)
    assert per_pair.sum() == 7905 and per_pair.min() == 3 and per_pair.max() == 4
    for world in (1, 2, 4, 8):
        ranges = [families.share_range(ch.G, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 2048 and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        assert {hi - lo for lo, hi in ranges} == {2048 // world}
    # the same rule as the library's map (pips_map_children_to_ranks)
    owner = pa.map_children_to_ranks(2048, 8) if hasattr(pa, "map_children_to_ranks") else None
    if owner is not None:
        for r in range(8):
            lo, hi = families.share_range(2048, r, 8)
            assert set(owner[lo:hi]) == {r}
    # rounds 3-4 measured the 256-block chain: 31 rows on every pair
    old = families.config3_chain(n_i=50000, G=256)
    assert old.S == 8000 and set(np.diff(old.row0)) == {31}


def test_a_rank_generates_its_own_blocks_only_and_they_do_not_depend_on_the_partition(monkeypatch):
    ch = families.config3_chain(n_i=400, G=16, S=95 + 60)
    calls = []
    orig = families.TimeCoupledChain.block

    def counted(self, i):
        calls.append(i)
        return orig(self, i)
    monkeypatch.setattr(families.TimeCoupledChain, "block", counted)
    whole = ch.blocks(0, 16)
    calls.clear()
    lo, hi = families.share_range(16, 5, 8)
    mine = ch.blocks(lo, hi)
    assert calls == [10, 11]
    for (W, T, F), (W2, T2, F2) in zip(mine, whole[lo:hi]):
        assert _same(W, W2) and _same(T, T2) and _same(F, F2)
    # every linking row has entries in exactly the two blocks of its pair, 3 in each
    count = np.zeros(ch.myl, int)
    for b, (W, T, F) in enumerate(whole):
        rows = np.nonzero(np.diff(F.rowptr))[0]
        want = np.concatenate([np.arange(*ch.pair_rows(p)) for p in (b - 1, b)])
        assert np.array_equal(rows, want) and set(np.diff(F.rowptr)[rows]) <= {1, 2, 3}
        count[rows] += 1
        assert np.array_equal(ch.border_columns(b, T)[-len(want):], ch.n0 + want)
    assert set(count) == {2}
    assert ch.F0().nrows == ch.myl and ch.F0().ncols == ch.n0


def test_prefix_is_the_sub_problem_of_the_first_blocks():
    ch = families.config3_chain(n_i=400)
    sub = ch.prefix(256)
    assert sub.n_blocks == 256 and sub.myl == int(ch.row0[256]) == 988 and sub.S == 1083 and ch.S == 8000
    for b in (0, 100, 255):
        W, T, F = ch.block(b)
        W2, T2, F2 = sub.block(b)
        assert _same(W, W2) and _same(T, T2) and F2.nrows == sub.myl
        assert np.array_equal(F2.val, F.val) and np.array_equal(F2.colidx, F.colidx)       # block 255 keeps its one-sided last pair
    with pytest.raises(IndexError):
        sub.block(256)
    F0, F0s = ch.F0().to_scipy(), sub.F0().to_scipy()
    assert (F0[:sub.myl] != F0s).nnz == 0
    assert ch.prefix(2048) is ch


def test_bench_labels_and_one_gpu_reference_come_from_the_problem_and_the_profiles():
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
    finally:
        sys.argv = sys_argv
    whole = families.config3_chain(50000)
    full = b.workload_label("time-coupled", 8, 256, 50000, 25000, 0, 8000, whole, 7905)
    assert "Schur dim 8000" in full and full.endswith("[BASELINE configs[3]]") and "all of it on 8 GPU(s)" in full
    part = whole.prefix(2 * 256)
    two = b.workload_label("time-coupled", 2, 256, 50000, 25000, 0, part.S, whole, part.myl)
    assert "[BASELINE configs[3] shape on 2 of its 8 GPUs]" in two and "Schur dim 8000" in two and f"(Schur dim {part.S})" in two
    ref = b.same_shape_on_one_gpu(b.shape_key("random", 64, 10000, 4000, None))
    assert ref is not None and ref["source"].startswith("profiles/") and ref["units_per_s"] > 0
    line = [l for l in open(os.path.join(ROOT, ref["source"])) if '"Schur dim 4000' in l or "Schur dim 4000," in l]
    assert any(abs(__import__("json").loads(l)["value"] - ref["units_per_s"]) < 1e-12 for l in line)
    assert b.same_shape_on_one_gpu(b.shape_key("random", 3, 777, 10, None)) is None
