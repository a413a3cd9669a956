"""Host-side symbolic analysis (constrained AMD, head/tail split, scatter maps) — runs without a GPU."""
import os
import numpy as np
import pytest

import pips_ipmpp_amd as pa
import families
from oracle import oracle as orc
from tests.util import Problem


@pytest.mark.parametrize("n_i,rho", [(40, 0.1), (300, 0.03), (2000, 0.005)])
def test_permutation_is_valid_and_respects_dual_constraint(n_i, rho):
    prob = Problem(3, 1, n_i, n_i // 2, 6, 6, rho)
    K = prob.blocks[0]["K"]
    info = pa.symbolic_probe(K, n_i, prob.blocks[0]["Bt"], want_perm=True)
    perm = info["perm"]
    n = K.nrows
    assert sorted(perm.tolist()) == list(range(n))
    # every dual row is eliminated after all its primal neighbours (static-pivot safety)
    pos = np.empty(n, dtype=np.int64)
    pos[perm] = np.arange(n)
    W = prob.blocks[0]["W"].to_scipy().tocsr()
    for r in range(W.shape[0]):
        cols = W.indices[W.indptr[r]:W.indptr[r + 1]]
        assert pos[n_i + r] > pos[cols].max()
    assert info["n_head"] + info["m"] == n
    assert info["nnzL"] >= K.nnz - n


def test_fill_not_worse_than_natural_order():
    prob = Problem(9, 1, 600, 300, 4, 4, 0.02)
    K = prob.blocks[0]["K"]
    info = pa.symbolic_probe(K, 600, want_perm=True, force_n_head=K.nrows)
    o_nat = orc.OracleLdl(prob.K_scipy(0))
    o_amd = orc.OracleLdl(prob.K_scipy(0), perm=info["perm"])
    assert o_amd.nnzL() <= o_nat.nnzL()
    # with everything in the head the stored factor is the exact symbolic count of the oracle plus the explicit zeros
    # admitted by the supernode amalgamation (at most 40% of any panel) ...
    exact = o_amd.nnzL()
    assert exact <= info["nnzL"] - K.nrows <= exact / 0.6
    # ... and exactly that count with fundamental supernodes
    os.environ["PIPS_HIP_RELAX_ZEROS"] = "0"
    try:
        fund = pa.symbolic_probe(K, 600, want_perm=True, force_n_head=K.nrows)
    finally:
        del os.environ["PIPS_HIP_RELAX_ZEROS"]
    assert fund["nnzL"] - K.nrows == orc.OracleLdl(prob.K_scipy(0), perm=fund["perm"]).nnzL()
    assert fund["n_sn"] >= info["n_sn"]


def test_cut_override_and_padding():
    prob = Problem(9, 1, 500, 250, 4, 4, 0.03)
    K = prob.blocks[0]["K"]
    all_tail = pa.symbolic_probe(K, 500, force_n_head=0)
    assert all_tail["n_head"] == 0 and all_tail["m"] == K.nrows and all_tail["ntc"] == (K.nrows + 127) // 128
    all_head = pa.symbolic_probe(K, 500, force_n_head=K.nrows)
    assert all_head["m"] == 0 and all_head["ntc"] == 0


def test_unconstrained_ordering_without_hint():
    prob = Problem(9, 1, 300, 150, 4, 4, 0.03)
    info = pa.symbolic_probe(prob.blocks[0]["K"], -1, want_perm=True)
    assert sorted(info["perm"].tolist()) == list(range(450))


def test_rejects_upper_triangular_input():
    K = pa.Csr(2, 2, [0, 2, 3], [0, 1, 1], [1.0, 2.0, 3.0])
    with pytest.raises(pa.capi.PipsHipError):
        pa.symbolic_probe(K)


def _banded_kkt(n_i, my_i, bw, seed=0):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(my_i):
        center = int(r * n_i / my_i)
        cs = np.union1d(np.clip(center + rng.integers(-bw, bw + 1, 9), 0, n_i - 1), [center])
        rows += [r] * len(cs)
        cols += list(cs)
    W = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(my_i, n_i))
    W.sum_duplicates()
    W.sort_indices()
    Wp = pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)
    K, _ = pa.kkt_leaf_assemble(n_i, Wp)
    return K, W


def test_partial_nested_dissection_on_time_coupled_block():
    """Banded W: dual-row separators cut the chain into independent segments (far fewer tree levels, no more fill), the
    static-pivot constraint (dual row after all its primal neighbours) still holds; random sparsity is left to minimum
    degree (the first separator is rejected)."""
    n_i, my_i = 6000, 3000
    K, W = _banded_kkt(n_i, my_i, 20)
    os.environ["PIPS_HIP_ND_DEPTH"] = "0"
    try:
        chain = pa.symbolic_probe(K, n_i, force_n_head=K.nrows)
    finally:
        del os.environ["PIPS_HIP_ND_DEPTH"]
    nd = pa.symbolic_probe(K, n_i, force_n_head=K.nrows, want_perm=True)
    assert nd["n_levels"] * 3 < chain["n_levels"], (nd["n_levels"], chain["n_levels"])
    assert nd["nnzL"] < 1.5 * chain["nnzL"]
    perm = nd["perm"]
    assert sorted(perm.tolist()) == list(range(K.nrows))
    pos = np.empty(K.nrows, dtype=np.int64)
    pos[perm] = np.arange(K.nrows)
    Wc = W.tocsr()
    for r in range(my_i):
        cols = Wc.indices[Wc.indptr[r]:Wc.indptr[r + 1]]
        assert pos[n_i + r] > pos[cols].max()
    # exact column counts: stored factor = symbolic count of the oracle for this order (fundamental supernodes)
    os.environ["PIPS_HIP_RELAX_ZEROS"] = "0"
    try:
        fund = pa.symbolic_probe(K, n_i, force_n_head=K.nrows, want_perm=True)
    finally:
        del os.environ["PIPS_HIP_RELAX_ZEROS"]
    import scipy.sparse as sp
    Ks = sp.csr_matrix((np.ones(len(K.colidx)), K.colidx, K.rowptr), shape=(K.nrows, K.ncols))
    assert fund["nnzL"] - K.nrows == orc.OracleLdl(Ks, perm=fund["perm"]).nnzL()
    assert int(fund["colcount"].sum()) == fund["nnzL"] - K.nrows
    # random sparsity: identical to the run without dissection
    prob = Problem(9, 1, 2000, 1000, 4, 4, 0.005)
    Kr = prob.blocks[0]["K"]
    a = pa.symbolic_probe(Kr, 2000, want_perm=True)
    os.environ["PIPS_HIP_ND_DEPTH"] = "0"
    try:
        b = pa.symbolic_probe(Kr, 2000, want_perm=True)
    finally:
        del os.environ["PIPS_HIP_ND_DEPTH"]
    assert np.array_equal(a["perm"], b["perm"])


def test_deep_dissection_beats_the_round_one_setting():
    """Time-coupled block of 20 000 variables: dissecting the dual rows down to 128-row segments (the default) gives a far
    shallower elimination tree than 4 levels / 512-row segments (round 1's cap) at about the same fill."""
    n_i, my_i = 20000, 10000
    K, W = _banded_kkt(n_i, my_i, 12, seed=3)
    os.environ["PIPS_HIP_ND_DEPTH"], os.environ["PIPS_HIP_ND_MIN"] = "4", "512"
    try:
        old = pa.symbolic_probe(K, n_i, force_n_head=K.nrows)
    finally:
        del os.environ["PIPS_HIP_ND_DEPTH"], os.environ["PIPS_HIP_ND_MIN"]
    new = pa.symbolic_probe(K, n_i, force_n_head=K.nrows, want_perm=True)
    assert new["n_levels"] * 3 < old["n_levels"], (new["n_levels"], old["n_levels"])
    assert new["nnzL"] <= 1.10 * old["nnzL"], (new["nnzL"], old["nnzL"])
    assert sorted(new["perm"].tolist()) == list(range(K.nrows))


def _two_link_schur_pattern(N, L, n0):
    """Lower-triangular pattern of the Schur complement of a 2-link arrowhead problem: x0 block dense, every linking row coupled to x0,
    to the rows of its own block pair and to those of the pair before."""
    import scipy.sparse as sp
    S = n0 + (N - 1) * L
    rows, cols = [], []
    for i in range(n0):
        rows += [i] * (i + 1); cols += list(range(i + 1))
    for p in range(N - 1):
        r0 = n0 + p * L
        for a in range(L):
            i = r0 + a
            c = list(range(n0)) + list(range(max(n0, r0 - L), i + 1))
            rows += [i] * len(c); cols += c
    A = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(S, S))
    A.sort_indices()
    return pa.Csr(S, S, A.indptr, A.indices, A.data)


def test_sparse_root_order_dissects_the_chain_around_the_hubs():
    """pips_symbolic_probe_hubs = the order pips_hip_kkt_create_sparse takes for a chain-like Schur complement: x0 last, the linking rows
    a tree.  Against the reference's root structure (sLinsysRootAug.C:1629-1739 hands the same pattern to PARDISO, which runs METIS on
    it): the constrained order of the leaves would eliminate x0 first and fill the matrix completely."""
    N, L, n0 = 64, 31, 95
    K = _two_link_schur_pattern(N, L, n0)
    S = K.nrows
    info = pa.capi.symbolic_probe_hubs(K, np.arange(n0), n_primal=n0)
    perm = info["perm"]
    assert sorted(perm.tolist()) == list(range(S))
    assert perm[-n0:].tolist() == list(range(n0))                    # the hubs last, in the given order
    assert info["n_head"] == S - n0 and info["m"] == n0              # every linking row in the head, x0 the dense tail
    assert info["colcount"].max() <= 2 * L + n0 + L                  # fronts: a block's rows + two neighbouring separators + x0
    # a tree, not a chain: depth ~ 2 log2(N) supernode levels (31 columns = two supernodes of <= 16) against N * L / 16 for the band
    assert info["n_levels"] <= 4 * int(np.ceil(np.log2(N))) and info["n_sn"] >= 2 * (N - 1)
    dense = pa.symbolic_probe(K, n0)                                  # x0 first (the leaves' constraint): all of it becomes a dense tail
    assert dense["n_head"] == 0 and dense["flops_factor"] > 20 * info["flops_factor"]
    # the fill under the order matches an explicit symbolic elimination (oracle)
    import scipy.sparse as sp
    A = K.to_scipy()
    A = (A + sp.tril(A, -1).T).tocsr()
    o = orc.OracleLdl(A, perm=perm)
    assert int(info["colcount"].sum()) == o.nnzL()


def test_sparse_root_order_reports_missing_separators():
    """a Schur complement without chain structure (every linking row touches every other): no separators, PIPS_ERR_STATE - the library
    then keeps the band / minimum-degree paths"""
    import scipy.sparse as sp
    S, n0 = 300, 4
    A = sp.csr_matrix(np.tril(np.ones((S, S))))
    K = pa.Csr(S, S, A.indptr, A.indices, A.data)
    with pytest.raises(RuntimeError):
        pa.capi.symbolic_probe_hubs(K, np.arange(n0), n_primal=n0)


def test_border_split_of_the_multifrontal_head(monkeypatch):
    """Symbolic side of the border split: on a time-coupled block with few border columns the fronts keep only the update
    columns of their rows of K - the update matrices shrink by more than half, the panels become compact (the border rows move to the
    border-row arena), the stored factor is the same; a block with more border columns than the LDS triangle takes, or the switch off,
    keeps whole update matrices and full panels."""
    import pips_ipmpp_amd as pa
    c3 = families.CONFIG3_SHARE
    blocks, F0, my_i, myl = families.time_coupled_blocks(3, 6000, c3["L"], c3["n0"], c3["bw"], c3["nnz_row"], 5)
    W, T, F = blocks[1]
    K, dpos = pa.kkt_leaf_assemble(6000, W)
    Bt = pa.border_assemble(6000, my_i, 0, c3["n0"], 0, A=T, F=F)
    on = pa.symbolic_probe(K, 6000, Bt=Bt)
    monkeypatch.setenv("PIPS_HIP_MF_KONLY", "1")
    konly = pa.symbolic_probe(K, 6000, Bt=Bt)
    monkeypatch.delenv("PIPS_HIP_MF_KONLY")
    monkeypatch.setenv("PIPS_HIP_MF_SPLIT", "0")
    off = pa.symbolic_probe(K, 6000, Bt=Bt)
    assert on["multifrontal"] == 1 and on["border_split"] == 1 and off["border_split"] == 0
    # round 5 (opt-in): fronts over the rows of K only - the update matrices lose their border rows as well, everything stored is the same
    assert konly["border_split"] == 2 and konly["update_matrix_doubles"] < 0.25 * on["update_matrix_doubles"]
    for k in ("nnzL", "n_sn", "n_levels", "arena_bytes", "border_row_arena_doubles"):
        assert konly[k] == on[k], k
    assert on["nnzL"] == off["nnzL"] and on["n_sn"] == off["n_sn"] and on["n_levels"] == off["n_levels"]
    assert on["update_matrix_doubles"] < 0.5 * off["update_matrix_doubles"]
    assert on["border_row_arena_doubles"] > 0 and off["border_row_arena_doubles"] == 0
    assert on["arena_bytes"] < off["arena_bytes"]                       # compact panels
    assert on["arena_bytes"] + 8 * on["border_row_arena_doubles"] <= off["arena_bytes"] + 8 * 16 * on["n_sn"]   # nothing stored twice
    monkeypatch.setenv("PIPS_HIP_MF_SPLIT", "20")                       # fewer border columns allowed than the block has
    few = pa.symbolic_probe(K, 6000, Bt=Bt)
    assert few["border_split"] == 0 and few["update_matrix_doubles"] == off["update_matrix_doubles"]
