"""Sparse root: 2-link structure - every linking row couples two neighbouring blocks - keeps the Schur
complement sparse; it is assembled as a CSR value array by the leaf kernels and factorised / solved by the sparse LDL^T
machinery instead of the dense root.  Checked against the oracle's dense restatement of the same system."""
import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem

pytestmark = pytest.mark.gpu


class TwoLinkProblem(Problem):
    """Problem whose linking rows are split among the N-1 neighbouring block pairs, L rows per pair."""

    def __init__(self, seed, N, n_i, my_i, n0, L, rho):
        myl = (N - 1) * L
        super().__init__(seed, N, n_i, my_i, n0, myl, rho)
        rng = np.random.default_rng(seed)
        for i, blk in enumerate(self.blocks):
            rows, cols, vals = [], [], []
            for pair in (i - 1, i):                       # pairs (i-1, i) and (i, i+1)
                if 0 <= pair < N - 1:
                    for r in range(pair * L, (pair + 1) * L):
                        cs = rng.choice(n_i, 3, replace=False)
                        rows += [r] * 3
                        cols += list(cs)
                        vals += list(rng.uniform(-1, 1, 3))
            F = sp.csr_matrix((vals, (rows, cols)), shape=(myl, n_i))
            F.sort_indices()
            Fp = pa.Csr(myl, n_i, F.indptr, F.indices, F.data)
            Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=blk["T"] if n0 else None, F=Fp)
            blk.update(F=Fp, Bt=Bt)
        # root rows: F0 couples every linking row to two first-stage variables
        F0 = sp.random(myl, n0, density=min(1.0, 2.0 / n0), random_state=seed, format="csr")
        F0.sort_indices()
        self.F0 = pa.Csr(myl, n0, F0.indptr, F0.indices, F0.data)


def _build(prob, sparse_root, **kw):
    bt = pa.LeafBatch(prob.N, prob.S)
    bt.set_schur_mode(1)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for b in range(prob.N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    return bt, pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, sparse_root=sparse_root, **kw)


@pytest.mark.parametrize("root_path", ["band", "amd", "dissected"])
@pytest.mark.parametrize("shape", [(6, 120, 60, 4, 3), (9, 300, 150, 6, 8), (24, 90, 45, 3, 6), (30, 60, 30, 5, 20)])
def test_sparse_root_matches_dense_oracle(shape, root_path, monkeypatch):
    """root_path: the banded all-tile elimination (linking rows, then x0), minimum degree with the head / tail split, or the linking
    rows dissected around x0 (a tree of fronts for the multifrontal head; needs >= 96 linking rows, else it falls back to the band) - the
    library picks by the thickness of the tile envelope, the test forces each.  The last shape spans five tiles."""
    import torch
    monkeypatch.setenv("PIPS_HIP_SPARSE_ROOT_BAND", {"band": "1", "amd": "0", "dissected": "2"}[root_path])
    N, n_i, my_i, n0, L = shape
    prob = TwoLinkProblem(77, N, n_i, my_i, n0, L, 5.0 / n_i)
    S = prob.S
    bt, kkt = _build(prob, True)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    ri = kkt.sparse_root_info()
    if root_path == "dissected":
        # (N - 1) L linking rows: dissected where there are enough of them, and then every linking row is in the head, x0 in the tail
        assert ri["order"] == ("dissected" if (N - 1) * L >= 96 else "band")
        if ri["order"] == "dissected":
            assert ri["n_head"] == (N - 1) * L and ri["m"] == n0 and ri["multifrontal_head"] == 1 and ri["n_levels"] >= 3
    else:
        assert ri["order"] == root_path
    kkt.factorize(diag, xd0)
    SCs = kkt.schur_sparse_to_host()
    want = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    # really sparse, and nothing of the true Schur complement falls outside the pattern
    if N >= 24:
        assert SCs.nnz < 0.3 * S * (S + 1) / 2
    mask = np.zeros((S, S), bool)
    mask[SCs.nonzero()] = True
    mask[np.arange(S), np.arange(S)] = True
    pat = np.zeros((S, S), bool)
    rp, ci = SCs.indptr, SCs.indices
    for r in range(S):
        pat[r, ci[rp[r]:rp[r + 1]]] = True
    assert np.abs(want[~pat]).max() == 0.0
    assert np.abs(SCs.toarray() - want).max() / np.abs(want).max() < 1e-9
    assert kkt.root_inertia() == (prob.n0, prob.myl, 0)
    # solveCompressed through the sparse root
    rng = np.random.default_rng(3)
    b0, bl = rng.standard_normal(S), rng.standard_normal(N * prob.n_leaf)
    b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
    kkt.solve_compressed(b0_d, bl_d)
    bt.sync()
    root = orc.DenseRootSolver(S)
    root.matrixChanged(want)
    b0_o, bs_o = b0.copy(), [bl.reshape(N, -1)[b].copy() for b in range(N)]
    orc.solve_compressed(b0_o, bs_o, [prob.oracle_leaf(b) for b in range(N)], [prob.Bt_scipy(b) for b in range(N)], root,
                         prob.n0, 0, 0, prob.myl, 0)
    assert np.linalg.norm(b0_d.cpu().numpy() - b0_o) / np.linalg.norm(b0_o) < 1e-8
    xl = bl_d.cpu().numpy().reshape(N, -1)
    for b in range(N):
        assert np.linalg.norm(xl[b] - bs_o[b]) / np.linalg.norm(bs_o[b]) < 1e-8
    # the dense root on the same problem gives the same answer
    bt2, kkt2 = _build(prob, False)
    kkt2.factorize(diag, xd0)
    c0_d, cl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
    kkt2.solve_compressed(c0_d, cl_d)
    bt2.sync()
    assert np.linalg.norm((c0_d - b0_d).cpu().numpy()) / np.linalg.norm(b0_o) < 1e-8


def test_sparse_root_reduction_with_global_pattern(monkeypatch):
    """n_ranks > 1 needs the border column sets of all blocks (every rank must reduce the same value array): driven here with
    a one-rank communicator, the global pattern passed explicitly."""
    import torch
    monkeypatch.setenv("PIPS_HIP_FORCE_REDUCE", "1")
    prob = TwoLinkProblem(78, 5, 150, 75, 4, 4, 5.0 / 150)
    seen = []

    def allreduce(ptr, n):
        seen.append(n)

    comm = pa.ExternalComm(allreduce)
    cols = [np.nonzero(np.diff(prob.blocks[b]["Bt"].rowptr) > 0)[0] for b in range(prob.N)]
    bt, kkt = _build(prob, True, comm=comm, rank=0, n_ranks=1, all_block_cols=cols)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    SCs = kkt.schur_sparse_to_host()
    want = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    assert seen == [SCs.nnz]
    assert np.abs(SCs.toarray() - want).max() / np.abs(want).max() < 1e-9
    comm.close()
