"""GPU parity on the GENERAL block structure the reference supports: inequality rows in the leaves (D_i, -Omega^-1
diagonal), C_i / G_i border parts, root equality rows A0 (my0 > 0), root inequality rows C0 (mz0 > 0, eliminated in
Dsolve) and linking inequalities G0 with their diagonal — against the oracle restatement (finalizeKKTdense
sLinsysRootAug.C:1769-1796, solveReducedLinkCons :384-466).  Also the inertia-correcting regularisation contract (a4)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu


def _rand(seed, blk, rows, cols, rho):
    """rows x cols random sparse CSR (about rho*cols, at least 1, entries per row; sorted distinct columns; U(-1,1))."""
    rng = np.random.default_rng(seed * 1000 + blk)
    k = int(min(cols, max(1, round(rho * cols))))
    colidx = np.concatenate([np.sort(rng.choice(cols, k, replace=False)) for _ in range(rows)]) if rows else np.zeros(0, np.int32)
    return pa.Csr(rows, cols, np.arange(rows + 1, dtype=np.int32) * k, colidx, rng.uniform(-1, 1, rows * k))


class GeneralProblem:
    def __init__(self, seed, N, nx, my, mz, n0, my0, mz0, myl, mzl, rho):
        self.dims = (N, nx, my, mz, n0, my0, mz0, myl, mzl)
        self.S = n0 + my0 + myl + mzl
        rng = np.random.default_rng(seed)
        self.blocks = []
        for b in range(1, N + 1):
            W = _rand(seed, b, my, nx, rho)
            Dm = _rand(seed + 1, b, mz, nx, rho)
            T = _rand(seed + 2, b, my, n0, 2.0 / n0)
            Cb = _rand(seed + 3, b, mz, n0, 2.0 / n0)
            F = _rand(seed + 4, b, myl, nx, 3.0 / nx)
            G = _rand(seed + 5, b, mzl, nx, 3.0 / nx)
            K, dpos = pa.kkt_leaf_assemble(nx, W, D=Dm)
            diag = np.concatenate([10 ** rng.uniform(-3, 3, nx), -1e-9 * np.ones(my), -10 ** rng.uniform(-3, 3, mz)])
            K.val[dpos] = diag
            Bt = pa.border_assemble(nx, my, mz, n0, my0, A=T, Cm=Cb, F=F, G=G)
            self.blocks.append(dict(K=K, Bt=Bt, diag=diag, W=W, Dm=Dm, T=T, Cb=Cb, F=F, G=G))
        self.A0 = _rand(seed + 6, 1, my0, n0, 3.0 / n0)
        self.C0 = _rand(seed + 7, 1, mz0, n0, 3.0 / n0)
        self.F0 = _rand(seed + 8, 1, myl, n0, 3.0 / n0)
        self.G0 = _rand(seed + 9, 1, mzl, n0, 3.0 / n0)
        self.x_diag0 = 10 ** rng.uniform(-2, 2, n0)
        self.z_diag0 = -10 ** rng.uniform(-2, 2, mz0)
        self.z_diag_link = -10 ** rng.uniform(-2, 2, mzl)

    def K_scipy(self, b):
        K = self.blocks[b]["K"]
        return sp.csr_matrix((K.val.copy(), K.colidx, K.rowptr), shape=(K.nrows, K.ncols))


def test_general_structure_factorize_and_solve_compressed():
    _check_general(17, (3, 200, 80, 40, 20, 6, 7, 9, 5), 0.03, 0)


@pytest.mark.parametrize("case", range(int(os.environ.get("PIPS_FUZZ_CASES", "30"))))
def test_general_structure_sweep(case, monkeypatch):
    """Seeded sweep over the block dimensions, including the empty parts (no leaf inequalities, no root equality /
    inequality rows, no linking rows of one kind) and the three Schur modes."""
    rng = np.random.default_rng(7000 + case)
    N = int(rng.integers(1, 4))
    nx = int(rng.choice([40, 120, 260]))
    my = int(nx * rng.choice([0.2, 0.4]))
    mz = int(nx * rng.choice([0.0, 0.1, 0.3]))
    n0 = int(rng.integers(2, 16))
    my0, mz0 = int(rng.integers(0, min(5, n0 // 2 + 1))), int(rng.integers(0, 6))   # A0 needs full row rank: my0 <= n0 / 2
    myl, mzl = int(rng.integers(0, 8)), int(rng.integers(0, 6))
    rho = max(float(rng.choice([0.03, 0.08])), 4.0 / nx)   # >= 4 entries per row: one-entry rows make [W; D] rank deficient
    sparse_root = bool(rng.integers(0, 2))      # the CSR Schur complement + one-block sparse engine as root
    monkeypatch.setenv("PIPS_HIP_SPARSE_ROOT_BAND", str(int(rng.integers(0, 2))))   # ... with either elimination path
    _check_general(900 + case, (N, nx, my, mz, n0, my0, mz0, myl, mzl), rho, 1 if sparse_root else int(rng.integers(0, 3)), sparse_root)


def _check_general(seed, dims, rho, schur_mode, sparse_root=False):
    import torch
    N, nx, my, mz, n0, my0, mz0, myl, mzl = dims
    gp = GeneralProblem(seed, N, nx, my, mz, n0, my0, mz0, myl, mzl, rho)
    S, nleaf = gp.S, nx + my + mz
    bt = pa.LeafBatch(N, S)
    bt.set_schur_mode(schur_mode)
    for b in range(N):
        bt.set_block(b, gp.blocks[b]["K"], nx, gp.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(N):
        bt.set_values(b, gp.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, n0, my0, myl, mzl, A0=gp.A0 if my0 else None, F0=gp.F0 if myl else None, G0=gp.G0 if mzl else None,
                       sparse_root=sparse_root)
    if mz0:
        kkt.set_root_inequalities(gp.C0)
        zd0 = torch.tensor(gp.z_diag0, device="cuda")
        kkt.set_zdiag0(zd0)
    kkt.factorize(torch.tensor(np.concatenate([b["diag"] for b in gp.blocks]), device="cuda"),
                  torch.tensor(gp.x_diag0, device="cuda"), torch.tensor(gp.z_diag_link, device="cuda") if mzl else None)
    got = kkt.schur_sparse_to_host().toarray() if sparse_root else hip_lower_as_rowmajor(kkt.schur_to_host(), S)
    # ---- oracle
    leaf, Bts = [], []
    SC = np.zeros((S, S))
    for b in range(N):
        s = orc.OracleLdl(gp.K_scipy(b), n_primal=nx)
        s.matrixChanged()
        Bt = gp.blocks[b]["Bt"].to_scipy()
        orc.add_term_to_schur_compl_blocked(SC, s, Bt)
        leaf.append(s)
        Bts.append(Bt)
        assert bt.inertia(b) == s.get_inertia() == (nx, my + mz, 0)
    SCf = orc.finalize_kkt_dense(SC, n0, my0, myl, mzl, gp.x_diag0, A0=gp.A0.to_scipy(), F0=gp.F0.to_scipy(), G0=gp.G0.to_scipy(),
                                 C0=gp.C0.to_scipy(), z_diag=gp.z_diag0, z_diag_link=gp.z_diag_link)
    want = np.tril(SCf)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-9, (dims, schur_mode, sparse_root)
    root = orc.DenseRootSolver(S)
    root.matrixChanged(want)
    assert kkt.root_inertia() == (n0, my0 + myl + mzl, 0)
    rng = np.random.default_rng(1)
    b0 = rng.standard_normal(S + mz0)
    bl = rng.standard_normal(N * nleaf)
    b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
    kkt.solve_compressed(b0_d, bl_d)
    bt.sync()
    b0_o, bs_o = b0.copy(), [bl.reshape(N, -1)[b].copy() for b in range(N)]
    orc.solve_compressed(b0_o, bs_o, leaf, Bts, root, n0, my0, mz0, myl, mzl, C0=gp.C0.to_scipy(), z_diag_reg=gp.z_diag0)
    assert np.linalg.norm(b0_d.cpu().numpy() - b0_o) / np.linalg.norm(b0_o) < 1e-8, (dims, schur_mode)
    xl = bl_d.cpu().numpy().reshape(N, -1)
    for b in range(N):
        assert np.linalg.norm(xl[b] - bs_o[b]) / np.linalg.norm(bs_o[b]) < 1e-8, (dims, schur_mode)


def test_inertia_contract_drives_regularisation_loop():
    """factorize_with_correct_inertia (LinearSystem.C:296-325): a rank-deficient equality block yields a wrong/zero inertia
    report, add_regularization_local_kkt + refactor repairs it — the host-side loop only needs get_inertia()."""
    nx, my = 120, 60
    W = _rand(5, 1, my, nx, 0.05)
    # make the last equality row an exact copy of the first one -> W rank deficient, K singular without regularisation
    s0, e0 = W.rowptr[0], W.rowptr[1]
    s1 = W.rowptr[my - 1]
    W.colidx[s1:s1 + (e0 - s0)] = W.colidx[s0:e0]
    W.val[s1:s1 + (e0 - s0)] = W.val[s0:e0]
    K, dpos = pa.kkt_leaf_assemble(nx, W)
    K.val[dpos] = np.concatenate([np.ones(nx), np.zeros(my)])
    bt = pa.LeafBatch(1, 0)
    bt.set_block(0, K, nx)
    bt.analyze(1)
    bt.set_values(0, K.val)
    bt.factor()
    pos, neg, zero = bt.inertia(0)
    assert (pos, neg, zero) != (nx, my, 0) and zero >= 1
    # Friedlander-Orban style: increase the dual regularisation until the inertia is (nx, my, 0)
    reg, tries = 1e-8, 0
    while bt.inertia(0) != (nx, my, 0) and tries < 12:
        bt.add_regularization(reg, reg)
        bt.factor()
        reg *= 100.0
        tries += 1
    assert bt.inertia(0) == (nx, my, 0) and tries <= 3
