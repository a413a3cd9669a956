"""Ltsolve from the augmented factor alone (Engine::solve_border_backward: u = L^-T (L21^T x0), one backward sweep, VERDICT r1 item
9a) against the border product + full solve it replaces: same solveCompressed result - dense tails, time-coupled blocks (sparse
head with spines), the general block structure - and the residual of the explicitly assembled arrowhead system."""
import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from tests.util import Problem

pytestmark = pytest.mark.gpu


def _solve_both(monkeypatch, build):
    import torch
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("PIPS_HIP_BORDER_BACKWARD", mode)
        bt, kkt, factor, S, nl = build()
        factor()
        rng = np.random.default_rng(9)
        res = []
        for rep in range(2):
            b0, bl = rng.standard_normal(S), rng.standard_normal(nl)
            b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
            kkt.solve_compressed(b0_d, bl_d)
            bt.sync()
            res.append((b0, bl, b0_d.cpu().numpy(), bl_d.cpu().numpy()))
        out.append(res)
        kkt.close()
        bt.close()
    return out


@pytest.mark.parametrize("shape", [(4, 1500, 750, 40, 30, 0.008), (9, 600, 300, 12, 20, 0.02), (2, 3000, 1500, 100, 60, 0.004)])
def test_border_backward_equals_border_product_plus_solve(shape, monkeypatch):
    import torch
    prob = Problem(55, *shape)

    def build():
        bt = pa.LeafBatch(prob.N, prob.S)
        for b in range(prob.N):
            bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
        bt.analyze(4)
        for b in range(prob.N):
            bt.set_values(b, prob.blocks[b]["K"].val)
        kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
        xd0 = torch.tensor(prob.x_diag0, device="cuda")
        return bt, kkt, (lambda: kkt.factorize(None, xd0)), prob.S, prob.N * prob.n_leaf

    new, old = _solve_both(monkeypatch, build)
    for (b0, bl, x0n, xln), (_, _, x0o, xlo) in zip(new, old):
        assert np.abs(x0n - x0o).max() <= 1e-10 * np.abs(x0o).max()
        assert np.abs(xln - xlo).max() <= 1e-10 * np.abs(xlo).max()
    # full arrowhead residual of the new path
    b0, bl, x0, xl = new[0]
    F0s = prob.F0.to_scipy()
    K0 = sp.bmat([[sp.diags(prob.x_diag0), F0s.T], [F0s, None]], format="csr")
    r0 = K0 @ x0 - b0
    nl = prob.n_leaf
    rmax = 0.0
    for b in range(prob.N):
        Bt = prob.Bt_scipy(b)
        xb = xl[b * nl:(b + 1) * nl]
        rmax = max(rmax, np.abs(prob.K_full(b) @ xb + Bt.T @ x0 - bl[b * nl:(b + 1) * nl]).max())
        r0 += Bt @ xb
    scale = max(np.abs(b0).max(), np.abs(bl).max())
    assert rmax <= 1e-9 * scale and np.abs(r0).max() <= 1e-9 * scale


def test_border_backward_time_coupled_blocks(monkeypatch):
    """banded blocks: everything is sparse head (levels, chains, spines), 2-link border"""
    import torch
    from tests.test_configs_gpu import energy_like_blocks
    N, n_i, L, n0, bw, nnz_row = 8, 5000, 6, 16, 12, 10
    blocks, F0, my_i, myl = energy_like_blocks(N, n_i, L, n0, bw, nnz_row, 5)
    S, nleaf = n0 + myl, n_i + my_i
    rng = np.random.default_rng(4)
    diags = [np.concatenate([10 ** rng.uniform(-2, 2, n_i), -1e-8 * np.ones(my_i)]) for _ in range(N)]
    xd0 = 10 ** rng.uniform(-1, 1, n0)

    def build():
        bt = pa.LeafBatch(N, S)
        Ks = []
        for b, (W, T, F) in enumerate(blocks):
            K, dpos = pa.kkt_leaf_assemble(n_i, W)
            K.val[dpos] = diags[b]
            Ks.append(K)
            bt.set_block(b, K, n_i, pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F))
        bt.analyze(4)
        for b in range(N):
            bt.set_values(b, Ks[b].val)
        kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
        return bt, kkt, (lambda: kkt.factorize(torch.tensor(np.concatenate(diags), device="cuda"), torch.tensor(xd0, device="cuda"))), S, N * nleaf

    new, old = _solve_both(monkeypatch, build)
    for (b0, bl, x0n, xln), (_, _, x0o, xlo) in zip(new, old):
        assert np.abs(x0n - x0o).max() <= 1e-9 * np.abs(x0o).max()
        assert np.abs(xln - xlo).max() <= 1e-9 * np.abs(xlo).max()


def test_border_backward_general_structure(monkeypatch):
    import torch
    from tests.test_general_gpu import GeneralProblem
    dims = (3, 400, 160, 80, 20, 6, 0, 9, 5)
    N, nx, my, mz, n0, my0, mz0, myl, mzl = dims
    gp = GeneralProblem(21, *dims, 0.02)
    nleaf = nx + my + mz

    def build():
        bt = pa.LeafBatch(N, gp.S)
        for b in range(N):
            bt.set_block(b, gp.blocks[b]["K"], nx, gp.blocks[b]["Bt"])
        bt.analyze(4)
        for b in range(N):
            bt.set_values(b, gp.blocks[b]["K"].val)
        kkt = pa.KktSystem(bt, n0, my0, myl, mzl, A0=gp.A0, F0=gp.F0, G0=gp.G0)
        f = lambda: kkt.factorize(torch.tensor(np.concatenate([b["diag"] for b in gp.blocks]), device="cuda"),
                                  torch.tensor(gp.x_diag0, device="cuda"), torch.tensor(gp.z_diag_link, device="cuda"))
        return bt, kkt, f, gp.S, N * nleaf

    new, old = _solve_both(monkeypatch, build)
    for (b0, bl, x0n, xln), (_, _, x0o, xlo) in zip(new, old):
        assert np.abs(x0n - x0o).max() <= 1e-9 * np.abs(x0o).max()
        assert np.abs(xln - xlo).max() <= 1e-9 * np.abs(xlo).max()


@pytest.mark.parametrize("case", range(12))
def test_border_backward_forced_on_the_general_structure_sweep(case, monkeypatch):
    """the seeded general-structure cases of test_general_gpu.py (root equality / inequality rows, linking rows of both kinds,
    empty parts) with the augmented-factor Ltsolve forced on, against the restatement"""
    from tests.test_general_gpu import _check_general
    monkeypatch.setenv("PIPS_HIP_BORDER_BACKWARD", "1")
    rng = np.random.default_rng(8100 + case)
    N = int(rng.integers(1, 4))
    nx = int(rng.choice([40, 120, 260, 500]))
    my = int(nx * rng.choice([0.2, 0.4]))
    mz = int(nx * rng.choice([0.0, 0.1, 0.3]))
    n0 = int(rng.integers(2, 16))
    my0, mz0 = int(rng.integers(0, min(5, n0 // 2 + 1))), int(rng.integers(0, 6))
    myl, mzl = int(rng.integers(0, 8)), int(rng.integers(0, 6))
    rho = max(float(rng.choice([0.03, 0.08])), 4.0 / nx)
    _check_general(700 + case, (N, nx, my, mz, n0, my0, mz0, myl, mzl), rho, 1, False)
