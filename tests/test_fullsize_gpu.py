"""BASELINE.json configs[1] at full size (64 blocks x 10 000 vars, Schur dim 2000) and the per-GPU share of configs[2]
(64 of the 512 blocks, Schur dim 4000) on one GPU, checked through size-independent properties: the full arrowhead residual of solveCompressed, linearity of the solve, exact inertia of every
leaf and of the root."""
import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schur_dim", [2000, 4000], ids=["config2", "config3_per_gpu_share"])
def test_full_size_properties(schur_dim):
    import torch
    N, n_i, my_i, rho, seed = 64, 10000, 5000, 1e-3, 20261002
    n0 = myl = schur_dim // 2
    S, nleaf = n0 + myl, n_i + my_i
    bt = pa.LeafBatch(N, S)
    Ks, Bts, diags, vals = [], [], [], []
    for b in range(N):
        W, T, F, c, xs = pa.gen_block(seed, b + 1, n_i, my_i, n0, myl, rho)
        K, dpos = pa.kkt_leaf_assemble(n_i, W)
        Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
        d = np.concatenate([pa.gen_diagonal(seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
        K.val[dpos] = d
        bt.set_block(b, K, n_i, Bt)
        low = sp.csr_matrix((K.val.copy(), K.colidx, K.rowptr), shape=(nleaf, nleaf))
        Ks.append((low + sp.tril(low, -1).T).tocsr())
        Bts.append(Bt.to_scipy())
        diags.append(d)
        vals.append(K.val)
    bt.analyze(16)
    assert bt.schur_mode() == 1   # dense factor: augmented partial factorisation (MFMA) is the cheaper way
    for b in range(N):
        bt.set_values(b, vals[b])
    F0, c0, x0s = pa.gen_root(seed, n0, myl)
    xd0 = pa.gen_diagonal(seed, 0, n0)
    kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
    kkt.factorize(torch.tensor(np.concatenate(diags), device="cuda"), torch.tensor(xd0, device="cuda"))
    for b in (0, 17, 63):
        assert bt.inertia(b) == (n_i, my_i, 0)
    assert kkt.root_inertia() == (n0, myl, 0)
    rng = np.random.default_rng(0)

    def solve(b0, bl):
        b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
        kkt.solve_compressed(b0_d, bl_d)
        bt.sync()
        return b0_d.cpu().numpy(), bl_d.cpu().numpy()

    b0, bl = rng.standard_normal(S), rng.standard_normal(N * nleaf)
    x0, xl = solve(b0, bl)
    # full arrowhead residual  [K_i Br_i ; Br_i^T K_0] x - b
    K0 = sp.bmat([[sp.diags(xd0), F0.to_scipy().T], [F0.to_scipy(), None]], format="csr")
    r0 = K0 @ x0 - b0
    num = 0.0
    for b in range(N):
        xb, rb = xl[b * nleaf:(b + 1) * nleaf], bl[b * nleaf:(b + 1) * nleaf]
        ri = Ks[b] @ xb + Bts[b].T @ x0 - rb
        r0 += Bts[b] @ xb
        num += ri @ ri
    num += r0 @ r0
    assert np.sqrt(num) / np.sqrt(b0 @ b0 + bl @ bl) < 1e-9
    # linearity: solve(2.5 b + c) = 2.5 solve(b) + solve(c)
    c0v, cl = rng.standard_normal(S), rng.standard_normal(N * nleaf)
    y0, yl = solve(c0v, cl)
    z0, zl = solve(2.5 * b0 + c0v, 2.5 * bl + cl)
    assert np.linalg.norm(z0 - (2.5 * x0 + y0)) / np.linalg.norm(z0) < 1e-8
    assert np.linalg.norm(zl - (2.5 * xl + yl)) / np.linalg.norm(zl) < 1e-8
