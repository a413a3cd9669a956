"""GPU parity tests of the leaf path: HIP supernodal LDL^T / solves / Schur contribution vs the oracle (through the C ABI)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu

RTOL_SOLVE = 1e-9   # relative to ||x||, after one refinement step on both sides (north_star: 1e-8 end to end)
RTOL_SC = 1e-9      # relative to max|SC|


def _solve_check(prob, b, force_n_head, rtol=RTOL_SOLVE):
    blk = prob.blocks[b]
    s = pa.HipLdlSolver(blk["K"], n_primal=prob.n_i)
    if force_n_head is not None:
        # single-handle path has no option hook; use a one-block batch for forced cuts
        s.close()
        bt = pa.LeafBatch(1, 0)
        bt.set_block(0, blk["K"], prob.n_i)
        bt.set_options(force_n_head=force_n_head)
        bt.analyze(1)
        bt.set_values(0, blk["K"].val)
        bt.factor()
        rng = np.random.default_rng(b)
        rhs = rng.standard_normal(prob.n_leaf)
        x = rhs.copy()
        bt.solve(x)
        inertia = bt.inertia(0)
        info = bt.info()
    else:
        s.matrixChanged()
        rng = np.random.default_rng(b)
        rhs = rng.standard_normal(prob.n_leaf)
        x = rhs.copy()
        s.solve(x)
        inertia = s.get_inertia()
        info = s.info()
    o = prob.oracle_leaf(b)
    xo = rhs.copy()
    o.solve(xo)
    Kf = prob.K_full(b)
    res = np.linalg.norm(Kf @ x - rhs) / np.linalg.norm(rhs)
    err = np.linalg.norm(x - xo) / np.linalg.norm(xo)
    assert inertia == o.get_inertia() == (prob.n_i, prob.my_i, 0), (inertia, o.get_inertia(), info)
    assert err < rtol and res < 1e-10, (err, res, info)


@pytest.mark.parametrize("n_i,rho", [(60, 0.1), (300, 0.02), (1000, 0.01)])
@pytest.mark.parametrize("cut", ["model", "all_head", "all_tail"])
def test_leaf_solve_matches_oracle(n_i, rho, cut):
    prob = Problem(11, 2, n_i, n_i // 2, 8, 8, rho)
    force = {"model": None, "all_head": prob.n_leaf, "all_tail": 0}[cut]
    for b in range(prob.N):
        _solve_check(prob, b, force)


@pytest.mark.parametrize("n_i", [200, 1500])
@pytest.mark.parametrize("scheme", ["separate_sweeps", "interleaved"])
def test_leaf_multi_rhs_rows(scheme, n_i, monkeypatch):
    """solve(nrhs, ...) with one right-hand side per row: both device schemes (one sweep per right-hand side, interleaved
    multi-vector layout) against SuperLU; n_i = 1500 has a dense tail of several tiles."""
    monkeypatch.setenv("PIPS_HIP_MULTI", "1" if scheme == "interleaved" else "0")
    prob = Problem(5, 1, n_i, n_i // 2, 4, 4, 6.0 / n_i)
    blk = prob.blocks[0]
    s = pa.HipLdlSolver(blk["K"], n_primal=prob.n_i)
    s.matrixChanged()
    rng = np.random.default_rng(0)
    R = rng.standard_normal((40, prob.n_leaf))   # 40 right-hand sides: two chunks of the batched multi-RHS path
    X = R.copy()
    s.solve(X)
    lu = spl.splu(prob.K_full(0))
    for k in range(40):
        xr = lu.solve(R[k])
        assert np.linalg.norm(X[k] - xr) / np.linalg.norm(xr) < 1e-9
    # empty right-hand sides stay out of the solve, as in PardisoSolver::solve (PardisoSolver.C:276-352): zero rows come back
    # as zeros and the non-zero ones are unaffected by the packing (chunks with one, several and no non-zero row)
    for keep in ([3], [0, 7, 8, 30], []):
        Rz = np.zeros((40, prob.n_leaf))
        Rz[keep] = R[keep]
        Xz = Rz.copy()
        s.solve(Xz)
        for k in range(40):
            if k in keep:
                assert np.linalg.norm(Xz[k] - X[k]) / np.linalg.norm(X[k]) < 1e-12
            else:
                assert not Xz[k].any()


@pytest.mark.parametrize("nrhs", [8, 33, 100, 257, 300])
@pytest.mark.parametrize("scheme", ["by_size", "half_panels", "whole_panels"])
def test_leaf_many_rhs_on_the_matrix_pipe(scheme, nrhs, monkeypatch):
    """solve(nrhs, ...) beyond one panel of 32 interleaved right-hand sides and beyond one pass of 256: every launch takes all panels, the
    dense tail's sweeps are one launch per direction with a workgroup per (tile row, quarter or half panel) - a single leaf, by the size of
    the launch - or per (tile row, panel) - what large batches take (PIPS_HIP_MULTI=4 / 2 force the two here); against SuperLU, and against the same handle one column at a time."""
    monkeypatch.setenv("PIPS_HIP_MULTI", {"by_size": "1", "half_panels": "4", "whole_panels": "2"}[scheme])
    n_i = 1500                                  # a dense tail of several tiles
    prob = Problem(5, 1, n_i, n_i // 2, 4, 4, 6.0 / n_i)
    s = pa.HipLdlSolver(prob.blocks[0]["K"], n_primal=prob.n_i)
    s.matrixChanged()
    rng = np.random.default_rng(nrhs)
    R = rng.standard_normal((nrhs, prob.n_leaf))
    R[nrhs // 2] *= 1e6                         # (the columns of a pass do not see each other)
    X = R.copy()
    s.solve(X)
    lu = spl.splu(prob.K_full(0))
    for k in range(nrhs):
        xr = lu.solve(R[k])
        assert np.linalg.norm(X[k] - xr) / np.linalg.norm(xr) < 1e-9
    for k in (0, nrhs // 2, nrhs - 1):
        one = R[k].copy()
        s.solve(one)
        assert np.linalg.norm(one - X[k]) / np.linalg.norm(one) < 1e-11


def test_leaf_many_rhs_refine_only_where_needed():
    """solve(nrhs) with the adapters' refinement (at most two steps, backward error 1e-15: iparm[7] = 2, PardisoProjectSolver.C:72): a chunk whose
    first solve is accurate takes no correction solve; a pivot rule that perturbs (threshold far above the small dual pivots) leaves factors
    the measure rejects - the steps are taken and bring the chunk to the accuracy of the refined single solve."""
    n_i = 1200
    prob = Problem(7, 1, n_i, n_i // 2, 4, 4, 6.0 / n_i)
    K = prob.blocks[0]["K"]
    rng = np.random.default_rng(3)
    R = rng.standard_normal((40, prob.n_leaf))
    lu = spl.splu(prob.K_full(0))
    s = pa.HipLdlSolver(K, n_primal=prob.n_i, refine_steps=2, refine_tol=1e-15, backward_error=True)
    s.matrixChanged()
    X = R.copy()
    s.solve(X)
    assert s.info()["last_refinement_steps"] == 0
    for k in range(40):
        xr = lu.solve(R[k])
        assert np.linalg.norm(X[k] - xr) / np.linalg.norm(xr) < 1e-9
    s.close()
    s = pa.HipLdlSolver(K, n_primal=prob.n_i, refine_steps=2, refine_tol=1e-15, backward_error=True)
    s.set_pivot_rule(1e-3, 1e-3)                 # pivots below 1e-3 max|K| are replaced: the factors are those of a perturbed matrix
    s.matrixChanged()
    if s.get_inertia()[2] == 0:
        pytest.skip("no pivot of this block falls under the threshold")
    X = R.copy()
    s.solve(X)
    assert s.info()["last_refinement_steps"] >= 1
    one = R[5].copy()
    s.solve(one)
    assert np.linalg.norm(one - X[5]) / np.linalg.norm(one) < 1e-6
    s.close()


def test_leaf_many_rhs_refine_per_column():
    """The correction solve of solve(nrhs) takes only the columns whose own measure asks for it: with a tolerance between the measures of the
    columns (border columns of a block: sparse right-hand sides, some of which come out of the first solve above 1e-16-ish) the result
    of every column equals either its refined or its unrefined solution, and all of them the reference."""
    n_i = 2000
    prob = Problem(9, 1, n_i, n_i // 2, 60, 40, 5.0 / n_i)
    K = prob.blocks[0]["K"]
    Bt = prob.Bt_scipy(0)
    cols = np.nonzero(np.diff(Bt.indptr) > 0)[0][:96]
    R = np.ascontiguousarray(Bt[cols].toarray())
    lu = spl.splu(prob.K_full(0))
    sols = {}
    for name, steps, tol in (("never", 0, 0.0), ("always", 1, 0.0), ("some", 2, 2e-16), ("none_needed", 2, 1e-10)):
        s = pa.HipLdlSolver(K, n_primal=prob.n_i, refine_steps=steps, refine_tol=tol, backward_error=tol > 0)
        s.set_deterministic()      # (so that the handles agree to the bit and "left as the first solve gave it" can be checked)
        s.matrixChanged()
        X = R.copy()
        s.solve(X)
        sols[name] = (X, s.info()["last_refinement_steps"])
        s.close()
    assert sols["none_needed"][1] == 0 and np.array_equal(sols["none_needed"][0], sols["never"][0])
    for k in range(len(cols)):
        xr = lu.solve(R[k])
        for name in sols:
            assert np.linalg.norm(sols[name][0][k] - xr) / np.linalg.norm(xr) < 1e-9
    X, steps = sols["some"]
    same_as_never = [np.array_equal(X[k], sols["never"][0][k]) for k in range(len(cols))]
    if steps > 0:      # (some column asked for it; the others were left as the first solve gave them)
        assert not all(same_as_never)
    print("columns left unrefined:", sum(same_as_never), "of", len(cols), "steps", steps)


def test_refactor_after_diagonal_change():
    """matrixChanged() after mutating the diagonal in place (a2/a3), pattern fixed."""
    prob = Problem(3, 1, 400, 200, 4, 4, 0.02)
    blk = prob.blocks[0]
    s = pa.HipLdlSolver(blk["K"], n_primal=prob.n_i)
    s.matrixChanged()
    for it in range(3):
        blk["K"].val[blk["dpos"][:prob.n_i]] = pa.gen_diagonal(100 + it, 1, prob.n_i, -6, 6)
        s.matrixChanged()
        rhs = np.random.default_rng(it).standard_normal(prob.n_leaf)
        x = rhs.copy()
        s.solve(x)
        Kf = prob.K_full(0)
        assert np.linalg.norm(Kf @ x - rhs) / np.linalg.norm(rhs) < 1e-10
        assert s.get_inertia() == (prob.n_i, prob.my_i, 0)


@pytest.mark.parametrize("schur_mode", [1, 2], ids=["augmented", "blocked_solves"])
@pytest.mark.parametrize("cut", ["model", "all_head", "all_tail"])
@pytest.mark.parametrize("shape", [(3, 120, 16, 12, 0.05), (4, 1000, 100, 100, 0.01)])
def test_schur_contribution_matches_oracle(shape, cut, schur_mode):
    """Both ways of forming SC -= Br^T K^-1 Br: the augmented partial factorisation and the reference's own blocked
    multi-RHS solves (K4-K6) on the device."""
    import torch
    N, n_i, n0, myl, rho = shape
    prob = Problem(21, N, n_i, n_i // 2, n0, myl, rho)
    S = prob.S
    bt = pa.LeafBatch(N, S)
    bt.set_schur_mode(schur_mode)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    force = {"model": -1, "all_head": prob.n_leaf, "all_tail": 0}[cut]
    bt.set_options(force_n_head=force)
    bt.analyze(4)
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    assert bt.schur_mode() == schur_mode
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    torch.cuda.synchronize()
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    want = np.tril(prob.oracle_schur())
    scale = np.abs(want).max()
    assert np.abs(got - want).max() / scale < RTOL_SC, (np.abs(got - want).max() / scale, bt.info())
    for b in range(N):
        assert bt.inertia(b) == (prob.n_i, prob.my_i, 0)
    # second factorisation with new diagonals accumulates on a fresh SC identically (pattern reuse)
    diag = np.concatenate([blk["diag"] for blk in prob.blocks])
    bt.set_diagonals(diag)
    SC.zero_()
    bt.factor(SC, S)
    bt.sync()
    got2 = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    assert np.abs(got2 - want).max() / scale < RTOL_SC


def test_batch_solve_and_border_products():
    import torch
    prob = Problem(33, 3, 500, 250, 40, 30, 0.02)
    S, N = prob.S, prob.N
    bt = pa.LeafBatch(N, S)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    bt.factor()
    rng = np.random.default_rng(1)
    rhs = rng.standard_normal(N * prob.n_leaf)
    x = torch.tensor(rhs, device="cuda")
    bt.solve(x)
    xs = x.cpu().numpy().reshape(N, -1)
    for b in range(N):
        r = rhs.reshape(N, -1)[b]
        assert np.linalg.norm(prob.K_full(b) @ xs[b] - r) / np.linalg.norm(r) < 1e-10
    # b0 -= sum Br^T z ;  t = Br x0
    z = torch.tensor(rng.standard_normal(N * prob.n_leaf), device="cuda")
    b0 = torch.zeros(S, dtype=torch.float64, device="cuda")
    bt.border_tmult(z, b0, -1.0)
    zz = z.cpu().numpy().reshape(N, -1)
    want = -sum(prob.Bt_scipy(b) @ zz[b] for b in range(N))
    assert np.allclose(b0.cpu().numpy(), want, rtol=1e-12, atol=1e-12)
    x0 = torch.tensor(rng.standard_normal(S), device="cuda")
    t = torch.zeros(N * prob.n_leaf, dtype=torch.float64, device="cuda")
    bt.border_mult(x0, t, 1.0)
    tt = t.cpu().numpy().reshape(N, -1)
    for b in range(N):
        assert np.allclose(tt[b], prob.Bt_scipy(b).T @ x0.cpu().numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,n_primal", [(50, 30), (128, 64), (300, 200), (1000, 500)])
def test_dense_root_matches_lapack(n, n_primal):
    """HIP tiled dense LDL^T vs the reference's own LAPACK path (dsytrf/dsytrs, DeSymIndefSolver.C:56-129)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(n)
    m = n - n_primal
    H = rng.standard_normal((n_primal, n_primal))
    H = H @ H.T + n_primal * np.eye(n_primal)
    A = rng.standard_normal((m, n_primal))
    E = rng.standard_normal((m, m))
    E = E @ E.T * 1e-3 + 1e-6 * np.eye(m)
    M = np.block([[H, A.T], [A, -E]])
    low = np.tril(M)                       # row-major, lower authoritative, garbage-free upper = 0
    o = orc.DenseRootSolver(n)
    o.matrixChanged(low)
    h = pa.HipDenseLdlSolver(n, n_primal)
    h.matrixChanged(low)
    rhs = rng.standard_normal(n)
    xo, xh = rhs.copy(), rhs.copy()
    o.solve(xo)
    h.solve(xh)
    assert np.linalg.norm(xh - xo) / np.linalg.norm(xo) < 1e-9
    assert np.linalg.norm(M @ xh - rhs) / np.linalg.norm(rhs) < 1e-10
    assert h.get_inertia() == (n_primal, m, 0)
    assert o.get_inertia()[:2] == (n_primal, m)


def test_dense_root_large_right_looking_with_lookahead():
    """S = 6400 (50 tile columns): the single-block root runs right-looking with the one-column lookahead on a side stream
    (DenseLdl, engine.hip); checked against LAPACK dsytrf/dsytrs like the small sizes, and for determinism across calls."""
    import torch
    from oracle import oracle as orc
    n, n_primal = 6400, 3200
    g = torch.Generator(device="cuda").manual_seed(3)
    M = torch.rand((n, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
    M = M + M.T
    d = torch.full((n,), float(n), dtype=torch.float64, device="cuda")
    d[n_primal:] = -float(n)
    M += torch.diag(d)
    Mh = M.cpu().numpy()
    h = pa.HipDenseLdlSolver(n, n_primal)
    xs = []
    for rep in range(2):
        work = M.clone()
        h.matrixChanged_dev(work, n)
        x = torch.ones(n, dtype=torch.float64, device="cuda") * (1.0 + torch.arange(n, device="cuda") % 7)
        rhs = x.clone().cpu().numpy()
        h.solve_dev(x)
        torch.cuda.synchronize()
        xs.append(x.cpu().numpy())
    assert h.get_inertia() == (n_primal, n - n_primal, 0)
    assert np.array_equal(xs[0], xs[1])          # no atomics in the dense root: bit-reproducible, also with two streams
    assert np.linalg.norm(Mh @ xs[0] - rhs) / np.linalg.norm(rhs) < 1e-12
    o = orc.DenseRootSolver(n)
    o.matrixChanged(np.tril(Mh))
    xo = rhs.copy()
    o.solve(xo)
    assert np.linalg.norm(xs[0] - xo) / np.linalg.norm(xo) < 1e-10


def test_heterogeneous_blocks_in_one_batch():
    """Blocks of different sizes and different head/tail splits (some without a dense tail) share every launch."""
    import torch
    shapes = [(40, 0.15), (700, 0.012), (150, 0.05), (1200, 0.008)]
    S = 24
    n0, myl = 14, 10
    probs = [Problem(50 + i, 1, n_i, n_i // 2, n0, myl, rho) for i, (n_i, rho) in enumerate(shapes)]
    bt = pa.LeafBatch(len(probs), S)
    for i, pr in enumerate(probs):
        bt.set_block(i, pr.blocks[0]["K"], pr.n_i, pr.blocks[0]["Bt"])
    bt.analyze(4)
    for i, pr in enumerate(probs):
        bt.set_values(i, pr.blocks[0]["K"].val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    want = np.zeros((S, S))
    for pr in probs:
        want += pr.oracle_schur()
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    assert np.abs(got - np.tril(want)).max() / np.abs(want).max() < RTOL_SC, bt.info()
    rhs = np.concatenate([np.random.default_rng(i).standard_normal(pr.n_leaf) for i, pr in enumerate(probs)])
    x = rhs.copy()
    bt.solve(x)
    off = 0
    for i, pr in enumerate(probs):
        xi, ri = x[off:off + pr.n_leaf], rhs[off:off + pr.n_leaf]
        assert np.linalg.norm(pr.K_full(0) @ xi - ri) / np.linalg.norm(ri) < 1e-10
        assert bt.inertia(i) == (pr.n_i, pr.my_i, 0)
        off += pr.n_leaf


def test_block_without_border_and_zero_schur_dim():
    prob = Problem(8, 2, 300, 150, 0, 0, 0.03)
    bt = pa.LeafBatch(2, 0)
    for b in range(2):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i)
    bt.analyze(2)
    for b in range(2):
        bt.set_values(b, prob.blocks[b]["K"].val)
    bt.factor()
    rhs = np.random.default_rng(3).standard_normal(2 * prob.n_leaf)
    x = rhs.copy()
    bt.solve(x)
    for b in range(2):
        r = rhs.reshape(2, -1)[b]
        assert np.linalg.norm(prob.K_full(b) @ x.reshape(2, -1)[b] - r) / np.linalg.norm(r) < 1e-10


@pytest.mark.parametrize("kind", ["spd_grid", "indefinite_banded"])
def test_structured_matrices_without_inertia_hint(kind):
    """No inertia hint (unconstrained ordering, sign-agnostic pivot rule) on structured matrices: a 2-D grid Laplacian
    (deep elimination tree, wide supernodes) and a banded quasi-definite matrix."""
    import scipy.sparse as sp
    if kind == "spd_grid":
        g = 40
        I = sp.identity(g)
        T = sp.diags([-1, 4.2, -1], [-1, 0, 1], shape=(g, g))
        A = (sp.kron(I, T) + sp.kron(sp.diags([-1, -1], [-1, 1], shape=(g, g)), I)).tocsr()
        want_inertia = (g * g, 0, 0)
    else:
        n, m = 900, 300
        rng = np.random.default_rng(0)
        W = sp.diags([rng.uniform(0.5, 1.5, m), rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)], [0, 1, 2], shape=(m, n))
        A = sp.bmat([[sp.diags(rng.uniform(0.5, 2.0, n)), W.T], [W, -1e-6 * sp.identity(m)]]).tocsr()
        want_inertia = (n, m, 0)
    low = sp.tril(A).tocsr()
    low.sort_indices()
    K = pa.Csr(low.shape[0], low.shape[1], low.indptr, low.indices, low.data)
    s = pa.HipLdlSolver(K, n_primal=-1)
    s.matrixChanged()
    assert s.get_inertia() == want_inertia, (s.get_inertia(), s.info())
    rhs = np.random.default_rng(1).standard_normal(A.shape[0])
    x = rhs.copy()
    s.solve(x)
    assert np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs) < 1e-11
    info = s.info()
    assert info["n_levels"] >= 3 or info["m"] > 0


class _TimeCoupledProblem(Problem):
    """Same as Problem but W_i is banded (time-coupled constraints): chain-like elimination trees, hundreds of levels,
    wide amalgamated supernodes, head-to-head update segments."""

    def __init__(self, seed, N, n_i, my_i, n0, myl, bw):
        import scipy.sparse as sp
        super().__init__(seed, N, n_i, my_i, n0, myl, 0.02)
        rng = np.random.default_rng(seed)
        for b, blk in enumerate(self.blocks):
            rows, cols = [], []
            for r in range(my_i):
                center = int(r * n_i / my_i)
                cs = np.union1d(np.clip(center + rng.integers(-bw, bw + 1, 5), 0, n_i - 1), [center])
                rows += [r] * len(cs)
                cols += list(cs)
            W = sp.csr_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(my_i, n_i))
            W.sum_duplicates()
            W.sort_indices()
            Wp = pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)
            K, dpos = pa.kkt_leaf_assemble(n_i, Wp)
            K.val[dpos] = blk["diag"]
            blk.update(W=Wp, K=K, dpos=dpos)


@pytest.mark.parametrize("head", ["multifrontal", "multifrontal_devmem", "multifrontal_k_only", "scatter"])
@pytest.mark.parametrize("cut", ["model", "all_head"])
@pytest.mark.parametrize("n_i", [600, 3000], ids=["chain_and_spine", "dissected"])
def test_time_coupled_blocks_match_oracle(cut, n_i, head, monkeypatch):
    """n_i = 600, dissection switched off: the head is one chain per block (level-scheduled bottom, spine kernels on top);
    n_i = 3000: the dual-row separators of the nested dissection cut each block into independent segments.
    head: the multifrontal head kernels (fronts in LDS / update matrices in device memory / fronts on the rows of K only) or the scattering ones."""
    import torch
    if n_i == 600:
        monkeypatch.setenv("PIPS_HIP_ND_DEPTH", "0")     # keeps the chain / spine kernels under test
    if head == "scatter":
        monkeypatch.setenv("PIPS_HIP_MF", "0")
    if head == "multifrontal_k_only":
        # fronts on the rows of K only, their border rows formed afterwards in gather form (opt-in)
        monkeypatch.setenv("PIPS_HIP_MF_KONLY", "1")

    prob = _TimeCoupledProblem(5, 3, n_i, n_i // 2, 10, 8, 6)
    S, N = prob.S, prob.N

    def analysed():
        bt = pa.LeafBatch(N, S)
        for b in range(N):
            bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
        if cut == "all_head":
            bt.set_options(force_n_head=prob.n_leaf)
        bt.analyze(2)
        return bt

    bt = analysed()
    info = bt.info()
    if head == "multifrontal_devmem":
        # an LDS budget one double short of the largest packed front: its panel columns still fit, its update matrix is
        # worked on in device memory (the variant fronts beyond ~200 rows take at the default budget)
        # (with the border split a front keeps fewer update columns than its packed triangle has: shrink the budget until one no longer fits)
        budget = info["max_front"] * (info["max_front"] + 1) // 2 + 7
        for _ in range(8):
            bt.close()
            monkeypatch.setenv("PIPS_HIP_MF_LDS", str(budget))
            bt = analysed()
            info = bt.info()
            if info["fronts_in_device_memory"] > 0 or not info["multifrontal_head"]:
                break
            budget = int(budget * 0.75)
        assert info["fronts_in_device_memory"] > 0, info
    assert bt.schur_mode() in (1, 2)
    assert info["multifrontal_head"] == (0 if head == "scatter" else 1), info
    assert info["blocks_with_k_only_fronts"] == (N if head == "multifrontal_k_only" else 0), info
    if cut == "all_head" and n_i == 600:
        assert info["n_levels"] >= 10, info          # really chain-like
        assert info["n_sn"] < 0.9 * info["n_head"]    # amalgamation merged columns
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    want = np.tril(prob.oracle_schur())
    assert np.abs(got - want).max() / np.abs(want).max() < RTOL_SC, info
    rhs = np.random.default_rng(3).standard_normal(N * prob.n_leaf)
    x = rhs.copy()
    bt.solve(x)
    for b in range(N):
        assert bt.inertia(b) == (prob.n_i, prob.my_i, 0)
        xo = rhs.reshape(N, -1)[b].copy()
        prob.oracle_leaf(b).solve(xo)
        assert np.linalg.norm(x.reshape(N, -1)[b] - xo) / np.linalg.norm(xo) < RTOL_SOLVE
    if cut == "all_head" and n_i == 600 and head == "scatter":
        # the narrow top levels are handled by the per-block spine kernels: far fewer launches than tree levels
        bt.set_timing(True)
        SC.zero_()
        bt.factor(SC, S)
        bt.sync()
        assert bt.get_timing()["head"][1] < info["n_levels"] // 2, (bt.get_timing(), info)
        bt.set_timing(False)
        assert np.abs(hip_lower_as_rowmajor(SC.cpu().numpy(), S) - want).max() / np.abs(want).max() < RTOL_SC
    # multi-RHS through the same chain kernels (grid.y = right-hand side), drop-in handle of block 0
    s0 = pa.HipLdlSolver(prob.blocks[0]["K"], n_primal=prob.n_i)
    s0.matrixChanged()
    X = np.random.default_rng(4).standard_normal((5, prob.n_leaf))
    R = X.copy()
    s0.solve(X)
    for i in range(5):
        assert np.linalg.norm(prob.K_full(0) @ X[i] - R[i]) / np.linalg.norm(R[i]) < 1e-10


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 257])
def test_dense_root_tile_boundaries(n):
    """Sizes around the 128-wide tiles (identity padding, single-tile and two-tile paths) including the trivial ones."""
    rng = np.random.default_rng(n)
    n_primal = (n + 1) // 2
    m = n - n_primal
    H = rng.standard_normal((n_primal, n_primal))
    H = H @ H.T + n_primal * np.eye(n_primal)
    A = rng.standard_normal((m, n_primal))
    M = np.block([[H, A.T], [A, -np.eye(m)]])
    h = pa.HipDenseLdlSolver(n, n_primal)
    h.matrixChanged(np.tril(M))
    rhs = rng.standard_normal((3, n))
    x = rhs.copy()
    h.solve(x)
    for k in range(3):
        assert np.linalg.norm(M @ x[k] - rhs[k]) / np.linalg.norm(rhs[k]) < 1e-12
    assert h.get_inertia() == (n_primal, m, 0)


def test_degenerate_leaf_shapes():
    """Blocks without dual rows (K_i is just the primal diagonal) next to an ordinary one, a single-variable block, and a
    solve with zero right-hand sides."""
    import scipy.sparse as sp
    import torch
    prob = Problem(3, 1, 300, 150, 6, 6, 0.03)
    S = prob.S
    # block 1: only primal variables, coupled to the root through F (linking rows) alone
    n1 = 40
    rng = np.random.default_rng(0)
    K1 = pa.Csr(n1, n1, np.arange(n1 + 1, dtype=np.int32), np.arange(n1, dtype=np.int32), rng.uniform(0.5, 2.0, n1))
    F1 = sp.random(prob.myl, n1, density=0.2, random_state=1, format="csr")
    Bt1 = sp.vstack([sp.csr_matrix((prob.n0, n1)), F1]).tocsr()
    Bt1.sort_indices()
    Bt1p = pa.Csr(S, n1, Bt1.indptr, Bt1.indices, Bt1.data)
    # block 2: one variable, no border entries at all
    K2 = pa.Csr(1, 1, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([3.0]))
    bt = pa.LeafBatch(3, S)
    bt.set_block(0, prob.blocks[0]["K"], prob.n_i, prob.blocks[0]["Bt"])
    bt.set_block(1, K1, n1, Bt1p)
    bt.set_block(2, K2, 1)
    bt.analyze(2)
    bt.set_values(0, prob.blocks[0]["K"].val)
    bt.set_values(1, K1.val)
    bt.set_values(2, K2.val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    want = prob.oracle_schur([0])
    want -= (Bt1 @ sp.diags(1.0 / K1.val) @ Bt1.T).toarray()
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    assert np.abs(got - np.tril(want)).max() / np.abs(want).max() < RTOL_SC
    assert bt.inertia(0) == (prob.n_i, prob.my_i, 0) and bt.inertia(1) == (n1, 0, 0) and bt.inertia(2) == (1, 0, 0)
    ntot = prob.n_leaf + n1 + 1
    rhs = rng.standard_normal(ntot)
    x = rhs.copy()
    bt.solve(x)
    assert np.allclose(x[prob.n_leaf:prob.n_leaf + n1], rhs[prob.n_leaf:prob.n_leaf + n1] / K1.val, rtol=1e-13)
    assert np.isclose(x[-1], rhs[-1] / 3.0, rtol=1e-14)
    s = pa.HipLdlSolver(K2, n_primal=1)
    s.matrixChanged()
    empty = np.zeros((0, 1))
    s.solve(empty)          # zero right-hand sides: a no-op, not an error
    assert s.get_inertia() == (1, 0, 0)


@pytest.mark.parametrize("shape", ["random", "time_coupled"])
def test_leaf_handle_schur_term_matches_the_blocked_loop(shape):
    """pips_hip_ldl_set_border + pips_hip_ldl_factor_schur (INTEGRATION.md level 1.5): the leaf's Schur term formed on the device from
    the CSR border, against the oracle's restatement of the K4-K6 chunk loop (addTermToSchurComplBlocked,
    DistributedLeafLinearSystem.C:214-252) and against that loop run through the drop-in solve(nrhs, ...) of the same handle."""
    if shape == "random":
        prob = Problem(11, 2, 700, 350, 30, 20, 0.01)
    else:
        prob = _TimeCoupledProblem(5, 2, 900, 450, 10, 8, 6)
    S = prob.S
    for b in range(prob.N):
        blk = prob.blocks[b]
        s = pa.HipLdlSolver(blk["K"], n_primal=prob.n_i)
        s.set_border(blk["Bt"])
        got = np.zeros((S, S))
        s.matrixChanged_with_schur_term(got)
        want = orc.add_term_to_schur_compl_blocked(np.zeros((S, S)), prob.oracle_leaf(b), prob.Bt_scipy(b))
        scale = np.abs(want).max()
        assert np.abs(np.tril(got) - np.tril(want)).max() / scale < RTOL_SC
        assert np.abs(np.triu(got, 1)).max() == 0.0              # only the lower triangle is touched
        assert s.get_inertia() == (prob.n_i, prob.my_i, 0)
        # the handle still solves (the adapter's solve(Vector&) / solve(nrhs, ...) after matrixChanged)
        Bt = prob.Bt_scipy(b)
        loop = np.zeros((S, S))
        cols = np.nonzero(np.diff(Bt.indptr) > 0)[0]
        for k in range(0, len(cols), 20):
            ids = cols[k:k + 20]
            dense = np.ascontiguousarray(Bt[ids].toarray())
            s.solve(dense)
            loop[ids, :] -= (Bt @ dense.T).T
        assert np.abs(np.tril(loop) - np.tril(got)).max() / scale < RTOL_SC
        s.close()


def test_border_split_that_would_not_fit_the_lds_is_taken_off_at_analyze_time(monkeypatch):
    """Round-4 advisor finding: nb = 176 (the border split's cap) under fronts up to 32 wide whose below-rows are nearly all border rows
    makes k_border_schur's triangle + staged batch + row positions exceed the 160 KB of LDS; the launch used to refuse it in EVERY
    factor().  The analysis now evaluates the launch's own formula and falls back to whole update matrices: the input factorises, and
    matches the oracle."""
    import torch
    monkeypatch.setenv("PIPS_HIP_SN_WIDTH", "32")
    prob = _TimeCoupledProblem(9, 2, 3000, 1500, 100, 76, 6)
    S, N = prob.S, prob.N
    probe = pa.symbolic_probe(prob.blocks[0]["K"], prob.n_i, Bt=prob.blocks[0]["Bt"])
    assert probe["border_split"] == 1                 # block by block the symbolic phase would split (nb = 176 is within the cap)
    bt = pa.LeafBatch(N, S)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    info = bt.info()
    assert info["nb"] == 176 * N and info["multifrontal_head"] == 1 and info["blocks_with_border_split"] == 0
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    want = np.tril(prob.oracle_schur())
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-9
    bt.close()
    # the same blocks at the default supernode width keep the split
    monkeypatch.delenv("PIPS_HIP_SN_WIDTH")
    bt = pa.LeafBatch(N, S)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    assert bt.info()["blocks_with_border_split"] == N
    bt.close()


def test_host_pointer_solves_leading_dimension_and_row_pack():
    """pips_hip_ldl_solve with rows longer than the system (ld > n: what lies behind a row's first n entries is not touched), zero columns found
    on the device and moved out of the pass; pips_hip_ldl_solve_sparse with a colSparsity that marks few rows (the pack on the host, taken
    below an eighth of the rows) and with one that marks most (whole columns travel) - all against the plain dense call."""
    n_i = 900
    prob = Problem(11, 1, n_i, n_i // 2, 4, 4, 6.0 / n_i)
    n = prob.n_leaf
    s = pa.HipLdlSolver(prob.blocks[0]["K"], n_primal=prob.n_i)
    s.matrixChanged()
    rng = np.random.default_rng(2)
    R = rng.standard_normal((24, n))
    R[[2, 9, 23]] = 0.0
    want = R.copy()
    s.solve(want)
    assert not want[[2, 9, 23]].any()
    wide = np.full((24, n + 7), 7.5)
    wide[:, :n] = R
    s.solve(wide)
    assert np.abs(wide[:, :n] - want).max() <= 1e-12 * np.abs(want).max() and np.all(wide[:, n:] == 7.5)
    for frac in (1.0 / 16.0, 0.6):
        rows = np.sort(rng.choice(n, size=int(frac * n), replace=False))
        cs = np.zeros(n, np.int32)
        cs[rows] = 1
        Rs = np.zeros((24, n))
        Rs[:, rows] = rng.standard_normal((24, len(rows)))
        Rs[[0, 11]] = 0.0
        ref = Rs.copy()
        s.solve(ref)
        got = Rs.copy()
        s.solve_sparse(got, cs)
        assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()
        assert not got[[0, 11]].any()
    s.close()
