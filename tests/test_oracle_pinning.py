"""Pins the oracle (the checker) before it is trusted.

The reference has no unit-level golden vectors for this path and its third-party leaf solvers are
absent, so the pins are: SuperLU and dense LAPACK (independent solvers), dense eigenvalue inertia, the reference's own
root routine dsytrf/dsytrs, MKL PARDISO driven with the reference's iparm settings where libmkl_rt exists, and the
end-to-end identity "solveCompressed solves the full arrowhead system"."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import oracle as orc
from oracle import pardiso_mkl as pm
from tests.util import Problem


@pytest.mark.parametrize("n_i,rho,lo,hi", [(50, 0.1, -2, 2), (400, 0.02, -4, 4), (400, 0.02, -8, 8)])
def test_leaf_ldl_against_superlu_and_eigen_inertia(n_i, rho, lo, hi):
    prob = Problem(5, 1, n_i, n_i // 2, 4, 4, rho, diag_lo=lo, diag_hi=hi)
    s = prob.oracle_leaf(0, refine_steps=2)
    Kf = prob.K_full(0)
    rhs = np.random.default_rng(0).standard_normal(Kf.shape[0])
    x = rhs.copy()
    s.solve(x)
    xr = spl.splu(Kf).solve(rhs)
    assert np.linalg.norm(Kf @ x - rhs) / np.linalg.norm(rhs) < 1e-11
    if hi <= 4:
        assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-7
    ev = np.linalg.eigvalsh(Kf.toarray())
    assert s.get_inertia() == (int((ev > 0).sum()), int((ev < 0).sum()), 0)


@pytest.mark.skipif(not pm.available(), reason="libmkl_rt.so not present")
def test_leaf_ldl_against_mkl_pardiso_with_reference_iparm():
    prob = Problem(6, 1, 800, 400, 8, 8, 0.01)
    K = prob.K_scipy(0)
    ref = pm.MklPardisoSolver(K)
    ref.matrixChanged()
    s = prob.oracle_leaf(0)
    R = np.random.default_rng(1).standard_normal((4, K.shape[0]))
    X1, X2 = R.copy(), R.copy()
    ref.solve(X1)
    s.solve(X2)
    assert np.abs(X1 - X2).max() / np.abs(X1).max() < 1e-9
    assert ref.get_inertia()[:2] == s.get_inertia()[:2] == (800, 400)


def test_schur_accumulation_against_dense_inverse():
    prob = Problem(7, 2, 200, 100, 12, 10, 0.04)
    SC = prob.oracle_schur()
    want = np.zeros_like(SC)
    for b in range(prob.N):
        Bt = prob.Bt_scipy(b).toarray()
        want -= Bt @ np.linalg.solve(prob.K_full(b).toarray(), Bt.T)
    assert np.abs(SC - want).max() / np.abs(want).max() < 1e-9
    assert np.abs(SC - SC.T).max() / np.abs(SC).max() < 1e-10


def test_dense_root_is_lapack_dsytrf():
    rng = np.random.default_rng(2)
    n, p = 60, 35
    H = rng.standard_normal((p, p)); H = H @ H.T + p * np.eye(p)
    A = rng.standard_normal((n - p, p))
    M = np.block([[H, A.T], [A, -1e-3 * np.eye(n - p)]])
    r = orc.DenseRootSolver(n)
    r.matrixChanged(np.tril(M))
    b = rng.standard_normal(n)
    x = b.copy()
    r.solve(x)
    assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1e-12
    assert r.get_inertia()[:2] == (p, n - p)


def test_solve_compressed_solves_the_full_arrowhead_system():
    prob = Problem(8, 3, 150, 75, 10, 8, 0.05)
    S, N = prob.S, prob.N
    leaf = [prob.oracle_leaf(b) for b in range(N)]
    SC = prob.oracle_finalize(prob.oracle_schur())
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SC))
    rng = np.random.default_rng(3)
    b0 = rng.standard_normal(S)
    bs = [rng.standard_normal(prob.n_leaf) for _ in range(N)]
    x0, xs = b0.copy(), [v.copy() for v in bs]
    orc.solve_compressed(x0, xs, leaf, [prob.Bt_scipy(b) for b in range(N)], root, prob.n0, 0, 0, prob.myl, 0)
    # assemble the full KKT matrix  [diag(K_i)  Br ; Br^T  K0]
    K0l = prob.oracle_finalize(np.zeros((S, S)))
    K0 = np.tril(K0l) + np.tril(K0l, -1).T
    rows = [[None] * (N + 1) for _ in range(N + 1)]
    for b in range(N):
        rows[b][b] = prob.K_full(b)
        rows[b][N] = prob.Bt_scipy(b).T
        rows[N][b] = prob.Bt_scipy(b)
    rows[N][N] = sp.csr_matrix(K0)
    Kbig = sp.bmat(rows, format="csc")
    xfull = spl.splu(Kbig).solve(np.concatenate(bs + [b0]))
    got = np.concatenate(xs + [x0])
    assert np.linalg.norm(got - xfull) / np.linalg.norm(xfull) < 1e-8


def test_ipm_oracle_against_highs():
    """The IPM restatement (oracle/ipm_oracle.py) reaches the optimum an independent LP solver finds."""
    from scipy.optimize import linprog
    from oracle import ipm_oracle as io
    import pips_ipmpp_amd as pa
    import scipy.sparse as sp
    N, n_i, my_i, n0, myl, rho = 3, 60, 30, 6, 5, 0.1
    F0, c0, x0s = pa.gen_root(2026, n0, myl)
    blocks, cs, xs = [], [c0], [x0s]
    for b in range(1, N + 1):
        W, T, F, c, x = pa.gen_block(2026, b, n_i, my_i, n0, myl, rho)
        blocks.append((W, T, F)); cs.append(c); xs.append(x)
    rows = [[F0.to_scipy()] + [F.to_scipy() for (_, _, F) in blocks]]
    for i, (W, T, F) in enumerate(blocks):
        r = [T.to_scipy()] + [None] * N
        r[1 + i] = W.to_scipy()
        rows.append(r)
    A = sp.bmat(rows, format="csr")
    c = np.concatenate(cs)
    b = A @ np.concatenate(xs)
    o = io.solve_lp(A, b, c, 100, 1e-9, 1e-9)
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert o["status"] == 0 and o["iterations"] <= 30
    assert abs(o["objective"] - ref.fun) / abs(ref.fun) < 1e-9
