"""Dense root with Bunch-Kaufman pivoting inside the diagonal tiles (k_tile_diag_bk) against LAPACK dsytrf / dsytrs, the routine
behind the reference's DeSymIndefSolver (DeSymIndefSolver.C:56-168): solutions, inertia (2 x 2 blocks counted as :135-160 does:
one positive and one negative eigenvalue), zero perturbed pivots - on matrices a static pivot order cannot take."""
import numpy as np
import pytest
import scipy.linalg as sla

import pips_ipmpp_amd as pa

pytestmark = pytest.mark.gpu


def lapack_solve_and_inertia(M, B):
    """dsytrf('L') + dsytrs; inertia from the 1 x 1 / 2 x 2 blocks of D exactly as DeSymIndefSolver::get_inertia reads ipiv."""
    ldu, ipiv, info = sla.lapack.dsytrf(M, lower=1)
    assert info == 0
    X, info = sla.lapack.dsytrs(ldu, ipiv, B, lower=1)
    assert info == 0
    n, pos, neg, k = M.shape[0], 0, 0, 0
    while k < n:
        if ipiv[k] > 0:
            pos += ldu[k, k] > 0
            neg += ldu[k, k] < 0
            k += 1
        else:
            ev = np.linalg.eigvalsh(np.array([[ldu[k, k], ldu[k + 1, k]], [ldu[k + 1, k], ldu[k + 1, k + 1]]]))
            pos += int((ev > 0).sum())
            neg += int((ev < 0).sum())
            k += 2
    return X, (int(pos), int(neg))


def hip_solve(M, B, n_primal=-1, pivoting=None):
    s = pa.HipDenseLdlSolver(M.shape[0], n_primal=n_primal)
    if pivoting is not None:
        s.set_pivoting(pivoting)
    s.matrixChanged(np.ascontiguousarray(np.tril(M)))     # row-major, lower triangle authoritative (DenseStorage.C:64-83)
    X = np.ascontiguousarray(B.T.copy())
    s.solve(X)
    inertia = s.get_inertia()
    s.close()
    return X.T, inertia


@pytest.mark.parametrize("n", [60, 128, 300, 1000])
def test_indefinite_with_zero_diagonal_needs_two_by_two_pivots(n):
    """Symmetric, dense, zero diagonal: no 1 x 1 pivot exists at the start of any tile."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    M = M + M.T
    np.fill_diagonal(M, 0.0)
    B = rng.standard_normal((n, 3))
    Xl, inl = lapack_solve_and_inertia(M, B)
    Xh, inh = hip_solve(M, B)
    ev = np.linalg.eigvalsh(M)
    assert inl == (int((ev > 0).sum()), int((ev < 0).sum()))
    assert inh == (inl[0], inl[1], 0), (inh, inl)
    assert np.linalg.norm(M @ Xh - B) / np.linalg.norm(B) < 1e-11
    assert np.linalg.norm(Xh - Xl) / np.linalg.norm(Xl) < 1e-8


@pytest.mark.parametrize("n0,m", [(100, 40), (260, 150)])
def test_root_with_eliminated_inequality_rows(n0, m):
    """The Schur complement the reference hands DeSymIndefSolver late in a run: the x0 block carries C0^T Omega^-1 C0 with
    Omega^-1 over fourteen decades (sLinsysRootAug.C:1276-1294; an active inequality row has Omega^-1 ~ 1e14).  Static pivots
    cancel to O(1) from 1e14 and are taken for zeros; Bunch-Kaufman factorises without a perturbed pivot, with LAPACK's inertia.
    The condition number is ~1e14: the two solutions are compared where that allows it - through the residual, and directly on a
    copy of the system whose Omega^-1 stops at 1e6."""
    rng = np.random.default_rng(n0)
    for top, direct in ((14, False), (6, True)):
        mz = n0 // 2
        C0 = rng.standard_normal((mz, n0)) * (rng.random((mz, n0)) < 0.2)
        om = 10.0 ** rng.uniform(0, top, mz)
        X = np.diag(10.0 ** rng.uniform(-2, 2, n0)) + C0.T @ (om[:, None] * C0)
        A = rng.standard_normal((m, n0)) * (rng.random((m, n0)) < 0.3)
        A[np.arange(m), rng.permutation(n0)[:m]] += 2.0        # full row rank
        M = np.block([[X, A.T], [A, -1e-8 * np.eye(m)]])
        B = rng.standard_normal((n0 + m, 2))
        Xl, inl = lapack_solve_and_inertia(M, B)
        Xh, inh = hip_solve(M, B)                               # no hint: pivots like dsytrf
        assert inl == (n0, m)
        assert inh == (n0, m, 0), inh
        scale = np.abs(M).max() * np.abs(Xh).max() + np.abs(B).max()
        assert np.abs(M @ Xh - B).max() / scale < 1e-13          # normwise backward error
        if direct:
            assert np.linalg.norm(Xh - Xl) / np.linalg.norm(Xl) < 1e-8
        else:
            # the static order on the same matrix: pivots lost to cancellation are reported as perturbed
            _, ins = hip_solve(M, B, n_primal=n0, pivoting=0)
            assert ins[2] > 0, ins


def test_static_order_still_default_with_hint():
    """With an inertia hint the handle keeps the static order (bit-identical to round 2 on quasi-definite input)."""
    rng = np.random.default_rng(3)
    n0, m = 150, 90
    A = rng.standard_normal((m, n0))
    M = np.block([[np.diag(rng.uniform(1, 2, n0)), A.T], [A, -1e-6 * np.eye(m)]])
    B = rng.standard_normal((n0 + m, 1))
    Xs, ins = hip_solve(M, B, n_primal=n0)
    Xb, inb = hip_solve(M, B, n_primal=n0, pivoting=1)
    assert ins == inb == (n0, m, 0)
    assert np.linalg.norm(Xs - Xb) / np.linalg.norm(Xs) < 1e-10
    assert np.linalg.norm(M @ Xb - B) / np.linalg.norm(B) < 1e-11


@pytest.mark.parametrize("case", ["zero_leading_tile", "zero_leading_two_tiles", "rank_deficient_leading_tile"])
def test_pivots_beyond_the_diagonal_tile(case):
    """dsytrf searches the whole column (DeSymIndefSolver.C:78); k_tile_diag_bk its 128 x 128 tile.  A leading tile that is singular on
    its own while the matrix is regular ([[0 A^T]; [A 0]]: every pivot of the zero block needs a row of A) is what the tile search cannot
    take: the indices it finds no pivot for are paired with the row of their column's largest entry below the tile, moved next to it
    (a symmetric permutation kept for later factorisations) and the matrix is factorised again - solution, inertia and zero perturbed
    pivots as LAPACK."""
    rng = np.random.default_rng(7)
    if case == "zero_leading_tile":
        n0, m = 128, 160
    elif case == "zero_leading_two_tiles":
        n0, m = 256, 300
    else:
        n0, m = 128, 200
    A = rng.standard_normal((m, n0)) * (rng.random((m, n0)) < 0.2)
    A[rng.permutation(m)[:n0], np.arange(n0)] += 3.0            # full column rank
    H = np.zeros((n0, n0))
    if case == "rank_deficient_leading_tile":
        V = rng.standard_normal((n0, 40))
        H = V @ V.T                                              # rank 40 of 128
    C = rng.standard_normal((m, m)) * 0.1
    M = np.block([[H, A.T], [A, -(C @ C.T) - 1e-3 * np.eye(m)]])
    B = rng.standard_normal((n0 + m, 3))
    Xl, inl = lapack_solve_and_inertia(M, B)
    s = pa.HipDenseLdlSolver(M.shape[0], n_primal=-1)
    for rep in range(2):                                         # the second factorisation starts from the order the first one found
        s.matrixChanged(np.ascontiguousarray(np.tril(M)))
        X = np.ascontiguousarray(B.T.copy())
        s.solve(X)
        inh = s.get_inertia()
        assert inh == (inl[0], inl[1], 0), (rep, inh, inl)
        assert np.linalg.norm(M @ X.T - B) / np.linalg.norm(B) < 1e-10
        assert np.linalg.norm(X.T - Xl) / np.linalg.norm(Xl) < 1e-8
    s.close()
