"""The dense root with the static pivot order as ONE dependency-driven launch (csrc/rootkernel.hip.h, csrc/rootplan.cpp) against LAPACK
dsytrf / dsytrs - the routine DeSymIndefSolver hands the Schur complement to (DeSymIndefSolver.C:56-129) - and against the
launch-per-step factorisation it replaces (PIPS_HIP_ROOT_LAUNCHES=1)."""
import numpy as np
import pytest
import scipy.linalg as sla

import pips_ipmpp_amd as pa

pytestmark = pytest.mark.gpu


def quasi_definite(n, n_primal, seed, spread=0.0):
    """[H A^T; A -G] with H, G symmetric positive definite: every leading block is nonsingular, the inertia is (n_primal, n - n_primal, 0)"""
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)) * 0.5
    M = M + M.T
    d = np.full(n, float(n))
    if spread:
        d *= 10.0 ** rng.uniform(-spread, spread, n)
    d[n_primal:] *= -1
    M[np.arange(n), np.arange(n)] = d
    # make the diagonal blocks definite whatever the spread did
    M[:n_primal, :n_primal] += np.diag(np.abs(M[:n_primal, :n_primal]).sum(axis=1))
    M[n_primal:, n_primal:] -= np.diag(np.abs(M[n_primal:, n_primal:]).sum(axis=1))
    return M


def factor_solve(M, n_primal, B):
    s = pa.HipDenseLdlSolver(M.shape[0], n_primal=n_primal)
    s.matrixChanged(np.ascontiguousarray(np.tril(M)))
    X = np.ascontiguousarray(B.T.copy())
    s.solve(X)
    inertia = s.get_inertia()
    s.close()
    return X.T, inertia


@pytest.mark.parametrize("n", [1, 37, 128, 129, 300, 1000, 2049])
def test_single_launch_matches_lapack_and_the_launch_per_step_path(n, monkeypatch):
    n_primal = (2 * n) // 3
    M = quasi_definite(n, n_primal, seed=n)
    B = np.random.default_rng(1).standard_normal((n, 3))
    lu, d, perm = sla.ldl(M, lower=True)
    want = np.linalg.solve(M, B)
    monkeypatch.delenv("PIPS_HIP_ROOT_LAUNCHES", raising=False)
    x1, in1 = factor_solve(M, n_primal, B)
    monkeypatch.setenv("PIPS_HIP_ROOT_LAUNCHES", "1")
    x2, in2 = factor_solve(M, n_primal, B)
    assert in1 == in2 == (n_primal, n - n_primal, 0)
    scale = np.abs(want).max()
    assert np.abs(x1 - want).max() <= 1e-10 * scale
    assert np.abs(x2 - want).max() <= 1e-10 * scale
    assert np.abs(M @ x1 - B).max() <= 1e-11 * np.abs(B).max() * max(1.0, np.linalg.cond(M) * 1e-3)


@pytest.mark.parametrize("variant", ["default", "one_list", "barrier_diagonal"])
def test_variants_of_the_launch_agree_bit_for_bit_where_the_arithmetic_is_the_same(variant, monkeypatch):
    """The lists decide where the K ranges of a tile are cut - the sums associate differently between schedules - but one schedule gives the
    same bits run after run and handle after handle (nothing in the launch depends on the order workgroups arrive in)."""
    env = {"default": {}, "one_list": {"PIPS_HIP_ROOT_CHAIN_CU": "0"}, "barrier_diagonal": {"PIPS_HIP_ROOT_DIAG_BARRIERS": "1"}}[variant]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, n_primal = 1500, 900
    M = quasi_definite(n, n_primal, seed=5, spread=3.0)
    B = np.random.default_rng(2).standard_normal((n, 2))
    want = np.linalg.solve(M, B)
    runs = [factor_solve(M, n_primal, B) for _ in range(3)]
    for x, inertia in runs:
        assert inertia == (n_primal, n - n_primal, 0)
        assert np.abs(x - want).max() <= 1e-9 * np.abs(want).max()
    assert all(np.array_equal(runs[0][0], r[0]) for r in runs[1:])
    s = pa.HipDenseLdlSolver(n, n_primal=n_primal)     # one handle, factorised three times
    outs = []
    for _ in range(3):
        s.matrixChanged(np.ascontiguousarray(np.tril(M)))
        X = np.ascontiguousarray(B.T.copy())
        s.solve(X)
        outs.append(X.copy())
    s.close()
    assert all(np.array_equal(outs[0], o) for o in outs[1:]) and np.array_equal(outs[0].T, runs[0][0])


def test_a_pivot_the_rule_rejects_sends_the_tile_to_the_careful_kernel(monkeypatch):
    """The blocked diagonal tile takes its pivots as they come and applies the pivot rule afterwards; a rejected pivot (here: exact zeros
    on the diagonal of the dual block with nothing coupling them) makes the tile start over in the kernel that replaces pivots one at a
    time - perturbed pivots are counted as `zero`, like the launch-per-step path counts them."""
    n, n_primal = 700, 400
    M = quasi_definite(n, n_primal, seed=9)
    dead = [450, 451, 600]
    for i in dead:
        M[i, :] = 0.0
        M[:, i] = 0.0
    res = {}
    for mode in ("single", "launches"):
        if mode == "launches":
            monkeypatch.setenv("PIPS_HIP_ROOT_LAUNCHES", "1")
        s = pa.HipDenseLdlSolver(n, n_primal=n_primal)
        s.matrixChanged(np.ascontiguousarray(np.tril(M)))
        res[mode] = s.get_inertia()
        x = np.ones(n)
        s.solve(x)
        assert np.isfinite(x).all()
        s.close()
    assert res["single"] == res["launches"]
    assert res["single"][2] == len(dead) and res["single"][0] == n_primal


def test_a_wait_that_gives_up_is_reported(monkeypatch):
    """Every wait inside the launch is bounded; with a poll limit of zero the first task that has to wait raises the error word, and the
    host hears of it at its next synchronisation point with the handle instead of getting factors of garbage with PIPS_OK."""
    monkeypatch.setenv("PIPS_HIP_ROOT_POLL_LIMIT", "0")
    n = 1500
    M = quasi_definite(n, 1000, seed=3)
    s = pa.HipDenseLdlSolver(n, n_primal=1000)
    with pytest.raises(pa.PipsHipError, match="gave up waiting"):
        for _ in range(20):                    # (a lucky schedule can satisfy every wait at once)
            s.matrixChanged(np.ascontiguousarray(np.tril(M)))
    s.close()
    monkeypatch.delenv("PIPS_HIP_ROOT_POLL_LIMIT")
    x, inertia = factor_solve(M, 1000, np.ones((n, 1)))
    assert inertia == (1000, 500, 0)
    assert np.abs(M @ x[:, 0] - 1.0).max() < 1e-9


@pytest.mark.parametrize("seed", range(8))
def test_random_sizes_and_splits(seed):
    """Dimensions that are no multiple of the tile or the sub-block, inertia hints anywhere from all-negative to all-positive, diagonals over
    six decades: solution against numpy, inertia exact."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 1400))
    n_primal = int(rng.integers(0, n + 1))
    M = quasi_definite(n, n_primal, seed=seed, spread=3.0)
    B = rng.standard_normal((n, 2))
    x, inertia = factor_solve(M, n_primal, B)
    want = np.linalg.solve(M, B)
    assert inertia == (n_primal, n - n_primal, 0)
    assert np.abs(x - want).max() <= 1e-9 * np.abs(want).max()

