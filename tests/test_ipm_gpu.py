"""End-to-end harness IPM (a18) on the GPU vs an independent LP solver (HiGHS through scipy): the final objective must
agree to the tolerance implied by the termination rule of the reference (mu <= mutol, ||r|| <= artol * dnorm)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa

pytestmark = pytest.mark.gpu


def build_lp(seed, N, n_i, my_i, n0, myl, rho):
    blocks, cs, xs_all = [], [], []
    F0, c0, x0s = pa.gen_root(seed, n0, myl)
    cs.append(c0)
    xs_all.append(x0s)
    for b in range(1, N + 1):
        W, T, F, c, xs = pa.gen_block(seed, b, n_i, my_i, n0, myl, rho)
        blocks.append((W, T, F))
        cs.append(c)
        xs_all.append(xs)
    c = np.concatenate(cs)
    xstar = np.concatenate(xs_all)
    # global A: rows [link | blocks], cols [x0 | x1..xN]
    rows = [[F0.to_scipy()] + [F.to_scipy() for (_, _, F) in blocks]]
    for i, (W, T, F) in enumerate(blocks):
        r = [T.to_scipy()] + [None] * N
        r[1 + i] = W.to_scipy()
        rows.append(r)
    A = sp.bmat(rows, format="csr")
    b = A @ xstar
    return blocks, F0, c, b, A


@pytest.mark.parametrize("shape", [(3, 60, 30, 6, 5, 0.1), (4, 1000, 500, 100, 100, 0.01)])
def test_ipm_objective_matches_highs(shape):
    from scipy.optimize import linprog
    N, n_i, my_i, n0, myl, rho = shape
    blocks, F0, c, b, A = build_lp(2026, N, n_i, my_i, n0, myl, rho)
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    # tightened tolerances (the reference terminates at mu <= 1e-6, ||r|| <= 1e-4 dnorm): north_star asks for 1e-8 relative
    res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    assert res["status"] == 0, res
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert ref.status == 0
    assert abs(res["objective"] - ref.fun) / max(1.0, abs(ref.fun)) < 1e-9, (res, ref.fun)
    # same algorithm on the CPU (oracle/ipm_oracle.py, KKT systems solved by SuperLU): identical iteration count, same path
    if n_i > 100:
        return   # the SuperLU-based oracle needs minutes at this size; the small shape covers the path comparison
    from oracle import ipm_oracle as io
    trace = []
    o = io.solve_lp(A, b, c, 100, 1e-9, 1e-8, trace)
    assert o["status"] == 0 and abs(o["iterations"] - res["iterations"]) <= 1, (o["iterations"], res["iterations"])
    assert abs(o["objective"] - res["objective"]) / abs(o["objective"]) < 1e-9
    # iterate-by-iterate: mu, ||r||inf, primal / dual objective, centering parameter and both step lengths follow the CPU
    # restatement (same start point, same predictor-corrector, InteriorPointMethod.cpp:68-234).  The two codes solve the
    # KKT systems differently (Schur complement + LDL^T vs one SuperLU factorisation), so the paths agree to solver
    # accuracy while the iterates are well conditioned and drift apart by a few digits close to the optimum.
    tr = ipm.trace()
    n_cmp = min(len(trace), tr.shape[0]) - 1
    assert n_cmp >= 8
    for k in range(n_cmp):
        it, mu, rnorm, pobj, dobj, sigma, ap, ad = trace[k]
        tol = 1e-6 if k < n_cmp - 4 else 1e-3
        assert abs(tr[k, 0] - mu) <= tol * mu, (k, tr[k], trace[k])
        assert abs(tr[k, 2] - pobj) <= tol * max(1.0, abs(pobj)), (k, tr[k], trace[k])
        assert abs(tr[k, 3] - dobj) <= tol * max(1.0, abs(dobj)), (k, tr[k], trace[k])
        # residual norms are compared down to the level the linear solves resolve (1e-10 relative to the data norm)
        assert abs(tr[k, 1] - rnorm) <= tol * rnorm + 1e-10 * o["dnorm"], (k, tr[k], trace[k])
        assert abs(tr[k, 4] - sigma) <= 10 * tol and abs(tr[k, 5] - ap) <= 10 * tol and abs(tr[k, 6] - ad) <= 10 * tol, (k, tr[k], trace[k])
    st = ipm.stats()
    assert st["factorizations"] >= res["iterations"] + 1 and st["factorizations"] - st["regularised_repeats"] == res["iterations"] + 1
    assert st["solve_compressed"] >= 2 * res["iterations"]
    x, y = ipm.solution()
    assert x.min() > -1e-9
    assert np.linalg.norm(A @ x - b, np.inf) <= 1e-8 * max(1.0, np.abs(b).max())
    # duality: c^T x ~ b^T y at the optimum
    assert abs(res["objective"] - res["dual_objective"]) / max(1.0, abs(ref.fun)) < 1e-6


@pytest.mark.parametrize("case", range(12))
def test_ipm_sweep_against_highs(case):
    """Seeded family of small arrowhead LPs (1-4 blocks, 24-400 variables per block): the device harness must reach the
    optimum of HiGHS (tools/ipm_sweep.py ran 400 of these without a failure)."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(31000 + case)
    N = int(rng.integers(1, 5))
    n_i = int(rng.choice([24, 60, 150, 400]))
    my_i = int(n_i * rng.choice([0.3, 0.5]))
    n0, myl = int(rng.integers(2, 12)), int(rng.integers(1, 10))
    rho = max(4.0 / n_i, float(rng.choice([0.02, 0.1])))
    blocks, F0, c, b, A = build_lp(4000 + case, N, n_i, my_i, n0, myl, rho)
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    res = ipm.solve(max_iter=150, mutol=1e-9, artol=1e-8)
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert res["status"] == 0 and ref.status == 0, (res, ref.status)
    assert abs(res["objective"] - ref.fun) / max(1.0, abs(ref.fun)) < 1e-7, (res, ref.fun)


GOLDEN_CFG1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipm_config1_trace.npz")


@pytest.mark.gpu
def test_config1_trace_follows_the_oracle_capture():
    """BASELINE configs[0] (4 blocks x 1000 variables, Schur dimension 200): the device harness against the committed
    per-iteration capture of the CPU restatement - iteration count (+10 % allowed, t_pips.cpp:119), final objective to
    1e-9, and mu, both objectives, sigma and both step lengths of every iterate (1e-6 relative while the iterates are well
    conditioned, 1e-3 over the last four)."""
    g = np.load(GOLDEN_CFG1)
    N, n_i, my_i, n0, myl = (int(v) for v in g["shape"][:5])
    blocks, F0, c, b, A = build_lp(int(g["seed"]), N, n_i, my_i, n0, myl, float(g["shape"][5]))
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    res = ipm.solve(max_iter=100, mutol=float(g["mutol"]), artol=float(g["artol"]))
    assert res["status"] == 0, res
    want_it = int(g["iterations"])
    assert res["iterations"] <= int(np.ceil(1.1 * want_it)) and res["iterations"] >= want_it - 2
    assert abs(res["objective"] - float(g["objective"])) / abs(float(g["objective"])) < 1e-9
    tr, gt = ipm.trace(), g["trace"]
    n_cmp = min(tr.shape[0], gt.shape[0]) - 1
    assert n_cmp >= 15
    for k in range(n_cmp):
        tol = 1e-6 if k < n_cmp - 4 else 1e-3
        mu, rnorm, pobj, dobj, sigma, ap, ad = gt[k]
        assert abs(tr[k, 0] - mu) <= tol * mu, (k, tr[k], gt[k])
        assert abs(tr[k, 2] - pobj) <= tol * max(1.0, abs(pobj)) and abs(tr[k, 3] - dobj) <= tol * max(1.0, abs(dobj)), (k, tr[k], gt[k])
        assert abs(tr[k, 4] - sigma) <= 10 * tol and abs(tr[k, 5] - ap) <= 10 * tol and abs(tr[k, 6] - ad) <= 10 * tol, (k, tr[k], gt[k])
    x, _ = ipm.solution()
    assert np.abs(x[:n0] - g["x0"]).max() <= 1e-6 * max(1.0, np.abs(g["x0"]).max())


GOLDEN_CONFIGS1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipm_configs1.json")


@pytest.mark.gpu
def test_configs1_matches_the_cpu_pardiso_path():
    """BASELINE.json north_star on the headline configuration: "final objective + primal/dual residuals matching the CPU PARDISO path to
    1e-8 relative".  configs[1] (64 blocks x 10 000 variables, Schur dimension 2000: 641 000 variables, 321 000 constraints - the LP
    bench.py's ipm_end_to_end solves) against tests/golden/ipm_configs1.json: the CPU restatement of the interior-point loop with every KKT
    system solved by MKL PARDISO on the assembled global matrix (tests/golden/make_ipm_configs1.py; none of the product's Schur-complement
    code is involved).  Iteration count within the reference's own allowance (t_pips.cpp:115-119: + 10 %)."""
    import json
    g = json.load(open(GOLDEN_CONFIGS1))
    N, n_i, my_i, n0, myl = (int(v) for v in g["shape"][:5])
    assert g["status"] == 0 and g["variables"] == n0 + N * n_i == 641000 and g["constraints"] == myl + N * my_i
    blocks, F0, c, b, A = build_lp(int(g["seed"]), N, n_i, my_i, n0, myl, float(g["shape"][5]))
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    res = ipm.solve(max_iter=150, mutol=float(g["mutol"]), artol=float(g["artol"]))
    assert res["status"] == 0, res
    # final objective: 1e-8 relative
    assert abs(res["objective"] - g["objective"]) <= 1e-8 * abs(g["objective"]), (res["objective"], g["objective"])
    assert abs(res["dual_objective"] - g["dual_objective"]) <= 1e-7 * abs(g["dual_objective"])
    assert g["iterations"] - 3 <= res["iterations"] <= int(np.ceil(1.1 * g["iterations"])), (res["iterations"], g["iterations"])
    # primal / dual residuals: recomputed on the host from the device's solution (not the harness' own norms), relative to the data norm
    # both codes terminate against (artol * dnorm), and at the level the CPU path reached
    x, y = ipm.solution()
    dnorm = float(g["dnorm"])
    assert abs(res["dnorm"] - dnorm) <= 1e-12 * dnorm
    rp = np.abs(A @ x - b).max()
    slack = c - A.T @ y                       # = gamma at a dual-feasible point: must be non-negative up to the dual residual
    rd = max(0.0, -slack.min())
    tol = float(g["artol"]) * dnorm
    assert rp <= max(tol, 10 * g["primal_residual_inf"]) and rd <= max(tol, 10 * g["dual_residual_inf"]), (rp, rd, tol)
    assert res["rnorm"] <= tol and g["rnorm"] <= tol
    assert x.min() >= -1e-9
    # complementarity at the level of the termination rule (mu <= mutol on both sides)
    assert res["mu"] <= float(g["mutol"]) and g["mu"] <= float(g["mutol"])
    assert abs(x @ slack) / x.size <= 10 * float(g["mutol"])
    ipm.close()


def _infeasible_lp():
    """Two blocks whose linking row cannot hold: x >= 0, every block row forces sum(x_i) = 1, the linking row asks for
    sum over all x = -3."""
    n_i, my_i, n0, myl = 6, 1, 2, 1
    blocks = []
    for _ in range(2):
        W = pa.Csr(my_i, n_i, np.array([0, n_i]), np.arange(n_i), np.ones(n_i))
        T = pa.Csr(my_i, n0, np.array([0, 1]), np.array([0]), np.ones(1))
        F = pa.Csr(myl, n_i, np.array([0, n_i]), np.arange(n_i), np.ones(n_i))
        blocks.append((W, T, F))
    F0 = pa.Csr(myl, n0, np.array([0, n0]), np.arange(n0), np.ones(n0))
    c = np.ones(n0 + 2 * n_i)
    b = np.array([-3.0, 1.0, 1.0])
    rows = [[F0.to_scipy(), blocks[0][2].to_scipy(), blocks[1][2].to_scipy()],
            [blocks[0][1].to_scipy(), blocks[0][0].to_scipy(), None], [blocks[1][1].to_scipy(), None, blocks[1][0].to_scipy()]]
    return n0, myl, blocks, F0, c, b, sp.bmat(rows, format="csr")


def test_infeasible_lp_is_reported():
    """The reference's infeasibility test (PIPSIPMppSolver.cpp:128-170: phi = (||r|| + |gap|) / dnorm ten iterations in and 1e4
    above its best value) on the device and in the CPU restatement: status 4 from both, HiGHS agrees that the LP is infeasible."""
    from scipy.optimize import linprog
    from oracle import ipm_oracle as io
    n0, myl, blocks, F0, c, b, A = _infeasible_lp()
    assert linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs").status == 2
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b, dual_reg=1e-9)
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
    assert res["status"] == 4, res
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o = io.solve_lp(A, b, c, 200, 1e-8, 1e-8, dual_reg=1e-9)
    assert o["status"] == 4
