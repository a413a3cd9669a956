"""ipm_oracle.py — TEST INFRASTRUCTURE ONLY.

numpy/scipy restatement of the IPM driver logic the product's host harness (pips-ipmpp_amd/csrc/harness.hip) mirrors, for the
generator's problem class  min c^T x, A x = b, x >= 0  (ixlow = 1 everywhere, no upper bounds, no inequality rows):
  start point        PIPSIPMppSolver::solve (PIPSIPMppSolver.cpp:36-42), Solver::solve_linear_system (Solver.cpp:19-31)
  residuals          Residuals::evaluate (Residuals.cpp:58-171): rQ = c - A^T y - gamma, rA = A x - b, rv = x - v
  linear system      LinearSystem::computeDiagonals/solve/solveXYZS (LinearSystem.C:262-294,327-447,449-548)
  predictor/corrector InteriorPointMethod.cpp:68-90,178-234: sigma = (mu_aff/mu)^3, corrector blended in with the weight
                     of the 10-point search (:486-523), Gondzio correctors (:236-358)
  step length        PrimalDualInteriorPointMethod::mehrotra_step_length (InteriorPointMethod.cpp:745-812)
  termination        PIPSIPMppSolver.cpp:143-149: mu <= mutol and ||r||inf <= artol * dnorm
The KKT system [dd A^T; A 0] is solved with SuperLU here (the arithmetic under test lives in the HIP path).
Not reproduced (neither here nor in the harness): the filter line search and the small-corrector heuristics.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl


def stepbound(v, dv):
    neg = dv < 0
    return np.min(-v[neg] / dv[neg]) if neg.any() else np.inf


def find_blocking(v, dv, g, dg):
    """(ratio, v_b, dv_b, g_b, dg_b) of the first entry attaining the step bound (DenseVector::find_blocking)."""
    neg = dv < 0
    if not neg.any():
        return np.inf, 0.0, 0.0, 0.0, 0.0
    r = np.where(neg, -v / np.where(neg, dv, -1.0), np.inf)
    i = int(np.argmin(r))
    return r[i], v[i], dv[i], g[i], dg[i]


def weight_search(v, dv, cv, g, dg, cg, apt, adt):
    wmin = apt * adt
    ape = ade = wp = wd = -1.0
    for k in range(11):
        w = min(1.0, wmin + (1.0 - wmin) / 10.0 * k)
        a1 = min(1.0, stepbound(v, dv + w * cv))
        a2 = min(1.0, stepbound(g, dg + w * cg))
        if a1 > ape:
            ape, wp = a1, w
        if a2 > ade:
            ade, wd = a2, w
    return ape, ade, wp, wd


def mehrotra_step_length(v, dv, g, dg, n_pairs=None):
    gamma_f, factor = 0.99, 0.99999999
    gamma_a = 1.0 / (1.0 - gamma_f)
    nx = len(v) if n_pairs is None else n_pairs
    pb, db = find_blocking(v, dv, g, dg), find_blocking(g, dg, v, dv)
    amax_p, amax_d = min(1.0, pb[0]), min(1.0, db[0])
    mufull = (v + amax_p * dv) @ (g + amax_d * dg) / nx / gamma_a
    a_p = a_d = 1.0
    if pb[0] < 1.0:
        est = pb[3] + amax_d * pb[4]
        a_p = 0.0 if est == 0.0 else (-pb[1] + mufull / est) / pb[2]
    if db[0] < 1.0:
        est = db[3] + amax_p * db[4]
        a_d = 0.0 if est == 0.0 else (-db[1] + mufull / est) / db[2]
    a_p = max(min(a_p, amax_p), gamma_f * amax_p) * factor
    a_d = max(min(a_d, amax_d), gamma_f * amax_d) * factor
    return a_p, a_d


def solve_lp(A, b, c, max_iter=100, mutol=1e-6, artol=1e-4, trace=None, dual_reg=0.0, gondzio=2, bounded=None, free_diag=0.0):
    """bounded: optional 0/1 mask, 0 = free variable (the reference's ixlow = ixupp = 0: no complementarity pair, dd_j = 0,
    LinearSystem.C:262-294); free entries carry the constant pair v = 1, gamma = 0 and masked rv, rgamma, dv, dgamma.
    free_diag: primal regularisation on the free entries of the KKT matrix (a free column without any coefficient makes the
    unregularised matrix exactly singular for SuperLU)."""
    A = sp.csr_matrix(A)
    ny, nx = A.shape
    fm = np.ones(nx) if bounded is None else np.asarray(bounded, dtype=float)
    free = fm == 0.0
    n_pairs = int(fm.sum())
    dnorm = max(np.abs(A.data).max(), np.abs(b).max(), np.abs(c).max())
    s0 = np.sqrt(dnorm)
    x, y = np.zeros(nx), np.zeros(ny)
    v, g = np.full(nx, s0), np.full(nx, s0)
    v[free], g[free] = 1.0, 0.0

    def residuals():
        return c - A.T @ y - g, A @ x - b, (x - v) * fm

    def solve(rQ, rA, rv, rg):
        dd = g / v
        rg = rg * fm
        rx = rQ + dd * rv + rg / v
        K = sp.bmat([[sp.diags(dd + free_diag * (1.0 - fm)), A.T], [A, -dual_reg * sp.identity(ny) if dual_reg else None]], format="csc")
        sol = spl.splu(K).solve(np.concatenate([rx, rA]))
        dx, dyp = sol[:nx], sol[nx:]
        dy = -dyp
        dv = (dx - rv) * fm
        dg = (rg - g * dv) / v * fm
        return -dx, -dy, -dv, -dg

    rQ, rA, rv = residuals()
    dx, dy, dv, dg = solve(rQ, rA, rv, v * g)
    x += dx; y += dy; v += dv; g += dg
    viol = max(0.0, -v.min(), -g.min())
    v += 1e3 + 2 * viol
    g += 1e3 + 2 * viol
    v[free], g[free] = 1.0, 0.0
    status, it = 1, 0
    for it in range(max_iter):
        rQ, rA, rv = residuals()
        rnorm = max(np.abs(rQ).max(), np.abs(rA).max(), np.abs(rv).max())
        mu = v @ g / n_pairs
        if trace is not None:
            trace.append((it, mu, rnorm, c @ x, b @ y))
        if mu <= mutol and rnorm <= artol * dnorm:
            status = 0
            break
        # "probably infeasible" (PIPSIPMppSolver.cpp:128-170)
        phi = (rnorm + abs(c @ x - b @ y)) / dnorm
        phi_min = phi if it == 0 else min(phi_min, phi)
        if it >= 10 and phi >= 1e-8 and phi >= 1e4 * phi_min:
            status = 4
            break
        dx, dy, dv, dg = solve(rQ, rA, rv, v * g)
        ap, ad = min(1.0, stepbound(v, dv)), min(1.0, stepbound(g, dg))
        mu_aff = (v + ap * dv) @ (g + ad * dg) / n_pairs
        sigma = (mu_aff / mu) ** 3
        z = np.zeros(nx)
        cx, cy, cv, cg = solve(z, np.zeros(ny), z, dv * dg - sigma * mu)
        ap, ad, wp, wd = weight_search(v, dv, cv, g, dg, cg, ap, ad)
        dx += wp * cx; dv += wp * cv
        dy += wd * cy; dg += wd * cg
        # Gondzio multiple centrality correctors (gondzio_correction_loop, InteriorPointMethod.cpp:236-358, primal-dual
        # variant; projection: DenseVector.cpp:405-420; weight search: InteriorPointMethod.cpp:486-523)
        rmin, rmax = sigma * mu * 0.1, sigma * mu * 10.0
        ng = 0
        while ng < gondzio and (ap < 1.0 or ad < 1.0):
            apt, adt = min(1.0, 1.5 * ap + 0.3), min(1.0, 1.5 * ad + 0.3)
            p = (v + apt * dv) * (g + adt * dg)
            t = np.where(p < rmin, rmin - p, np.where(p > rmax, rmax - p, 0.0))
            t = np.maximum(t, -rmax)
            cx, cy, cv, cg = solve(z, np.zeros(ny), z, -t)
            ape, ade, wp, wd = weight_search(v, dv, cv, g, dg, cg, apt, adt)
            both_one = ape >= 1.0 and ade >= 1.0
            p_better, d_better = ape >= 1.01 * ap, ade >= 1.01 * ad
            if not (both_one or p_better or d_better):
                break
            if both_one or p_better:
                dx += wp * cx; dv += wp * cv; ap = ape
            if both_one or d_better:
                dy += wd * cy; dg += wd * cg; ad = ade
            ng += 1
            if both_one:
                break
        ap, ad = mehrotra_step_length(v, dv, g, dg, n_pairs)
        x += ap * dx; v += ap * dv
        y += ad * dy; g += ad * dg
        if trace is not None:   # the step that leaves iterate `it`: (sigma, alpha_primal, alpha_dual) appended to its row
            trace[-1] = trace[-1] + (sigma, ap, ad)
    return dict(objective=c @ x, iterations=it, mu=mu, rnorm=rnorm, status=status, dual_objective=b @ y, x=x, y=y, dnorm=dnorm)
