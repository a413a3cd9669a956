"""ipm_oracle.py — TEST INFRASTRUCTURE ONLY.

numpy/scipy restatement of the IPM driver logic the product's host harness (pips-ipmpp_amd/csrc/harness.hip) mirrors, for the
reference's full problem class
     min c^T x   s.t.  A x = b,  clow <= C x <= cupp,  xlow <= x <= xupp     (every bound optional: indicator vectors)
  start point        PIPSIPMppSolver::solve (PIPSIPMppSolver.cpp:36-42), Solver::solve_linear_system (Solver.cpp:19-31),
                     Variables::push_to_interior / violation / shift_bound_variables (Variables.C:310-403)
  residuals          Residuals::evaluate (Residuals.cpp:58-171): rQ = c - A^T y - C^T z - gamma + phi, rA = A x - b, rC = C x - s,
                     rz = z - lambda + pi, rt = s - clow - t, ru = s - cupp + u, rv = x - xlow - v, rw = x - xupp + w
  linear system      LinearSystem::computeDiagonals / solve / solveXYZS (LinearSystem.C:262-294,327-548)
  predictor/corrector InteriorPointMethod.cpp:68-90,178-234: sigma = (mu_aff/mu)^3, corrector blended in with the weight
                     of the 10-point search (:486-523), Gondzio correctors (:236-358)
  step length        PrimalDualInteriorPointMethod::mehrotra_step_length (InteriorPointMethod.cpp:745-812)
  termination        PIPSIPMppSolver.cpp:143-149: mu <= mutol and ||r||inf <= artol * dnorm
The four complementarity pairs are kept as two flat vectors G = [t|u|v|w], L = [lambda|pi|gamma|phi] with the mask
M = [iclow|icupp|ixlow|ixupp] - every Variables method of the reference loops over the four pairs with this mask semantics.
The reduced KKT system [dd A^T C^T; A 0 0; C 0 nOmegaInv] is solved with SuperLU here (the arithmetic under test lives in the
HIP path).  Not reproduced (neither here nor in the harness): the filter line search and the small-corrector heuristics.
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


def assemble(blocks):
    """Global data of a block LP given in the reader's layout (list of GMSPIPSBlockData_t dicts, block 0 = root) in the orders the
    harness uses: x = [x0 | x_1..x_N]; equality rows [A0 | linking | blocks]; inequality rows [C0 | linking | blocks]."""
    def mat(d, rows, cols):
        if d is None:
            return sp.csr_matrix((rows, cols))
        return sp.csr_matrix((np.asarray(d["val"], float), np.asarray(d["colidx"], int), np.asarray(d["rowptr"], int)), shape=(d["rows"], d["cols"]))

    root = blocks[0]
    n0, myl, mzl = root["n0"], root["mBL"], root["mDL"]
    ns = [n0] + [b["ni"] for b in blocks[1:]]
    Arows = [[mat(root["A"], root["mA"], n0)] + [None] * (len(blocks) - 1)]
    Arows.append([mat(b["BL"], myl, n) for b, n in zip(blocks, ns)])
    Crows = [[mat(root["C"], root["mC"], n0)] + [None] * (len(blocks) - 1)]
    Crows.append([mat(b["DL"], mzl, n) for b, n in zip(blocks, ns)])
    for k, b in enumerate(blocks[1:], start=1):
        ra, rc = [None] * len(blocks), [None] * len(blocks)
        ra[0], ra[k] = mat(b["A"], b["mA"], n0), mat(b["B"], b["mA"], b["ni"])
        rc[0], rc[k] = mat(b["C"], b["mC"], n0), mat(b["D"], b["mC"], b["ni"])
        Arows.append(ra)
        Crows.append(rc)

    def stack(rows):
        nx = sum(ns)
        out = []
        for r in rows:
            m = next(x.shape[0] for x in r if x is not None)
            out.append(sp.hstack([x if x is not None else sp.csr_matrix((m, n)) for x, n in zip(r, ns)], format="csr"))
        return sp.vstack(out, format="csr") if out else sp.csr_matrix((0, nx))

    cat = lambda f: np.concatenate([np.asarray(b[f], float) for b in blocks])   # noqa: E731
    zcat = lambda fr, fl, fb: np.concatenate([np.asarray(root[fr], float), np.asarray(root[fl], float)] + [np.asarray(b[fb], float) for b in blocks[1:]])   # noqa: E731
    return dict(A=stack(Arows), C=stack(Crows), c=cat("c"), xlow=cat("xlow"), ixlow=cat("ixlow"), xupp=cat("xupp"), ixupp=cat("ixupp"),
                b=zcat("b", "bL", "b"), clow=zcat("clow", "dlow", "clow"), iclow=zcat("iclow", "idlow", "iclow"),
                cupp=zcat("cupp", "dupp", "cupp"), icupp=zcat("icupp", "idupp", "icupp"))


def solve_general(A, b, C, clow, iclow, cupp, icupp, c, xlow, ixlow, xupp, ixupp, max_iter=100, mutol=1e-6, artol=1e-4, trace=None,
                  dual_reg=0.0, gondzio=2, free_diag=0.0, kkt_solver=None):
    """kkt_solver: None = SuperLU on the assembled reduced KKT matrix; else a callable K (csc, symmetric) -> (rhs -> solution): how the
    large fixtures plug MKL PARDISO in (tests/golden/make_ipm_configs1.py: the reference's leaf-solver class on the global matrix)."""
    A, C = sp.csr_matrix(A), sp.csr_matrix(C)
    my, nx = A.shape
    mz = C.shape[0]
    M = np.concatenate([iclow, icupp, ixlow, ixupp]).astype(float)
    Bd = np.concatenate([clow, cupp, xlow, xupp]).astype(float) * M
    sgn = np.concatenate([np.ones(mz), -np.ones(mz), np.ones(nx), -np.ones(nx)])
    oU, oV, oW = mz, 2 * mz, 2 * mz + nx
    ncp = 2 * mz + 2 * nx
    n_pairs = int(M.sum())
    free = (np.asarray(ixlow) == 0) & (np.asarray(ixupp) == 0)
    dnorm = max([np.abs(c).max(initial=0.0), np.abs(b).max(initial=0.0), np.abs(A.data).max(initial=0.0), np.abs(C.data).max(initial=0.0),
                 np.abs(Bd).max(initial=0.0)])
    dnorm = dnorm if dnorm > 0 else 1.0
    s0 = np.sqrt(dnorm)
    x, s, y, z = np.zeros(nx), np.zeros(mz), np.zeros(my), np.zeros(mz)
    G, L = s0 * M, s0 * M

    def residuals():
        rQ = c - A.T @ y - C.T @ z - L[oV:oW] + L[oW:]
        rA = A @ x - b
        rC = C @ x - s
        rz = z - L[:oU] + L[oU:oV]
        rG = np.concatenate([(s - Bd[:oU]) * M[:oU] - G[:oU], (s - Bd[oU:oV]) * M[oU:oV] + G[oU:oV],
                             (x - Bd[oV:oW]) * M[oV:oW] - G[oV:oW], (x - Bd[oW:]) * M[oW:] + G[oW:]])
        return rQ, rA, rC, rz, rG

    def objectives():
        return c @ x, b @ y + (sgn * Bd) @ L

    def solve(res, rL):
        rQ, rA, rC, rz, rG = res
        on = M != 0
        q = np.zeros(ncp)                 # (L/G rG + sgn rL / G) on the pairs
        q[on] = (L[on] * rG[on] + sgn[on] * rL[on]) / G[on]
        ratio = np.zeros(ncp)
        ratio[on] = L[on] / G[on]
        dd = ratio[oV:oW] + ratio[oW:]
        om = ratio[:oU] + ratio[oU:oV]
        nom = np.where(om != 0, -1.0 / np.where(om != 0, om, 1.0), 0.0)
        rx = rQ + q[oV:oW] + q[oW:]
        rs = rz + q[:oU] + q[oU:oV]
        rzz = rC - nom * rs
        K = sp.bmat([[sp.diags(dd + free_diag * free), A.T, C.T],
                     [A, -dual_reg * sp.identity(my) if dual_reg else None, None],
                     [C, None, sp.diags(nom)]], format="csc")
        if kkt_solver is not None:
            key = (dd.tobytes(), nom.tobytes())           # (one factorisation per iterate: predictor, corrector and Gondzio steps share it)
            if solve.key != key:
                solve.key, solve.fn = key, kkt_solver(K)
            sol = solve.fn(np.concatenate([rx, rA, rzz]))
        else:
            sol = spl.splu(K).solve(np.concatenate([rx, rA, rzz]))
        dx, dy, dz = sol[:nx], -sol[nx:nx + my], -sol[nx + my:]
        ds = -(nom * (rs - dz))
        dG = np.concatenate([ds - rG[:oU], rG[oU:oV] - ds, dx - rG[oV:oW], rG[oW:] - dx]) * M
        dL = np.zeros(ncp)
        dL[on] = (rL[on] - L[on] * dG[on]) / G[on]
        return -dx, -ds, -dy, -dz, -dG, -dL

    solve.key = solve.fn = None
    zero_res = (np.zeros(nx), np.zeros(my), np.zeros(mz), np.zeros(mz), np.zeros(ncp))
    dx, ds, dy, dz, dG, dL = solve(residuals(), G * L)
    x += dx; s += ds; y += dy; z += dz; G = G + dG; L = L + dL
    on = M != 0
    viol = max(0.0, -G[on].min(initial=0.0), -L[on].min(initial=0.0))
    G = G + (1e3 + 2 * viol) * M
    L = L + (1e3 + 2 * viol) * M
    status, it = 1, 0
    for it in range(max_iter):
        res = residuals()
        rnorm = max(np.abs(r).max(initial=0.0) for r in res)
        mu = G @ L / n_pairs if n_pairs else 0.0
        pobj, dobj = objectives()
        if trace is not None:
            trace.append((it, mu, rnorm, pobj, dobj))
        if mu <= mutol and rnorm <= artol * dnorm:
            status = 0
            break
        phi = (rnorm + abs(pobj - dobj)) / dnorm   # "probably infeasible" (PIPSIPMppSolver.cpp:128-170)
        phi_min = phi if it == 0 else min(phi_min, phi)
        if it >= 10 and phi >= 1e-8 and phi >= 1e4 * phi_min:
            status = 4
            break
        dx, ds, dy, dz, dG, dL = solve(res, G * L)
        ap, ad = min(1.0, stepbound(G, dG)), min(1.0, stepbound(L, dL))
        mu_aff = (G + ap * dG) @ (L + ad * dL) / n_pairs
        sigma = (mu_aff / mu) ** 3
        cx, cs, cy, cz, cG, cL = solve(zero_res, dG * dL - sigma * mu * M)
        ap, ad, wp, wd = weight_search(G, dG, cG, L, dL, cL, ap, ad)
        dx = dx + wp * cx; ds = ds + wp * cs; dG = dG + wp * cG
        dy = dy + wd * cy; dz = dz + wd * cz; dL = dL + wd * cL
        # Gondzio multiple centrality correctors (gondzio_correction_loop, InteriorPointMethod.cpp:236-358, primal-dual
        # variant; projection: DenseVector.cpp:405-420; weight search: InteriorPointMethod.cpp:486-523)
        rmin, rmax = sigma * mu * 0.1, sigma * mu * 10.0
        ng = 0
        while ng < gondzio and (ap < 1.0 or ad < 1.0):
            apt, adt = min(1.0, 1.5 * ap + 0.3), min(1.0, 1.5 * ad + 0.3)
            p = (G + apt * dG) * (L + adt * dL)
            t = np.where(p < rmin, rmin - p, np.where(p > rmax, rmax - p, 0.0))
            t = np.maximum(t, -rmax)
            cx, cs, cy, cz, cG, cL = solve(zero_res, -t * M)
            ape, ade, wp, wd = weight_search(G, dG, cG, L, dL, cL, apt, adt)
            both_one = ape >= 1.0 and ade >= 1.0
            p_better, d_better = ape >= 1.01 * ap, ade >= 1.01 * ad
            if not (both_one or p_better or d_better):
                break
            if both_one or p_better:
                dx = dx + wp * cx; ds = ds + wp * cs; dG = dG + wp * cG; ap = ape
            if both_one or d_better:
                dy = dy + wd * cy; dz = dz + wd * cz; dL = dL + wd * cL; ad = ade
            ng += 1
            if both_one:
                break
        ap, ad = mehrotra_step_length(G, dG, L, dL, n_pairs)
        x += ap * dx; s += ap * ds; G = G + ap * dG
        y += ad * dy; z += ad * dz; L = L + ad * dL
        if trace is not None:   # the step that leaves iterate `it`: (sigma, alpha_primal, alpha_dual) appended to its row
            trace[-1] = trace[-1] + (sigma, ap, ad)
    pobj, dobj = objectives()
    return dict(objective=pobj, iterations=it, mu=mu, rnorm=rnorm, status=status, dual_objective=dobj, x=x, s=s, y=y, z=z, dnorm=dnorm,
                t=G[:oU], u=G[oU:oV], v=G[oV:oW], w=G[oW:], lam=L[:oU], pi=L[oU:oV], gamma=L[oV:oW], phi=L[oW:])


def solve_blocks(blocks, **kw):
    """The general IPM on a block LP in the reader's layout."""
    d = assemble(blocks)
    return solve_general(d["A"], d["b"], d["C"], d["clow"], d["iclow"], d["cupp"], d["icupp"], d["c"], d["xlow"], d["ixlow"], d["xupp"],
                         d["ixupp"], **kw)


def solve_lp(A, b, c, max_iter=100, mutol=1e-6, artol=1e-4, trace=None, dual_reg=0.0, gondzio=2, bounded=None, free_diag=0.0, kkt_solver=None):
    """The generator's class  min c^T x, A x = b, x >= 0  (bounded: optional 0/1 mask, 0 = free variable: ixlow = ixupp = 0, no
    complementarity pair, dd_j = 0, LinearSystem.C:262-294) through the general routine.  free_diag: primal regularisation on the
    free entries of the KKT matrix (a free column without any coefficient makes the unregularised matrix exactly singular)."""
    A = sp.csr_matrix(A)
    ny, nx = A.shape
    ixlow = np.ones(nx) if bounded is None else np.asarray(bounded, dtype=float)
    z0 = np.zeros(0)
    return solve_general(A, b, sp.csr_matrix((0, nx)), z0, z0, z0, z0, c, np.zeros(nx), ixlow, np.zeros(nx), np.zeros(nx), max_iter=max_iter,
                         mutol=mutol, artol=artol, trace=trace, dual_reg=dual_reg, gondzio=gondzio, free_diag=free_diag, kkt_solver=kkt_solver)


# ----------------------------------------------------------------------------------------------------------------------------
# f-1: the outer BiCGStab
# ----------------------------------------------------------------------------------------------------------------------------
BICG_STATUS = {1: "converged", 2: "skipped", 3: "max iterations", 4: "breakdown", 5: "diverged", 6: "stagnation"}


def bicgstab(matvec, precond, b, tol=1e-10, max_iter=75, max_div=4, max_stag=4, eps=1e-15, history=None):
    """LinearSystem::solveCompressedBiCGStab (LinearSystem.C:550-798) restated: right-preconditioned BiCGStab with
    precond = solveCompressed, convergence tests on the predicted residual confirmed by the true one (:671-690,:722-738), the
    best-iterate rollback (:692-697,:741-760), divergence (:741-753) and stagnation (:625-630,:763-775) counters, breakdown tests
    with PIPSisZero(pips_eps0 = 1e-40) on rho, beta, r0^T v, t^T t, omega.  Returns (x, status, iterations, residual norm)."""
    def is_zero(v):
        return abs(v) < 1e-40

    def true_res(x):
        r = b - matvec(x)
        return r, np.linalg.norm(r)

    bn = np.linalg.norm(b)
    target = max(bn * tol, eps)
    x = precond(b.copy())
    r, rn = true_res(x)
    min_rn, best_x = rn, x.copy()
    if rn <= target:
        return x, 2, 0, rn
    r0 = r / rn
    rho = alpha = omega = 1.0
    ndiv = nstag = 0
    status = 3
    p = v = None
    it = 0

    def stagn(step, dxn):
        nonlocal nstag
        nstag = nstag + 1 if abs(step) * dxn <= eps * np.linalg.norm(x) else 0

    for it in range(max_iter):
        rho_last, rho = rho, r0 @ r
        if is_zero(rho):
            status = 4
            break
        if it == 0:
            p = r.copy()
        else:
            beta = (rho / rho_last) * (alpha / omega)
            if is_zero(beta):
                status = 4
                break
            p = r + beta * (p - omega * v)
        dx = precond(p.copy())
        v = matvec(dx)
        rtv = r0 @ v
        if is_zero(rtv):
            status = 4
            break
        alpha = rho / rtv
        stagn(alpha, np.linalg.norm(dx))
        x = x + alpha * dx
        r = r - alpha * v
        rn = np.linalg.norm(r)
        if rn <= target:
            r, tr = true_res(x)
            if tr <= target:
                rn, status = tr, 1
                break
            min_rn = true_res(best_x)[1]
            rn = tr
        if rn < min_rn:
            min_rn, best_x = rn, x.copy()
        dx = precond(r.copy())
        t = matvec(dx)
        tt = t @ t
        if is_zero(tt):
            status = 4
            break
        omega = (t @ r) / tt
        stagn(omega, np.linalg.norm(dx))
        x = x + omega * dx
        r = r - omega * t
        rn = np.linalg.norm(r)
        if rn <= target or nstag >= max_stag:
            r, tr = true_res(x)
            if tr <= target:
                rn, status = tr, 1
                break
            min_rn = true_res(best_x)[1]
            rn = tr
        else:
            ndiv = ndiv + 1 if rn >= min_rn else 0
            if ndiv > max_div:
                x, rn, status = best_x.copy(), min_rn, 5
                break
        if rn < min_rn:
            min_rn, best_x = rn, x.copy()
        if history is not None:
            history.append(rn)
        if nstag >= max_stag:
            if min_rn < rn:
                rn, x = min_rn, best_x.copy()
            status = 6
            break
        if is_zero(omega):
            status = 4
            break
    else:
        it = max_iter - 1
    return x, status, it + 1, rn
