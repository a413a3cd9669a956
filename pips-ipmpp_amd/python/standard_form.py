"""Block data in the reference's reader layout (GMSPIPSBlockData_t, gmspipsio.h:5-58; one dict per block as returned by
capi.gdx_read_block / pips_ipmpp_amd.gdx.read_block or stored in tests/golden/gamssmall.json) -> the harness' problem class
    min c^T y + offset,  A y = b,  y >= 0,   A block-angular with linking variables and linking rows.

Per variable:   fixed -> constant;  lower bound only -> x = l + x';  upper bound only -> x = u - x';  both -> x = l + x' and a
                bound row x' + s = u - l in the variable's own block;  free -> x = x+ - x-.
Per inequality: lower only -> r - s = low;  upper only -> r + s = upp;  both -> r - s = low and a range row s + s2 = upp - low.
Rows that only involve block-0 variables (the root's own rows, its bound rows) join the linking rows: the harness has no
separate root equality block in this problem class.
`general_lp` assembles the same data as one bounded LP (for HiGHS), so that the conversion itself can be checked."""
import numpy as np
import scipy.sparse as sp

from . import capi as pa


def _csr(d, rows, cols):
    if d is None:
        return sp.csr_matrix((rows, cols))
    return sp.csr_matrix((np.asarray(d["val"], dtype=float), np.asarray(d["colidx"], dtype=int), np.asarray(d["rowptr"], dtype=int)),
                         shape=(rows, cols))


def _to_pa(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    return pa.Csr(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)


def _var_transform(blk, n, split_free=True):
    """x = xc + M x', the list of (column of x', width) that need a bound row x'_j + s = width, and the 0/1 mask of the columns
    of x' that are sign-constrained (0 = a free variable kept as one column, split_free=False)."""
    xlow, xupp = np.asarray(blk["xlow"], dtype=float), np.asarray(blk["xupp"], dtype=float)
    il, iu = np.asarray(blk["ixlow"]), np.asarray(blk["ixupp"])
    xc = np.zeros(n)
    ri, ci, vv, ranges, free_cols = [], [], [], [], []
    ncol = 0
    for j in range(n):
        if il[j] and iu[j] and xlow[j] == xupp[j]:
            xc[j] = xlow[j]
        elif il[j]:
            xc[j] = xlow[j]
            ri.append(j); ci.append(ncol); vv.append(1.0)
            if iu[j]:
                ranges.append((ncol, xupp[j] - xlow[j]))
            ncol += 1
        elif iu[j]:
            xc[j] = xupp[j]
            ri.append(j); ci.append(ncol); vv.append(-1.0)
            ncol += 1
        elif split_free:
            ri += [j, j]; ci += [ncol, ncol + 1]; vv += [1.0, -1.0]
            ncol += 2
        else:
            ri.append(j); ci.append(ncol); vv.append(1.0)
            free_cols.append(ncol)
            ncol += 1
    mask = np.ones(ncol)
    mask[free_cols] = 0.0
    return xc, sp.csr_matrix((vv, (ri, ci)), shape=(n, ncol)), ranges, mask


def _ineq_rows(low, ilow, upp, iupp):
    """Slack sign per row (+1: r + s = upp, -1: r - s = low, 0: the row is an equality r = low), rhs, and the range rows
    (slack index, width)."""
    m = len(ilow)
    sign, rhs, ranges = np.zeros(m), np.zeros(m), []
    for r in range(m):
        if ilow[r] and iupp[r]:
            rhs[r] = low[r]
            if upp[r] != low[r]:
                sign[r] = -1.0
                ranges.append((r, upp[r] - low[r]))
        elif ilow[r]:
            sign[r], rhs[r] = -1.0, low[r]
        else:
            sign[r], rhs[r] = 1.0, upp[r]
    return sign, rhs, ranges


def _select(cols_widths, ncols):
    """Rows e_j^T for the given columns and the widths as rhs."""
    k = len(cols_widths)
    E = sp.csr_matrix((np.ones(k), (np.arange(k), [c for c, _ in cols_widths])), shape=(k, ncols))
    return E, np.array([w for _, w in cols_widths], dtype=float)


def block_standard_form(blocks, split_free=True):
    """split_free=False keeps a free variable as one column and reports it in `bounded_mask` (0 there) for
    IpmSolver.set_free_variables; the default splits it into x+ - x-."""
    root, kids = blocks[0], blocks[1:]
    n0 = root["n0"]
    mBL, mDL = root["mBL"], root["mDL"]
    xc0, M0, vr0, mask0 = _var_transform(root, n0, split_free)
    n0p = M0.shape[1]
    # ---- root rows: [link eq | link ineq | A0 | C0 | range rows of link ineq and C0 | bound rows of x0]
    sgnL, rhsL, rngL = _ineq_rows(root["dlow"], root["idlow"], root["dupp"], root["idupp"])
    sgnC, rhsC, rngC = _ineq_rows(root["clow"], root["iclow"], root["cupp"], root["icupp"])
    mA0, mC0 = root["mA"], root["mC"]
    # first-stage variable vector: [x0' | sL (mDL) | sC (mC0) | s2 for rngL | s2 for rngC | bound slacks]
    nsl, nsc, n2l, n2c, nb0 = mDL, mC0, len(rngL), len(rngC), len(vr0)
    n0s = n0p + nsl + nsc + n2l + n2c + nb0
    off = np.cumsum([0, n0p, nsl, nsc, n2l, n2c, nb0])

    def pad0(M, at=None, D=None):
        """[M | 0 ...] over the first-stage vector, optionally with block D placed at column offset `at`."""
        M = sp.csr_matrix(M)
        out = sp.lil_matrix((M.shape[0], n0s))
        out[:, :n0p] = M
        if D is not None:
            out[:, at:at + D.shape[1]] = D
        return out.tocsr()

    BL0, DL0 = _csr(root["BL"], mBL, n0), _csr(root["DL"], mDL, n0)
    A0, C0 = _csr(root["A"], mA0, n0), _csr(root["C"], mC0, n0)
    E0, w0 = _select(vr0, n0p)
    rows0 = [pad0(BL0 @ M0), pad0(DL0 @ M0, off[1], sp.diags(sgnL)), pad0(A0 @ M0), pad0(C0 @ M0, off[2], sp.diags(sgnC))]
    rhs0 = [np.asarray(root["bL"], dtype=float) - BL0 @ xc0, rhsL - DL0 @ xc0, np.asarray(root["b"], dtype=float) - A0 @ xc0, rhsC - C0 @ xc0]
    # range rows: s + s2 = width
    R = sp.lil_matrix((n2l, n0s))
    for q, (r, w) in enumerate(rngL):
        R[q, off[1] + r] = 1.0; R[q, off[3] + q] = 1.0
    rows0.append(R.tocsr()); rhs0.append(np.array([w for _, w in rngL], dtype=float))
    R = sp.lil_matrix((n2c, n0s))
    for q, (r, w) in enumerate(rngC):
        R[q, off[2] + r] = 1.0; R[q, off[4] + q] = 1.0
    rows0.append(R.tocsr()); rhs0.append(np.array([w for _, w in rngC], dtype=float))
    rows0.append(pad0(E0, off[5], sp.identity(nb0))); rhs0.append(w0)
    F0 = sp.vstack(rows0, format="csr")
    b_link = np.concatenate(rhs0)
    n_link = F0.shape[0]
    c0 = np.concatenate([M0.T @ np.asarray(root["c"], dtype=float), np.zeros(n0s - n0p)])
    offset = float(np.asarray(root["c"], dtype=float) @ xc0)
    out_blocks, cs, bs, recover = [], [], [], [(xc0, M0)]
    row_layout = [dict(mBL=mBL, mDL=mDL, mA=mA0, mC=mC0, rows=n_link)]
    masks = [np.concatenate([mask0, np.ones(n0s - n0p)])]
    for k in kids:
        ni, mA, mC = k["ni"], k["mA"], k["mC"]
        xc, M, vr, maskk = _var_transform(k, ni, split_free)
        n1 = M.shape[1]
        sgn, rhs, rng = _ineq_rows(k["clow"], k["iclow"], k["cupp"], k["icupp"])
        n2, nb = len(rng), len(vr)
        nloc = n1 + mC + n2 + nb
        A, B = _csr(k["A"], mA, n0), _csr(k["B"], mA, ni)
        C, D = _csr(k["C"], mC, n0), _csr(k["D"], mC, ni)
        BL, DL = _csr(k["BL"], mBL, ni), _csr(k["DL"], mDL, ni)
        E, w = _select(vr, n1)
        R = sp.lil_matrix((n2, nloc))
        for q, (r, _) in enumerate(rng):
            R[q, n1 + r] = 1.0; R[q, n1 + mC + q] = 1.0
        W = sp.vstack([sp.hstack([B @ M, sp.csr_matrix((mA, nloc - n1))]),
                       sp.hstack([D @ M, sp.diags(sgn), sp.csr_matrix((mC, n2 + nb))]),
                       R.tocsr(),
                       sp.hstack([E, sp.csr_matrix((nb, mC + n2)), sp.identity(nb)])], format="csr")
        T = sp.vstack([pad0(A @ M0), pad0(C @ M0), sp.csr_matrix((n2 + nb, n0s))], format="csr")
        F = sp.vstack([sp.hstack([BL @ M, sp.csr_matrix((mBL, nloc - n1))]),
                       sp.hstack([DL @ M, sp.csr_matrix((mDL, nloc - n1))]),
                       sp.csr_matrix((n_link - mBL - mDL, nloc))], format="csr")
        b_link[:mBL] -= BL @ xc
        b_link[mBL:mBL + mDL] -= DL @ xc
        out_blocks.append((_to_pa(W), _to_pa(T), _to_pa(F)))
        cs.append(np.concatenate([M.T @ np.asarray(k["c"], dtype=float), np.zeros(nloc - n1)]))
        bs.append(np.concatenate([np.asarray(k["b"], dtype=float) - B @ xc - A @ xc0, rhs - D @ xc - C @ xc0,
                                  np.array([ww for _, ww in rng], dtype=float), w]))
        offset += float(np.asarray(k["c"], dtype=float) @ xc)
        recover.append((xc, M))
        row_layout.append(dict(mA=mA, mC=mC, rows=W.shape[0]))
        masks.append(np.concatenate([maskk, np.ones(nloc - n1)]))
    c = np.concatenate([c0] + cs)
    b = np.concatenate([b_link] + bs)
    rows = [[F0] + [f.to_scipy() for (_, _, f) in out_blocks]]
    for i, (W, T, F) in enumerate(out_blocks):
        r = [T.to_scipy()] + [None] * len(out_blocks)
        r[1 + i] = W.to_scipy()
        rows.append(r)
    Afull = sp.bmat(rows, format="csr")
    return dict(n0=n0s, myl=n_link, blocks=out_blocks, F0=_to_pa(F0), c=c, b=b, A=Afull, offset=offset, recover=recover,
                bounded_mask=np.concatenate(masks), row_layout=row_layout,
                sizes=[n0s] + [f.to_scipy().shape[1] for (_, _, f) in out_blocks])


def recover_solution(sf, y):
    """Solution of the standard form (flat vector y, blocks in order) -> the original variables of every block:
    x_k = xc_k + M_k y_k[:n'_k]  (slack columns dropped)."""
    out, at = [], 0
    for (xc, M), size in zip(sf["recover"], sf["sizes"]):
        out.append(xc + M @ y[at:at + M.shape[1]])
        at += size
    return out


def general_lp(blocks):
    """The same data as one LP with bounds and two-sided rows, for scipy.optimize.linprog:
    returns c, A_eq, b_eq, A_ub, b_ub, bounds."""
    root, kids = blocks[0], blocks[1:]
    n0, mBL, mDL = root["n0"], root["mBL"], root["mDL"]
    sizes = [n0] + [k["ni"] for k in kids]
    offs = np.cumsum([0] + sizes)
    ntot = offs[-1]

    def place(M, k):
        return sp.hstack([sp.csr_matrix((M.shape[0], offs[k])), M, sp.csr_matrix((M.shape[0], ntot - offs[k + 1]))], format="csr")

    eq, beq, ub, bub = [], [], [], []

    def add_ineq(G, low, ilow, upp, iupp):
        for r in range(G.shape[0]):
            if iupp[r]:
                ub.append(G[r]); bub.append(upp[r])
            if ilow[r]:
                ub.append(-G[r]); bub.append(-low[r])

    eq.append(place(_csr(root["A"], root["mA"], n0), 0)); beq.append(np.asarray(root["b"], dtype=float))
    add_ineq(place(_csr(root["C"], root["mC"], n0), 0), root["clow"], root["iclow"], root["cupp"], root["icupp"])
    L, Dl = place(_csr(root["BL"], mBL, n0), 0), place(_csr(root["DL"], mDL, n0), 0)
    for i, k in enumerate(kids, start=1):
        ni = k["ni"]
        eq.append(place(_csr(k["A"], k["mA"], n0), 0) + place(_csr(k["B"], k["mA"], ni), i)); beq.append(np.asarray(k["b"], dtype=float))
        add_ineq(place(_csr(k["C"], k["mC"], n0), 0) + place(_csr(k["D"], k["mC"], ni), i), k["clow"], k["iclow"], k["cupp"], k["icupp"])
        L = L + place(_csr(k["BL"], mBL, ni), i)
        Dl = Dl + place(_csr(k["DL"], mDL, ni), i)
    eq.append(L); beq.append(np.asarray(root["bL"], dtype=float))
    add_ineq(sp.csr_matrix(Dl), root["dlow"], root["idlow"], root["dupp"], root["idupp"])
    c = np.concatenate([np.asarray(b["c"], dtype=float) for b in blocks])
    bounds = []
    for b in blocks:
        n = b["n0"] if b["blockID"] == 0 else b["ni"]
        for j in range(n):
            bounds.append((b["xlow"][j] if b["ixlow"][j] else None, b["xupp"][j] if b["ixupp"][j] else None))
    A_eq = sp.vstack(eq, format="csr")
    A_ub = sp.vstack(ub, format="csr") if ub else None
    return c, A_eq, np.concatenate(beq), A_ub, (np.array(bub, dtype=float) if ub else None), bounds


def recover_duals(sf, y):
    """Multipliers of the original rows from the dual solution y of the standard form (rows: [linking | block 1 | block 2 ...]).
    Per block: `eq` (its own equality rows) and `ineq` (its own inequality rows); block 0 also has `link_eq` and `link_ineq`.
    Convention: the multiplier is d objective / d right-hand side (for an inequality: of the side that is active) - what
    GAMS calls the marginal; an equality row turned slack row keeps its multiplier (stationarity in the slack ties it to the
    slack's reduced cost), range and bound rows are internal and dropped."""
    out, at = [], 0
    for k, lay in enumerate(sf["row_layout"]):
        seg = y[at:at + lay["rows"]]
        if k == 0:
            a = lay["mBL"] + lay["mDL"]
            out.append(dict(link_eq=seg[:lay["mBL"]], link_ineq=seg[lay["mBL"]:a], eq=seg[a:a + lay["mA"]], ineq=seg[a + lay["mA"]:a + lay["mA"] + lay["mC"]]))
        else:
            out.append(dict(eq=seg[:lay["mA"]], ineq=seg[lay["mA"]:lay["mA"] + lay["mC"]]))
        at += lay["rows"]
    return out


def general_rows(blocks):
    """The rows of the original problem as one matrix in the order recover_duals reports multipliers:
    [link_eq | link_ineq | block 0 eq | block 0 ineq | block 1 eq | block 1 ineq | ...] over the variables [x0 | x1 | ...]."""
    root, kids = blocks[0], blocks[1:]
    n0, mBL, mDL = root["n0"], root["mBL"], root["mDL"]
    sizes = [n0] + [k["ni"] for k in kids]
    offs = np.cumsum([0] + sizes)
    ntot = offs[-1]

    def place(M, k):
        return sp.hstack([sp.csr_matrix((M.shape[0], offs[k])), M, sp.csr_matrix((M.shape[0], ntot - offs[k + 1]))], format="csr")

    L = place(_csr(root["BL"], mBL, n0), 0)
    Dl = place(_csr(root["DL"], mDL, n0), 0)
    rows = [None, None, place(_csr(root["A"], root["mA"], n0), 0), place(_csr(root["C"], root["mC"], n0), 0)]
    for i, k in enumerate(kids, start=1):
        ni = k["ni"]
        L = L + place(_csr(k["BL"], mBL, ni), i)
        Dl = Dl + place(_csr(k["DL"], mDL, ni), i)
        rows.append(place(_csr(k["A"], k["mA"], n0), 0) + place(_csr(k["B"], k["mA"], ni), i))
        rows.append(place(_csr(k["C"], k["mC"], n0), 0) + place(_csr(k["D"], k["mC"], ni), i))
    rows[0], rows[1] = L, sp.csr_matrix(Dl)
    return sp.vstack(rows, format="csr")


def kkt_violation(blocks, x_blocks, duals, tol=1e-6):
    """Largest violation of the optimality conditions of the ORIGINAL problem (bounded variables, two-sided rows) by a primal
    point (list of per-block x) and the multipliers of recover_duals: stationarity with the right sign of every reduced cost
    (>= 0 at a lower bound, <= 0 at an upper bound, 0 inside), sign and complementarity of the inequality multipliers
    (<= 0 on a row at its upper side, >= 0 at its lower side, 0 strictly inside), primal feasibility."""
    x = np.concatenate(x_blocks)
    lam = np.concatenate([duals[0]["link_eq"], duals[0]["link_ineq"], duals[0]["eq"], duals[0]["ineq"]] +
                         [np.concatenate([d["eq"], d["ineq"]]) for d in duals[1:]])
    G = general_rows(blocks)
    c = np.concatenate([np.asarray(b["c"], dtype=float) for b in blocks])
    rc = c - G.T @ lam
    act = G @ x
    worst = 0.0
    at = 0
    for b in blocks:
        n = b["n0"] if b["blockID"] == 0 else b["ni"]
        for j in range(n):
            xi, r = x[at + j], rc[at + j]
            lo = b["xlow"][j] if b["ixlow"][j] else None
            up = b["xupp"][j] if b["ixupp"][j] else None
            scale = max(1.0, abs(xi))
            at_lo = lo is not None and abs(xi - lo) <= tol * scale
            at_up = up is not None and abs(xi - up) <= tol * scale
            if lo is not None:
                worst = max(worst, lo - xi)
            if up is not None:
                worst = max(worst, xi - up)
            if at_lo and at_up:
                continue
            worst = max(worst, max(0.0, -r) if at_lo else (max(0.0, r) if at_up else abs(r)))
        at += n
    # rows, in the order of general_rows
    root = blocks[0]
    specs = [("eq", root["bL"], None, None, None), ("in", root["dlow"], root["idlow"], root["dupp"], root["idupp"]),
             ("eq", root["b"], None, None, None), ("in", root["clow"], root["iclow"], root["cupp"], root["icupp"])]
    for k in blocks[1:]:
        specs += [("eq", k["b"], None, None, None), ("in", k["clow"], k["iclow"], k["cupp"], k["icupp"])]
    r0 = 0
    for kind, a1, i1, a2, i2 in specs:
        m = len(a1)
        for q in range(m):
            v, mult = act[r0 + q], lam[r0 + q]
            if kind == "eq":
                worst = max(worst, abs(v - a1[q]))
                continue
            scale = max(1.0, abs(v))
            at_lo = bool(i1[q]) and abs(v - a1[q]) <= tol * scale
            at_up = bool(i2[q]) and abs(v - a2[q]) <= tol * scale
            if i1[q]:
                worst = max(worst, a1[q] - v)
            if i2[q]:
                worst = max(worst, v - a2[q])
            if at_lo and at_up:
                continue
            worst = max(worst, max(0.0, -mult) if at_lo else (max(0.0, mult) if at_up else abs(mult)))
        r0 += m
    return worst
