"""Random block-structured LPs in the reference's reader layout (GMSPIPSBlockData_t fields, as pips_gdx_read_block returns them):
every bound kind on variables (lower, upper, boxed, fixed, free), equality / <= / >= / ranged rows, own and linking rows.
Feasible and bounded by construction: the right-hand sides come from a point x* inside the bounds, the objective is made dual
feasible (c = A^T y + reduced costs whose signs match the active bounds of x*)."""
import numpy as np
import scipy.sparse as sp


def _csr(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    return dict(rows=M.shape[0], cols=M.shape[1], rowptr=M.indptr.tolist(), colidx=M.indices.tolist(), val=M.data.tolist())


def _rand(rng, m, n, density):
    M = sp.random(m, n, density=min(1.0, density), random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
    M.data = rng.choice([1.0, -1.0, 2.0, 0.5, 3.0, -2.5], size=M.nnz)
    return M


def random_block_lp(seed, num_blocks, n0, ni, mA, mC, mBL, mDL, free_fraction=0.15):
    rng = np.random.default_rng(seed)
    sizes = [n0] + [ni] * (num_blocks - 1)
    xs, bounds = [], []
    for n in sizes:
        kind = rng.choice(5, size=n, p=[0.45, 0.1, 0.2, 0.25 - free_fraction, free_fraction])   # lower, upper, boxed, fixed, free
        lo = rng.integers(-2, 3, size=n).astype(float)
        up = lo + rng.integers(1, 6, size=n)
        x = lo + (up - lo) * rng.choice([0.0, 0.5, 1.0, 0.3], size=n)
        x[kind == 3] = lo[kind == 3]
        x[kind == 4] = rng.standard_normal((kind == 4).sum())
        xs.append(x)
        bounds.append((kind, lo, up))
    blocks = []
    link_eq = np.zeros(mBL)
    link_in = np.zeros(mDL)
    BLs, DLs = [], []
    for k, n in enumerate(sizes):
        BL, DL = _rand(rng, mBL, n, 3.0 / n), _rand(rng, mDL, n, 3.0 / n)
        BLs.append(BL); DLs.append(DL)
        link_eq += BL @ xs[k]
        link_in += DL @ xs[k]
    dkind = rng.integers(0, 3, size=mDL)
    dlow = np.where(dkind != 0, link_in - rng.integers(0, 3, size=mDL), 0.0)
    dupp = np.where(dkind != 1, link_in + rng.integers(0, 3, size=mDL), 0.0)
    for k, n in enumerate(sizes):
        kind, lo, up = bounds[k]
        ma, mc = (mA, mC) if k else (max(1, mA // 2), max(1, mC // 2))
        A = _rand(rng, ma, n0, 2.0 / n0)
        C = _rand(rng, mc, n0, 2.0 / n0)
        if k:
            B = sp.hstack([sp.identity(ma), _rand(rng, ma, n - ma, 3.0 / n)], format="csr") if n > ma else _rand(rng, ma, n, 0.5)
            D = _rand(rng, mc, n, 3.0 / n)
            ra, rc = A @ xs[0] + B @ xs[k], C @ xs[0] + D @ xs[k]
        else:
            A = sp.hstack([sp.identity(ma), _rand(rng, ma, n0 - ma, 3.0 / n0)], format="csr") if n0 > ma else A
            B = D = None
            ra, rc = A @ xs[0], C @ xs[0]
        ckind = rng.integers(0, 3, size=mc)     # 0: <= only, 1: >= only, 2: range
        clow = np.where(ckind != 0, rc - rng.integers(0, 3, size=mc), 0.0)
        cupp = np.where(ckind != 1, rc + rng.integers(0, 3, size=mc), 0.0)
        blocks.append(dict(numBlocks=num_blocks, blockID=k, n0=n0, ni=n, mA=ma, mC=mc, mBL=mBL, mDL=mDL,
                           c=np.zeros(n), xlow=np.where(kind != 1, lo, 0.0) * (kind != 4), xupp=np.where(np.isin(kind, [1, 2]), up, np.where(kind == 3, lo, 0.0)),
                           ixlow=np.isin(kind, [0, 2, 3]).astype(np.int16), ixupp=np.isin(kind, [1, 2, 3]).astype(np.int16),
                           b=ra, clow=clow, cupp=cupp, iclow=(ckind != 0).astype(np.int16), icupp=(ckind != 1).astype(np.int16),
                           bL=link_eq.copy(), dlow=dlow, dupp=dupp, idlow=(dkind != 0).astype(np.int16), idupp=(dkind != 1).astype(np.int16),
                           A=_csr(A), B=_csr(B) if B is not None else None, C=_csr(C), D=_csr(D) if D is not None else None,
                           BL=_csr(BLs[k]), DL=_csr(DLs[k])))
    # upper-bound-only variables sit at their upper bound in x*
    for k, (kind, lo, up) in enumerate(bounds):
        xs[k][kind == 1] = up[kind == 1]
    # objective: bounded because every variable either is boxed / fixed or gets a cost pushing it towards its only bound;
    # free variables get cost zero plus a row combination (they appear in equality rows with an identity block where possible)
    for k, (kind, lo, up) in enumerate(bounds):
        cst = rng.integers(1, 4, size=sizes[k]).astype(float)
        cst[kind == 1] *= -1.0
        cst[kind == 4] = 0.0
        cst[np.isin(kind, [2, 3])] *= rng.choice([-1.0, 1.0], size=int(np.isin(kind, [2, 3]).sum()))
        blocks[k]["c"] = cst
    return blocks
