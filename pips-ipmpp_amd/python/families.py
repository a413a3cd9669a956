"""Synthetic block families of the harness (next to the generator of SURVEY 8d, which lives in csrc/gen.cpp).

time_coupled_blocks: the shape BASELINE.json configs[3] ("energy-system scale") presumes - time-coupled rows inside a block
(banded W_i, ~10 non-zeros per row as SURVEY 8d asks), a handful of first-stage variables, 2-link rows between neighbouring
blocks.  Uniformly random fill has no counterpart at 50 000 variables per block (the factor of one block would be dense: 5 GB).
Used by bench.py --family time-coupled, tools/config3_probe.py and the tests."""
import numpy as np
import scipy.sparse as sp

from . import capi as pa


def _csr(M):
    M = sp.csr_matrix(M)
    M.sum_duplicates()
    M.sort_indices()
    return pa.Csr(M.shape[0], M.shape[1], M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64))


def time_coupled_blocks(N, n_i, L, n0, bw, nnz_row, seed):
    """Time-coupled blocks: W_i banded (band half-width bw, nnz_row entries per row, the diagonal-like entry always present),
    T_i with ~2 entries per row on the n0 first-stage variables, 2-link rows: L linking equalities between every pair of
    neighbouring blocks with 3 entries per block."""
    rng = np.random.default_rng(seed)
    my_i, myl = n_i // 2, (N - 1) * L
    out = []
    for i in range(N):
        rows = np.repeat(np.arange(my_i), nnz_row)
        center = (np.arange(my_i) * n_i // my_i)[:, None]
        cols = np.clip(center + rng.integers(-bw, bw + 1, (my_i, nnz_row)), 0, n_i - 1)
        cols[:, 0] = center[:, 0]
        W = sp.csr_matrix((rng.uniform(-1, 1, rows.size), (rows, cols.ravel())), shape=(my_i, n_i))
        tr = np.repeat(np.arange(my_i), 2)
        T = sp.csr_matrix((rng.uniform(-1, 1, tr.size), (tr, rng.integers(0, n0, tr.size))), shape=(my_i, n0))
        fr, fc, fv = [np.zeros(0, int)], [np.zeros(0, int)], [np.zeros(0)]
        for pair in (i - 1, i):
            if 0 <= pair < N - 1:
                r = np.repeat(np.arange(pair * L, (pair + 1) * L), 3)
                fr.append(r)
                fc.append(rng.integers(0, n_i, r.size))
                fv.append(rng.uniform(-1, 1, r.size))
        F = sp.csr_matrix((np.concatenate(fv), (np.concatenate(fr), np.concatenate(fc))), shape=(myl, n_i))
        out.append((_csr(W), _csr(T), _csr(F)))
    F0 = sp.random(myl, n0, density=min(1.0, 2.0 / n0), random_state=seed, format="csr")
    return out, _csr(F0), my_i, myl


CONFIG3_SHARE = dict(L=31, n0=95, bw=12, nnz_row=10, seed=20261004)   # 256 blocks -> S = 95 + 255 * 31 = 8000

