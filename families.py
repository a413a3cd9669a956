"""Synthetic block families of the harness (next to the generator of SURVEY 8d, which lives in csrc/gen.cpp).

The time-coupled chain: the shape BASELINE.json configs[3] ("energy-system scale") presumes - time-coupled rows inside a block
(banded W_i, ~10 non-zeros per row as SURVEY 8d asks), a handful of first-stage variables, 2-link rows between neighbouring
blocks.  Uniformly random fill has no counterpart at 50 000 variables per block (the factor of one block would be dense: 5 GB).

A chain is ONE fixed problem (`TimeCoupledChain`): G blocks, `link_rows` two-link rows spread evenly over the G - 1 neighbouring
pairs, n0 first-stage variables, Schur dimension S = n0 + link_rows whatever the number of ranks that work on it.  Every block
draws from its own generator seeded by (seed, block index), so a rank produces exactly the blocks of its range
(`blocks(lo, hi)`) - the reference maps a fixed tree onto ranks the same way (Readers/Distributed/DistributedTree.C:62-89) - and
host memory per rank does not grow with the number of ranks.  `CONFIG3` is BASELINE configs[3]: 2048 blocks, S = 8000 (95 + 7905
linking rows: 3 or 4 per pair).  `prefix(k)` is the sub-problem of the first k blocks (their linking rows only: every one of
them is touched by a block that is present), what `bench.py` runs on fewer than the 8 GPUs of configs[3].

Used by bench.py --family time-coupled, the tools and the tests; lives beside them, not in the product package."""
import numpy as np
import scipy.sparse as sp

import pips_ipmpp_amd as pa


def _csr(M):
    M = sp.csr_matrix(M)
    M.sum_duplicates()
    M.sort_indices()
    return pa.Csr(M.shape[0], M.shape[1], M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64))


class TimeCoupledChain:
    """G blocks of n_i variables (n_i // 2 banded equality rows each), n0 first-stage variables, link_rows two-link equalities:
    pair p (between blocks p and p + 1) owns rows [row0[p], row0[p + 1]) with row0[p] = floor(p * link_rows / (G - 1)) - equal
    shares up to rounding; every linking row has 3 entries in each of its two blocks."""

    def __init__(self, G, n_i, link_rows, n0, bw, nnz_row, seed):
        if G < 2 or link_rows < 0:
            raise ValueError("a chain has at least two blocks")
        self.G, self.n_i, self.my_i, self.n0, self.bw, self.nnz_row, self.seed = G, n_i, n_i // 2, n0, bw, nnz_row, seed
        self.myl = int(link_rows)
        self.row0 = (np.arange(G, dtype=np.int64) * self.myl) // (G - 1)       # row0[G - 1] = link_rows
        self.S = n0 + self.myl

    def pair_rows(self, p):
        """[first, last) linking rows of pair p; empty outside the chain"""
        if p < 0 or p >= self.G - 1:
            return 0, 0
        return int(self.row0[p]), int(self.row0[p + 1])

    def prefix(self, k):
        """the chain of the first k blocks with the linking rows they touch (pairs 0 .. k - 1; the last pair is one-sided unless
        k = G): a fixed sub-problem of this chain, its Schur dimension n0 + row0[min(k, G - 1)]"""
        if k >= self.G:
            return self
        sub = TimeCoupledChain.__new__(TimeCoupledChain)
        sub.__dict__.update(self.__dict__)
        sub.G_present = k
        sub.myl = int(self.row0[k])
        sub.S = self.n0 + sub.myl
        return sub

    @property
    def n_blocks(self):
        return getattr(self, "G_present", self.G)

    def block(self, i):
        """(W_i, T_i, F_i) of block i of the chain, F_i with self.myl rows"""
        if not 0 <= i < self.n_blocks:
            raise IndexError(i)
        rng = np.random.default_rng([self.seed, i])
        n_i, my_i, n0 = self.n_i, self.my_i, self.n0
        rows = np.repeat(np.arange(my_i), self.nnz_row)
        center = (np.arange(my_i) * n_i // my_i)[:, None]
        cols = np.clip(center + rng.integers(-self.bw, self.bw + 1, (my_i, self.nnz_row)), 0, n_i - 1)
        cols[:, 0] = center[:, 0]
        W = sp.csr_matrix((rng.uniform(-1, 1, rows.size), (rows, cols.ravel())), shape=(my_i, n_i))
        tr = np.repeat(np.arange(my_i), 2)
        T = sp.csr_matrix((rng.uniform(-1, 1, tr.size), (tr, rng.integers(0, n0, tr.size))), shape=(my_i, n0))
        fr, fc, fv = [np.zeros(0, int)], [np.zeros(0, int)], [np.zeros(0)]
        for pair in (i - 1, i):
            lo, hi = self.pair_rows(pair)
            hi = min(hi, self.myl)
            if hi > lo:
                r = np.repeat(np.arange(lo, hi), 3)
                fr.append(r)
                fc.append(rng.integers(0, n_i, r.size))
                fv.append(rng.uniform(-1, 1, r.size))
        F = sp.csr_matrix((np.concatenate(fv), (np.concatenate(fr), np.concatenate(fc))), shape=(self.myl, n_i))
        return _csr(W), _csr(T), _csr(F)

    def blocks(self, lo, hi):
        return [self.block(i) for i in range(lo, hi)]

    def F0(self):
        """root part of the linking rows (~2 entries per row), drawn for the whole chain and cut to the rows present"""
        full = int(self.row0[self.G - 1])
        M = sp.random(full, self.n0, density=min(1.0, 2.0 / self.n0), random_state=self.seed, format="csr")
        return _csr(M[:self.myl])

    def border_columns(self, i, T=None):
        """non-empty border columns of block i in the Schur complement's numbering: the first-stage columns T_i touches, then its
        linking rows.  The linking rows follow from the chain alone; the first-stage columns need T_i (pass it if it is at hand)."""
        if T is None:
            T = self.block(i)[1]
        rows = [np.arange(*self.pair_rows(p)) for p in (i - 1, i)]
        link = np.concatenate(rows)
        return np.concatenate([np.unique(T.colidx), self.n0 + link[link < self.myl]]).astype(np.int32)


def share_range(n_blocks, rank, n_ranks):
    """contiguous, even block range of a rank - the rule of pips_map_children_to_ranks (DistributedTree.C:62-89)"""
    base, extra = divmod(n_blocks, n_ranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def time_coupled_blocks(N, n_i, L, n0, bw, nnz_row, seed):
    """All blocks of an N-block chain with L linking rows per neighbouring pair (the small cases of the tests and tools):
    -> [(W, T, F)], F0, my_i, myl"""
    ch = TimeCoupledChain(N, n_i, (N - 1) * L, n0, bw, nnz_row, seed)
    return ch.blocks(0, N), ch.F0(), ch.my_i, ch.myl


CONFIG3 = dict(G=2048, S=8000, n0=95, bw=12, nnz_row=10, seed=20261004)          # BASELINE configs[3]: 95 + 7905 linking rows
CONFIG3_SHARE = dict(L=31, n0=95, bw=12, nnz_row=10, seed=20261004)   # rounds 3-4: a 256-block chain, S = 95 + 255 * 31 = 8000


def config3_chain(n_i=50000, G=None, S=None):
    """the configs[3] chain (G = 2048, S = 8000), or the same shape with another block count / Schur dimension: G = 256 is the
    256-block chain rounds 3 and 4 measured (31 linking rows per pair)"""
    c = CONFIG3
    G = G or c["G"]
    S = S or c["S"]
    return TimeCoupledChain(G, n_i, S - c["n0"], c["n0"], c["bw"], c["nnz_row"], c["seed"])
