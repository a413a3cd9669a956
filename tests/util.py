"""Shared helpers for the tests: synthetic arrowhead problems in both the product's and the oracle's
representation.  The generator is the product's host harness (pips_gen_*); the oracle consumes the same matrices."""
import numpy as np
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from oracle import oracle as orc


class Problem:
    """N blocks, each n_i vars / my_i equalities, n0 first-stage vars, myl linking equalities."""

    def __init__(self, seed, N, n_i, my_i, n0, myl, rho, dual_reg=1e-8, diag_lo=-4.0, diag_hi=4.0):
        self.N, self.n_i, self.my_i, self.n0, self.myl = N, n_i, my_i, n0, myl
        self.S = n0 + myl
        self.blocks = []
        for b in range(1, N + 1):
            W, T, F, c, xs = pa.gen_block(seed, b, n_i, my_i, n0, myl, rho)
            K, dpos = pa.kkt_leaf_assemble(n_i, W)
            Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T if n0 else None, F=F if myl else None)
            d = pa.gen_diagonal(seed, b, n_i, diag_lo, diag_hi)
            diag = np.concatenate([d, -dual_reg * np.ones(my_i)])
            K.val[dpos] = diag
            self.blocks.append(dict(W=W, T=T, F=F, K=K, dpos=dpos, Bt=Bt, diag=diag, c=c, xs=xs))
        self.F0, self.c0, self.x0s = pa.gen_root(seed, n0, myl)
        self.x_diag0 = pa.gen_diagonal(seed, 0, n0, diag_lo, diag_hi)

    @property
    def n_leaf(self):
        return self.n_i + self.my_i

    def K_scipy(self, b):
        K = self.blocks[b]["K"]
        return sp.csr_matrix((K.val.copy(), K.colidx, K.rowptr), shape=(K.nrows, K.ncols))

    def K_full(self, b):
        K = self.K_scipy(b)
        return (K + sp.tril(K, -1).T).tocsc()

    def Bt_scipy(self, b):
        return self.blocks[b]["Bt"].to_scipy()

    def oracle_leaf(self, b, **kw):
        s = orc.OracleLdl(self.K_scipy(b), n_primal=self.n_i, **kw)
        s.matrixChanged()
        return s

    def oracle_schur(self, blocks=None):
        """SC after assembleLocalKKT over the given blocks (row-major, lower authoritative)."""
        SC = np.zeros((self.S, self.S))
        for b in (range(self.N) if blocks is None else blocks):
            orc.add_term_to_schur_compl_blocked(SC, self.oracle_leaf(b), self.Bt_scipy(b))
        return SC

    def oracle_finalize(self, SC):
        return orc.finalize_kkt_dense(SC, self.n0, 0, self.myl, 0, self.x_diag0, F0=self.F0.to_scipy())


def hip_lower_as_rowmajor(buf, S):
    """The HIP path writes SC column-major with the lower triangle valid; return the row-major lower triangle."""
    A = np.asarray(buf).reshape(S, S)   # A[c][r] = SC(r, c)
    return np.tril(A.T)
