"""oracle.py — TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy/scipy + oracle_ldl.c) of PIPS-IPM++'s KKT factor+solve hot path, function by function, used as
the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (pips-ipmpp_amd/) never does.

PARITY STATUS: the third-party arithmetic the reference delegates to (PARDISO / MA27 / MA57 for the leaves) is not in
/root/reference and has no golden vectors there: the leaf factorisation restated in oracle_ldl.c is
pinned against scipy.sparse.linalg.splu (SuperLU), dense numpy eigenvalue inertia and, where libmkl_rt is present,
against MKL PARDISO called with the reference's own iparm settings (tests/test_oracle_pinning.py).  The dense root uses
the very LAPACK routine the reference calls (dsytrf_/dsytrs_, DeSymIndefSolver.C:78,112) through scipy.  The assembly
logic (Schur accumulation, finalize, three-phase solve) follows the reference line by line as cited per function.

Conventions (identical to the reference):
  * matrices: scipy CSR, 0-based;  K_i lower-triangular CSR with an explicit diagonal in every row
  * SC: dense S x S numpy array, row-major, LOWER triangle authoritative (DenseSymmetricMatrix, DenseStorage.C:64-83)
  * S = n0 + my0 + myl + mzl, border column order [x0 | y0 (empty) | linking eq | linking ineq]
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.linalg.lapack as lapack
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    """gcc -O2 oracle_ldl.c -> liboracle.so (idempotent)."""
    src = os.path.join(_HERE, "oracle_ldl.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_ldl_analyze.restype = C.c_void_p
        _lib.oracle_ldl_nnzL.restype = C.c_long
        _lib.oracle_ldl_nnzL.argtypes = [C.c_void_p]
        _lib.oracle_ldl_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------------------------------------------------
# a1 / a2  leaf KKT assembly and diagonal updates
# ----------------------------------------------------------------------------------------------------------------------
def leaf_kkt(nx, B, D=None, Q=None):
    """DistributedLeafLinearSystem::create_kkt (DistributedLeafLinearSystem.C:44-72): lower CSR of
    [Q+Dx B^T D^T; B 0 0; D 0 0] with a (zero) diagonal entry stored in every row."""
    my = B.shape[0] if B is not None else 0
    mz = D.shape[0] if D is not None else 0
    n = nx + my + mz
    rows = [[sp.tril(Q, format="csr") if Q is not None else sp.csr_matrix((nx, nx)), sp.csr_matrix((nx, my + mz))]]
    if my:
        rows.append([B, sp.csr_matrix((my, my + mz))])
    if mz:
        rows.append([D, sp.csr_matrix((mz, my + mz))])
    lower = sp.bmat(rows, format="csr")
    lower.sort_indices()
    indptr, indices, data = [0], [], []
    for i in range(n):
        c = lower.indices[lower.indptr[i]:lower.indptr[i + 1]]
        v = lower.data[lower.indptr[i]:lower.indptr[i + 1]]
        indices.extend(c)
        data.extend(v)
        if i not in c:  # explicit (zero) diagonal entry, always the last of its row in a lower-triangular matrix
            indices.append(i)
            data.append(0.0)
        indptr.append(len(indices))
    return sp.csr_matrix((np.array(data, dtype=np.float64), np.array(indices, dtype=np.int32),
                          np.array(indptr, dtype=np.int32)), shape=(n, n))


def diag_positions(K):
    n = K.shape[0]
    pos = np.zeros(n, dtype=np.int64)
    for i in range(n):
        s, e = K.indptr[i], K.indptr[i + 1]
        j = np.nonzero(K.indices[s:e] == i)[0]
        pos[i] = s + j[0]
    return pos


def put_diagonals(K, dpos, nx, my, mz, primal_diag, nomega_inv=None, primal_reg=0.0, dual_y_reg=0.0, dual_z_reg=0.0):
    """put_primal_diagonal / clear_dual_equality_diagonal / put_dual_inequalites_diagonal /
    add_regularization_local_kkt (DistributedLeafLinearSystem.C:88-143)."""
    K.data[dpos[:nx]] = primal_diag + primal_reg
    K.data[dpos[nx:nx + my]] = 0.0 - dual_y_reg
    if mz:
        K.data[dpos[nx + my:]] = nomega_inv - dual_z_reg
    return K


# ----------------------------------------------------------------------------------------------------------------------
# a3 / a7  leaf solver (DoubleLinearSolver): matrixChanged / solve(nrhs) / get_inertia
# ----------------------------------------------------------------------------------------------------------------------
class OracleLdl:
    def __init__(self, K, perm=None, n_primal=-1, thr_rel=1e-13, repl_rel=1e-8, refine_steps=1):
        self.K = K
        self.n = K.shape[0]
        self.krow = np.ascontiguousarray(K.indptr, dtype=np.int32)
        self.jcol = np.ascontiguousarray(K.indices, dtype=np.int32)
        self.perm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        self.n_primal = n_primal
        self.psign = None
        if n_primal >= 0:
            self.psign = np.where(np.arange(self.n) < n_primal, 1, -1).astype(np.int8)
        self.thr_rel, self.repl_rel, self.refine_steps = thr_rel, repl_rel, refine_steps
        self._f = C.c_void_p(lib().oracle_ldl_analyze(C.c_int(self.n), _p(self.krow), _p(self.jcol), _p(self.perm)))
        self.vals = None

    def nnzL(self):
        return int(lib().oracle_ldl_nnzL(self._f))

    def matrixChanged(self):
        self.vals = np.ascontiguousarray(self.K.data, dtype=np.float64).copy()
        amax = float(np.abs(self.vals).max()) if self.vals.size else 1.0
        amax = amax if amax > 0 else 1.0
        lib().oracle_ldl_factor(self._f, _p(self.vals), _p(self.psign), C.c_double(self.thr_rel), C.c_double(self.repl_rel),
                                C.c_double(self.repl_rel * amax), _p(self._pivot_reference()))

    def _pivot_reference(self):
        """|a_kk| for primal rows, |a_kk| + sum_j K_kj^2/|K_jj| over the primal neighbours for dual rows (the magnitude of the
        normal-equation diagonal the pivot is built from)."""
        K = self.K
        diag = np.abs(K.diagonal())
        pref = diag.copy()
        if self.n_primal >= 0:
            for i in range(self.n_primal, self.n):
                s, e = K.indptr[i], K.indptr[i + 1]
                cols, vals = K.indices[s:e], K.data[s:e]
                m = (cols < self.n_primal) & (diag[cols] > 0)
                pref[i] += np.sum(vals[m] ** 2 / diag[cols[m]])
        return np.ascontiguousarray(pref)

    def _solve_raw(self, x):
        nrhs = 1 if x.ndim == 1 else x.shape[0]
        lib().oracle_ldl_solve(self._f, C.c_int(nrhs), _p(x), C.c_int(x.shape[-1]))

    def solve(self, x):
        """in place; x (n,) or (nrhs, n) — one right-hand side per row (PardisoSolver.C:276-352).  Iterative refinement
        like PARDISO's iparm[7] (PardisoProjectSolver.C:72)."""
        assert x.dtype == np.float64 and x.flags.c_contiguous
        X = x.reshape(-1, self.n)
        B = X.copy()
        self._solve_raw(X)
        for _ in range(self.refine_steps):
            R = np.empty_like(X)
            for k in range(X.shape[0]):
                lib().oracle_sym_residual(C.c_int(self.n), _p(self.krow), _p(self.jcol), _p(self.vals), _p(X[k]), _p(B[k]),
                                          _p(R[k]))
            self._solve_raw(R)
            X += R
        return x

    def get_inertia(self):
        out = np.zeros(3, dtype=np.int32)
        lib().oracle_ldl_inertia(self._f, _p(out))
        return int(out[0]), int(out[1]), int(out[2])

    def __del__(self):
        try:
            if self._f:
                lib().oracle_ldl_free(self._f)
                self._f = None
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# a5 / a6  Schur complement contribution of one leaf, blocked multi-RHS formulation
# ----------------------------------------------------------------------------------------------------------------------
def border_transposed(nx, my, mz, n0, n_empty, R=None, A=None, Cm=None, F=None, G=None):
    """Br_i^T as CSR (S x N_i): BorderBiBlock {R,A,C | n_empty | F^T, G^T} (RACFG_BLOCK.h:13-53,
    DistributedLeafLinearSystem.C:214-252)."""
    N = nx + my + mz
    top = []
    top.append(R.T.tocsr() if R is not None else sp.csr_matrix((n0, nx)))
    top.append(A.T.tocsr() if A is not None else sp.csr_matrix((n0, my)))
    if mz:
        top.append(Cm.T.tocsr() if Cm is not None else sp.csr_matrix((n0, mz)))
    rows = [sp.hstack(top, format="csr")]
    if n_empty:
        rows.append(sp.csr_matrix((n_empty, N)))
    if F is not None and F.shape[0]:
        rows.append(sp.hstack([F, sp.csr_matrix((F.shape[0], my + mz))], format="csr"))
    if G is not None and G.shape[0]:
        rows.append(sp.hstack([G, sp.csr_matrix((G.shape[0], my + mz))], format="csr"))
    return sp.vstack(rows, format="csr")


def add_term_to_schur_compl_blocked(SC, solver, Bt, blocksize=20):
    """DistributedLeafLinearSystem::addTermToSchurComplBlocked (DistributedLeafLinearSystem.C:214-252) →
    addBiTLeftKiBiRightToResBlockedParallelSolvers (DistributedLinearSystem.C:766-1047): walk the non-empty border
    columns in chunks of <= blocksize, dense-ify them ("columns lie as rows", :895-901), multi-RHS solve (:903), then
    addLeftBorderTimesDenseColsToResTranspDense (:1115-1175):  SC[col_id][:] -= Br^T * (K^-1 Br e_col)."""
    nnz_per_col = np.diff(Bt.indptr)
    cols = np.nonzero(nnz_per_col > 0)[0]  # empty columns are skipped (:870-874)
    for s in range(0, len(cols), blocksize):
        ids = cols[s:s + blocksize]
        dense = np.ascontiguousarray(Bt[ids].toarray())       # (chunk, N_i): one border column per row
        solver.solve(dense)                                    # in place
        SC[ids, :] -= (Bt @ dense.T).T                         # row = Schur column id
    return SC


# ----------------------------------------------------------------------------------------------------------------------
# a8 / a9 / a10  root: zero, reduce, finalize
# ----------------------------------------------------------------------------------------------------------------------
def reduce_kkt_dense(parts):
    """DistributedRootLinearSystem::reduceKKTdense (:860-881): MPI_Allreduce(SUM) of the lower triangle."""
    out = np.zeros_like(parts[0])
    for p in parts:
        out += np.tril(p)
    return out


def finalize_kkt_dense(SC, n0, my0, myl, mzl, x_diag, A0=None, F0=None, G0=None, C0=None, z_diag=None, z_diag_link=None):
    """sLinsysRootAug::finalizeKKTdense (sLinsysRootAug.C:1769-1796): SC += diag(xDiag) on the x0 block
    (schur_complement_put_primal_block :229-268), -= C0^T diag(zDiag)^-1 C0 (lower part only, :1276-1294), += A0 at row
    n0 (:270-320), += F0 at row n0+my0, += G0 at row n0+my0+myl and zDiagLinkCons on that diagonal."""
    idx = np.arange(n0)
    SC[idx, idx] += x_diag
    if C0 is not None and C0.shape[0]:
        ctdc = (C0.T @ sp.diags(1.0 / z_diag) @ C0).toarray()
        SC[:n0, :n0] -= np.tril(ctdc)
    if A0 is not None and my0:
        SC[n0:n0 + my0, :n0] += A0.toarray()
    if F0 is not None and myl:
        SC[n0 + my0:n0 + my0 + myl, :n0] += F0.toarray()
    if G0 is not None and mzl:
        r0 = n0 + my0 + myl
        SC[r0:r0 + mzl, :n0] += G0.toarray()
        SC[np.arange(r0, r0 + mzl), np.arange(r0, r0 + mzl)] += z_diag_link
    return SC


# ----------------------------------------------------------------------------------------------------------------------
# a11  dense root solver = the reference's own LAPACK calls
# ----------------------------------------------------------------------------------------------------------------------
class DenseRootSolver:
    """DeSymIndefSolver (DeSymIndefSolver.C:56-168): copy SC, dsytrf_('U') on the column-major view of the row-major
    lower triangle, dsytrs_, inertia from ipiv / D."""

    def __init__(self, n):
        self.n = n

    def matrixChanged(self, SC_rowmajor_lower):
        a = np.asfortranarray(SC_rowmajor_lower.T)          # column-major view: lower of row-major == upper here
        self.ldu, self.ipiv, info = lapack.dsytrf(a, lower=0)
        if info != 0:
            print("DenseRootSolver: dsytrf info", info)

    def solve(self, x):
        sol, info = lapack.dsytrs(self.ldu, self.ipiv, x.reshape(self.n, 1) if x.ndim == 1 else x.T, lower=0)
        x[...] = sol.reshape(x.shape) if x.ndim == 1 else sol.T
        return x

    def get_inertia(self):
        """DeSymIndefSolver::get_inertia (:135-168): 1x1 pivots by sign, 2x2 pivots contribute one of each sign."""
        n, ipiv, a = self.n, self.ipiv, self.ldu
        pos = neg = zero = 0
        k = 0
        # scipy returns 0-based-agnostic ipiv in LAPACK convention (1-based, negative for 2x2 blocks)
        while k < n:
            if ipiv[k] > 0:
                d = a[k, k]
                pos += d > 0
                neg += d < 0
                zero += d == 0
                k += 1
            else:
                pos += 1
                neg += 1
                k += 2
        return int(pos), int(neg), int(zero)


# ----------------------------------------------------------------------------------------------------------------------
# a12 / a13  three-phase solve
# ----------------------------------------------------------------------------------------------------------------------
def solve_compressed(b0, bs, leaf_solvers, Bts, root_solver, n0, my0, mz0, myl, mzl, C0=None, z_diag_reg=None):
    """DistributedLinearSystem::solveCompressed (DistributedLinearSystem.C:409-420).
    b0: [x0 | y0 | z0 | ylink | zlink] (length n0+my0+mz0+myl+mzl), bs: list of per-leaf vectors; all modified in place.
      Lsolve  (sLinsysRootAug.C:323-344)  z_i = K_i^-1 b_i ; b0 -= Br_i^T z_i   (addLniziLinkCons, Leaf.C:171-212)
      Dsolve  (:347-354, 384-466)         eliminate z0 through C0, solve with SC, recover z0
      Ltsolve (:356-365)                  b_i -= K_i^-1 Br_i x0                   (LniTransMult, DistributedLinearSystem.C:430-483)
    The reduced border vector order is [x0 | y0 | ylink | zlink] (z0 dropped)."""
    S = n0 + my0 + myl + mzl
    red = np.concatenate([np.arange(n0 + my0), np.arange(n0 + my0 + mz0, n0 + my0 + mz0 + myl + mzl)])
    for bi, sol, Bt in zip(bs, leaf_solvers, Bts):
        sol.solve(bi)
        b0[red] -= Bt @ bi
    # Dsolve / solveReducedLinkCons
    rhs = b0[red].copy()
    if mz0:
        b3 = b0[n0 + my0:n0 + my0 + mz0] / z_diag_reg
        rhs[:n0] -= C0.T @ b3
    root_solver.solve(rhs)
    assert rhs.shape[0] == S
    b0[red] = rhs
    if mz0:
        x3 = b0[n0 + my0:n0 + my0 + mz0] - C0 @ rhs[:n0]
        b0[n0 + my0:n0 + my0 + mz0] = x3 / z_diag_reg
    x0 = b0[red]
    for bi, sol, Bt in zip(bs, leaf_solvers, Bts):
        t = Bt.T @ x0
        sol.solve(t)
        bi -= t
    return b0, bs
