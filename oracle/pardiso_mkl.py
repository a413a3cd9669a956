"""pardiso_mkl.py — TEST INFRASTRUCTURE ONLY.

Restatement of the reference's PARDISO call sequence (PardisoSolver.C:51-135 first call / CSR-lower -> upper transpose,
:141-205 matrixChanged = phase 12, :207-352 solve = phase 33 incl. the multi-RHS form, :361-367 inertia from
iparm[21..22]; settings PardisoProjectSolver.C:68-77: iparm[1]=2 METIS, iparm[7]=2 refinement steps, iparm[10]=1 scaling,
iparm[12]=2 matching; mtype = -2) on top of the PARDISO that ships with MKL (libmkl_rt.so), which is present in this
image but is NOT the Schenk PARDISO the reference links (that one is licence-gated and absent).  Used (a) to pin
oracle_ldl.c against an independent production solver of the same algorithm class and (b) as the CPU baseline in
bench.py when the library can be loaded.  Never imported by the product.
"""
import ctypes as C
import os

import numpy as np

_CANDIDATES = ["/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "libmkl_rt.so", "libmkl_rt.so.1", "libmkl_rt.so.2"]
_mkl = None


def available():
    global _mkl
    if _mkl is not None:
        return _mkl is not False
    # MKL's default Intel threading layer needs libiomp5, which this image lacks: with it every multi-threaded PARDISO
    # call fails with error -2/-3.  The GNU layer (libgomp) works.
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    for c in _CANDIDATES:
        try:
            _mkl = C.CDLL(c, mode=C.RTLD_GLOBAL)
            _mkl.pardiso  # noqa: B018
            return True
        except (OSError, AttributeError):
            continue
    _mkl = False
    return False


def set_threads(n):
    if available():
        try:
            _mkl.MKL_Set_Num_Threads(C.c_int(n))
        except AttributeError:
            pass


class MklPardisoSolver:
    """DoubleLinearSolver-shaped wrapper: matrixChanged() / solve(x or (nrhs,n)) / get_inertia()."""

    def __init__(self, K_lower_csr, num_threads=1, matching=True, reuse_analysis=False):
        """matching=False: iparm[12] = 0 (scaling stays) - on a whole arrowhead KKT matrix MKL's matching triples the fill (8 blocks of
        configs[1]: 364 M against 166 M factor entries); the analysis then does not depend on the values and reuse_analysis=True runs
        phase 11 once and phase 22 per matrixChanged()."""
        if not available():
            raise RuntimeError("libmkl_rt.so not found")
        self.K = K_lower_csr
        self.n = K_lower_csr.shape[0]
        # PardisoSolver::firstCall: transpose lower CSR -> upper CSR (fortran-indexed), remember where values go
        import scipy.sparse as sp
        low = sp.csr_matrix(K_lower_csr)
        tag = sp.csr_matrix((np.arange(1, low.nnz + 1, dtype=np.float64), low.indices, low.indptr), shape=low.shape)
        up = sp.csr_matrix(tag.T)
        up.sort_indices()
        self.map = up.data.astype(np.int64) - 1
        self.ia = (up.indptr + 1).astype(np.int32)
        self.ja = (up.indices + 1).astype(np.int32)
        self.a = np.zeros(low.nnz)
        self.pt = np.zeros(64, dtype=np.int64)
        self.iparm = np.zeros(64, dtype=np.int32)
        self.mtype = C.c_int(-2)
        _mkl.pardisoinit(self.pt.ctypes, C.byref(self.mtype), self.iparm.ctypes)
        ip = self.iparm
        ip[0] = 1
        ip[1] = 2      # METIS
        ip[2] = 0
        ip[7] = 2      # max iterative refinement steps
        ip[9] = 8      # pivot perturbation 1e-8 (default for mtype -2)
        ip[10] = 1     # scaling
        ip[12] = 1 if matching else 0     # matching (MKL supports 0/1; the reference asks Schenk-PARDISO for 2)
        ip[17] = -1
        ip[20] = 1     # Bunch-Kaufman 1x1/2x2 pivoting (default for mtype -2)
        ip[34] = 0     # fortran indexing, like the reference
        set_threads(num_threads)
        self.first = True
        self.reuse_analysis = bool(reuse_analysis) and not matching

    def _call(self, phase, nrhs, b, x):
        err = C.c_int(0)
        one = C.c_int(1)
        msg = C.c_int(0)
        n = C.c_int(self.n)
        ph = C.c_int(phase)
        nr = C.c_int(nrhs)
        dummy = np.zeros(1, dtype=np.int32)
        _mkl.pardiso(self.pt.ctypes, C.byref(one), C.byref(one), C.byref(self.mtype), C.byref(ph), C.byref(n),
                     self.a.ctypes, self.ia.ctypes, self.ja.ctypes, dummy.ctypes, C.byref(nr), self.iparm.ctypes,
                     C.byref(msg), b.ctypes if b is not None else dummy.ctypes, x.ctypes if x is not None else dummy.ctypes,
                     C.byref(err))
        if err.value != 0:
            raise RuntimeError(f"MKL pardiso phase {phase} error {err.value}")

    def matrixChanged(self):
        self.a[:] = np.asarray(self.K.data)[self.map]
        if self.reuse_analysis:
            if self.first:
                self._call(11, 1, None, None)
            self._call(22, 1, None, None)
        else:
            self._call(12, 1, None, None)    # analysis + numerical factorisation every time, as the reference does
        self.first = False

    def solve(self, x):
        X = x.reshape(-1, self.n)
        sol = np.zeros_like(X)
        self._call(33, X.shape[0], X, sol)
        X[...] = sol
        return x

    def get_inertia(self):
        return int(self.iparm[21]), int(self.iparm[22]), 0

    def close(self):
        if self.pt is not None:
            try:
                self._call(-1, 1, None, None)
            except Exception:
                pass
            self.pt = None

    def __del__(self):
        self.close()
