"""ctypes binding of libpipship.so (the C ABI in include/pips_hip.h).

This module is plumbing for the tests and bench.py: it mirrors the reference's plug-in surface
(`DoubleLinearSolver`: matrixChanged / solve / get_inertia, PIPS-IPM/Core/LinearSolvers/DoubleLinearSolver.h:24-72)
on top of the C entry points.  There is NO CPU fallback: if the shared library is missing, or no GPU is visible when a
compute call is made, the call raises.
"""
import ctypes as C
import sys
import os

import numpy as np

try:  # torch (if present) must load its HIP runtime first so that libpipship binds to the same libamdhip64.so.7
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIPS_HIP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "libpipship.so")   # override: experiment builds


class PipsHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise PipsHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The MI355X backend has no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_vp = C.c_void_p

lib.pips_hip_last_error.restype = C.c_char_p
lib.pips_hip_device_count.restype = C.c_int
lib.pips_hip_host_wait_count.restype = C.c_longlong

# every symbol include/pips_hip.h declares (tests/test_capi_symbols.py checks the header against this list)
SYMBOLS = [
    "pips_hip_last_error", "pips_hip_device_count",
    "pips_hip_ldl_create", "pips_hip_ldl_set_inertia_hint", "pips_hip_ldl_set_pivot_rule", "pips_hip_ldl_set_refinement", "pips_hip_ldl_set_refinement_backward_error", "pips_hip_ldl_set_deterministic",
    "pips_hip_ldl_analyze", "pips_hip_ldl_factor", "pips_hip_ldl_solve", "pips_hip_ldl_inertia", "pips_hip_ldl_info",
    "pips_hip_ldl_get_perm", "pips_hip_ldl_set_border", "pips_hip_ldl_factor_schur", "pips_hip_ldl_destroy",
    "pips_hip_ldl_solve_dev", "pips_hip_ldl_solve_sparse", "pips_hip_ldl_factor_schur_batch", "pips_hip_ldl_solve_batch", "pips_hip_ldl_solve_batch_dev",
    "pips_hip_ldl_inertia_batch",
    "pips_hip_dense_ldl_create", "pips_hip_dense_ldl_factor", "pips_hip_dense_ldl_factor_dev", "pips_hip_dense_ldl_solve",
    "pips_hip_dense_ldl_solve_dev", "pips_hip_dense_ldl_inertia", "pips_hip_dense_ldl_set_pivoting", "pips_hip_dense_ldl_set_distributed", "pips_hip_dense_ldl_destroy", "pips_root_plan_build", "pips_hip_host_wait_count", "pips_hip_host_wait_sites",
    "pips_hip_batch_create", "pips_hip_batch_set_block", "pips_hip_batch_set_options", "pips_hip_batch_set_schur_mode", "pips_hip_batch_set_deterministic", "pips_hip_batch_get_schur_mode", "pips_hip_batch_add_regularization", "pips_hip_batch_set_refinement",
    "pips_hip_batch_last_refinement_steps", "pips_hip_batch_set_refinement_backward_error",
    "pips_hip_batch_last_refinement_measure", "pips_hip_batch_analyze",
    "pips_hip_batch_set_values", "pips_hip_batch_set_diagonals_dev", "pips_hip_batch_set_diagonals", "pips_hip_batch_factor",
    "pips_hip_batch_solve_dev", "pips_hip_batch_solve", "pips_hip_batch_border_tmult_dev", "pips_hip_batch_border_mult_dev",
    "pips_hip_batch_inertia", "pips_hip_batch_info", "pips_hip_batch_sync", "pips_hip_batch_set_timing",
    "pips_hip_batch_get_timing", "pips_hip_batch_destroy",
    "pips_hip_kkt_create", "pips_hip_kkt_create_sparse", "pips_hip_kkt_get_schur_sparse", "pips_hip_kkt_sparse_root_info", "pips_hip_kkt_factorize", "pips_hip_kkt_set_root_regularization", "pips_hip_kkt_solve_compressed", "pips_hip_kkt_get_schur",
    "pips_hip_kkt_set_root_inequalities", "pips_hip_kkt_set_zdiag0_dev",
    "pips_hip_kkt_root_inertia", "pips_hip_kkt_get_timing", "pips_hip_kkt_last_ltsolve_from_factor", "pips_hip_kkt_last_solve_path", "pips_hip_kkt_set_solve_check", "pips_hip_kkt_solve_check_counts", "pips_hip_kkt_set_solve_graph", "pips_hip_kkt_set_root_stream", "pips_hip_kkt_solve_graph_stats", "pips_hip_kkt_set_root_pivoting", "pips_hip_kkt_destroy",
    "pips_hip_malloc", "pips_hip_free", "pips_hip_memcpy_h2d", "pips_hip_memcpy_d2h", "pips_hip_memset",
    "pips_hip_comm_unique_id", "pips_hip_comm_create", "pips_hip_comm_create_external", "pips_hip_comm_set_external_rsag", "pips_hip_comm_set_external_broadcast", "pips_hip_broadcast", "pips_hip_comm_has_broadcast", "pips_hip_allreduce_sum_rsag", "pips_hip_all_gather", "pips_hip_comm_size", "pips_hip_allreduce_sum", "pips_hip_comm_destroy",
    "pips_hip_vec_axpy", "pips_hip_vec_axpby", "pips_hip_vec_scale", "pips_hip_vec_copy", "pips_hip_vec_set",
    "pips_hip_vec_add_const", "pips_hip_vec_mul", "pips_hip_vec_div", "pips_hip_vec_add_product", "pips_hip_vec_add_quotient",
    "pips_hip_vec_divide_some", "pips_hip_vec_select_nonzeros", "pips_hip_vec_safe_invert", "pips_hip_vec_gondzio_projection", "pips_hip_vec_dot",
    "pips_hip_vec_one_norm", "pips_hip_vec_inf_norm", "pips_hip_vec_min", "pips_hip_vec_sumsq_scaled", "pips_hip_vec_stepbound",
    "pips_hip_vec_find_blocking", "pips_hip_vec_weighted_stepbounds", "pips_hip_vec_dot_shifted", "pips_ipm_create", "pips_ipm_create_rank", "pips_ipm_create_general", "pips_ipm_get_dims", "pips_ipm_get_iterate", "pips_ipm_get_stats2", "pips_ipm_mult", "pips_ipm_outer_solve", "pips_ipm_solve", "pips_ipm_set_gondzio", "pips_ipm_set_option", "pips_ipm_set_free_variables", "pips_ipm_get_solution", "pips_ipm_get_trace", "pips_ipm_get_stats", "pips_ipm_destroy",
    "pips_gdx_read_block", "pips_gdx_block_counts", "pips_gdx_block_vector", "pips_gdx_block_matrix", "pips_gdx_block_destroy",
    "pips_gen_row_nnz", "pips_gen_block", "pips_gen_root", "pips_gen_diagonal", "pips_kkt_leaf_assemble",
    "pips_border_assemble", "pips_symbolic_probe", "pips_symbolic_probe_hubs", "pips_map_children_to_ranks",
]


def _check(rc, what):
    if rc != 0:
        msg = lib.pips_hip_last_error()
        raise PipsHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def device_count():
    return int(lib.pips_hip_device_count())


def host_wait_count():
    """host waits for the device inside the library so far (a diagnostic: take the difference around a call sequence)"""
    return int(lib.pips_hip_host_wait_count())


def host_wait_sites():
    """{"file:line": count} of the host waits inside the library so far"""
    buf = C.create_string_buffer(16384)
    lib.pips_hip_host_wait_sites(buf, 16384)
    out = {}
    for ln in buf.value.decode().splitlines():
        k, v = ln.rsplit(" ", 1)
        out[k] = int(v)
    return out


def _ptr(a):
    """numpy array / torch tensor / int / None -> void*"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    if torch is not None and isinstance(a, torch.Tensor):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------------------------------------------------------------
# host harness helpers: synthetic arrowhead LP, leaf KKT and border assembly
# ----------------------------------------------------------------------------------------------------------------------
class Csr:
    """Plain CSR triple, row-major, 0-based, int32/fp64 (SparseStorage.h:45-50)."""

    def __init__(self, nrows, ncols, rowptr, colidx, val):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.rowptr, self.colidx, self.val = _i32(rowptr), _i32(colidx), _f64(val)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.colidx, self.rowptr), shape=(self.nrows, self.ncols))


def gen_block(seed, block, n_i, my_i, n0, myl, rho=1e-3):
    """W_i (my_i x n_i), T_i (my_i x n0), F_i (myl x n_i), c_i, x*_i of block `block` >= 1."""
    kw = int(lib.pips_gen_row_nnz(C.c_int(n_i), C.c_double(rho)))
    kw = min(kw, n_i)
    W = Csr(my_i, n_i, np.zeros(my_i + 1, np.int32), np.zeros(my_i * kw, np.int32), np.zeros(my_i * kw))
    kt = min(2, n0)
    T = Csr(my_i, n0, np.zeros(my_i + 1, np.int32), np.zeros(my_i * kt, np.int32), np.zeros(my_i * kt))
    kf = min(4, n_i)
    F = Csr(myl, n_i, np.zeros(myl + 1, np.int32), np.zeros(myl * kf, np.int32), np.zeros(myl * kf))
    c = np.zeros(n_i)
    xs = np.zeros(n_i)
    rc = lib.pips_gen_block(C.c_uint64(seed), C.c_int(block), C.c_int(n_i), C.c_int(my_i), C.c_int(n0), C.c_int(myl),
                            C.c_double(rho), _ptr(W.rowptr), _ptr(W.colidx), _ptr(W.val), _ptr(T.rowptr), _ptr(T.colidx),
                            _ptr(T.val), _ptr(F.rowptr), _ptr(F.colidx), _ptr(F.val), _ptr(c), _ptr(xs))
    _check(rc, "pips_gen_block")
    return W, T, F, c, xs


def gen_root(seed, n0, myl):
    kf = min(2, n0)
    F0 = Csr(myl, n0, np.zeros(myl + 1, np.int32), np.zeros(myl * kf, np.int32), np.zeros(myl * kf))
    c0 = np.zeros(n0)
    x0 = np.zeros(n0)
    _check(lib.pips_gen_root(C.c_uint64(seed), C.c_int(n0), C.c_int(myl), _ptr(F0.rowptr), _ptr(F0.colidx), _ptr(F0.val),
                             _ptr(c0), _ptr(x0)), "pips_gen_root")
    return F0, c0, x0


def gen_diagonal(seed, block, n, lo=-4.0, hi=4.0):
    d = np.zeros(n)
    _check(lib.pips_gen_diagonal(C.c_uint64(seed), C.c_int(block), C.c_int(n), C.c_double(lo), C.c_double(hi), _ptr(d)),
           "pips_gen_diagonal")
    return d


def kkt_leaf_assemble(nx, B, D=None, Q=None):
    """Lower CSR of K_i = [Q+Dx B^T D^T; B 0 0; D 0 0] with explicit diagonal (create_kkt,
    DistributedLeafLinearSystem.C:44-72).  Returns (Csr, diag_pos)."""
    my = B.nrows if B is not None else 0
    mz = D.nrows if D is not None else 0
    n = nx + my + mz
    rowptr = np.zeros(n + 1, np.int32)

    def trip(M):
        return (None, None, None) if M is None else (_ptr(M.rowptr), _ptr(M.colidx), _ptr(M.val))

    args = [C.c_int(nx), C.c_int(my), C.c_int(mz), *trip(Q), *trip(B), *trip(D)]
    _check(lib.pips_kkt_leaf_assemble(*args, _ptr(rowptr), None, None, None), "pips_kkt_leaf_assemble")
    nnz = int(rowptr[n])
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    diag_pos = np.zeros(n, np.int32)
    _check(lib.pips_kkt_leaf_assemble(*args, _ptr(rowptr), _ptr(colidx), _ptr(val), _ptr(diag_pos)),
           "pips_kkt_leaf_assemble")
    return Csr(n, n, rowptr, colidx, val), diag_pos


def border_assemble(nx, my, mz, n0, n_empty, R=None, A=None, Cm=None, F=None, G=None):
    """Br_i^T as CSR with S = n0 + n_empty + myl + mzl rows over N_i = nx+my+mz columns (BorderBiBlock, RACFG_BLOCK.h)."""
    myl = F.nrows if F is not None else 0
    mzl = G.nrows if G is not None else 0
    S = n0 + n_empty + myl + mzl

    def trip(M):
        return (None, None, None) if M is None else (_ptr(M.rowptr), _ptr(M.colidx), _ptr(M.val))

    rowptr = np.zeros(S + 1, np.int32)
    args = [C.c_int(nx), C.c_int(my), C.c_int(mz), C.c_int(n0), C.c_int(n_empty), C.c_int(myl), C.c_int(mzl), *trip(R),
            *trip(A), *trip(Cm), *trip(F), *trip(G)]
    _check(lib.pips_border_assemble(*args, _ptr(rowptr), None, None), "pips_border_assemble")
    nnz = int(rowptr[S])
    colidx = np.zeros(nnz, np.int32)
    val = np.zeros(nnz)
    _check(lib.pips_border_assemble(*args, _ptr(rowptr), _ptr(colidx), _ptr(val)), "pips_border_assemble")
    return Csr(S, nx + my + mz, rowptr, colidx, val)


def map_children_to_ranks(n_children, n_ranks):
    """Block -> rank map with the contract of DistributedTree::assignProcesses (contiguous, monotone, balanced)."""
    m = np.zeros(max(n_children, 1), np.int32)
    _check(lib.pips_map_children_to_ranks(C.c_int(n_children), C.c_int(n_ranks), _ptr(m)), "pips_map_children_to_ranks")
    return m[:n_children]


def symbolic_probe(K, n_primal=-1, Bt=None, force_n_head=-1, want_perm=False):
    what = np.zeros(17, np.int64)
    perm = np.zeros(K.nrows, np.int32) if want_perm else None
    cc = np.zeros(K.nrows, np.int32) if want_perm else None
    S = Bt.nrows if Bt is not None else 0
    _check(lib.pips_symbolic_probe(C.c_int(K.nrows), C.c_int(n_primal), _ptr(K.rowptr), _ptr(K.colidx), C.c_int(S),
                                   _ptr(Bt.rowptr) if Bt is not None else None,
                                   _ptr(Bt.colidx) if Bt is not None else None, C.c_int(force_n_head), _ptr(what),
                                   C.c_int(17), _ptr(perm), _ptr(cc)), "pips_symbolic_probe")
    info = dict(nnzL=int(what[0]), n=int(what[1]), n_head=int(what[2]), m=int(what[3]), n_sn=int(what[4]),
                n_levels=int(what[5]), flops_factor=int(what[6]), flops_border=int(what[7]), arena_bytes=int(what[8]),
                ntc=int(what[9]), upd_bytes=int(what[10]), multifrontal=int(what[13]), border_split=int(what[14]),
                update_matrix_doubles=int(what[15]), border_row_arena_doubles=int(what[16]))
    if want_perm:
        info["perm"] = perm
        info["colcount"] = cc
    return info


def symbolic_probe_hubs(K, hubs, n_primal=-1, min_size=48):
    """Symbolic analysis under the sparse root's dissection order (hubs last); returns the info dict with perm and colcount."""
    what = np.zeros(11, np.int64)
    perm = np.zeros(K.nrows, np.int32)
    cc = np.zeros(K.nrows, np.int32)
    hubs = _i32(hubs)
    _check(lib.pips_symbolic_probe_hubs(C.c_int(K.nrows), C.c_int(n_primal), _ptr(K.rowptr), _ptr(K.colidx), C.c_int(len(hubs)),
                                        _ptr(hubs), C.c_int(min_size), _ptr(what), C.c_int(11), _ptr(perm), _ptr(cc)),
           "pips_symbolic_probe_hubs")
    return dict(nnzL=int(what[0]), n=int(what[1]), n_head=int(what[2]), m=int(what[3]), n_sn=int(what[4]), n_levels=int(what[5]),
                flops_factor=int(what[6]), flops_border=int(what[7]), arena_bytes=int(what[8]), ntc=int(what[9]),
                upd_bytes=int(what[10]), perm=perm, colcount=cc)


# ----------------------------------------------------------------------------------------------------------------------
# DoubleLinearSolver mirror: sparse leaf solver
# ----------------------------------------------------------------------------------------------------------------------
class HipLdlSolver:
    """Mirror of `DoubleLinearSolver` for a leaf KKT block (replaces PardisoProjectSolver / Ma27Solver / Ma57Solver).

    The solver keeps a reference to the caller's CSR value array like the reference keeps a pointer to the
    SparseSymmetricMatrix (PardisoSolver.h:49-50): mutate `K.val` in place, then call matrixChanged().
    """

    def __init__(self, K, n_primal=-1, device=-1, refine_steps=1, refine_tol=0.0, backward_error=False):
        self.K = K
        self.n = K.nrows
        self._h = C.c_void_p()
        _check(lib.pips_hip_ldl_create(C.byref(self._h), C.c_int(K.nrows), _ptr(K.rowptr), _ptr(K.colidx), C.c_int(device),
                                       C.c_int(0)), "pips_hip_ldl_create")
        if n_primal >= 0:
            _check(lib.pips_hip_ldl_set_inertia_hint(self._h, C.c_int(n_primal)), "pips_hip_ldl_set_inertia_hint")
        # backward_error: stop on the normwise backward error (what the adapters set: at most 2 steps, 1e-15)
        _check((lib.pips_hip_ldl_set_refinement_backward_error if backward_error else lib.pips_hip_ldl_set_refinement)(
            self._h, C.c_int(refine_steps), C.c_double(refine_tol)), "pips_hip_ldl_set_refinement")

    def set_pivot_rule(self, thr_rel, repl_rel):
        _check(lib.pips_hip_ldl_set_pivot_rule(self._h, C.c_double(thr_rel), C.c_double(repl_rel)), "set_pivot_rule")

    def set_deterministic(self, on=True):
        """before the first factorisation: factor / solve / solve(nrhs) of this leaf repeat to the bit"""
        _check(lib.pips_hip_ldl_set_deterministic(self._h, C.c_int(1 if on else 0)), "pips_hip_ldl_set_deterministic")

    def analyze(self):
        _check(lib.pips_hip_ldl_analyze(self._h), "pips_hip_ldl_analyze")

    def matrixChanged(self):
        _check(lib.pips_hip_ldl_factor(self._h, _ptr(self.K.val)), "pips_hip_ldl_factor")

    def set_border(self, Bt):
        """Declare Br^T (Csr, S rows = Schur column ids over the n rows of K) before analyze(): level 1.5 of INTEGRATION.md."""
        self.Bt = Bt
        _check(lib.pips_hip_ldl_set_border(self._h, C.c_int(Bt.nrows), _ptr(Bt.rowptr), _ptr(Bt.colidx)), "pips_hip_ldl_set_border")

    def matrixChanged_with_schur_term(self, SC):
        """matrixChanged() + addTermToSchurComplBlocked(): SC (S x S float64 C-contiguous, lower triangle) -= Br^T K^-1 Br."""
        assert SC.dtype == np.float64 and SC.flags.c_contiguous and SC.shape[0] == SC.shape[1] == self.Bt.nrows
        _check(lib.pips_hip_ldl_factor_schur(self._h, _ptr(self.K.val), _ptr(self.Bt.val), _ptr(SC), C.c_int(SC.shape[1])),
               "pips_hip_ldl_factor_schur")

    def diagonalChanged(self, idiag=0, extent=0):
        self.matrixChanged()

    def solve(self, x):
        """x: (n,) or (nrhs, n) float64 C-contiguous; overwritten by the solution (one RHS per row, PardisoSolver.C:276)."""
        assert x.dtype == np.float64 and x.flags.c_contiguous
        nrhs = 1 if x.ndim == 1 else x.shape[0]
        ld = x.shape[-1]
        _check(lib.pips_hip_ldl_solve(self._h, C.c_int(nrhs), _ptr(x), C.c_int(ld)), "pips_hip_ldl_solve")
        return x

    def solve_sparse(self, x, col_sparsity):
        """DoubleLinearSolver::solve(int nrhss, double* rhss, int* colSparsity): only the rows flagged in col_sparsity (int32, length n) travel."""
        assert x.dtype == np.float64 and x.flags.c_contiguous
        nrhs = 1 if x.ndim == 1 else x.shape[0]
        cs = None if col_sparsity is None else _i32(col_sparsity)
        _check(lib.pips_hip_ldl_solve_sparse(self._h, C.c_int(nrhs), _ptr(x), C.c_int(x.shape[-1]), _ptr(cs) if cs is not None else None),
               "pips_hip_ldl_solve_sparse")
        return x

    def solve_dev(self, x_dev, nrhs=1, ld=None):
        """right-hand sides in device memory (a torch tensor), overwritten"""
        _check(lib.pips_hip_ldl_solve_dev(self._h, C.c_int(nrhs), _ptr(x_dev), C.c_longlong(self.n if ld is None else ld)), "pips_hip_ldl_solve_dev")
        return x_dev

    @staticmethod
    def _handle_array(solvers):
        return (C.c_void_p * len(solvers))(*[s._h for s in solvers])

    @staticmethod
    def factor_schur_batch(solvers, SC=None):
        """All leaf solvers of a rank as one batch (the host's loop over its children handed over): every K_i factorised, SC (S x S row-major,
        lower triangle) += sum_i -Br_i^T K_i^-1 Br_i.  The solvers' K.val / Bt.val are read."""
        hs = HipLdlSolver._handle_array(solvers)
        n = len(solvers)
        kv = (C.c_void_p * n)(*[s.K.val.ctypes.data for s in solvers])
        bv = (C.c_void_p * n)(*[(s.Bt.val.ctypes.data if getattr(s, "Bt", None) is not None else None) for s in solvers])
        _check(lib.pips_hip_ldl_factor_schur_batch(hs, C.c_int(n), kv, bv, _ptr(SC) if SC is not None else None,
                                                   C.c_int(SC.shape[1] if SC is not None else 0)), "pips_hip_ldl_factor_schur_batch")

    @staticmethod
    def solve_batch(solvers, rhs_list):
        """one right-hand side per leaf (numpy arrays of length n_i, or None), overwritten by the solutions"""
        hs = HipLdlSolver._handle_array(solvers)
        ptrs = (C.c_void_p * len(solvers))(*[(r.ctypes.data if r is not None else None) for r in rhs_list])
        _check(lib.pips_hip_ldl_solve_batch(hs, C.c_int(len(solvers)), ptrs), "pips_hip_ldl_solve_batch")

    @staticmethod
    def solve_batch_dev(solvers, x_dev):
        _check(lib.pips_hip_ldl_solve_batch_dev(HipLdlSolver._handle_array(solvers), C.c_int(len(solvers)), _ptr(x_dev)), "pips_hip_ldl_solve_batch_dev")

    @staticmethod
    def inertia_batch(solvers):
        n = len(solvers)
        p, q, z = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        _check(lib.pips_hip_ldl_inertia_batch(HipLdlSolver._handle_array(solvers), C.c_int(n), _ptr(p), _ptr(q), _ptr(z)), "pips_hip_ldl_inertia_batch")
        return [(int(p[i]), int(q[i]), int(z[i])) for i in range(n)]

    def reports_inertia(self):
        return True

    def get_inertia(self):
        p, n, z = C.c_int(), C.c_int(), C.c_int()
        _check(lib.pips_hip_ldl_inertia(self._h, C.byref(p), C.byref(n), C.byref(z)), "pips_hip_ldl_inertia")
        return p.value, n.value, z.value

    def info(self):
        what = np.zeros(8, np.int64)
        _check(lib.pips_hip_ldl_info(self._h, _ptr(what), C.c_int(8)), "pips_hip_ldl_info")
        return dict(nnzL=int(what[0]), n_head=int(what[1]), m=int(what[2]), n_sn=int(what[3]), n_levels=int(what[4]),
                    flops=int(what[5]), last_refinement_steps=int(what[6]), last_multi_path=int(what[7]))

    def close(self):
        if self._h:
            lib.pips_hip_ldl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipDenseLdlSolver:
    """Mirror of DeSymIndefSolver (DeSymIndefSolver.C:56-168): dense symmetric indefinite root solver."""

    def __init__(self, n, n_primal=-1, device=-1):
        self.n = n
        self._h = C.c_void_p()
        _check(lib.pips_hip_dense_ldl_create(C.byref(self._h), C.c_int(n), C.c_int(n_primal), C.c_int(device)),
               "pips_hip_dense_ldl_create")

    def set_pivoting(self, mode):
        """0: static pivot order (needs the inertia hint), 1: Bunch-Kaufman 1 x 1 / 2 x 2 pivots inside the diagonal tiles."""
        _check(lib.pips_hip_dense_ldl_set_pivoting(self._h, C.c_int(mode)), "pips_hip_dense_ldl_set_pivoting")

    def set_distributed(self, comm, rank, n_ranks):
        """Column-cyclic factorisation over the ranks of `comm` (Comm / ExternalComm); every rank calls matrixChanged with the same matrix."""
        self._comm = comm
        _check(lib.pips_hip_dense_ldl_set_distributed(self._h, comm._h if comm is not None else None, C.c_int(rank), C.c_int(n_ranks)),
               "pips_hip_dense_ldl_set_distributed")

    def matrixChanged(self, A_rowmajor_lower):
        A = _f64(A_rowmajor_lower)
        _check(lib.pips_hip_dense_ldl_factor(self._h, _ptr(A), C.c_int(A.shape[1])), "pips_hip_dense_ldl_factor")

    def matrixChanged_dev(self, A_dev, lda):
        _check(lib.pips_hip_dense_ldl_factor_dev(self._h, _ptr(A_dev), C.c_int(lda)), "pips_hip_dense_ldl_factor_dev")

    def solve(self, x):
        assert x.dtype == np.float64 and x.flags.c_contiguous
        nrhs = 1 if x.ndim == 1 else x.shape[0]
        _check(lib.pips_hip_dense_ldl_solve(self._h, C.c_int(nrhs), _ptr(x), C.c_int(x.shape[-1])), "dense solve")
        return x

    def solve_dev(self, x_dev):
        _check(lib.pips_hip_dense_ldl_solve_dev(self._h, _ptr(x_dev)), "pips_hip_dense_ldl_solve_dev")

    def get_inertia(self):
        p, n, z = C.c_int(), C.c_int(), C.c_int()
        _check(lib.pips_hip_dense_ldl_inertia(self._h, C.byref(p), C.byref(n), C.byref(z)), "dense inertia")
        return p.value, n.value, z.value

    def close(self):
        if self._h:
            lib.pips_hip_dense_ldl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# batched device-resident path
# ----------------------------------------------------------------------------------------------------------------------
class LeafBatch:
    """All leaf blocks owned by one GPU (the children of sLinsysRootAug that are not dummies on this rank)."""

    def __init__(self, n_blocks, S, device=-1, stream=None):
        self.n_blocks, self.S = n_blocks, S
        self._h = C.c_void_p()
        _check(lib.pips_hip_batch_create(C.byref(self._h), C.c_int(n_blocks), C.c_int(S), C.c_int(device), _ptr(stream)),
               "pips_hip_batch_create")
        self.n = [0] * n_blocks

    def set_block(self, b, K, n_primal, Bt=None):
        self.n[b] = K.nrows
        _check(lib.pips_hip_batch_set_block(self._h, C.c_int(b), C.c_int(K.nrows), C.c_int(n_primal), _ptr(K.rowptr),
                                            _ptr(K.colidx), _ptr(Bt.rowptr) if Bt is not None else None,
                                            _ptr(Bt.colidx) if Bt is not None else None,
                                            _ptr(Bt.val) if Bt is not None else None), "pips_hip_batch_set_block")

    def set_options(self, force_n_head=-1, refine_steps=-1, thr_rel=-1.0, repl_rel=-1.0):
        _check(lib.pips_hip_batch_set_options(self._h, C.c_int(force_n_head), C.c_int(refine_steps), C.c_double(thr_rel),
                                              C.c_double(repl_rel)), "pips_hip_batch_set_options")

    def add_regularization(self, primal, dual):
        _check(lib.pips_hip_batch_add_regularization(self._h, C.c_double(primal), C.c_double(dual)), "add_regularization")

    def set_refinement(self, max_steps, tol):
        """tol > 0: adaptive (stop once ||r||inf <= tol ||rhs||inf, at most max_steps steps); tol = 0: always max_steps."""
        _check(lib.pips_hip_batch_set_refinement(self._h, C.c_int(max_steps), C.c_double(tol)), "pips_hip_batch_set_refinement")

    def set_refinement_backward_error(self, max_steps, tol):
        _check(lib.pips_hip_batch_set_refinement_backward_error(self._h, C.c_int(max_steps), C.c_double(tol)),
               "pips_hip_batch_set_refinement_backward_error")

    def last_refinement_measure(self):
        lib.pips_hip_batch_last_refinement_measure.restype = C.c_double
        return float(lib.pips_hip_batch_last_refinement_measure(self._h))

    def last_refinement_steps(self):
        return int(lib.pips_hip_batch_last_refinement_steps(self._h))

    def analyze(self, n_threads=8):
        _check(lib.pips_hip_batch_analyze(self._h, C.c_int(n_threads)), "pips_hip_batch_analyze")

    def set_values(self, b, vals):
        _check(lib.pips_hip_batch_set_values(self._h, C.c_int(b), _ptr(_f64(vals))), "pips_hip_batch_set_values")

    def set_schur_mode(self, mode):
        """0 auto, 1 augmented partial factorisation, 2 blocked multi-RHS solves (the reference's K4-K6); before analyze."""
        _check(lib.pips_hip_batch_set_schur_mode(self._h, C.c_int(mode)), "pips_hip_batch_set_schur_mode")

    def set_deterministic(self, on=True):
        """Before analyze(): no FP64 atomics on the path - bit-identical results from run to run."""
        _check(lib.pips_hip_batch_set_deterministic(self._h, C.c_int(1 if on else 0)), "pips_hip_batch_set_deterministic")

    def schur_mode(self):
        m = C.c_int()
        _check(lib.pips_hip_batch_get_schur_mode(self._h, C.byref(m)), "pips_hip_batch_get_schur_mode")
        return m.value

    def set_diagonals(self, diag):
        if torch is not None and isinstance(diag, torch.Tensor):
            _check(lib.pips_hip_batch_set_diagonals_dev(self._h, _ptr(diag)), "pips_hip_batch_set_diagonals_dev")
        else:
            _check(lib.pips_hip_batch_set_diagonals(self._h, _ptr(_f64(diag))), "pips_hip_batch_set_diagonals")

    def factor(self, SC_dev=None, ldSC=0):
        _check(lib.pips_hip_batch_factor(self._h, _ptr(SC_dev), C.c_int(ldSC)), "pips_hip_batch_factor")

    def solve(self, x):
        if torch is not None and isinstance(x, torch.Tensor):
            _check(lib.pips_hip_batch_solve_dev(self._h, _ptr(x)), "pips_hip_batch_solve_dev")
        else:
            assert x.dtype == np.float64 and x.flags.c_contiguous
            _check(lib.pips_hip_batch_solve(self._h, _ptr(x)), "pips_hip_batch_solve")
        return x

    def border_tmult(self, z_dev, b0_dev, alpha):
        _check(lib.pips_hip_batch_border_tmult_dev(self._h, _ptr(z_dev), _ptr(b0_dev), C.c_double(alpha)), "border_tmult")

    def border_mult(self, x0_dev, t_dev, alpha):
        _check(lib.pips_hip_batch_border_mult_dev(self._h, _ptr(x0_dev), _ptr(t_dev), C.c_double(alpha)), "border_mult")

    def inertia(self, b):
        p, n, z = C.c_int(), C.c_int(), C.c_int()
        _check(lib.pips_hip_batch_inertia(self._h, C.c_int(b), C.byref(p), C.byref(n), C.byref(z)), "batch inertia")
        return p.value, n.value, z.value

    def info(self):
        what = np.zeros(26, np.int64)
        _check(lib.pips_hip_batch_info(self._h, _ptr(what), C.c_int(26)), "pips_hip_batch_info")
        keys = ["nnzL", "n", "n_head", "m", "n_sn", "n_levels", "flops_factor", "flops_border", "arena_bytes", "ntc", "upd_table_bytes", "nb", "nnzK",
                "ltsolve_from_augmented_factor", "multifrontal_head", "max_front", "update_matrix_bytes", "fronts_in_device_memory", "nnzL_head", "head_row_indices",
                "nnzL_border", "augmented_sweeps", "augmented_passes", "tail_border_entries", "blocks_with_border_split", "blocks_with_k_only_fronts"]
        return {k: int(v) for k, v in zip(keys, what)}

    def sync(self):
        _check(lib.pips_hip_batch_sync(self._h), "pips_hip_batch_sync")

    def set_timing(self, on=True):
        _check(lib.pips_hip_batch_set_timing(self._h, C.c_int(1 if on else 0)), "set_timing")

    def get_timing(self):
        ms = np.zeros(16)
        cnt = np.zeros(16, np.int64)
        _check(lib.pips_hip_batch_get_timing(self._h, _ptr(ms), _ptr(cnt), C.c_int(16)), "get_timing")
        names = ["scatter", "head", "tail_update", "tail_diag", "tail_trsm", "schur", "total", "solve_permute", "solve_head_fwd", "solve_tail",
                 "solve_head_bwd", "solve_refine"]
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(names)}

    def close(self):
        if self._h:
            lib.pips_hip_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator over the GPUs of one node (replaces the MPI communicator of the reference's collectives)."""

    def __init__(self, id_bytes, n_ranks, rank, device):
        self._h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(id_bytes)
        _check(lib.pips_hip_comm_create(C.byref(self._h), buf, C.c_int(n_ranks), C.c_int(rank), C.c_int(device)),
               "pips_hip_comm_create")

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib.pips_hip_comm_unique_id(buf), "pips_hip_comm_unique_id")
        return bytes(buf)

    def allreduce_sum(self, t, n=None, stream=None):
        n = t.numel() if n is None else n
        _check(lib.pips_hip_allreduce_sum(self._h, _ptr(t), C.c_size_t(n), _ptr(stream)), "pips_hip_allreduce_sum")

    def broadcast(self, t, root, n=None, stream=None):
        n = t.numel() if n is None else n
        _check(lib.pips_hip_broadcast(self._h, _ptr(t), C.c_size_t(n), C.c_int(root), _ptr(stream)), "pips_hip_broadcast")

    def has_broadcast(self):
        return bool(lib.pips_hip_comm_has_broadcast(self._h))

    def close(self):
        if self._h:
            lib.pips_hip_comm_destroy(self._h)
            self._h = C.c_void_p()


_ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class _DeviceDoubles:
    """Zero-copy view of n doubles of device memory for torch.as_tensor (CUDA array interface, also honoured on ROCm)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class ExternalComm(Comm):
    """Communicator whose all-reduce is supplied by the host program (the reference would hand in its MPI communicator's
    `PIPS_MPIsumArrayInPlace`); `torch_distributed()` builds one on an initialised torch.distributed process group."""

    def __init__(self, allreduce, reduce_scatter=None, all_gather=None, n_ranks=1, rank=0):
        """allreduce(ptr, n): in-place sum of n device doubles at ptr over all ranks.  Optional pair for the reduce-scatter +
        all-gather formulation (pips_hip_allreduce_sum_rsag): reduce_scatter(ptr, chunk) leaves the summed slice `rank` at
        ptr + rank * chunk doubles, all_gather(ptr, chunk) replicates every rank's slice."""
        def _cb(_user, ptr, n):
            try:
                allreduce(int(ptr), int(n))
                return 0
            except Exception as e:  # never unwind through the C frame
                sys.stderr.write(f"external all-reduce failed: {e}\n")
                return 1

        self._cb = _ALLREDUCE_CB(_cb)   # keep the trampoline alive as long as the communicator
        self._h = C.c_void_p()
        _check(lib.pips_hip_comm_create_external(C.byref(self._h), self._cb, None), "pips_hip_comm_create_external")
        if reduce_scatter is not None and all_gather is not None:
            def _wrap(fn, what):
                def _f(_user, ptr, chunk):
                    try:
                        fn(int(ptr), int(chunk))
                        return 0
                    except Exception as e:
                        sys.stderr.write(f"external {what} failed: {e}\n")
                        return 1
                return _ALLREDUCE_CB(_f)
            self._rs, self._ag = _wrap(reduce_scatter, "reduce-scatter"), _wrap(all_gather, "all-gather")
            _check(lib.pips_hip_comm_set_external_rsag(self._h, C.c_int(n_ranks), C.c_int(rank), self._rs, self._ag), "pips_hip_comm_set_external_rsag")

    def set_broadcast(self, broadcast, n_ranks, rank):
        """broadcast(ptr, n, root): n device doubles at ptr go from rank root to every rank (MPI_Bcast on device pointers in a host program);
        without one pips_hip_broadcast falls back to an all-reduce of zeros."""
        def _f(_user, ptr, n, root):
            try:
                broadcast(int(ptr), int(n), int(root))
                return 0
            except Exception as e:
                sys.stderr.write(f"external broadcast failed: {e}\n")
                return 1
        self._bc = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)(_f)
        _check(lib.pips_hip_comm_set_external_broadcast(self._h, C.c_int(n_ranks), C.c_int(rank), self._bc), "pips_hip_comm_set_external_broadcast")

    @classmethod
    def torch_distributed(cls, group=None):
        import torch
        import torch.distributed as dist

        def allreduce(ptr, n):
            t = torch.as_tensor(_DeviceDoubles(ptr, n), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.current_stream().synchronize()

        return cls(allreduce)


class KktSystem:
    """Mirror of the root linear system's factor2()/solveCompressed() for the blocks one rank owns."""

    def __init__(self, batch, n0, my0, myl, mzl, A0=None, F0=None, G0=None, comm=None, rank=0, n_ranks=1, sparse_root=False,
                 all_block_cols=None):
        self.batch = batch
        self.S = n0 + my0 + myl + mzl
        self._h = C.c_void_p()

        def trip(M):
            return (None, None, None) if M is None else (_ptr(M.rowptr), _ptr(M.colidx), _ptr(M.val))

        self.sparse_root = bool(sparse_root)
        if sparse_root:
            cp = cc = None
            nglob = 0
            if all_block_cols is not None:     # border column sets of all blocks of the problem (needed with n_ranks > 1)
                nglob = len(all_block_cols)
                cp = np.zeros(nglob + 1, np.int32)
                cp[1:] = np.cumsum([len(x) for x in all_block_cols])
                cc = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in all_block_cols]) if nglob else np.zeros(0, np.int32))
            self._keep_cols = (cp, cc)
            _check(lib.pips_hip_kkt_create_sparse(C.byref(self._h), batch._h, C.c_int(n0), C.c_int(my0), C.c_int(myl), C.c_int(mzl),
                                                  *trip(A0), *trip(F0), *trip(G0), C.c_int(nglob), _ptr(cp), _ptr(cc),
                                                  comm._h if comm is not None else None, C.c_int(rank), C.c_int(n_ranks)),
                   "pips_hip_kkt_create_sparse")
        else:
            _check(lib.pips_hip_kkt_create(C.byref(self._h), batch._h, C.c_int(n0), C.c_int(my0), C.c_int(myl), C.c_int(mzl),
                                           *trip(A0), *trip(F0), *trip(G0), comm._h if comm is not None else None,
                                           C.c_int(rank), C.c_int(n_ranks)), "pips_hip_kkt_create")

    def set_root_inequalities(self, C0):
        self.mz0 = C0.nrows
        _check(lib.pips_hip_kkt_set_root_inequalities(self._h, C.c_int(C0.nrows), _ptr(C0.rowptr), _ptr(C0.colidx), _ptr(C0.val)),
               "pips_hip_kkt_set_root_inequalities")

    def set_zdiag0(self, zdiag0_dev):
        self._zdiag0 = zdiag0_dev   # keep alive: the library stores the pointer only
        _check(lib.pips_hip_kkt_set_zdiag0_dev(self._h, _ptr(zdiag0_dev)), "pips_hip_kkt_set_zdiag0_dev")

    def factorize(self, leaf_diag_dev, xdiag0_dev, zdiag_link_dev=None):
        _check(lib.pips_hip_kkt_factorize(self._h, _ptr(leaf_diag_dev), _ptr(xdiag0_dev), _ptr(zdiag_link_dev)),
               "pips_hip_kkt_factorize")

    def set_root_regularization(self, primal, dual):
        """+primal on the x0 diagonal, -dual on the dual rows of the Schur complement in the factorizations that follow."""
        _check(lib.pips_hip_kkt_set_root_regularization(self._h, C.c_double(primal), C.c_double(dual)), "pips_hip_kkt_set_root_regularization")

    def solve_compressed(self, b0_dev, b_leaf_dev):
        _check(lib.pips_hip_kkt_solve_compressed(self._h, _ptr(b0_dev), _ptr(b_leaf_dev)), "pips_hip_kkt_solve_compressed")

    def schur_sparse_nnz(self):
        """Sparse-root systems: stored entries of the Schur complement (= doubles every rank reduces per factorisation)."""
        nnz = C.c_int()
        _check(lib.pips_hip_kkt_get_schur_sparse(self._h, C.byref(nnz), None, None, None), "pips_hip_kkt_get_schur_sparse")
        return int(nnz.value)

    def schur_sparse_to_host(self):
        """Sparse-root systems: the Schur complement as a scipy CSR matrix (lower triangle)."""
        import scipy.sparse as sp
        nnz = C.c_int()
        _check(lib.pips_hip_kkt_get_schur_sparse(self._h, C.byref(nnz), None, None, None), "pips_hip_kkt_get_schur_sparse")
        rp, ci, v = np.zeros(self.S + 1, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value)
        _check(lib.pips_hip_kkt_get_schur_sparse(self._h, C.byref(nnz), _ptr(rp), _ptr(ci), _ptr(v)), "pips_hip_kkt_get_schur_sparse")
        return sp.csr_matrix((v, ci, rp), shape=(self.S, self.S))

    def sparse_root_info(self):
        """Sparse-root systems: elimination order taken ("amd" | "band" | "dissected") and the symbolic figures of the root's engine."""
        what = np.zeros(16, np.int64)
        _check(lib.pips_hip_kkt_sparse_root_info(self._h, _ptr(what), C.c_int(16)), "pips_hip_kkt_sparse_root_info")
        return dict(order=("amd", "band", "dissected")[int(what[0])], nnzL=int(what[1]), n=int(what[2]), n_head=int(what[3]), m=int(what[4]),
                    n_sn=int(what[5]), n_levels=int(what[6]), flops_factor=int(what[7]), multifrontal_head=int(what[15]))

    def schur_ptr(self):
        p = C.c_void_p()
        ld = C.c_int()
        _check(lib.pips_hip_kkt_get_schur(self._h, C.byref(p), C.byref(ld)), "pips_hip_kkt_get_schur")
        return p.value, ld.value

    def schur_to_host(self):
        p, ld = self.schur_ptr()
        out = np.zeros(self.S * self.S)
        self.batch.sync()
        _check(lib.pips_hip_memcpy_d2h(_ptr(out), C.c_void_p(p), C.c_size_t(out.nbytes)), "memcpy_d2h")
        return out

    def get_timing(self):
        """Phase times (ms, launches) of the last factorize and the solveCompressed calls since (batch timing switch on)."""
        ms = np.zeros(16)
        cnt = np.zeros(16, np.int64)
        _check(lib.pips_hip_kkt_get_timing(self._h, _ptr(ms), _ptr(cnt), C.c_int(16)), "kkt get_timing")
        names = ["diag_zero", "leaf_factor", "reduce", "finalize", "root_factor", "lsolve_leaf", "lsolve_border_reduce", "dsolve", "ltsolve", "combine",
                 "reduce_panels", "root_wait", "solve_check", "root_factor_main_stream"]
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(names)}

    def last_ltsolve_from_factor(self):
        f = C.c_int()
        _check(lib.pips_hip_kkt_last_ltsolve_from_factor(self._h, C.byref(f)), "pips_hip_kkt_last_ltsolve_from_factor")
        return bool(f.value)

    def set_solve_check(self, every):
        """measure every `every`-th solveCompressed that goes by sweeps of the augmented factor against the leaf rows (1: all, the default; 0: witness only)"""
        _check(lib.pips_hip_kkt_set_solve_check(self._h, C.c_int(every)), "pips_hip_kkt_set_solve_check")

    def solve_check_counts(self):
        a, b = C.c_longlong(), C.c_longlong()
        _check(lib.pips_hip_kkt_solve_check_counts(self._h, C.byref(a), C.byref(b)), "pips_hip_kkt_solve_check_counts")
        return a.value, b.value

    def last_solve_path(self):
        """0: two refined leaf solves; 1: refined Lsolve + Ltsolve from the factor; 2: one forward + one backward sweep of the augmented factor;
        3: the same sweeps, their result checked against the leaf rows (the witness of a factorisation on one rank)"""
        f = C.c_int()
        _check(lib.pips_hip_kkt_last_solve_path(self._h, C.byref(f)), "pips_hip_kkt_last_solve_path")
        return int(f.value)

    def set_solve_graph(self, on=True):
        """solveCompressed as a captured / replayed HIP graph (fixed refinement, one rank, dense root; else launch by launch)."""
        _check(lib.pips_hip_kkt_set_solve_graph(self._h, C.c_int(1 if on else 0)), "pips_hip_kkt_set_solve_graph")

    def set_root_stream(self, own_stream):
        """dense root on a stream of its own (default) or on the main stream (callers that query the inertia after every factorisation)"""
        _check(lib.pips_hip_kkt_set_root_stream(self._h, C.c_int(1 if own_stream else 0)), "pips_hip_kkt_set_root_stream")

    def solve_graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        _check(lib.pips_hip_kkt_solve_graph_stats(self._h, C.byref(c), C.byref(r)), "pips_hip_kkt_solve_graph_stats")
        return int(c.value), int(r.value)

    def set_root_pivoting(self, mode):
        _check(lib.pips_hip_kkt_set_root_pivoting(self._h, C.c_int(mode)), "pips_hip_kkt_set_root_pivoting")

    def root_inertia(self):
        p, n, z = C.c_int(), C.c_int(), C.c_int()
        _check(lib.pips_hip_kkt_root_inertia(self._h, C.byref(p), C.byref(n), C.byref(z)), "root inertia")
        return p.value, n.value, z.value

    def close(self):
        if self._h:
            lib.pips_hip_kkt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# flat-arena vector kernels (DistributedVector / DenseVector operations) on torch CUDA tensors
# ----------------------------------------------------------------------------------------------------------------------
class vec:
    """Thin wrappers: every function takes float64 CUDA tensors of equal length and works in place on `y`."""

    @staticmethod
    def _n(t):
        return C.c_longlong(t.numel())

    @staticmethod
    def axpy(a, x, y):
        _check(lib.pips_hip_vec_axpy(vec._n(y), C.c_double(a), _ptr(x), _ptr(y), None), "vec_axpy")

    @staticmethod
    def axpby(a, x, b, y):
        _check(lib.pips_hip_vec_axpby(vec._n(y), C.c_double(a), _ptr(x), C.c_double(b), _ptr(y), None), "vec_axpby")

    @staticmethod
    def scale(a, y):
        _check(lib.pips_hip_vec_scale(vec._n(y), C.c_double(a), _ptr(y), None), "vec_scale")

    @staticmethod
    def add_const(a, y):
        _check(lib.pips_hip_vec_add_const(vec._n(y), C.c_double(a), _ptr(y), None), "vec_add_const")

    @staticmethod
    def mul(x, y):
        _check(lib.pips_hip_vec_mul(vec._n(y), _ptr(x), _ptr(y), None), "vec_mul")

    @staticmethod
    def div(x, y):
        _check(lib.pips_hip_vec_div(vec._n(y), _ptr(x), _ptr(y), None), "vec_div")

    @staticmethod
    def add_product(a, x, z, y):
        _check(lib.pips_hip_vec_add_product(vec._n(y), C.c_double(a), _ptr(x), _ptr(z), _ptr(y), None), "vec_add_product")

    @staticmethod
    def add_quotient(a, x, z, mask, y):
        _check(lib.pips_hip_vec_add_quotient(vec._n(y), C.c_double(a), _ptr(x), _ptr(z), _ptr(mask), _ptr(y), None),
               "vec_add_quotient")

    @staticmethod
    def divide_some(x, mask, y):
        _check(lib.pips_hip_vec_divide_some(vec._n(y), _ptr(x), _ptr(mask), _ptr(y), None), "vec_divide_some")

    @staticmethod
    def select_nonzeros(mask, y):
        _check(lib.pips_hip_vec_select_nonzeros(vec._n(y), _ptr(mask), _ptr(y), None), "vec_select_nonzeros")

    @staticmethod
    def safe_invert(y):
        _check(lib.pips_hip_vec_safe_invert(vec._n(y), _ptr(y), None), "vec_safe_invert")

    @staticmethod
    def find_blocking(x, dx, y, dy):
        out = np.zeros(5)
        _check(lib.pips_hip_vec_find_blocking(vec._n(x), _ptr(x), _ptr(dx), _ptr(y), _ptr(dy), _ptr(out), None), "vec_find_blocking")
        return out

    @staticmethod
    def weighted_stepbounds(x, dx, cx, y, dy, cy, wmin, nw=11):
        out = np.zeros(2 * nw)
        _check(lib.pips_hip_vec_weighted_stepbounds(vec._n(x), _ptr(x), _ptr(dx), _ptr(cx), _ptr(y), _ptr(dy), _ptr(cy), C.c_double(wmin),
                                                    C.c_int(nw), _ptr(out), None), "vec_weighted_stepbounds")
        return out[:nw], out[nw:]

    @staticmethod
    def gondzio_projection(rmin, rmax, y):
        _check(lib.pips_hip_vec_gondzio_projection(vec._n(y), C.c_double(rmin), C.c_double(rmax), _ptr(y), None), "vec_gondzio_projection")

    @staticmethod
    def _red(fn, *args):
        out = C.c_double()
        _check(fn(*args, C.byref(out), None), fn.__name__)
        return out.value

    @staticmethod
    def dot(x, y, skip_root=0):
        return vec._red(lib.pips_hip_vec_dot, vec._n(x), C.c_longlong(skip_root), _ptr(x), _ptr(y))

    @staticmethod
    def one_norm(x, skip_root=0):
        return vec._red(lib.pips_hip_vec_one_norm, vec._n(x), C.c_longlong(skip_root), _ptr(x))

    @staticmethod
    def inf_norm(x):
        return vec._red(lib.pips_hip_vec_inf_norm, vec._n(x), _ptr(x))

    @staticmethod
    def min(x):
        return vec._red(lib.pips_hip_vec_min, vec._n(x), _ptr(x))

    @staticmethod
    def two_norm(x):
        """DistributedVector::two_norm (DistributedVector.C:424-437): s * sqrt(sum (x/s)^2), s = inf_norm."""
        s = vec.inf_norm(x)
        if s == 0.0:
            return 0.0
        q = vec._red(lib.pips_hip_vec_sumsq_scaled, vec._n(x), C.c_longlong(0), C.c_double(1.0 / s), _ptr(x))
        return s * q ** 0.5

    @staticmethod
    def stepbound(x, dx, mask=None):
        return vec._red(lib.pips_hip_vec_stepbound, vec._n(x), _ptr(x), _ptr(dx), _ptr(mask))

    @staticmethod
    def dot_shifted(x, a, dx, y, b, dy, skip_root=0):
        return vec._red(lib.pips_hip_vec_dot_shifted, vec._n(x), C.c_longlong(skip_root), _ptr(x), C.c_double(a), _ptr(dx),
                        _ptr(y), C.c_double(b), _ptr(dy))


class IpmSolver:
    """The host harness: Mehrotra predictor-corrector on the device for the generator's LP class."""

    def __init__(self, n0, myl, blocks, F0, c, b, dual_reg=0.0, device=-1, comm=None, rank=0, n_ranks=1):
        """blocks: list of (W, T, F) Csr triples; c: [x0 | x1..xN]; b: [link | block rows].
        Several ranks: blocks / c / b hold this rank's blocks behind the replicated root parts; comm: Comm or ExternalComm."""
        N = len(blocks)
        n_i = _i32([w.ncols for (w, t, f) in blocks])
        my_i = _i32([w.nrows for (w, t, f) in blocks])

        def cat(ms):
            return (_i32(np.concatenate([m.rowptr for m in ms])), _i32(np.concatenate([m.colidx for m in ms])),
                    _f64(np.concatenate([m.val for m in ms])))

        W = cat([w for (w, t, f) in blocks])
        T = cat([t for (w, t, f) in blocks]) if n0 > 0 else (None, None, None)
        F = cat([f for (w, t, f) in blocks]) if myl > 0 else (None, None, None)
        self._keep = (n_i, my_i, W, T, F, _f64(c), _f64(b))
        self.nx = n0 + int(n_i.sum())
        self.ny = myl + int(my_i.sum())
        self._h = C.c_void_p()
        self._comm = comm
        _check(lib.pips_ipm_create_rank(C.byref(self._h), C.c_int(N), C.c_int(n0), C.c_int(myl), _ptr(n_i), _ptr(my_i),
                                        *[_ptr(a) for a in W], *[_ptr(a) for a in T], *[_ptr(a) for a in F],
                                        _ptr(F0.rowptr) if F0 is not None else None, _ptr(F0.colidx) if F0 is not None else None,
                                        _ptr(F0.val) if F0 is not None else None, _ptr(self._keep[5]), _ptr(self._keep[6]),
                                        C.c_double(dual_reg), C.c_int(device), comm._h if comm is not None else None, C.c_int(rank),
                                        C.c_int(n_ranks)), "pips_ipm_create_rank")

    def set_gondzio(self, max_correctors):
        _check(lib.pips_ipm_set_gondzio(self._h, C.c_int(max_correctors)), "pips_ipm_set_gondzio")

    def solve(self, max_iter=100, mutol=1e-6, artol=1e-4, verbose=False):
        res = np.zeros(7)
        _check(lib.pips_ipm_solve(self._h, C.c_int(max_iter), C.c_double(mutol), C.c_double(artol), C.c_int(int(verbose)),
                                  _ptr(res)), "pips_ipm_solve")
        return dict(objective=res[0], iterations=int(res[1]), mu=res[2], rnorm=res[3], status=int(res[4]), dual_objective=res[5],
                    dnorm=res[6])

    def solution(self):
        x = np.zeros(self.nx)
        y = np.zeros(self.ny)
        _check(lib.pips_ipm_get_solution(self._h, _ptr(x), _ptr(y)), "pips_ipm_get_solution")
        return x, y

    def trace(self):
        """(n_iterates, 7) history of the last solve: mu, ||r||inf, primal obj, dual obj, sigma, alpha_p, alpha_d."""
        n = C.c_int()
        _check(lib.pips_ipm_get_trace(self._h, None, C.c_int(0), C.byref(n)), "pips_ipm_get_trace")
        out = np.zeros((n.value, 7))
        _check(lib.pips_ipm_get_trace(self._h, _ptr(out), C.c_int(n.value), C.byref(n)), "pips_ipm_get_trace")
        return out

    def set_free_variables(self, bounded_mask):
        """bounded_mask: 1 where x_j >= 0, 0 where x_j is free (no complementarity pair, dd_j = 0 like the reference's computeDiagonals)."""
        m = _f64(bounded_mask)
        if m.shape[0] != self.nx:
            raise ValueError("bounded_mask must have nx entries")
        _check(lib.pips_ipm_set_free_variables(self._h, _ptr(m)), "pips_ipm_set_free_variables")

    def set_option(self, name, value):
        """Harness setting under the reference's option identifier (GONDZIO_MAX_CORRECTORS, OUTER_SOLVE, OUTER_BICG_MAX_ITER, REGULARIZATION)."""
        _check(lib.pips_ipm_set_option(self._h, name.encode(), C.c_double(float(value))), "pips_ipm_set_option")

    def stats(self):
        """Counters of the last solve: KKT factorisations, repeats with added dual regularisation (inertia loop), solveCompressed
        calls, accepted Gondzio correctors."""
        out = (C.c_longlong * 4)()
        _check(lib.pips_ipm_get_stats(self._h, out), "pips_ipm_get_stats")
        return dict(factorizations=out[0], regularised_repeats=out[1], solve_compressed=out[2], gondzio_correctors=out[3])

    def close(self):
        if self._h:
            lib.pips_ipm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CsrView(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("val", C.c_void_p)]


class _IpmBlock(C.Structure):
    _fields_ = [("n", C.c_int), ("my", C.c_int), ("mz", C.c_int)] + [(k, _CsrView) for k in ("A", "B", "C", "D", "BL", "DL")] + \
               [(k, C.c_void_p) for k in ("c", "xlow", "xupp", "ixlow", "ixupp", "b", "clow", "cupp", "iclow", "icupp")]


class GeneralIpmSolver(IpmSolver):
    """The device IPM on the reference's full problem class (bounds of either side on variables and rows, inequality rows, root
    and linking rows), fed with the reader's per-block dicts (fields of GMSPIPSBlockData_t as gdx.read_block / gdx_read_block
    return them): blocks[0] is the root.  Several ranks: blocks = [root] + this rank's blocks."""

    def __init__(self, blocks, dual_reg=0.0, device=-1, comm=None, rank=0, n_ranks=1):
        keep = []

        def arr(a, dtype=np.float64):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p).value if a.size else None

        def view(m):
            if m is None:
                return _CsrView(0, 0, None, None, None)
            if isinstance(m, Csr):
                m = dict(rows=m.nrows, cols=m.ncols, rowptr=m.rowptr, colidx=m.colidx, val=m.val)
            rp = np.ascontiguousarray(m["rowptr"], dtype=np.int32)
            keep.append(rp)
            return _CsrView(int(m["rows"]), int(m["cols"]), rp.ctypes.data_as(C.c_void_p).value, arr(m["colidx"], np.int32), arr(m["val"]))

        root = blocks[0]
        myl, mzl = int(root["mBL"]), int(root["mDL"])
        cb = (_IpmBlock * len(blocks))()
        for k, b in enumerate(blocks):
            n = int(b["n0"] if k == 0 else b["ni"])
            cb[k].n, cb[k].my, cb[k].mz = n, int(b["mA"]), int(b["mC"])
            cb[k].A, cb[k].C, cb[k].BL, cb[k].DL = view(b["A"]), view(b["C"]), view(b["BL"]), view(b["DL"])
            cb[k].B, cb[k].D = (view(None), view(None)) if k == 0 else (view(b["B"]), view(b["D"]))
            for f in ("c", "xlow", "xupp", "ixlow", "ixupp", "b", "clow", "cupp", "iclow", "icupp"):
                setattr(cb[k], f, arr(b[f]))
        self.n_blocks = len(blocks)
        self._keep = (keep, cb)
        self._comm = comm
        self._h = C.c_void_p()
        _check(lib.pips_ipm_create_general(C.byref(self._h), C.c_int(len(blocks)), cb, C.c_int(myl), C.c_int(mzl),
                                           C.c_void_p(arr(root["bL"])), C.c_void_p(arr(root["dlow"])), C.c_void_p(arr(root["dupp"])),
                                           C.c_void_p(arr(root["idlow"])), C.c_void_p(arr(root["idupp"])), C.c_double(dual_reg), C.c_int(device),
                                           comm._h if comm is not None else None, C.c_int(rank), C.c_int(n_ranks)), "pips_ipm_create_general")
        d = (C.c_longlong * 4)()
        _check(lib.pips_ipm_get_dims(self._h, d), "pips_ipm_get_dims")
        self.nx, self.ny, self.nzr, self.n_pairs = int(d[0]), int(d[1]), int(d[2]), int(d[3])

    def iterate(self):
        """All twelve parts of the current iterate as a dict of host arrays."""
        nx, my, mz = self.nx, self.ny, self.nzr
        out = dict(x=np.zeros(nx), s=np.zeros(mz), y=np.zeros(my), z=np.zeros(mz), t=np.zeros(mz), u=np.zeros(mz), v=np.zeros(nx), w=np.zeros(nx),
                   lam=np.zeros(mz), pi=np.zeros(mz), gamma=np.zeros(nx), phi=np.zeros(nx))
        _check(lib.pips_ipm_get_iterate(self._h, *[_ptr(out[k]) for k in ("x", "s", "y", "z", "t", "u", "v", "w", "lam", "pi", "gamma", "phi")]),
               "pips_ipm_get_iterate")
        return out

    def mult(self, vec, transposed=False):
        """J vec ([A x | C x]) or J^T vec for J = [A; C] in the harness' row / column order (DistributedMatrix::mult / transpose_mult)."""
        vec = _f64(vec)
        out = np.zeros(self.nx if transposed else self.ny + getattr(self, "nzr", 0))
        _check(lib.pips_ipm_mult(self._h, C.c_int(1 if transposed else 0), _ptr(vec), _ptr(out)), "pips_ipm_mult")
        return out

    def outer_solve(self, G, L, rhs, tol=1e-10):
        """Outer solve of [dd J^T; J diag(0, nOmegaInv)] sol = rhs with the diagonals of the pair vectors G = [t|u|v|w], L = [lambda|pi|gamma|phi]."""
        G, L, rhs = _f64(G), _f64(L), _f64(rhs)
        sol, info = np.zeros(self.nx + self.ny + self.nzr), np.zeros(6)
        _check(lib.pips_ipm_outer_solve(self._h, _ptr(G), _ptr(L), _ptr(rhs), C.c_double(tol), _ptr(sol), _ptr(info)), "pips_ipm_outer_solve")
        return sol, dict(status=int(info[0]), iterations=int(info[1]), residual=info[2], rhs_norm=info[3], preconditioner_calls=int(info[4]),
                         host_syncs=int(info[5]))

    def stats2(self):
        out = (C.c_longlong * 2)()
        _check(lib.pips_ipm_get_stats2(self._h, out), "pips_ipm_get_stats2")
        return dict(bicgstab_iterations=out[0], host_syncs=out[1])


def gdx_read_block(path, num_blocks, act_block, offset=1):
    """One block of a jacobian GDX file through the library's reader (pips_gdx_read_block): the same dict as
    pips_ipmpp_amd.gdx.read_block (fields of GMSPIPSBlockData_t, gmspipsio.h:5-58)."""
    h = C.c_void_p()
    _check(lib.pips_gdx_read_block(C.byref(h), str(path).encode(), C.c_int(num_blocks), C.c_int(act_block), C.c_int(offset)), "pips_gdx_read_block")
    try:
        cnt = (C.c_longlong * 14)()
        _check(lib.pips_gdx_block_counts(h, cnt), "pips_gdx_block_counts")
        out = dict(numBlocks=int(cnt[12]), blockID=int(cnt[13]), n0=int(cnt[0]), ni=int(cnt[1]), mA=int(cnt[2]), mC=int(cnt[3]), mBL=int(cnt[4]),
                   mDL=int(cnt[5]))
        names = ["c", "xlow", "xupp", "ixlow", "ixupp", "b", "clow", "cupp", "iclow", "icupp", "bL", "dlow", "dupp", "idlow", "idupp"]
        for which, name in enumerate(names):
            n = C.c_int()
            _check(lib.pips_gdx_block_vector(h, C.c_int(which), None, C.c_int(0), C.byref(n)), "pips_gdx_block_vector")
            v = np.zeros(n.value)
            _check(lib.pips_gdx_block_vector(h, C.c_int(which), _ptr(v), C.c_int(n.value), C.byref(n)), "pips_gdx_block_vector")
            out[name] = v.astype(np.int16) if name.startswith("i") else v
        for which, name in enumerate(["A", "B", "C", "D", "BL", "DL"]):
            present, rows, cols = C.c_int(), C.c_int(), C.c_int()
            _check(lib.pips_gdx_block_matrix(h, C.c_int(which), C.byref(present), C.byref(rows), C.byref(cols), None, None, None), "pips_gdx_block_matrix")
            if not present.value:
                out[name] = None
                continue
            nnz = int(cnt[6 + which])
            rp, ci, va = np.zeros(rows.value + 1, dtype=np.int32), np.zeros(nnz, dtype=np.int32), np.zeros(nnz)
            _check(lib.pips_gdx_block_matrix(h, C.c_int(which), None, None, None, _ptr(rp), _ptr(ci), _ptr(va)), "pips_gdx_block_matrix")
            out[name] = dict(rows=rows.value, cols=cols.value, rowptr=rp.tolist(), colidx=ci.tolist(), val=va.tolist())
        return out
    finally:
        lib.pips_gdx_block_destroy(h)
