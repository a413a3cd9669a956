// Host harness: a Mehrotra predictor-corrector interior-point loop for the synthetic arrowhead LPs, driving the device
// KKT path end to end.  This is the build's own counterpart of the reference callers that do not travel to the GPU box
// (SURVEY.md §8 a18):
//   PIPSIPMppSolver::solve            (InteriorPointMethod/PIPSIPMppSolver.cpp:29-83)   start point, loop, termination
//   Solver::solve_linear_system       (InteriorPointMethod/Solver.cpp:19-31)            initial affine solve + shift
//   InteriorPointMethod predictor/corrector (InteriorPointMethod.cpp:68-90,178-234)     sigma = (mu_aff/mu)^3
//   LinearSystem::computeDiagonals / solve / solveXYZS (LinearSystem.C:262-294,327-447,449-548)  rhs reduction, recovery
//   Residuals::evaluate / set_complementarity_residual (Residuals.cpp:58-171,220-256)
// restricted to the problem class of the generator: min c^T x, A x = b, x >= 0 (ixlow = 1, no upper bounds, no
// inequality rows), A block-angular.  The outer Krylov wrapper of the reference (solveCompressedBiCGStab, OUTER_SOLVE 2,
// LinearSystem.C:550-798) is reproduced with solveCompressed as the preconditioner; OUTER_SOLVE 1 (iterative refinement,
// :877-966) is available too.  Not reproduced: Gondzio correctors, Mehrotra's step-length heuristic, the filter line search.
// Single rank.  Everything numeric runs on the device; the host sees scalars only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "pips_hip.h"

// vector layer (vecops.hip)
extern "C" {
int pips_hip_vec_axpy(long long, double, const double*, double*, void*);
int pips_hip_vec_axpby(long long, double, const double*, double, double*, void*);
int pips_hip_vec_sumsq_scaled(long long, long long, double, const double*, double*, void*);
int pips_hip_vec_copy(long long, const double*, double*, void*);
int pips_hip_vec_set(long long, double, double*, void*);
int pips_hip_vec_scale(long long, double, double*, void*);
int pips_hip_vec_add_const(long long, double, double*, void*);
int pips_hip_vec_mul(long long, const double*, double*, void*);
int pips_hip_vec_div(long long, const double*, double*, void*);
int pips_hip_vec_add_product(long long, double, const double*, const double*, double*, void*);
int pips_hip_vec_add_quotient(long long, double, const double*, const double*, const double*, double*, void*);
int pips_hip_vec_dot(long long, long long, const double*, const double*, double*, void*);
int pips_hip_vec_inf_norm(long long, const double*, double*, void*);
int pips_hip_vec_min(long long, const double*, double*, void*);
int pips_hip_vec_stepbound(long long, const double*, const double*, const double*, double*, void*);
int pips_hip_vec_dot_shifted(long long, long long, const double*, double, const double*, const double*, double, const double*,
                             double*, void*);
}

namespace pips {

#define HIP_TRYH(expr)                                                                                   \
   do {                                                                                                  \
      hipError_t _e = (expr);                                                                            \
      if (_e != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
   } while (0)
#define TRY(expr)                \
   do {                          \
      int _rc = (expr);          \
      if (_rc) return _rc;       \
   } while (0)

// y = alpha * A x + beta * y, CSR, one thread per row (SparseStorage::mult, SparseStorage.C:818-845; the rows have ~10 entries)
// Rows longer than CSR_LONG_ROW are left to k_csr_mult_long: with few first-stage variables a row of A^T that belongs to
// x_0 collects the T_i entries of every block (80 000 entries at n0 = 8, 64 blocks x 5000 rows) and would keep one thread busy
// for milliseconds.
constexpr int CSR_LONG_ROW = 512;

__global__ void k_csr_mult(int nrows, const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                           const double* __restrict__ x, double alpha, double beta, double* __restrict__ y) {
   for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
      if (rp[r + 1] - rp[r] > CSR_LONG_ROW) continue;
      double s = 0.0;
      for (int p = rp[r]; p < rp[r + 1]; ++p) s += v[p] * x[ci[p]];
      y[r] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[r]);
   }
}

// one workgroup per long row (rows listed in long_rows)
__global__ __launch_bounds__(256) void k_csr_mult_long(const int* __restrict__ long_rows, const int* __restrict__ rp,
                                                      const int* __restrict__ ci, const double* __restrict__ v,
                                                      const double* __restrict__ x, double alpha, double beta,
                                                      double* __restrict__ y) {
   __shared__ double red[256];
   const int r = long_rows[blockIdx.x];
   double s = 0.0;
   for (int p = rp[r] + threadIdx.x; p < rp[r + 1]; p += 256) s += v[p] * x[ci[p]];
   red[threadIdx.x] = s;
   __syncthreads();
   for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
   }
   if (threadIdx.x == 0) y[r] = alpha * red[0] + (beta == 0.0 ? 0.0 : beta * y[r]);
}

// pack (rx, ry) into the KKT right-hand sides: b0 = [rx_0 | ry_link], leaf block i = [rx_i | ry_i]; unpack is the inverse
__global__ void k_kkt_pack(int N, int n0, int myl, const int* __restrict__ xoff, const int* __restrict__ yoff,
                           const long long* __restrict__ koff, const double* __restrict__ rx, const double* __restrict__ ry,
                           double* __restrict__ b0, double* __restrict__ bl, int unpack) {
   const int b = blockIdx.y;   // 0 = root, 1..N = leaves
   const int nx = b == 0 ? n0 : xoff[b + 1] - xoff[b];
   const int ny = b == 0 ? myl : yoff[b + 1] - yoff[b];
   const int x0 = b == 0 ? 0 : xoff[b];
   const int y0 = b == 0 ? 0 : yoff[b];
   double* dst = b == 0 ? b0 : bl + koff[b];
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx + ny; i += gridDim.x * blockDim.x) {
      double* vec = const_cast<double*>(i < nx ? rx + x0 + i : ry + y0 + (i - nx));
      if (unpack) *vec = dst[i]; else dst[i] = *vec;
   }
}

// K diagonals from the primal diagonal dd = gamma/v (computeDiagonals) and the dual regularisation (clear_dual_equality_diagonal)
__global__ void k_leaf_diag(int N, const int* __restrict__ xoff, const int* __restrict__ yoff, const long long* __restrict__ koff,
                            const double* __restrict__ dd, double primal_reg, double dual_reg, double* __restrict__ leaf_diag) {
   const int b = blockIdx.y + 1;
   const int nx = xoff[b + 1] - xoff[b], ny = yoff[b + 1] - yoff[b];
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx + ny; i += gridDim.x * blockDim.x)
      leaf_diag[koff[b] + i] = i < nx ? dd[xoff[b] + i] + primal_reg : -dual_reg;
}

// free entries: v = 1, gamma = 0
__global__ void k_fix_free(long long n, const double* __restrict__ fmask, double* __restrict__ v, double* __restrict__ g) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      if (fmask[i] == 0.0) { v[i] = 1.0; g[i] = 0.0; }
}
// preconditioner diagonal: ddp = dd + free_reg on free entries
__global__ void k_precond_diag(long long n, const double* __restrict__ fmask, const double* __restrict__ dd, double free_reg,
                               double* __restrict__ ddp) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      ddp[i] = dd[i] + (fmask[i] == 0.0 ? free_reg : 0.0);
}

struct Ipm {
   int device = 0;
   hipStream_t stream = nullptr;
   int N = 0, n0 = 0, myl = 0, nx = 0, ny = 0;
   long long nleaf = 0;
   double dnorm = 1.0, dual_reg = 0.0;
   void* batch = nullptr;
   void* kkt = nullptr;
   // global A (rows: [link | blocks], cols: [x0 | x_1..x_N]) and its transpose, CSR
   int *A_rp = nullptr, *A_ci = nullptr, *At_rp = nullptr, *At_ci = nullptr, *d_xoff = nullptr, *d_yoff = nullptr;
   long long* d_koff = nullptr;
   double *A_v = nullptr, *At_v = nullptr;
   // vectors
   double *c = nullptr, *b = nullptr, *x = nullptr, *y = nullptr, *v = nullptr, *g = nullptr;
   double *rQ = nullptr, *rA = nullptr, *rv = nullptr, *rg = nullptr, *dd = nullptr;
   double *dx = nullptr, *dy = nullptr, *dv = nullptr, *dg = nullptr, *cx = nullptr, *cy = nullptr, *cv = nullptr, *cg = nullptr;
   double *tx = nullptr, *ty = nullptr, *b0 = nullptr, *bl = nullptr, *leaf_diag = nullptr, *zx = nullptr, *zy = nullptr;
   double *bz = nullptr, *xz = nullptr, *w_r = nullptr, *w_r0 = nullptr, *w_best = nullptr, *w_v = nullptr, *w_t = nullptr, *w_p = nullptr,
          *w_dx = nullptr;
   double *gv = nullptr, *gg = nullptr;   // Gondzio trial vectors
   double *bx = nullptr, *bv = nullptr, *bg = nullptr, *by = nullptr;   // best iterate so far (numerical-trouble fallback)
   // Free variables (no bound: ixlow = ixupp = 0 in the reference, whose computeDiagonals gives them dd = 0, LinearSystem.C:
   // 262-294): fmask is 1 on x >= 0 entries and 0 on free ones.  A free entry carries the constant pair v = 1, gamma = 0, takes
   // no part in the complementarity terms (its rv, rgamma, dv, dgamma are masked to zero) and gets the proximal term free_reg
   // on the diagonal of the *preconditioner* only (ddp); the outer solve works with dd = 0 there.
   double *fmask = nullptr, *ddp = nullptr;
   bool has_free = false;
   double free_reg = 1e-6;
   int max_gondzio = 2;   // multiple centrality correctors per iteration (InteriorPointMethod.cpp:236-358)
   long long n_gondzio = 0;
   int outer_mode = 2;   // 1 = iterative refinement, 2 = BiCGStab (the reference's OUTER_SOLVE default)
   int outer_max = 10, last_outer_steps = 0;
   int bicg_max_iter = 75;      // OUTER_BICG_MAX_ITER
   bool regularize = true;      // REGULARIZATION: the inertia-correcting loop of factorize()
   long long n_precond = 0;
   double outer_tol = 1e-10, last_outer_res = 0.0, last_outer_abs = 0.0;
   std::vector<void*> owned;
   double last[8] = {0};

   ~Ipm() {
      if (kkt) pips_hip_kkt_destroy(kkt);
      if (batch) pips_hip_batch_destroy(batch);
      for (void* p : owned)
         if (p) (void)hipFree(p);
   }
   template <class T>
   int up(T** d, const std::vector<T>& h) {
      HIP_TRYH(hipMalloc((void**)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
      owned.push_back(*d);
      if (!h.empty()) HIP_TRYH(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
      return PIPS_OK;
   }
   int alloc(double** d, long long n) {
      HIP_TRYH(hipMalloc((void**)d, std::max<long long>(n, 1) * sizeof(double)));
      HIP_TRYH(hipMemset(*d, 0, std::max<long long>(n, 1) * sizeof(double)));
      owned.push_back(*d);
      return PIPS_OK;
   }
   // ---- several ranks: the blocks are sharded over the ranks, the root parts of every vector (x0 at the head of
   // the x-type vectors, y_link at the head of the y-type ones) are replicated.  Sums count a replicated part on rank 0 only
   // (iAmSpecial, DistributedVector.C:1293-1303), maxima / minima travel in one slot per rank of a summed vector, the link
   // rows of A x and the x0 rows of A^T y are summed over the ranks.  All scalars that steer the iteration are therefore
   // identical on every rank.
   void* comm = nullptr;
   int rank = 0, n_ranks = 1;
   long long nx_global = 0;
   double* d_red = nullptr;
   enum Kind { KX, KY, KZ };
   int reduce_host(double* vals, int n) {
      HIP_TRYH(hipMemcpyAsync(d_red, vals, n * sizeof(double), hipMemcpyHostToDevice, stream));
      TRY(pips_hip_allreduce_sum(comm, d_red, (size_t)n, stream));
      HIP_TRYH(hipMemcpyAsync(vals, d_red, n * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRYH(hipStreamSynchronize(stream));
      return PIPS_OK;
   }
   int gsum(double* val) { return n_ranks > 1 ? reduce_host(val, 1) : PIPS_OK; }
   int gext(double* val, bool want_max) {
      if (n_ranks == 1) return PIPS_OK;
      std::vector<double> slots(n_ranks, 0.0);
      slots[rank] = *val;
      TRY(reduce_host(slots.data(), n_ranks));
      for (int r = 0; r < n_ranks; ++r) *val = r == 0 ? slots[0] : (want_max ? std::max(*val, slots[r]) : std::min(*val, slots[r]));
      return PIPS_OK;
   }
   long long skx() const { return rank ? n0 : 0; }
   long long sky() const { return rank ? myl : 0; }
   int gdot(Kind k, const double* a, const double* b2, double* out) {
      if (k == KX) TRY(pips_hip_vec_dot(nx, skx(), a, b2, out, stream));
      else if (k == KY) TRY(pips_hip_vec_dot(ny, sky(), a, b2, out, stream));
      else if (rank == 0) TRY(pips_hip_vec_dot(nz(), 0, a, b2, out, stream));
      else {
         double p1, p2;
         TRY(pips_hip_vec_dot(nx, skx(), a, b2, &p1, stream));
         TRY(pips_hip_vec_dot(ny, sky(), a + nx, b2 + nx, &p2, stream));
         *out = p1 + p2;
      }
      return gsum(out);
   }
   int ginf(long long len, const double* z, double* out) {
      TRY(pips_hip_vec_inf_norm(len, z, out, stream));
      return gext(out, true);
   }
   int gvmin(const double* z, double* out) {
      TRY(pips_hip_vec_min(nx, z, out, stream));
      return gext(out, false);
   }
   int gstepbound(const double* z, const double* dz, double* out) {
      TRY(pips_hip_vec_stepbound(nx, z, dz, nullptr, out, stream));
      return gext(out, false);
   }
   int gdot_shifted(const double* a, double sa, const double* da, const double* b2, double sb, const double* db, double* out) {
      TRY(pips_hip_vec_dot_shifted(nx, skx(), a, sa, da, b2, sb, db, out, stream));
      return gsum(out);
   }
   // blocking entry over all ranks: the smallest ratio wins, the lowest rank on ties
   int gfind_blocking(const double* a, const double* da, const double* b2, const double* db, double* out5) {
      TRY(pips_hip_vec_find_blocking(nx, a, da, b2, db, out5, stream));
      if (n_ranks == 1) return PIPS_OK;
      std::vector<double> slots(5 * (size_t)n_ranks, 0.0);
      const bool none = !(out5[0] < INFINITY);
      for (int q = 0; q < 5; ++q) slots[5 * rank + q] = (q == 0 && none) ? -1.0 : out5[q];   // infinities do not travel through a sum
      TRY(reduce_host(slots.data(), 5 * n_ranks));
      int best = -1;
      for (int r = 0; r < n_ranks; ++r)
         if (slots[5 * r] >= 0.0 && (best < 0 || slots[5 * r] < slots[5 * best])) best = r;
      if (best < 0) { out5[0] = INFINITY; out5[1] = out5[2] = out5[3] = out5[4] = 0.0; }
      else for (int q = 0; q < 5; ++q) out5[q] = slots[5 * best + q];
      return PIPS_OK;
   }
   int root_sum(double* part, int n) { return (n_ranks > 1 && n > 0) ? pips_hip_allreduce_sum(comm, part, (size_t)n, stream) : PIPS_OK; }

   int *A_long = nullptr, *At_long = nullptr;   // rows of A / A^T with more than CSR_LONG_ROW entries
   int nA_long = 0, nAt_long = 0;
   void mult(const int* rp, const int* ci, const double* vals, int nrows, const int* long_rows, int n_long, const double* xin, double alpha,
             double beta, double* yout) {
      const int g = std::min(2048, (nrows + 255) / 256 > 0 ? (nrows + 255) / 256 : 1);
      hipLaunchKernelGGL(k_csr_mult, dim3(g), dim3(256), 0, stream, nrows, rp, ci, vals, xin, alpha, beta, yout);
      if (n_long > 0)
         hipLaunchKernelGGL(k_csr_mult_long, dim3(n_long), dim3(256), 0, stream, long_rows, rp, ci, vals, xin, alpha, beta, yout);
   }
   // y = alpha A x + beta y and x = alpha A^T y + beta x; with several ranks the local A holds F0 on rank 0 only, the other ranks
   // start their replicated output part from zero and the part is summed (DistributedMatrix::mult / transpose_mult,
   // DistributedMatrix.C:224-326)
   int Amult(const double* xin, double alpha, double beta, double* yout) {
      if (rank && beta != 0.0 && myl > 0) HIP_TRYH(hipMemsetAsync(yout, 0, (size_t)myl * sizeof(double), stream));
      mult(A_rp, A_ci, A_v, ny, A_long, nA_long, xin, alpha, beta, yout);
      return root_sum(yout, myl);
   }
   int ATmult(const double* yin, double alpha, double beta, double* xout) {
      if (rank && beta != 0.0 && n0 > 0) HIP_TRYH(hipMemsetAsync(xout, 0, (size_t)n0 * sizeof(double), stream));
      mult(At_rp, At_ci, At_v, nx, At_long, nAt_long, yin, alpha, beta, xout);
      return root_sum(xout, n0);
   }

   // Residuals::evaluate for this problem class: rQ = c - A^T y - gamma, rA = A x - b, rv = x - v; returns the inf-norm
   int residuals(double* rnorm, double* pobj, double* dobj) {
      TRY(pips_hip_vec_copy(nx, c, rQ, stream));
      TRY(ATmult(y, -1.0, 1.0, rQ));
      TRY(pips_hip_vec_axpy(nx, -1.0, g, rQ, stream));
      TRY(pips_hip_vec_copy(ny, b, rA, stream));
      TRY(Amult(x, 1.0, -1.0, rA));
      TRY(pips_hip_vec_copy(nx, x, rv, stream));
      TRY(pips_hip_vec_axpy(nx, -1.0, v, rv, stream));
      if (has_free) TRY(pips_hip_vec_mul(nx, fmask, rv, stream));
      double a1, a2, a3;
      TRY(ginf(nx, rQ, &a1));
      TRY(ginf(ny, rA, &a2));
      TRY(ginf(nx, rv, &a3));
      *rnorm = std::max(a1, std::max(a2, a3));
      TRY(gdot(KX, c, x, pobj));
      TRY(gdot(KY, b, y, dobj));
      return PIPS_OK;
   }
   int mu(double* out) {
      double s;
      TRY(gdot(KX, v, g, &s));
      *out = s / nx_global;
      return PIPS_OK;
   }
   // LinearSystem::factorize: dd = gamma / v, K diagonals, factor2 of the two-level system
   // KKT factorisation with the inertia contract of LinearSystem::factorize_with_correct_inertia (LinearSystem.C:295-325):
   // factor once as is; while a leaf or the root reports perturbed pivots, add primal and dual regularisation (1e-8, times
   // 100 per try; leaves: add_regularization_local_kkt DistributedLeafLinearSystem.C:108-143, root: sLinsysRootAug.C:
   // 1545-1600) and factor again.  Two sources were seen: close to a vertex the dual pivots are differences of 1e10-sized
   // terms and come out with the wrong sign, and split free variables (x = x+ - x-, both drifting) leave primal pivots of
   // 1e-10 - the reference's GAMSsmall instances are full of them.  The regularised factors only precondition: the outer
   // solve works on the unregularised system.
   int n_regularised = 0, n_factorize = 0, n_refactor_outer = 0, verbose_run = 0;
   double last_reg = 0.0;
   int perturbed_pivots(int* total) {
      int p_, n_, z_;
      TRY(pips_hip_kkt_root_inertia(kkt, &p_, &n_, &z_));
      *total = z_;
      if (verbose_run > 1) printf("   inertia: root (%d %d %d) leaves", p_, n_, z_);
      double leaves = 0.0;
      for (int b = 0; b < N; ++b) {
         TRY(pips_hip_batch_inertia(batch, b, &p_, &n_, &z_));
         leaves += z_;
         if (verbose_run > 1) printf(" (%d %d %d)", p_, n_, z_);
      }
      if (verbose_run > 1) printf("\n");
      TRY(gsum(&leaves));
      *total += (int)leaves;
      return PIPS_OK;
   }
   int factorize(double reg_start = 0.0) {
      TRY(pips_hip_vec_copy(nx, g, dd, stream));
      TRY(pips_hip_vec_div(nx, v, dd, stream));
      const double* dfac = dd;
      if (has_free) {
         hipLaunchKernelGGL(k_precond_diag, dim3(std::min<long long>(2048, (nx + 255) / 256)), dim3(256), 0, stream, (long long)nx, fmask, dd, free_reg, ddp);
         dfac = ddp;
      }
      double reg = reg_start;
      for (int attempt = 0;; ++attempt) {
         hipLaunchKernelGGL(k_leaf_diag, dim3(32, N), dim3(256), 0, stream, N, d_xoff, d_yoff, d_koff, dfac, reg, dual_reg + reg, leaf_diag);
         TRY(pips_hip_kkt_set_root_regularization(kkt, reg, dual_reg + reg));
         TRY(pips_hip_kkt_factorize(kkt, leaf_diag, dfac, nullptr));
         ++n_factorize;
         int pert;
         TRY(perturbed_pivots(&pert));
         if (verbose_run && (pert || reg > 0.0)) printf("   factorize: dual regularisation %.1e, %d perturbed pivots\n", dual_reg + reg, pert);
         if (pert == 0 || attempt == 4 || !regularize) break;
         reg = reg == 0.0 ? 1e-8 : reg * 100.0;
         ++n_regularised;
      }
      last_reg = reg;
      return PIPS_OK;
   }
   // ---- outer solve machinery on concatenated vectors z = [x | y] of length nz = nx + ny ------------------------------
   // z := M^-1 z with M^-1 = solveCompressed (the Schur-complement decomposition as preconditioner)
   int precond(double* z) {
      hipLaunchKernelGGL(k_kkt_pack, dim3(32, N + 1), dim3(256), 0, stream, N, n0, myl, d_xoff, d_yoff, d_koff, z, z + nx, b0, bl, 0);
      TRY(pips_hip_kkt_solve_compressed(kkt, b0, bl));
      hipLaunchKernelGGL(k_kkt_pack, dim3(32, N + 1), dim3(256), 0, stream, N, n0, myl, d_xoff, d_yoff, d_koff, z, z + nx, b0, bl, 1);
      ++n_precond;
      return PIPS_OK;
   }
   // out = K z, K = [dd A^T; A 0]  (LinearSystem::system_mult, LinearSystem.C:808-844, for this problem class)
   int kmult(const double* z, double* out) {
      TRY(pips_hip_vec_set(nx, 0.0, out, stream));
      TRY(pips_hip_vec_add_product(nx, 1.0, dd, z, out, stream));
      TRY(ATmult(z + nx, 1.0, 1.0, out));
      TRY(Amult(z, 1.0, 0.0, out + nx));
      return PIPS_OK;
   }
   int residual(const double* rhs, const double* z, double* r, double* nrm) {   // r = rhs - K z, two-norm
      TRY(kmult(z, r));
      TRY(pips_hip_vec_axpby(nz(), 1.0, rhs, -1.0, r, stream));
      return two_norm(r, nrm);
   }
   long long nz() const { return (long long)nx + ny; }
   int two_norm(const double* z, double* out) {   // DistributedVector::two_norm: s * sqrt(sum (z/s)^2), s = inf_norm
      double s, q;
      TRY(ginf(nz(), z, &s));
      if (s == 0.0) { *out = 0.0; return PIPS_OK; }
      if (rank == 0) TRY(pips_hip_vec_sumsq_scaled(nz(), 0, 1.0 / s, z, &q, stream));
      else {
         double q1, q2;
         TRY(pips_hip_vec_sumsq_scaled(nx, skx(), 1.0 / s, z, &q1, stream));
         TRY(pips_hip_vec_sumsq_scaled(ny, sky(), 1.0 / s, z + nx, &q2, stream));
         q = q1 + q2;
      }
      TRY(gsum(&q));
      *out = s * std::sqrt(q);
      return PIPS_OK;
   }
   // OUTER_SOLVE 1 (solveCompressedIterRefin, LinearSystem.C:877-966): x += M^-1 (b - K x) while the residual decreases
   int iter_refine(const double* b_, double* x_) {
      double bn, rn, best = INFINITY;
      TRY(two_norm(b_, &bn));
      const double target = std::max(bn * outer_tol, 1e-15);
      TRY(pips_hip_vec_set(nz(), 0.0, x_, stream));
      TRY(pips_hip_vec_copy(nz(), b_, w_r, stream));
      last_outer_steps = 0;
      for (int it = 0; it <= outer_max; ++it) {
         TRY(precond(w_r));
         TRY(pips_hip_vec_axpy(nz(), 1.0, w_r, x_, stream));
         TRY(residual(b_, x_, w_r, &rn));
         if (!(rn < best)) { TRY(pips_hip_vec_copy(nz(), w_best, x_, stream)); break; }
         best = rn;
         last_outer_res = bn > 0 ? rn / bn : rn;
         last_outer_abs = rn;
         TRY(pips_hip_vec_copy(nz(), x_, w_best, stream));
         if (rn <= target) break;
         ++last_outer_steps;
      }
      return PIPS_OK;
   }
   // OUTER_SOLVE 2, the reference's default: BiCGStab right-preconditioned by solveCompressed
   // (LinearSystem::solveCompressedBiCGStab, LinearSystem.C:550-798): same half-step structure, best-iterate rollback,
   // divergence (4) and stagnation (4) counters, <= 75 iterations, tolerance max(tol ||b||_2, 1e-15).
   int bicgstab(const double* b_, double* x_) {
      const double eps = 1e-15;
      double bn, rn;
      TRY(two_norm(b_, &bn));
      const double target = std::max(bn * outer_tol, eps);
      TRY(pips_hip_vec_copy(nz(), b_, x_, stream));
      TRY(precond(x_));
      TRY(residual(b_, x_, w_r, &rn));
      double min_rn = rn;
      TRY(pips_hip_vec_copy(nz(), x_, w_best, stream));
      last_outer_steps = 0;
      last_outer_res = bn > 0 ? rn / bn : rn;
      last_outer_abs = rn;
      if (rn <= target) return PIPS_OK;                      // "skipped": the common case (LinearSystem.C:591-600)
      TRY(pips_hip_vec_copy(nz(), w_r, w_r0, stream));
      TRY(pips_hip_vec_scale(nz(), 1.0 / rn, w_r0, stream));
      double rho = 1.0, omega = 1.0, alpha = 1.0;
      int ndiv = 0, nstag = 0;
      auto is_zero = [](double v) { return std::fabs(v) < 1e-40; };   // PIPSisZero with pips_eps0 (pipsdef.h:35,108)
      // breakdown guard: near the optimum rho / (r0, v) can lose all digits and alpha, omega overflow; a non-finite quantity
      // ends the iteration and the best iterate so far is returned (NaN compares false, so the roll-backs below would miss it)
      auto bad = [](double v) { return !(v == v) || std::fabs(v) > 1e300; };
      auto stagn = [&](double step, double step_norm, double xn) { if (std::fabs(step) * step_norm <= eps * xn) ++nstag; else nstag = 0; };
      int it = 0;
      for (; it < bicg_max_iter; ++it) {
         const double rho_last = rho;
         TRY(gdot(KZ, w_r0, w_r, &rho));
         if (is_zero(rho) || bad(rho)) break;
         if (it == 0) TRY(pips_hip_vec_copy(nz(), w_r, w_p, stream));
         else {
            const double beta = (rho / rho_last) * (alpha / omega);
            if (is_zero(beta)) break;
            TRY(pips_hip_vec_axpy(nz(), -omega, w_v, w_p, stream));
            TRY(pips_hip_vec_axpby(nz(), 1.0, w_r, beta, w_p, stream));
         }
         TRY(pips_hip_vec_copy(nz(), w_p, w_dx, stream));
         TRY(precond(w_dx));
         TRY(kmult(w_dx, w_v));
         double rtv, dxn, xn;
         TRY(gdot(KZ, w_r0, w_v, &rtv));
         if (is_zero(rtv) || bad(rtv)) break;
         alpha = rho / rtv;
         if (bad(alpha)) break;
         TRY(two_norm(w_dx, &dxn));
         TRY(two_norm(x_, &xn));
         stagn(alpha, dxn, xn);
         TRY(pips_hip_vec_axpy(nz(), alpha, w_dx, x_, stream));   // half-way iterate
         TRY(pips_hip_vec_axpy(nz(), -alpha, w_v, w_r, stream));
         TRY(two_norm(w_r, &rn));
         if (bad(rn)) break;
         if (rn <= target) {
            TRY(residual(b_, x_, w_r, &rn));
            if (rn <= target) break;
         }
         if (rn < min_rn) { min_rn = rn; TRY(pips_hip_vec_copy(nz(), x_, w_best, stream)); }
         TRY(pips_hip_vec_copy(nz(), w_r, w_dx, stream));
         TRY(precond(w_dx));
         TRY(kmult(w_dx, w_t));
         double tt, tr;
         TRY(gdot(KZ, w_t, w_t, &tt));
         if (is_zero(tt) || bad(tt)) break;
         TRY(gdot(KZ, w_t, w_r, &tr));
         omega = tr / tt;
         if (bad(omega)) break;
         TRY(two_norm(w_dx, &dxn));
         TRY(two_norm(x_, &xn));
         stagn(omega, dxn, xn);
         TRY(pips_hip_vec_axpy(nz(), omega, w_dx, x_, stream));
         TRY(pips_hip_vec_axpy(nz(), -omega, w_t, w_r, stream));
         TRY(two_norm(w_r, &rn));
         if (bad(rn)) break;
         if (rn <= target || nstag >= 4) {
            TRY(residual(b_, x_, w_r, &rn));
            if (rn <= target) break;
         } else {
            if (rn >= min_rn) ++ndiv; else ndiv = 0;
            if (ndiv > 4) { TRY(pips_hip_vec_copy(nz(), w_best, x_, stream)); rn = min_rn; break; }   // diverged: roll back
         }
         if (rn < min_rn) { min_rn = rn; TRY(pips_hip_vec_copy(nz(), x_, w_best, stream)); }
         if (nstag >= 4) { if (min_rn < rn) { TRY(pips_hip_vec_copy(nz(), w_best, x_, stream)); rn = min_rn; } break; }
         if (is_zero(omega)) break;
      }
      if (min_rn < rn || bad(rn)) { TRY(pips_hip_vec_copy(nz(), w_best, x_, stream)); rn = min_rn; }
      last_outer_steps = it + 1;
      last_outer_res = bn > 0 ? rn / bn : rn;
      last_outer_abs = rn;
      return PIPS_OK;
   }

   // LinearSystem::solve + step.negate(): (sx, sy, sv, sg) := -solution for the residual set (rQ_, rA_, rv_, rg_)
   int solve(const double* rQ_, const double* rA_, const double* rv_, const double* rg_, double* sx, double* sy, double* sv, double* sg) {
      // rx = rQ + Gamma/V rv + rgamma/V ; ry = rA
      TRY(pips_hip_vec_copy(nx, rQ_, tx, stream));
      TRY(pips_hip_vec_add_product(nx, 1.0, dd, rv_, tx, stream));
      TRY(pips_hip_vec_add_quotient(nx, 1.0, rg_, v, has_free ? fmask : nullptr, tx, stream));
      TRY(pips_hip_vec_copy(ny, rA_, ty, stream));
      // joinRHS: z = [rx | ry]; outer solve on the ORIGINAL system [dd A^T; A 0] preconditioned by solveCompressed
      TRY(pips_hip_vec_copy(nx, tx, bz, stream));
      TRY(pips_hip_vec_copy(ny, ty, bz + nx, stream));
      // A pivot whose value is rounding noise can keep the right sign and pass the inertia test; the factors are then useless as
      // a preconditioner and the outer solve does not reach its tolerance.  The reference treats a failed outer solve as
      // numerical trouble of the factorisation; here the system is factorised again with (more) regularisation - which also
      // serves the later solves of the iteration - and the outer solve repeated, at most twice.
      for (int retry = 0;; ++retry) {
         if (outer_mode == 2) TRY(bicgstab(bz, xz));
         else TRY(iter_refine(bz, xz));
         const bool reached = last_outer_res <= std::max(1e3 * outer_tol, 1e-7) || last_outer_abs <= 1e-12;   // relative, or the absolute floor
         if (!regularize || retry == 2 || reached || last_reg >= 1e-2) break;
         if (verbose_run) printf("   outer solve stopped at rel.res %.1e: factorising again with regularisation\n", last_outer_res);
         ++n_refactor_outer;
         TRY(factorize(last_reg > 0.0 ? last_reg * 100.0 : 1e-8));
      }
      TRY(pips_hip_vec_copy(nx, xz, sx, stream));       // separateVars
      TRY(pips_hip_vec_copy(ny, xz + nx, sy, stream));
      // solveXYZS: stepy.negate()
      TRY(pips_hip_vec_scale(ny, -1.0, sy, stream));
      // Dv = Dx - rv ; Dgamma = (rgamma - Gamma Dv) / V
      TRY(pips_hip_vec_copy(nx, sx, sv, stream));
      TRY(pips_hip_vec_axpy(nx, -1.0, rv_, sv, stream));
      TRY(pips_hip_vec_copy(nx, rg_, sg, stream));
      TRY(pips_hip_vec_add_product(nx, -1.0, g, sv, sg, stream));
      TRY(pips_hip_vec_div(nx, v, sg, stream));
      if (has_free) {
         TRY(pips_hip_vec_mul(nx, fmask, sv, stream));
         TRY(pips_hip_vec_mul(nx, fmask, sg, stream));
      }
      // step.negate()
      TRY(pips_hip_vec_scale(nx, -1.0, sx, stream));
      TRY(pips_hip_vec_scale(ny, -1.0, sy, stream));
      TRY(pips_hip_vec_scale(nx, -1.0, sv, stream));
      TRY(pips_hip_vec_scale(nx, -1.0, sg, stream));
      return PIPS_OK;
   }
   int step_lengths(const double* sv, const double* sg, double tau, double* ap, double* ad) {
      double bp, bd;
      TRY(gstepbound(v, sv, &bp));
      TRY(gstepbound(g, sg, &bd));
      *ap = std::min(1.0, tau * bp);
      *ad = std::min(1.0, tau * bd);
      return PIPS_OK;
   }

   // 11-point search (one fused device pass) for the corrector weight in [alpha_p alpha_d, 1] that allows the longest steps
   // (calculate_alpha_pd_weight_candidate, InteriorPointMethod.cpp:486-523); step bounds are plain ratios capped at 1
   int weight_search(double apt, double adt, double* ape, double* ade, double* wp, double* wd) {
      constexpr int NW = 11;
      const double wmin = apt * adt;
      double bounds[2 * NW];   // one fused pass: primal bounds of the 11 blends, then the dual ones
      TRY(pips_hip_vec_weighted_stepbounds(nx, v, dv, cv, g, dg, cg, wmin, NW, bounds, stream));
      if (n_ranks > 1) {       // minimum over the ranks, one slot per rank and value (infinities do not travel through a sum)
         std::vector<double> slots((size_t)2 * NW * n_ranks, 0.0);
         for (int q = 0; q < 2 * NW; ++q) slots[(size_t)2 * NW * rank + q] = bounds[q] < INFINITY ? bounds[q] : -1.0;
         TRY(reduce_host(slots.data(), 2 * NW * n_ranks));
         for (int q = 0; q < 2 * NW; ++q) {
            bounds[q] = INFINITY;
            for (int r = 0; r < n_ranks; ++r) {
               const double t = slots[(size_t)2 * NW * r + q];
               if (t >= 0.0) bounds[q] = std::min(bounds[q], t);
            }
         }
      }
      *ape = *ade = *wp = *wd = -1.0;
      for (int k = 0; k < NW; ++k) {
         const double w = std::min(1.0, wmin + (1.0 - wmin) / (NW - 1) * k);
         const double a1 = std::min(1.0, bounds[k]), a2 = std::min(1.0, bounds[NW + k]);
         if (a1 > *ape) { *ape = a1; *wp = w; }
         if (a2 > *ade) { *ade = a2; *wd = w; }
      }
      return PIPS_OK;
   }

   // Mehrotra's step length heuristic (PrimalDualInteriorPointMethod::mehrotra_step_length, InteriorPointMethod.cpp:745-812):
   // let the blocking pair land on the complementarity value mu_full / gamma_a instead of on the boundary, stay within
   // [gamma_f, 1] of the maximal step, back off by 1e-8.
   int mehrotra_step_length(double* ap, double* ad) {
      const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f), steplength_factor = 0.99999999;
      double pb[5], db[5];
      TRY(gfind_blocking(v, dv, g, dg, pb));   // primal blocking: [ratio, v_b, dv_b, g_b, dg_b]
      TRY(gfind_blocking(g, dg, v, dv, db));   // dual blocking:   [ratio, g_b, dg_b, v_b, dv_b]
      const double amax_p = std::min(1.0, pb[0]), amax_d = std::min(1.0, db[0]);
      double mufull;
      TRY(gdot_shifted(v, amax_p, dv, g, amax_d, dg, &mufull));
      mufull = mufull / nx_global / gamma_a;
      double a_p = 1.0, a_d = 1.0;
      if (pb[0] < 1.0) {
         const double est = pb[3] + amax_d * pb[4];
         a_p = est == 0.0 ? 0.0 : (-pb[1] + mufull / est) / pb[2];
      }
      if (db[0] < 1.0) {
         const double est = db[3] + amax_p * db[4];
         a_d = est == 0.0 ? 0.0 : (-db[1] + mufull / est) / db[2];
      }
      a_p = std::max(std::min(a_p, amax_p), gamma_f * amax_p) * steplength_factor;
      a_d = std::max(std::min(a_d, amax_d), gamma_f * amax_d) * steplength_factor;
      *ap = a_p; *ad = a_d;
      return PIPS_OK;
   }

   // Gondzio's multiple centrality correctors (gondzio_correction_loop, InteriorPointMethod.cpp:236-358, primal-dual variant):
   // aim at longer steps (1.5 alpha + 0.3), look at the complementarity products of that trial point, pull the outliers back
   // into [beta_min, beta_max] * sigma * mu (Residuals::project_r3), solve for the corrector, blend it in with the weight in
   // [alpha_p alpha_d, 1] that gives the longest steps (10-point search, :486-523), keep it if a step grows by >= 1 %.
   int gondzio_loop(double sigma, double mu_now, double tau, double* ap, double* ad) {
      const double beta_min = 0.1, beta_max = 10.0, step_factor0 = 0.3, step_factor1 = 1.5, accept = 0.01;
      const double rmin = sigma * mu_now * beta_min, rmax = sigma * mu_now * beta_max;
      int ng = 0;
      while (ng < max_gondzio && (*ap < 1.0 || *ad < 1.0)) {
         const double apt = std::min(1.0, step_factor1 * *ap + step_factor0), adt = std::min(1.0, step_factor1 * *ad + step_factor0);
         // rg = -(projection step of the trial products)
         TRY(pips_hip_vec_copy(nx, v, gv, stream));
         TRY(pips_hip_vec_axpy(nx, apt, dv, gv, stream));
         TRY(pips_hip_vec_copy(nx, g, rg, stream));
         TRY(pips_hip_vec_axpy(nx, adt, dg, rg, stream));
         TRY(pips_hip_vec_mul(nx, gv, rg, stream));
         TRY(pips_hip_vec_gondzio_projection(nx, rmin, rmax, rg, stream));
         TRY(pips_hip_vec_scale(nx, -1.0, rg, stream));
         TRY(solve(zx, zy, zx, rg, cx, cy, cv, cg));
         double ape, ade, wp, wd;
         TRY(weight_search(apt, adt, &ape, &ade, &wp, &wd));
         const bool both_one = ape >= 1.0 && ade >= 1.0;
         const bool p_better = ape >= (1.0 + accept) * *ap, d_better = ade >= (1.0 + accept) * *ad;
         if (!both_one && !p_better && !d_better) break;
         if (both_one || p_better) {
            TRY(pips_hip_vec_axpy(nx, wp, cx, dx, stream));
            TRY(pips_hip_vec_axpy(nx, wp, cv, dv, stream));
            *ap = ape;
         }
         if (both_one || d_better) {
            TRY(pips_hip_vec_axpy(ny, wd, cy, dy, stream));
            TRY(pips_hip_vec_axpy(nx, wd, cg, dg, stream));
            *ad = ade;
         }
         ++ng;
         ++n_gondzio;
         if (both_one) break;
      }
      return PIPS_OK;
   }

   std::vector<double> trace;   // per iterate: mu, ||r||inf, primal obj, dual obj, then the step taken from it: sigma, alpha_p, alpha_d
   int run(int max_iter, double mutol, double artol, int verbose, double* result) {
      HIP_TRYH(hipSetDevice(device));
      verbose_run = verbose = rank == 0 ? verbose : 0;
      n_gondzio = n_precond = 0;
      n_regularised = n_factorize = n_refactor_outer = 0;
      // ---- start point: push_to_interior(sqrt(dnorm)), one affine solve, full step, shift (PIPSIPMppSolver.cpp:36-42, Solver.cpp:19-31)
      const double s0 = std::sqrt(dnorm);
      TRY(pips_hip_vec_set(nx, 0.0, x, stream));
      TRY(pips_hip_vec_set(ny, 0.0, y, stream));
      TRY(pips_hip_vec_set(nx, s0, v, stream));
      TRY(pips_hip_vec_set(nx, s0, g, stream));
      auto fix_free = [&]() {
         if (has_free) hipLaunchKernelGGL(k_fix_free, dim3(std::min<long long>(2048, (nx + 255) / 256)), dim3(256), 0, stream, (long long)nx, fmask, v, g);
      };
      fix_free();
      double rnorm, pobj, dobj, m;
      TRY(residuals(&rnorm, &pobj, &dobj));
      TRY(pips_hip_vec_copy(nx, v, rg, stream));
      TRY(pips_hip_vec_mul(nx, g, rg, stream));
      TRY(factorize());
      TRY(solve(rQ, rA, rv, rg, dx, dy, dv, dg));
      TRY(pips_hip_vec_axpy(nx, 1.0, dx, x, stream));
      TRY(pips_hip_vec_axpy(ny, 1.0, dy, y, stream));
      TRY(pips_hip_vec_axpy(nx, 1.0, dv, v, stream));
      TRY(pips_hip_vec_axpy(nx, 1.0, dg, g, stream));
      double vmin, gmin;
      TRY(gvmin(v, &vmin));
      TRY(gvmin(g, &gmin));
      const double viol = std::max(0.0, std::max(-vmin, -gmin));
      const double shift = 1e3 + 2.0 * viol;
      TRY(pips_hip_vec_add_const(nx, shift, v, stream));
      TRY(pips_hip_vec_add_const(nx, shift, g, stream));
      fix_free();

      int it = 0, status = 1;  // 1 = max iterations
      trace.clear();
      // Numerical-trouble fallback.  Far below the reference's default accuracy (mu 1e-6) the leaf diagonals span sixteen
      // decades and a step can come out useless (step lengths of 1e-16) or harmful (a full step that throws the residual
      // from 1e-11 to 1); the reference answers with its "numerical troubles" logic (InteriorPointMethod.cpp:264-274,
      // PIPSIPMppSolver.cpp:163-185).  Here the iterate with the best merit max(mu / mutol, ||r|| / (artol dnorm)) is kept
      // and returned with status 3 when the iteration breaks down (NaN, residual blow-up, two stalled steps).
      double best_merit = INFINITY, best_rnorm = INFINITY, phi_min = INFINITY;
      int n_stall = 0, n_rstall = 0;
      double prev_rnorm = INFINITY;
      auto merit = [&](double mm, double rr) { return std::max(mm / mutol, rr / (artol * dnorm)); };
      auto save_best = [&]() -> int {
         TRY(pips_hip_vec_copy(nx, x, bx, stream)); TRY(pips_hip_vec_copy(nx, v, bv, stream));
         TRY(pips_hip_vec_copy(nx, g, bg, stream)); TRY(pips_hip_vec_copy(ny, y, by, stream));
         return PIPS_OK;
      };
      auto restore_best = [&]() -> int {
         TRY(pips_hip_vec_copy(nx, bx, x, stream)); TRY(pips_hip_vec_copy(nx, bv, v, stream));
         TRY(pips_hip_vec_copy(nx, bg, g, stream)); TRY(pips_hip_vec_copy(ny, by, y, stream));
         TRY(residuals(&rnorm, &pobj, &dobj));
         TRY(mu(&m));
         return PIPS_OK;
      };
      for (; it < max_iter; ++it) {
         TRY(residuals(&rnorm, &pobj, &dobj));
         TRY(mu(&m));
         const bool is_nan = !(m == m) || !(rnorm == rnorm) || !(pobj == pobj);
         const bool blown = !is_nan && best_merit < INFINITY && rnorm > 1e4 * std::max(best_rnorm, artol * dnorm);
         // complementarity long converged, residual not moving any more (seen with the inexact preconditioner of the native
         // free-variable route): nothing further will come of it
         n_rstall = (!is_nan && m <= 1e-3 * mutol && rnorm > artol * dnorm && rnorm >= 0.99 * prev_rnorm) ? n_rstall + 1 : 0;
         prev_rnorm = rnorm;
         if ((is_nan || blown || n_stall >= 2 || n_rstall >= 3) && best_merit < INFINITY) {
            if (verbose)
               printf("ipm it %3d  numerical troubles (%s: mu %.3e ||r||inf %.3e), falling back to the best iterate\n", it,
                      is_nan ? "nan" : (blown ? "residual blow-up" : (n_rstall >= 3 ? "residual stagnates, mu far below its tolerance" : "stalled")), m, rnorm);
            TRY(restore_best());
            trace.insert(trace.end(), {m, rnorm, pobj, dobj, 0.0, 0.0, 0.0});
            status = (m <= mutol && rnorm <= artol * dnorm) ? 0 : 3;
            break;
         }
         if (!is_nan && merit(m, rnorm) < best_merit) { best_merit = merit(m, rnorm); best_rnorm = rnorm; TRY(save_best()); }
         trace.insert(trace.end(), {m, rnorm, pobj, dobj, 0.0, 0.0, 0.0});   // step data filled in below
         if (verbose)
            printf("ipm it %3d  mu %.3e  ||r||inf %.3e  pobj %.10e  dobj %.10e  (last solve: %d outer its, rel.res %.1e)\n", it, m, rnorm, pobj,
                   dobj, last_outer_steps, last_outer_res);
         if (verbose) fflush(stdout);
         if (is_nan) { status = 2; break; }                                 // numerical breakdown before any usable iterate
         if (m <= mutol && rnorm <= artol * dnorm) { status = 0; break; }   // PIPSIPMppSolver.cpp:143-149
         // "probably infeasible" (PIPSIPMppSolver.cpp:128-170): phi = (||r|| + |gap|) / dnorm, ten iterations in and four
         // orders of magnitude above the best value seen
         {
            const double phi = (rnorm + std::fabs(pobj - dobj)) / dnorm;
            phi_min = it == 0 ? phi : std::min(phi_min, phi);
            if (it >= 10 && phi >= 1e-8 && phi >= 1e4 * phi_min) { status = 4; break; }
         }
         // outer tolerance schedule (InteriorPointMethod.cpp:655-669): 1e-8 up to iteration 3, 1e-9 up to 7, then 1e-10
         outer_tol = it <= 3 ? 1e-8 : (it <= 7 ? 1e-9 : 1e-10);
         // ---- predictor (affine scaling): rgamma = V Gamma e
         TRY(pips_hip_vec_copy(nx, v, rg, stream));
         TRY(pips_hip_vec_mul(nx, g, rg, stream));
         TRY(factorize());
         TRY(solve(rQ, rA, rv, rg, dx, dy, dv, dg));
         double ap, ad;
         TRY(step_lengths(dv, dg, 1.0, &ap, &ad));
         double maff;
         TRY(gdot_shifted(v, ap, dv, g, ad, dg, &maff));
         maff /= nx_global;
         const double sigma = std::pow(maff / m, 3.0);
         // ---- corrector: linear residuals cleared, rgamma = dV_aff dGamma_aff - sigma mu  (set_complementarity_residual(step, -sigma mu))
         TRY(pips_hip_vec_copy(nx, dv, rg, stream));
         TRY(pips_hip_vec_mul(nx, dg, rg, stream));
         TRY(pips_hip_vec_add_const(nx, -sigma * m, rg, stream));
         TRY(solve(zx, zy, zx, rg, cx, cy, cv, cg));   // zx / zy: constant zero vectors (clear_linear_residuals)
         // weighted predictor-corrector step (compute_corrector_step, InteriorPointMethod.cpp:178-206), Gondzio loop, then
         // the step length by Mehrotra's heuristic
         double wp, wd;
         TRY(weight_search(ap, ad, &ap, &ad, &wp, &wd));
         TRY(pips_hip_vec_axpy(nx, wp, cx, dx, stream));
         TRY(pips_hip_vec_axpy(nx, wp, cv, dv, stream));
         TRY(pips_hip_vec_axpy(ny, wd, cy, dy, stream));
         TRY(pips_hip_vec_axpy(nx, wd, cg, dg, stream));
         TRY(gondzio_loop(sigma, m, 1.0, &ap, &ad));
         TRY(mehrotra_step_length(&ap, &ad));
         n_stall = (ap < 1e-10 && ad < 1e-10) ? n_stall + 1 : 0;
         { double* row = trace.data() + trace.size() - 7; row[4] = sigma; row[5] = ap; row[6] = ad; }
         TRY(pips_hip_vec_axpy(nx, ap, dx, x, stream));
         TRY(pips_hip_vec_axpy(nx, ap, dv, v, stream));
         TRY(pips_hip_vec_axpy(ny, ad, dy, y, stream));
         TRY(pips_hip_vec_axpy(nx, ad, dg, g, stream));
      }
      last[0] = pobj; last[1] = it; last[2] = m; last[3] = rnorm; last[4] = status; last[5] = dobj; last[6] = dnorm;
      if (result)
         for (int i = 0; i < 7; ++i) result[i] = last[i];
      return PIPS_OK;
   }
};

}  // namespace pips

using namespace pips;

extern "C" {

int pips_ipm_create(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                    const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                    const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                    const double* F0_val, const double* c, const double* b, double dual_reg, int device) {
   return pips_ipm_create_rank(handle, N, n0, myl, n_i, my_i, W_rowptr, W_colidx, W_val, T_rowptr, T_colidx, T_val, F_rowptr, F_colidx, F_val,
                               F0_rowptr, F0_colidx, F0_val, c, b, dual_reg, device, nullptr, 0, 1);
}

int pips_ipm_create_rank(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                         const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                         const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                         const double* F0_val, const double* c, const double* b, double dual_reg, int device, void* comm, int rank,
                         int n_ranks) {
   if (!handle || N <= 0 || n0 < 0 || myl < 0 || !n_i || !my_i || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !comm))
      PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create: bad arguments");
   auto p = std::make_unique<Ipm>();
   p->N = N; p->n0 = n0; p->myl = myl; p->dual_reg = dual_reg;
   p->comm = comm; p->rank = rank; p->n_ranks = n_ranks;
   std::vector<int> xoff(N + 2, 0), yoff(N + 2, 0);
   std::vector<long long> koff(N + 2, 0);
   xoff[1] = n0; yoff[1] = myl;
   for (int i = 1; i <= N; ++i) {
      xoff[i + 1] = xoff[i] + n_i[i - 1];
      yoff[i + 1] = yoff[i] + my_i[i - 1];
      koff[i + 1] = koff[i] + n_i[i - 1] + my_i[i - 1];
   }
   p->nx = xoff[N + 1]; p->ny = yoff[N + 1]; p->nleaf = koff[N + 1];
   const int S = n0 + myl;
   int rc = pips_hip_batch_create(&p->batch, N, S, device, nullptr);
   if (rc) return rc;
   // ---- per block: K_i pattern/values, border, and rows of the global A
   std::vector<std::vector<std::pair<int, double>>> Arows(p->ny);
   double dn = 0.0;
   long long wp = 0, tp = 0, fp = 0;   // running offsets into the concatenated CSR arrays
   long long wr = 0, tr = 0, fr = 0;   // running row-pointer offsets
   std::vector<std::vector<double>> kvals(N);
   for (int i = 0; i < N; ++i) {
      const int nxi = n_i[i], myi = my_i[i];
      const int* Wrp = W_rowptr + wr; const int* Trp = T_rowptr ? T_rowptr + tr : nullptr; const int* Frp = F_rowptr ? F_rowptr + fr : nullptr;
      const int* Wci = W_colidx + wp; const double* Wv = W_val + wp;
      const int* Tci = T_colidx ? T_colidx + tp : nullptr; const double* Tv = T_val ? T_val + tp : nullptr;
      const int* Fci = F_colidx ? F_colidx + fp : nullptr; const double* Fv = F_val ? F_val + fp : nullptr;
      std::vector<int> Krp(nxi + myi + 1), dpos(nxi + myi);
      rc = pips_kkt_leaf_assemble(nxi, myi, 0, nullptr, nullptr, nullptr, Wrp, Wci, Wv, nullptr, nullptr, nullptr, Krp.data(), nullptr, nullptr, nullptr);
      if (rc) return rc;
      std::vector<int> Kci(Krp[nxi + myi]);
      kvals[i].assign(Krp[nxi + myi], 0.0);
      rc = pips_kkt_leaf_assemble(nxi, myi, 0, nullptr, nullptr, nullptr, Wrp, Wci, Wv, nullptr, nullptr, nullptr, Krp.data(), Kci.data(), kvals[i].data(), dpos.data());
      if (rc) return rc;
      std::vector<int> Brp(S + 1);
      rc = pips_border_assemble(nxi, myi, 0, n0, 0, myl, 0, nullptr, nullptr, nullptr, Trp, Tci, Tv, nullptr, nullptr, nullptr, Frp, Fci, Fv, nullptr, nullptr, nullptr, Brp.data(), nullptr, nullptr);
      if (rc) return rc;
      std::vector<int> Bci(Brp[S]);
      std::vector<double> Bv(Brp[S]);
      rc = pips_border_assemble(nxi, myi, 0, n0, 0, myl, 0, nullptr, nullptr, nullptr, Trp, Tci, Tv, nullptr, nullptr, nullptr, Frp, Fci, Fv, nullptr, nullptr, nullptr, Brp.data(), Bci.data(), Bv.data());
      if (rc) return rc;
      rc = pips_hip_batch_set_block(p->batch, i, nxi + myi, nxi, Krp.data(), Kci.data(), Brp.data(), Bci.data(), Bv.data());
      if (rc) return rc;
      for (int r = 0; r < myi; ++r) {
         auto& row = Arows[yoff[i + 1] + r];
         if (Trp) for (int q = Trp[r] - Trp[0]; q < Trp[r + 1] - Trp[0]; ++q) { row.push_back({Tci[q], Tv[q]}); dn = std::max(dn, std::fabs(Tv[q])); }
         for (int q = Wrp[r] - Wrp[0]; q < Wrp[r + 1] - Wrp[0]; ++q) { row.push_back({xoff[i + 1] + Wci[q], Wv[q]}); dn = std::max(dn, std::fabs(Wv[q])); }
      }
      if (Frp)
         for (int l = 0; l < myl; ++l)
            for (int q = Frp[l] - Frp[0]; q < Frp[l + 1] - Frp[0]; ++q) { Arows[l].push_back({xoff[i + 1] + Fci[q], Fv[q]}); dn = std::max(dn, std::fabs(Fv[q])); }
      wp += Wrp[myi] - Wrp[0]; wr += myi + 1;
      if (Trp) { tp += Trp[myi] - Trp[0]; tr += myi + 1; }
      if (Frp) { fp += Frp[myl] - Frp[0]; fr += myl + 1; }
   }
   if (F0_rowptr)
      for (int l = 0; l < myl; ++l)
         for (int q = F0_rowptr[l]; q < F0_rowptr[l + 1]; ++q) {
            if (rank == 0) Arows[l].push_back({F0_colidx[q], F0_val[q]});   // the replicated root block enters the summed products once
            dn = std::max(dn, std::fabs(F0_val[q]));
         }
   for (int j = 0; j < p->nx; ++j) dn = std::max(dn, std::fabs(c[j]));
   for (int r = 0; r < p->ny; ++r) dn = std::max(dn, std::fabs(b[r]));
   p->dnorm = dn > 0 ? dn : 1.0;
   // CSR of A and A^T
   std::vector<int> Arp(p->ny + 1, 0), Atrp(p->nx + 1, 0);
   for (int r = 0; r < p->ny; ++r) {
      std::sort(Arows[r].begin(), Arows[r].end());
      Arp[r + 1] = Arp[r] + (int)Arows[r].size();
      for (auto& e : Arows[r]) ++Atrp[e.first + 1];
   }
   std::vector<int> Aci(Arp[p->ny]), Atci(Arp[p->ny]);
   std::vector<double> Av(Arp[p->ny]), Atv(Arp[p->ny]);
   for (int j = 0; j < p->nx; ++j) Atrp[j + 1] += Atrp[j];
   {
      std::vector<int> fill(Atrp.begin(), Atrp.end() - 1);
      for (int r = 0; r < p->ny; ++r) {
         int q = Arp[r];
         for (auto& e : Arows[r]) {
            Aci[q] = e.first; Av[q] = e.second; ++q;
            const int t = fill[e.first]++;
            Atci[t] = r; Atv[t] = e.second;
         }
      }
   }
   // PIPS_IPM_SPARSE_ROOT=1: keep the Schur complement sparse and factorise it with the sparse engine (2-link problems)
   const bool sparse_root = getenv("PIPS_IPM_SPARSE_ROOT") && atoi(getenv("PIPS_IPM_SPARSE_ROOT")) != 0;
   if (sparse_root && n_ranks > 1)
      PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_rank: the sparse root needs the border column sets of all blocks on every rank (pips_hip_kkt_create_sparse); "
                              "the harness passes only its own - use the dense root with several ranks");
   if (sparse_root && (rc = pips_hip_batch_set_schur_mode(p->batch, 1))) return rc;
   rc = pips_hip_batch_analyze(p->batch, 16);
   if (rc) return rc;
   for (int i = 0; i < N; ++i)
      if ((rc = pips_hip_batch_set_values(p->batch, i, kvals[i].data()))) return rc;
   if ((rc = pips_hip_batch_set_refinement_backward_error(p->batch, 2, 1e-15))) return rc;   // PARDISO iparm[7]=2 semantics
   if (sparse_root)
      rc = pips_hip_kkt_create_sparse(&p->kkt, p->batch, n0, 0, myl, 0, nullptr, nullptr, nullptr, F0_rowptr, F0_colidx, F0_val, nullptr, nullptr,
                                      nullptr, 0, nullptr, nullptr, comm, rank, n_ranks);
   else
      rc = pips_hip_kkt_create(&p->kkt, p->batch, n0, 0, myl, 0, nullptr, nullptr, nullptr, F0_rowptr, F0_colidx, F0_val, nullptr, nullptr, nullptr,
                               comm, rank, n_ranks);
   if (rc) return rc;
   HIP_TRYH(hipGetDevice(&p->device));
   if ((rc = p->alloc(&p->d_red, 24 * (long long)n_ranks))) return rc;
   // data norm and number of complementarity pairs over all ranks
   if ((rc = p->gext(&p->dnorm, true))) return rc;
   {
      double pairs = (double)(p->nx - n0);
      if ((rc = p->gsum(&pairs))) return rc;
      p->nx_global = (long long)pairs + n0;
   }
   {
      std::vector<int> la, lat;
      for (int r = 0; r < p->ny; ++r) if (Arp[r + 1] - Arp[r] > CSR_LONG_ROW) la.push_back(r);
      for (int r = 0; r < p->nx; ++r) if (Atrp[r + 1] - Atrp[r] > CSR_LONG_ROW) lat.push_back(r);
      p->nA_long = (int)la.size(); p->nAt_long = (int)lat.size();
      la.push_back(0); lat.push_back(0);   // never upload an empty array
      if ((rc = p->up(&p->A_long, la)) || (rc = p->up(&p->At_long, lat))) return rc;
   }
   if ((rc = p->up(&p->A_rp, Arp)) || (rc = p->up(&p->A_ci, Aci)) || (rc = p->up(&p->A_v, Av)) || (rc = p->up(&p->At_rp, Atrp)) ||
       (rc = p->up(&p->At_ci, Atci)) || (rc = p->up(&p->At_v, Atv)) || (rc = p->up(&p->d_xoff, xoff)) || (rc = p->up(&p->d_yoff, yoff)) ||
       (rc = p->up(&p->d_koff, koff)))
      return rc;
   std::vector<double> hc(c, c + p->nx), hb(b, b + p->ny);
   if ((rc = p->up(&p->c, hc)) || (rc = p->up(&p->b, hb))) return rc;
   double** xs[] = {&p->x, &p->v, &p->g, &p->rQ, &p->rv, &p->rg, &p->dd, &p->dx, &p->dv, &p->dg, &p->cx, &p->cv, &p->cg, &p->tx, &p->zx, &p->gv, &p->gg, &p->bx, &p->bv, &p->bg};
   for (auto d : xs)
      if ((rc = p->alloc(d, p->nx))) return rc;
   double** ys[] = {&p->y, &p->rA, &p->dy, &p->cy, &p->ty, &p->zy, &p->by};
   for (auto d : ys)
      if ((rc = p->alloc(d, p->ny))) return rc;
   double** zs[] = {&p->bz, &p->xz, &p->w_r, &p->w_r0, &p->w_best, &p->w_v, &p->w_t, &p->w_p, &p->w_dx};
   for (auto d : zs)
      if ((rc = p->alloc(d, (long long)p->nx + p->ny))) return rc;
   if ((rc = p->alloc(&p->b0, S)) || (rc = p->alloc(&p->bl, p->nleaf)) || (rc = p->alloc(&p->leaf_diag, p->nleaf))) return rc;
   *handle = p.release();
   return PIPS_OK;
}

int pips_ipm_solve(void* handle, int max_iter, double mutol, double artol, int verbose, double* result7) {
   Ipm* p = (Ipm*)handle;
   if (!p) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   return p->run(max_iter, mutol, artol, verbose, result7);
}

int pips_ipm_get_solution(void* handle, double* x_host, double* y_host) {
   Ipm* p = (Ipm*)handle;
   if (!p) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (x_host) HIP_TRYH(hipMemcpy(x_host, p->x, (size_t)p->nx * sizeof(double), hipMemcpyDeviceToHost));
   if (y_host) HIP_TRYH(hipMemcpy(y_host, p->y, (size_t)p->ny * sizeof(double), hipMemcpyDeviceToHost));
   return PIPS_OK;
}

int pips_ipm_set_gondzio(void* handle, int max_correctors) {
   Ipm* p = (Ipm*)handle;
   if (!p || max_correctors < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_gondzio: bad arguments");
   p->max_gondzio = max_correctors;
   return PIPS_OK;
}

int pips_ipm_set_free_variables(void* handle, const double* bounded_mask_host) {
   Ipm* p = (Ipm*)handle;
   if (!p || !bounded_mask_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: bad arguments");
   HIP_TRYH(hipSetDevice(p->device));
   double local_bounded = 0.0;
   bool any_free = false;
   for (int j = 0; j < p->nx; ++j) {
      if (bounded_mask_host[j] != 0.0 && bounded_mask_host[j] != 1.0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: mask entries must be 0 or 1");
      any_free |= bounded_mask_host[j] == 0.0;
      if (j >= p->n0 || p->rank == 0) local_bounded += bounded_mask_host[j];
   }
   int rc;
   if (!p->fmask && ((rc = p->alloc(&p->fmask, p->nx)) || (rc = p->alloc(&p->ddp, p->nx)))) return rc;
   HIP_TRYH(hipMemcpy(p->fmask, bounded_mask_host, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice));
   // the number of complementarity pairs (divisor of mu) and whether any rank has free entries: over all ranks
   double flags[2] = {local_bounded, any_free ? 1.0 : 0.0};
   if (p->n_ranks > 1 && (rc = p->reduce_host(flags, 2))) return rc;
   p->nx_global = (long long)flags[0];
   p->has_free = flags[1] > 0.0;
   if (p->nx_global <= 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: no bounded variable left");
   return PIPS_OK;
}

int pips_ipm_set_option(void* handle, const char* name, double value) {
   Ipm* p = (Ipm*)handle;
   if (!p || !name) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_option: bad arguments");
   const std::string key(name);
   // identifiers of the reference's option tables (Options.C:18-73, PIPSIPMppOptions.C:170-264)
   if (key == "GONDZIO_MAX_CORRECTORS") {
      if (value < 0) PIPS_FAIL(PIPS_ERR_ARG, "GONDZIO_MAX_CORRECTORS must be >= 0");
      p->max_gondzio = (int)value;
   }
   else if (key == "OUTER_SOLVE") {
      if (value != 1 && value != 2) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_SOLVE: 1 (iterative refinement) or 2 (BiCGStab) - the harness always runs an outer solve");
      p->outer_mode = (int)value;
   }
   else if (key == "OUTER_BICG_MAX_ITER") {
      if (value < 1) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_BICG_MAX_ITER must be >= 1");
      p->bicg_max_iter = (int)value;
   }
   else if (key == "REGULARIZATION") p->regularize = value != 0.0;
   else if (key == "FREE_VARIABLE_PROXIMAL_TERM") {   // not a reference identifier: diagonal of the free variables in the preconditioner
      if (!(value > 0.0)) PIPS_FAIL(PIPS_ERR_ARG, "FREE_VARIABLE_PROXIMAL_TERM must be > 0");
      p->free_reg = value;
   }
   else PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_option: unknown or unsupported identifier %s", name);
   return PIPS_OK;
}

int pips_ipm_get_trace(void* handle, double* rows7, int max_rows, int* n_rows) {
   Ipm* p = (Ipm*)handle;
   if (!p || !n_rows) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_get_trace: bad arguments");
   const int have = (int)(p->trace.size() / 7);
   *n_rows = have;
   if (rows7)
      for (int i = 0; i < std::min(have, max_rows) * 7; ++i) rows7[i] = p->trace[i];
   return PIPS_OK;
}

int pips_ipm_get_stats(void* handle, long long* stats4) {
   Ipm* p = (Ipm*)handle;
   if (!p || !stats4) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_get_stats: bad arguments");
   stats4[0] = p->n_factorize; stats4[1] = p->n_regularised + p->n_refactor_outer; stats4[2] = p->n_precond; stats4[3] = p->n_gondzio;
   return PIPS_OK;
}

void pips_ipm_destroy(void* handle) { delete (Ipm*)handle; }

}  // extern "C"
