// Host harness: a Mehrotra predictor-corrector interior-point loop with Gondzio correctors driving the device KKT path end to
// end, for the reference's full problem class
//      min c^T x   s.t.  A x = b,   clow <= C x <= cupp (each side optional per row),   xlow <= x <= xupp (optional per entry)
// with block-angular A and C (root rows, block rows, linking rows).  It is the build's counterpart of the reference callers
// that do not travel to the GPU box:
//   PIPSIPMppSolver::solve            (InteriorPointMethod/PIPSIPMppSolver.cpp:29-83)   start point, loop, termination
//   Solver::solve_linear_system       (InteriorPointMethod/Solver.cpp:19-31)            initial affine solve + shift
//   InteriorPointMethod predictor / corrector / Gondzio loop / Mehrotra step length (InteriorPointMethod.cpp:68-90,178-358,745-812)
//   LinearSystem::computeDiagonals / solve / solveXYZS / system_mult (LinearSystem.C:262-294,327-548,808-844)
//   LinearSystem::solveCompressedBiCGStab (LinearSystem.C:550-798)
//   Residuals::evaluate / set_complementarity_residual / project_r3 (Residuals.cpp:58-171,220-290)
//   DistributedMatrix::mult / transpose_mult (LinearAlgebra/Distributed/DistributedMatrix.C:224-326)
//   Variables::mu / mustep_pd / stepbound_pd / find_blocking / push_to_interior / shift_bound_variables (Variables.C:88-403)
// Everything numeric runs on the device; the host sees scalars only.
//
// Device layout.  One iterate (and likewise a step, a corrector, the best iterate) is ONE array
//      [ x | s | t | u | v | w ]  [ y | z | lambda | pi | gamma | phi ]
//        primal part (step length alpha_p)          dual part (alpha_d)
// x: [x0 | x_1 .. x_N];  y: [y0 | y_link | y_1 .. y_N];  s, z, t, u, lambda, pi: [z0 | z_link | z_1 .. z_N] rows;
// v, w, gamma, phi: like x.  The four complementarity pairs are therefore two flat vectors G = [t|u|v|w] and
// L = [lambda|pi|gamma|phi] with the 0/1 mask M = [iclow|icupp|ixlow|ixupp]: mu, step bounds, the blocking entry, the weight
// search and the Gondzio projection are single passes over G and L - every Variables method above loops over the four pairs
// with exactly this mask semantics.  The constraint matrices live as one CSR J = [A; C] (rows [y rows | z rows]) and J^T.
//
// Kernel count per Residuals::evaluate: 2 SpMV with fused epilogues (rQ; [rA|rC]), 1 element-wise kernel (rz, rt, ru, rv, rw),
// 1 multi-reduction (+ its finishing workgroup).  The outer BiCGStab keeps every scalar on the device: reductions feed
// single-thread "logic" kernels, vector updates read alpha / omega / beta from device memory, branches are predicated by
// flags; the host reads one status record per iteration.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "pips_hip.h"

extern "C" {
int pips_hip_vec_axpy(long long, double, const double*, double*, void*);
int pips_hip_vec_copy(long long, const double*, double*, void*);
int pips_hip_vec_set(long long, double, double*, void*);
int pips_hip_vec_mul(long long, const double*, double*, void*);
int pips_hip_vec_add_product(long long, double, const double*, const double*, double*, void*);
int pips_hip_vec_gondzio_projection(long long, double, double, double*, void*);
int pips_hip_vec_weighted_stepbounds(long long, const double*, const double*, const double*, const double*, const double*, const double*,
                                     double, int, double*, void*);
int pips_hip_vec_find_blocking(long long, const double*, const double*, const double*, const double*, double*, void*);
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

static inline int egrid(long long n) { return (int)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256)); }

// ---------------------------------------------------------------------------------------------------------------------------
// block-angular SpMV with fused epilogues (f-2).  One thread per row (rows have ~10 entries); rows longer than CSR_LONG_ROW
// get a workgroup each (the x0 rows of J^T collect the T_i entries of every block).  `lin`: this rank is not the special
// one - on replicated output rows (ranges rep0 = [0, r0e), rep1 = [r1b, r1e)) only the linear term is produced, the rest of
// the epilogue is added once by rank 0 and the rows are summed over the ranks afterwards (DistributedMatrix.C:224-326).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CSR_LONG_ROW = 512;
enum SpmvMode : int { SP_RQ = 0, SP_RAC, SP_KX, SP_KYZ, SP_RES_X, SP_RES_YZ };

struct SpmvArgs {
   int nrows;
   const int* rp; const int* ci; const double* v;   // CSR
   const double* in;                                 // multiplied vector
   double* out;
   const double *e0, *e1, *e2, *e3;                  // epilogue operands, meaning per mode
   int split;                                        // SP_RAC: rows < split subtract e0[r], rows >= split subtract e1[r - split]
   int lin, r0e, r1b, r1e;
   const int* pred;                                  // run only if *pred != 0 (nullptr: always)
};

template <int MODE>
__device__ __forceinline__ double spmv_epilogue(const SpmvArgs& a, int r, double s) {
   const bool rep = a.lin && (r < a.r0e || (r >= a.r1b && r < a.r1e));
   switch (MODE) {
      case SP_RQ:     return rep ? -s : a.e0[r] - s - a.e1[r] + a.e2[r];                       // rQ = c - J^T[y;z] - gamma + phi
      case SP_RAC:    return rep ? s : s - (r < a.split ? a.e0[r] : a.e1[r - a.split]);        // [rA|rC] = J x - [b|s]
      case SP_KX:     return rep ? s : a.e0[r] * a.e1[r] + s;                                  // dd .* x + J^T[y;z]
      case SP_KYZ:    return rep ? s : s + a.e0[r] * a.e1[r];                                  // J x + [0|nOmegaInv] .* [y;z]
      case SP_RES_X:  return rep ? -s : a.e2[r] - (a.e0[r] * a.e1[r] + s);                     // rhs_x - (K z)_x
      default:        return rep ? -s : a.e2[r] - (s + a.e0[r] * a.e1[r]);                     // rhs_yz - (K z)_yz
   }
}

// Eight lanes per row (rows of the constraint Jacobian and of its transpose hold a handful of entries: a thread per row left seven
// of eight lanes' worth of every 64-byte request unused and serialised a row's gathers): a wave reads the entries of eight consecutive
// rows as one contiguous piece; two rows per lane group are in flight.  Bandwidth: profiles/r4_harness_spmv.txt.
template <int MODE, int LPR = 8>   // LPR lanes per row: 8, or 4 where the rows average fewer than six entries (J^T of an LP)
__global__ __launch_bounds__(256) void k_spmv(SpmvArgs a) {
   if (a.pred && *a.pred == 0) return;
   constexpr int RU = 2, SH = LPR == 8 ? 3 : 2;
   const int l = threadIdx.x & (LPR - 1);
   const long long step = ((long long)gridDim.x * blockDim.x) >> SH, chunk = step * RU;
   const long long i_end = (a.nrows + chunk - 1) / chunk * chunk;   // every lane of a wave makes the same trips (shuffles below)
   for (long long r0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> SH; r0 < i_end; r0 += chunk) {
      int b[RU], e[RU];
      double s[RU];
      bool mine[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
         const long long r = r0 + u * step;
         mine[u] = r < a.nrows;
         b[u] = mine[u] ? a.rp[r] : 0;
         e[u] = mine[u] ? a.rp[r + 1] : 0;
         mine[u] = mine[u] && e[u] - b[u] <= CSR_LONG_ROW;          // the others: k_spmv_long
         s[u] = 0.0;
      }
#pragma unroll
      for (int u = 0; u < RU; ++u)
         if (mine[u])
            for (int p = b[u] + l; p < e[u]; p += LPR) s[u] += a.v[p] * a.in[a.ci[p]];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
         s[u] += __shfl_xor(s[u], 1);
         s[u] += __shfl_xor(s[u], 2);
         if (LPR == 8) s[u] += __shfl_xor(s[u], 4);
         if (mine[u] && l == 0) { const int r = (int)(r0 + u * step); a.out[r] = spmv_epilogue<MODE>(a, r, s[u]); }
      }
   }
}

// Long rows (the columns of the first-stage variables in J^T: ~135 000 entries each on the time-coupled share, 95 of them): a row is cut
// into LONG_PARTS pieces, one workgroup each (a workgroup per ROW left all but 95 compute units idle: 425 us, as long as the rest of the
// product); the pieces' sums go to a scratch array and are added in a fixed order by the finishing kernel, which applies the epilogue.
constexpr int LONG_PARTS = 32;
__global__ __launch_bounds__(256) void k_spmv_long_part(SpmvArgs a, const int* __restrict__ long_rows, double* __restrict__ scratch) {
   if (a.pred && *a.pred == 0) return;
   __shared__ double red[256];
   const int r = long_rows[blockIdx.x];
   const long long b = a.rp[r], e = a.rp[r + 1], len = (e - b + LONG_PARTS - 1) / LONG_PARTS;
   const long long p0 = b + len * blockIdx.y, p1 = p0 + len < e ? p0 + len : e;
   double s = 0.0;
   for (long long p = p0 + threadIdx.x; p < p1; p += 256) s += a.v[p] * a.in[a.ci[p]];
   red[threadIdx.x] = s;
   __syncthreads();
   for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
   }
   if (threadIdx.x == 0) scratch[(long long)blockIdx.x * LONG_PARTS + blockIdx.y] = red[0];
}
template <int MODE>
__global__ __launch_bounds__(256) void k_spmv_long_finish(SpmvArgs a, const int* __restrict__ long_rows, int n_long, const double* __restrict__ scratch) {
   if (a.pred && *a.pred == 0) return;
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_long) return;
   double s = 0.0;
   for (int k = 0; k < LONG_PARTS; ++k) s += scratch[(long long)i * LONG_PARTS + k];
   const int r = long_rows[i];
   a.out[r] = spmv_epilogue<MODE>(a, r, s);
}

// ---------------------------------------------------------------------------------------------------------------------------
// fused multi-reduction: up to RED_MAX terms in one launch (grid.y = term), partials per workgroup, one finishing workgroup
// per term.  The result stays on the device (BiCGStab) or is copied to the host in one transfer (IPM loop).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int RED_MAX = 12;
constexpr int RED_GRID = 256;
enum RedKind : int {
   R_DOT = 0,       // sum w a b
   R_ABSMAX,        // max |a|
   R_MIN_MASKED,    // min a over entries with c != 0
   R_STEPBOUND,     // min -a / b over b < 0
   R_DOT_SHIFTED,   // sum w (a + p c)(b + q d)
};
struct RedTerm {
   int kind; long long n;
   const double *a, *b, *c, *d, *w;
   double p, q;
};
struct RedPack { int n_terms; RedTerm t[RED_MAX]; };

__device__ __forceinline__ double red_id(int kind) { return (kind == R_MIN_MASKED || kind == R_STEPBOUND) ? INFINITY : 0.0; }
__device__ __forceinline__ double red_comb(int kind, double u, double v) {
   return kind == R_ABSMAX ? fmax(u, v) : ((kind == R_MIN_MASKED || kind == R_STEPBOUND) ? fmin(u, v) : u + v);
}

__global__ __launch_bounds__(256) void k_multi_reduce(RedPack pk, double* __restrict__ partial, const int* pred) {
   if (pred && *pred == 0) return;
   const RedTerm t = pk.t[blockIdx.y];
   double acc = red_id(t.kind);
   // the kind is uniform over the launch row: one loop per kind, four entries per thread and trip with all their loads issued before the
   // first use (a scalar loop with the switch inside streamed 12.8 M-entry vectors at 2 TB/s)
   const long long stride = (long long)gridDim.x * blockDim.x, i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
   auto sweep = [&](auto term) {
      long long i = i0;
      for (; i + 3 * stride < t.n; i += 4 * stride) {
         const double v0 = term(i), v1 = term(i + stride), v2 = term(i + 2 * stride), v3 = term(i + 3 * stride);
         acc = red_comb(t.kind, acc, red_comb(t.kind, red_comb(t.kind, v0, v1), red_comb(t.kind, v2, v3)));
      }
      for (; i < t.n; i += stride) acc = red_comb(t.kind, acc, term(i));
   };
   switch (t.kind) {
      case R_DOT:
         if (t.w) sweep([&](long long i) { return t.a[i] * t.b[i] * t.w[i]; });
         else sweep([&](long long i) { return t.a[i] * t.b[i]; });
         break;
      case R_ABSMAX: sweep([&](long long i) { return fabs(t.a[i]); }); break;
      case R_MIN_MASKED: sweep([&](long long i) { return t.c[i] != 0.0 ? t.a[i] : (double)INFINITY; }); break;
      case R_STEPBOUND: sweep([&](long long i) { return t.b[i] < 0.0 ? -t.a[i] / t.b[i] : (double)INFINITY; }); break;
      default:
         if (t.w) sweep([&](long long i) { return (t.a[i] + t.p * t.c[i]) * (t.b[i] + t.q * t.d[i]) * t.w[i]; });
         else sweep([&](long long i) { return (t.a[i] + t.p * t.c[i]) * (t.b[i] + t.q * t.d[i]); });
         break;
   }
   __shared__ double red[256];
   red[threadIdx.x] = acc;
   __syncthreads();
   for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = red_comb(t.kind, red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0) partial[blockIdx.y * RED_GRID + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_multi_reduce_final(RedPack pk, int grid_x, const double* __restrict__ partial,
                                                           double* __restrict__ out, const int* pred) {
   if (pred && *pred == 0) return;
   const int kind = pk.t[blockIdx.x].kind;
   double acc = red_id(kind);
   for (int i = threadIdx.x; i < grid_x; i += blockDim.x) acc = red_comb(kind, acc, partial[blockIdx.x * RED_GRID + i]);
   __shared__ double red[256];
   red[threadIdx.x] = acc;
   __syncthreads();
   for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = red_comb(kind, red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------------------
// element-wise kernels of Residuals / LinearSystem
// ---------------------------------------------------------------------------------------------------------------------------
struct Lay {   // sizes and offsets shared by the kernels
   int nx, my, mz;
   long long ncp;   // 2 mz + 2 nx
};

// rz = z - lambda + pi ; rt = (s - clow) iclow - t ; ru = (s - cupp) icupp + u ; rv = (x - xlow) ixlow - v ; rw = (x - xupp) ixupp + w
// (Residuals.cpp:90-170).  G = [t|u|v|w], L = [lambda|pi|gamma|phi], M the masks, Bd = [clow|cupp|xlow|xupp], rG = [rt|ru|rv|rw]
__global__ void k_bound_residuals(Lay d, const double* __restrict__ x, const double* __restrict__ s, const double* __restrict__ z,
                                  const double* __restrict__ G, const double* __restrict__ L, const double* __restrict__ M,
                                  const double* __restrict__ Bd, double* __restrict__ rz, double* __restrict__ rG) {
   const long long n = d.nx > d.mz ? d.nx : d.mz;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (i < d.mz) {
         rz[i] = z[i] - L[i] + L[d.mz + i];
         rG[i] = (s[i] - Bd[i]) * M[i] - G[i];
         rG[d.mz + i] = (s[i] - Bd[d.mz + i]) * M[d.mz + i] + G[d.mz + i];
      }
      if (i < d.nx) {
         const long long o = 2LL * d.mz;
         rG[o + i] = (x[i] - Bd[o + i]) * M[o + i] - G[o + i];
         rG[o + d.nx + i] = (x[i] - Bd[o + d.nx + i]) * M[o + d.nx + i] + G[o + d.nx + i];
      }
   }
}

// computeDiagonals (LinearSystem.C:262-294): dd = gamma/v [ixlow] + phi/w [ixupp]; nOmegaInv = -safe_invert(lambda/t + pi/u);
// ddp = dd + free_reg on entries without any bound (preconditioner only); dyz = [0 (my) | nOmegaInv (mz)]
__global__ void k_diagonals(Lay d, const double* __restrict__ G, const double* __restrict__ L, const double* __restrict__ M,
                            double free_reg, double* __restrict__ dd, double* __restrict__ ddp, double* __restrict__ dyz) {
   const long long n = d.nx > d.mz ? d.nx : d.mz;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (i < d.mz) {
         double om = 0.0;
         if (M[i] != 0.0) om += L[i] / G[i];
         if (M[d.mz + i] != 0.0) om += L[d.mz + i] / G[d.mz + i];
         dyz[d.my + i] = om != 0.0 ? -1.0 / om : 0.0;
      }
      if (i < d.nx) {
         const long long o = 2LL * d.mz;
         double v = 0.0;
         const bool lo = M[o + i] != 0.0, up = M[o + d.nx + i] != 0.0;
         if (lo) v += L[o + i] / G[o + i];
         if (up) v += L[o + d.nx + i] / G[o + d.nx + i];
         dd[i] = v;
         ddp[i] = v + ((lo || up) ? 0.0 : free_reg);
      }
   }
}

// OUTER_BICG_TEST_PRECOND (test hook, see Ipm::precond): z := 0, or z[i] *= 1 - 2 i / (n - 1)
__global__ void k_test_distort(long long n, int mode, double* __restrict__ z) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      z[i] = mode == 1 ? 0.0 : z[i] * (n > 1 ? 1.0 - 2.0 * (double)i / (double)(n - 1) : 1.0);
}

// out[k] = in[k] - reg: the regularised dual diagonal of the eliminated root inequality rows (sLinsysRootAug.C:384-466 eliminates z0 with
// dual_inequality_diagonal_regularized)
__global__ void k_shift_diag(const double* __restrict__ in, double reg, double* __restrict__ out, int n) {
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = in[k] - reg;
}

// K_i diagonals of every leaf from the primal diagonal, the dual regularisation and nOmegaInv: code[p] >= 0 an x index,
// -1 an equality row, <= -2 the inequality row -2 - code[p]
__global__ void k_leaf_diag(long long nleaf, const long long* __restrict__ code, const double* __restrict__ ddp,
                            const double* __restrict__ nomega, double primal_reg, double dual_reg, double reg,
                            double* __restrict__ leaf_diag) {
   for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < nleaf; p += (long long)gridDim.x * blockDim.x) {
      const long long c = code[p];
      leaf_diag[p] = c >= 0 ? ddp[c] + primal_reg : (c == -1 ? -dual_reg : nomega[-2 - c] - reg);
   }
}

// diagonals of the REGULARISED system the factors belong to: x rows dd (or ddp) + reg, equality rows -(dual_reg + reg), inequality rows
// nOmegaInv - reg (k_leaf_diag / pips_hip_kkt_set_root_regularization put the same terms into the factorised matrix)
__global__ void k_reg_operator(long long nx, long long my, long long mz, const double* __restrict__ ddp, const double* __restrict__ dyz,
                               double reg, double dual_reg, double* __restrict__ dop_r, double* __restrict__ dyz_r) {
   const long long n = nx > my + mz ? nx : my + mz;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (i < nx) dop_r[i] = ddp[i] + reg;
      if (i < my) dyz_r[i] = -(dual_reg + reg);
      else if (i < my + mz) dyz_r[i] = dyz[i] - reg;
   }
}

// LinearSystem::solve, reduction of the right-hand side (LinearSystem.C:327-395) for one residual set:
//   rx = rQ + G/V rv + rgamma/V + Phi/W rw - rphi/W ; rs = rz + L/T rt + rlambda/T + Pi/U ru - rpi/U ; rhs = [rx | rA | rC - nOmegaInv rs]
// zero_lin: the linear residuals (rQ, rA, rC, rz, rt, ru, rv, rw) are zero (corrector / Gondzio right-hand sides:
// clear_linear_residuals); rs is kept for the recovery
__global__ void k_rhs_reduce(Lay d, int zero_lin, const double* __restrict__ rQ, const double* __restrict__ rAC, const double* __restrict__ rz,
                             const double* __restrict__ rG, const double* __restrict__ rL, const double* __restrict__ G,
                             const double* __restrict__ L, const double* __restrict__ M, const double* __restrict__ dyz,
                             double* __restrict__ rhs, double* __restrict__ rs) {
   const long long n = d.nx > d.mz ? (d.nx > d.my ? d.nx : d.my) : (d.mz > d.my ? d.mz : d.my);
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (i < d.nx) {
         const long long o = 2LL * d.mz;
         double r = zero_lin ? 0.0 : rQ[i];
         if (M[o + i] != 0.0) r += ((zero_lin ? 0.0 : L[o + i] * rG[o + i]) + rL[o + i]) / G[o + i];
         if (M[o + d.nx + i] != 0.0) r += ((zero_lin ? 0.0 : L[o + d.nx + i] * rG[o + d.nx + i]) - rL[o + d.nx + i]) / G[o + d.nx + i];
         rhs[i] = r;
      }
      if (i < d.my) rhs[d.nx + i] = zero_lin ? 0.0 : rAC[i];
      if (i < d.mz) {
         double r = zero_lin ? 0.0 : rz[i];
         if (M[i] != 0.0) r += ((zero_lin ? 0.0 : L[i] * rG[i]) + rL[i]) / G[i];
         if (M[d.mz + i] != 0.0) r += ((zero_lin ? 0.0 : L[d.mz + i] * rG[d.mz + i]) - rL[d.mz + i]) / G[d.mz + i];
         rs[i] = r;
         rhs[d.nx + d.my + i] = (zero_lin ? 0.0 : rAC[d.my + i]) - dyz[d.my + i] * r;
      }
   }
}

// recovery (LinearSystem.C:396-447 and the tail of solveXYZS :542-547) + step.negate(): sol = [Dx | Dy' | Dz'] of the reduced
// system; step = -(Dx, Ds, Dt, Du, Dv, Dw | Dy, Dz, Dlambda, Dpi, Dgamma, Dphi) with Dy = -Dy', Dz = -Dz'
__global__ void k_recover(Lay d, int zero_lin, const double* __restrict__ sol, const double* __restrict__ rs, const double* __restrict__ rG,
                          const double* __restrict__ rL, const double* __restrict__ G, const double* __restrict__ L,
                          const double* __restrict__ M, const double* __restrict__ dyz, double* __restrict__ sP, double* __restrict__ sD) {
   // sP = [dx | ds | dt | du | dv | dw], sD = [dy | dz | dlambda | dpi | dgamma | dphi]
   const long long n = d.nx > d.mz ? (d.nx > d.my ? d.nx : d.my) : (d.mz > d.my ? d.mz : d.my);
   double* dG = sP + d.nx + d.mz;
   double* dL = sD + d.my + d.mz;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (i < d.nx) {
         const long long o = 2LL * d.mz;
         const double dx = sol[i];
         sP[i] = -dx;
         double dv = 0.0, dg = 0.0, dw = 0.0, dp = 0.0;
         if (M[o + i] != 0.0) {
            dv = dx - (zero_lin ? 0.0 : rG[o + i]);
            dg = (rL[o + i] - L[o + i] * dv) / G[o + i];
         }
         if (M[o + d.nx + i] != 0.0) {
            dw = (zero_lin ? 0.0 : rG[o + d.nx + i]) - dx;
            dp = (rL[o + d.nx + i] - L[o + d.nx + i] * dw) / G[o + d.nx + i];
         }
         dG[o + i] = -dv; dL[o + i] = -dg; dG[o + d.nx + i] = -dw; dL[o + d.nx + i] = -dp;
      }
      if (i < d.my) sD[i] = sol[d.nx + i];            // -(Dy) = -(-Dy')
      if (i < d.mz) {
         const double dzp = sol[d.nx + d.my + i];     // Dz'
         const double dz = -dzp;
         sD[d.my + i] = dzp;                           // -(Dz)
         const double ds = -(dyz[d.my + i] * (rs[i] - dz));
         sP[d.nx + i] = -ds;
         double dt = 0.0, dl = 0.0, du = 0.0, dpi = 0.0;
         if (M[i] != 0.0) {
            dt = ds - (zero_lin ? 0.0 : rG[i]);
            dl = (rL[i] - L[i] * dt) / G[i];
         }
         if (M[d.mz + i] != 0.0) {
            du = (zero_lin ? 0.0 : rG[d.mz + i]) - ds;
            dpi = (rL[d.mz + i] - L[d.mz + i] * du) / G[d.mz + i];
         }
         dG[i] = -dt; dL[i] = -dl; dG[d.mz + i] = -du; dL[d.mz + i] = -dpi;
      }
   }
}

// complementarity right-hand sides over the flat pair vectors: mode 0  r = G .* L ; mode 1  r = dG .* dL + alpha M ;
// mode 2 (Gondzio)  r = -proj((G + ap dG) .* (L + ad dL)) .* M with the projection step onto [rmin, rmax], floor -rmax
__global__ void k_compl_rhs(long long n, int mode, const double* __restrict__ G, const double* __restrict__ L, const double* __restrict__ dG,
                            const double* __restrict__ dL, const double* __restrict__ M, double alpha, double ap, double ad, double rmin,
                            double rmax, double* __restrict__ r) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      double v;
      if (mode == 0) v = G[i] * L[i];
      else if (mode == 1) v = dG[i] * dL[i] + alpha * M[i];
      else {
         const double p = (G[i] + ap * dG[i]) * (L[i] + ad * dL[i]);
         double t = p < rmin ? rmin - p : (p > rmax ? rmax - p : 0.0);
         t = t < -rmax ? -rmax : t;
         v = -t * M[i];
      }
      r[i] = v;
   }
}

// y[i] (+)= a where mask: push_to_interior / shift_bound_variables
__global__ void k_masked_const(long long n, int add, double a, const double* __restrict__ M, double* __restrict__ y) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      y[i] = M[i] != 0.0 ? (add ? y[i] + a : a) : 0.0;
}

// gather / scatter between the flat [x|y|z] vector and the KKT right-hand sides (b0 = [x0|y0|z0|ylink|zlink], leaves [x_i|y_i|z_i])
__global__ void k_gather(long long n, const long long* __restrict__ src, const double* __restrict__ in, double* __restrict__ out, int scatter) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      if (scatter) out[src[i]] = in[i]; else out[i] = in[src[i]];
   }
}

// ---------------------------------------------------------------------------------------------------------------------------
// device-resident BiCGStab (f-1): scalar state, logic kernels, predicated vector kernels
// ---------------------------------------------------------------------------------------------------------------------------
enum BicgSlot : int {
   B_RHO = 0, B_RHO_LAST, B_ALPHA, B_OMEGA, B_BETA, B_RN, B_MIN_RN, B_TARGET, B_BN, B_TRUE_RN, B_EPS,
   B_FLAG,        // IterativeSolverSolutionStatus of the reference: see BicgFlag
   B_ACTIVE, B_IT, B_NDIV, B_NSTAG, B_MAX_DIV, B_MAX_STAG,
   B_RED0,        // 8 reduction results
   B_SLOTS = B_RED0 + 8
};
enum BicgFlag : int { BF_RUNNING = 0, BF_CONVERGED = 1, BF_SKIPPED = 2, BF_MAX_ITER = 3, BF_BREAKDOWN = 4, BF_DIVERGED = 5, BF_STAGNATION = 6 };
// integer predicates read by the vector kernels
enum BicgPred : int { P_ACTIVE = 0, P_NEED_TRUE, P_NEED_BEST, P_COPY_BEST, P_ROLLBACK, P_FIRST, P_COUNT };

__device__ __forceinline__ bool bz(double v) { return fabs(v) < 1e-40; }                       // PIPSisZero with pips_eps0 (pipsdef.h:35,108)
__device__ __forceinline__ bool bbad(double v) { return !(v == v) || fabs(v) > 1e300; }

// phase numbers follow the order of the kernels in Ipm::bicgstab
__global__ void k_bicg_logic(int phase, double* __restrict__ st, int* __restrict__ pr) {
   if (threadIdx.x != 0 || blockIdx.x != 0) return;
   const double* red = st + B_RED0;
   auto stop = [&](int flag) { st[B_FLAG] = flag; st[B_ACTIVE] = 0.0; pr[P_ACTIVE] = 0; };
   auto stagnation = [&](double step, double dxn2, double xn2) {
      if (fabs(step) * sqrt(dxn2) <= st[B_EPS] * sqrt(xn2)) st[B_NSTAG] += 1.0; else st[B_NSTAG] = 0.0;
   };
   if (phase == 0) {   // after the initial residual: red[0] = ||b||^2, red[1] = ||r||^2
      const double bn = sqrt(red[0]), rn = sqrt(red[1]);
      st[B_BN] = bn; st[B_RN] = rn; st[B_MIN_RN] = rn; st[B_TRUE_RN] = rn;
      st[B_RHO] = st[B_ALPHA] = st[B_OMEGA] = 1.0;
      st[B_NDIV] = st[B_NSTAG] = st[B_IT] = 0.0;
      st[B_FLAG] = BF_RUNNING; st[B_ACTIVE] = 1.0;
      for (int k = 0; k < P_COUNT; ++k) pr[k] = 0;
      pr[P_ACTIVE] = 1; pr[P_FIRST] = 1;
      if (rn <= st[B_TARGET]) stop(BF_SKIPPED);
      if (bbad(rn)) stop(BF_BREAKDOWN);
      return;
   }
   if (st[B_ACTIVE] == 0.0) return;
   switch (phase) {
      case 1: {   // red[0] = <r0, r>
         st[B_RHO_LAST] = st[B_RHO];
         st[B_RHO] = red[0];
         if (bz(st[B_RHO]) || bbad(st[B_RHO])) { stop(BF_BREAKDOWN); break; }
         if (st[B_IT] > 0.0) {
            st[B_BETA] = (st[B_RHO] / st[B_RHO_LAST]) * (st[B_ALPHA] / st[B_OMEGA]);
            if (bz(st[B_BETA]) || bbad(st[B_BETA])) stop(BF_BREAKDOWN);
         }
         break;
      }
      case 2: {   // red[0] = <r0, v>, red[1] = ||dx||^2, red[2] = ||x||^2
         if (bz(red[0]) || bbad(red[0])) { stop(BF_BREAKDOWN); break; }
         st[B_ALPHA] = st[B_RHO] / red[0];
         if (bbad(st[B_ALPHA])) { stop(BF_BREAKDOWN); break; }
         stagnation(st[B_ALPHA], red[1], red[2]);
         break;
      }
      case 3:     // first half done: red[0] = ||r||^2 (predicted)
      case 7: {   // second half done
         const double rn = sqrt(red[0]);
         if (bbad(rn)) { stop(BF_BREAKDOWN); break; }
         st[B_RN] = rn;
         pr[P_NEED_TRUE] = 0; pr[P_NEED_BEST] = 0; pr[P_COPY_BEST] = 0; pr[P_ROLLBACK] = 0;
         if (phase == 3) {
            if (rn <= st[B_TARGET]) pr[P_NEED_TRUE] = 1;
         } else {
            if (rn <= st[B_TARGET] || st[B_NSTAG] >= st[B_MAX_STAG]) pr[P_NEED_TRUE] = 1;
            else {
               if (rn >= st[B_MIN_RN]) st[B_NDIV] += 1.0; else st[B_NDIV] = 0.0;
               if (st[B_NDIV] > st[B_MAX_DIV]) {   // rollback to the best iterate
                  pr[P_ROLLBACK] = 1;
                  st[B_RN] = st[B_MIN_RN];
                  stop(BF_DIVERGED);
               }
            }
         }
         break;
      }
      case 4:     // true residual of x computed (if requested): red[0] = ||b - K x||^2
      case 8: {
         if (pr[P_NEED_TRUE]) {
            const double tr = sqrt(red[0]);
            st[B_TRUE_RN] = tr;
            if (tr <= st[B_TARGET]) { st[B_RN] = tr; stop(BF_CONVERGED); break; }
            pr[P_NEED_BEST] = 1;   // the guess was bad: go on with the actual residual, re-evaluate the rollback candidate
            st[B_RN] = tr;
         }
         break;
      }
      case 5:     // residual of best_x recomputed (if requested): red[0]; then the rollback bookkeeping
      case 9: {
         if (pr[P_NEED_BEST]) st[B_MIN_RN] = sqrt(red[0]);
         pr[P_COPY_BEST] = 0;
         if (st[B_RN] < st[B_MIN_RN]) { st[B_MIN_RN] = st[B_RN]; pr[P_COPY_BEST] = 1; }
         break;
      }
      case 6: {   // red[0] = <t,t>, red[1] = <t,r>, red[2] = ||dx||^2, red[3] = ||x||^2
         if (bz(red[0]) || bbad(red[0])) { stop(BF_BREAKDOWN); break; }
         st[B_OMEGA] = red[1] / red[0];
         if (bbad(st[B_OMEGA])) { stop(BF_BREAKDOWN); break; }
         stagnation(st[B_OMEGA], red[2], red[3]);
         break;
      }
      case 10: {  // end of the iteration
         pr[P_ROLLBACK] = 0;
         if (st[B_NSTAG] >= st[B_MAX_STAG]) {
            if (st[B_MIN_RN] < st[B_RN]) { st[B_RN] = st[B_MIN_RN]; pr[P_ROLLBACK] = 1; }
            stop(BF_STAGNATION);
            break;
         }
         if (bz(st[B_OMEGA])) { stop(BF_BREAKDOWN); break; }
         st[B_IT] += 1.0;
         pr[P_FIRST] = 0;
         break;
      }
   }
}

// p = r (first iteration) or r + beta (p - omega v)
__global__ void k_bicg_p(long long n, const double* __restrict__ st, const int* __restrict__ pr, const double* __restrict__ r,
                         const double* __restrict__ v, double* __restrict__ p) {
   if (!pr[P_ACTIVE]) return;
   const bool first = pr[P_FIRST] != 0;
   const double beta = st[B_BETA], omega = st[B_OMEGA];
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      p[i] = first ? r[i] : r[i] + beta * (p[i] - omega * v[i]);
}
// x += a dx ; r -= a q   with a = st[slot]
__global__ void k_bicg_update(long long n, int slot, const double* __restrict__ st, const int* __restrict__ pr, const double* __restrict__ dx,
                              const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r) {
   if (!pr[P_ACTIVE]) return;
   const double a = st[slot];
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      x[i] += a * dx[i];
      r[i] -= a * q[i];
   }
}
__global__ void k_copy_pred(long long n, const int* __restrict__ flag, const double* __restrict__ src, double* __restrict__ dst) {
   if (flag && *flag == 0) return;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_scale_by_inv(long long n, const double* __restrict__ st, int slot, double* __restrict__ y) {
   const double a = 1.0 / st[slot];
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] *= a;
}

// target = max(tol ||b||, eps) from red[0] = ||b||^2
__global__ void k_bicg_set_target(double* st, double tol) {
   if (threadIdx.x == 0 && blockIdx.x == 0) st[B_TARGET] = fmax(sqrt(st[B_RED0]) * tol, st[B_EPS]);
}

// ---------------------------------------------------------------------------------------------------------------------------
struct Vars {   // pointers into one iterate-shaped array
   double *base = nullptr, *P = nullptr, *D = nullptr;
   double *x = nullptr, *s = nullptr, *G = nullptr, *yz = nullptr, *y = nullptr, *z = nullptr, *L = nullptr;
   void bind(double* b, int nx, int my, int mz, long long ncp) {
      base = b; P = b; x = b; s = b + nx; G = b + nx + mz;
      D = b + nx + mz + ncp; yz = D; y = D; z = D + my; L = D + my + mz;
   }
};

struct Csr { std::vector<int> rp, ci; std::vector<double> v; };

struct Ipm {
   int device = 0;
   hipStream_t stream = nullptr;
   // dimensions
   int N = 0, n0 = 0, my0 = 0, mz0 = 0, myl = 0, mzl = 0;
   int e_mz0 = 0, e_mzl = 0;   // what the KKT engine is told: root inequality rows either eliminated (e_mz0 = mz0) or kept as rows among the linking inequalities
   int nx = 0, my = 0, mz = 0;
   long long ncp = 0, nxyz = 0, NP = 0, ND = 0, nleaf = 0;
   int S = 0;
   Lay lay{};
   double dnorm = 1.0, dual_reg = 0.0, free_reg = 1e-6;
   double n_pairs = 0.0;      // complementarity pairs over all ranks (replicated root parts counted once)
   void* batch = nullptr;
   void* kkt = nullptr;
   // matrices J = [A; C] and J^T
   int *J_rp = nullptr, *J_ci = nullptr, *Jt_rp = nullptr, *Jt_ci = nullptr, *J_long = nullptr, *Jt_long = nullptr;
   double *J_v = nullptr, *Jt_v = nullptr;
   int nJ_long = 0, nJt_long = 0;
   double* long_scratch = nullptr;   // LONG_PARTS partial sums per long row (k_spmv_long_part)
   long long J_nnz = 0, Jt_nnz = 0;
   // data
   double *c = nullptr, *bA = nullptr, *M = nullptr, *Bd = nullptr, *wG = nullptr, *wXYZ = nullptr, *wX = nullptr, *wY = nullptr;
   long long *d_pack = nullptr, *d_code = nullptr;
   long long npack = 0;
   // state
   Vars it, st, co, best;
   double *rQ = nullptr, *rAC = nullptr, *rz = nullptr, *rG = nullptr, *rL = nullptr, *rs = nullptr;
   double *dd = nullptr, *ddp = nullptr, *dyz = nullptr, *leaf_diag = nullptr, *b0 = nullptr, *bl = nullptr;
   double *dop_r = nullptr, *dyz_r = nullptr;   // diagonals of the regularised system (k_reg_operator)
   double* zd0 = nullptr;                       // eliminated root inequality rows: nOmegaInv - dual regularisation (dual_inequality_diagonal_regularized)
   double *rhs = nullptr, *sol = nullptr, *w_r = nullptr, *w_r0 = nullptr, *w_best = nullptr, *w_v = nullptr, *w_t = nullptr, *w_p = nullptr,
          *w_dx = nullptr, *w_tmp = nullptr;
   double *d_partial = nullptr, *d_out = nullptr, *h_out = nullptr, *d_bst = nullptr, *h_bst = nullptr;
   int* d_pred = nullptr;
   // options / counters
   int max_gondzio = 2;
   long long n_gondzio = 0, n_precond = 0, n_bicg_iter = 0, n_host_syncs = 0;
   int outer_mode = 2, outer_max = 10, last_outer_steps = 0, last_outer_flag = 0;
   int bicg_max_iter = 75, bicg_max_div = 4, bicg_max_stag = 4;
   double bicg_eps = 1e-15;   // OUTER_BICG_EPSILON (PIPSIPMppOptions.C:303-310): absolute floor of the target, scale of the stagnation test
   bool regularize = true;
   double outer_tol = 1e-10, last_outer_res = 0.0, last_outer_abs = 0.0;
   int n_regularised = 0, n_factorize = 0, n_refactor_outer = 0, verbose_run = 0;
   double last_reg = 0.0;
   std::vector<void*> owned;
   double last[8] = {0};
   std::vector<double> trace;
   // ranks
   void* comm = nullptr;
   int rank = 0, n_ranks = 1;
   double* d_red = nullptr;
   int ry = 0, rzr = 0;   // replicated leading rows of y-type (my0 + myl) and z-type (mz0 + mzl) vectors

   ~Ipm() {
      if (kkt) pips_hip_kkt_destroy(kkt);
      if (batch) pips_hip_batch_destroy(batch);
      for (void* p : owned)
         if (p) (void)hipFree(p);
      if (h_out) (void)hipHostFree(h_out);
      if (h_bst) (void)hipHostFree(h_bst);
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

   // ---- scalars over the ranks: sums are added, maxima / minima travel in one slot per rank of a summed vector -------------
   int reduce_host(double* vals, int n) {
      HIP_TRYH(hipMemcpyAsync(d_red, vals, n * sizeof(double), hipMemcpyHostToDevice, stream));
      TRY(pips_hip_allreduce_sum(comm, d_red, (size_t)n, stream));
      HIP_TRYH(hipMemcpyAsync(vals, d_red, n * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRYH(hipStreamSynchronize(stream));
      return PIPS_OK;
   }
   int root_sum(double* part, int n) { return (n_ranks > 1 && n > 0) ? pips_hip_allreduce_sum(comm, part, (size_t)n, stream) : PIPS_OK; }

   // one fused reduction -> host.  out[k] for every term; over the ranks: sums added, extrema combined
   int reduce(RedPack& pk, double* out) {
      hipLaunchKernelGGL(k_multi_reduce, dim3(RED_GRID, pk.n_terms), dim3(256), 0, stream, pk, d_partial, (const int*)nullptr);
      hipLaunchKernelGGL(k_multi_reduce_final, dim3(pk.n_terms), dim3(256), 0, stream, pk, RED_GRID, d_partial, d_out, (const int*)nullptr);
      HIP_TRYH(hipMemcpyAsync(h_out, d_out, pk.n_terms * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRYH(hipStreamSynchronize(stream));
      ++n_host_syncs;
      for (int k = 0; k < pk.n_terms; ++k) out[k] = h_out[k];
      if (n_ranks > 1) {
         // one all-reduce for everything: sums are added in row 0, extrema travel in one row per rank (infinities do not survive
         // a sum: "no entry" is sent as 1e300)
         const int nt = pk.n_terms;
         std::vector<double> slots((size_t)nt * n_ranks, 0.0);
         auto is_ext = [](int kind) { return kind == R_ABSMAX || kind == R_MIN_MASKED || kind == R_STEPBOUND; };
         for (int k = 0; k < nt; ++k) {
            if (is_ext(pk.t[k].kind)) slots[(size_t)rank * nt + k] = out[k] < 1e300 ? out[k] : 1e300;
            else slots[k] = out[k];
         }
         TRY(reduce_host(slots.data(), nt * n_ranks));
         for (int k = 0; k < nt; ++k) {
            const int kind = pk.t[k].kind;
            if (!is_ext(kind)) { out[k] = slots[k]; continue; }
            double m = slots[k];
            for (int r = 1; r < n_ranks; ++r) m = kind == R_ABSMAX ? std::max(m, slots[(size_t)r * nt + k]) : std::min(m, slots[(size_t)r * nt + k]);
            out[k] = (kind != R_ABSMAX && m >= 1e300) ? INFINITY : m;
         }
      }
      return PIPS_OK;
   }
   static RedTerm term(int kind, long long n, const double* a, const double* b = nullptr, const double* c = nullptr, const double* d = nullptr,
                       const double* w = nullptr, double p = 0.0, double q = 0.0) {
      RedTerm t; t.kind = kind; t.n = n; t.a = a; t.b = b; t.c = c; t.d = d; t.w = w; t.p = p; t.q = q;
      return t;
   }

   // ---- SpMV ------------------------------------------------------------------------------------------------------------------
   template <int MODE>
   int spmv(bool transposed, const double* in, double* out, const double* e0, const double* e1, const double* e2, const double* e3,
            const int* pred = nullptr) {
      SpmvArgs a;
      a.nrows = transposed ? nx : my + mz;
      a.rp = transposed ? Jt_rp : J_rp; a.ci = transposed ? Jt_ci : J_ci; a.v = transposed ? Jt_v : J_v;
      a.in = in; a.out = out; a.e0 = e0; a.e1 = e1; a.e2 = e2; a.e3 = e3; a.split = my;
      a.lin = rank != 0;
      if (transposed) { a.r0e = n0; a.r1b = a.r1e = 0; }
      else { a.r0e = ry; a.r1b = my; a.r1e = my + rzr; }
      a.pred = pred;
      // several ranks + predicate: the all-reduce of the replicated rows runs on every rank whatever the predicate says, so the
      // product goes to a scratch vector and is copied over under the predicate afterwards
      const bool staged = n_ranks > 1 && pred != nullptr;
      if (staged) a.out = transposed ? w_tmp : w_tmp + nx;
      // eight lanes per row (four where the rows are short), two rows per lane group
      const long long nnz_m = transposed ? Jt_nnz : J_nnz;
      if (a.nrows > 0 && nnz_m < 6LL * a.nrows) hipLaunchKernelGGL((k_spmv<MODE, 4>), dim3(egrid(2LL * a.nrows)), dim3(256), 0, stream, a);
      else if (a.nrows > 0) hipLaunchKernelGGL((k_spmv<MODE, 8>), dim3(egrid(4LL * a.nrows)), dim3(256), 0, stream, a);
      const int nl = transposed ? nJt_long : nJ_long;
      if (nl > 0) {
         hipLaunchKernelGGL(k_spmv_long_part, dim3(nl, LONG_PARTS), dim3(256), 0, stream, a, transposed ? Jt_long : J_long, long_scratch);
         hipLaunchKernelGGL(k_spmv_long_finish<MODE>, dim3((nl + 255) / 256), dim3(256), 0, stream, a, transposed ? Jt_long : J_long, nl, long_scratch);
      }
      HIP_TRYH(hipGetLastError());
      if (n_ranks > 1) {   // replicated rows: sum of the ranks' contributions
         if (transposed) TRY(root_sum(a.out, n0));
         else { TRY(root_sum(a.out, ry)); TRY(root_sum(a.out + my, rzr)); }
      }
      if (staged) hipLaunchKernelGGL(k_copy_pred, dim3(egrid(a.nrows)), dim3(256), 0, stream, (long long)a.nrows, pred, a.out, out);
      return PIPS_OK;
   }

   // ---- Residuals::evaluate -----------------------------------------------------------------------------------------------------
   int residuals(double* rnorm, double* pobj, double* dobj, double* mu_out) {
      TRY((spmv<SP_RQ>(true, it.yz, rQ, c, it.L + 2 * mz, it.L + 2 * mz + nx, nullptr)));
      TRY((spmv<SP_RAC>(false, it.x, rAC, bA, it.s, nullptr, nullptr)));
      hipLaunchKernelGGL(k_bound_residuals, dim3(egrid(std::max(nx, mz))), dim3(256), 0, stream, lay, it.x, it.s, it.z, it.G, it.L, M, Bd, rz, rG);
      RedPack pk;
      pk.n_terms = 8;
      pk.t[0] = term(R_ABSMAX, nx, rQ);
      pk.t[1] = term(R_ABSMAX, my + mz, rAC);
      pk.t[2] = term(R_ABSMAX, mz, rz);
      pk.t[3] = term(R_ABSMAX, ncp, rG);
      pk.t[4] = term(R_DOT, nx, c, it.x, nullptr, nullptr, wX);          // primal objective
      pk.t[5] = term(R_DOT, my, bA, it.y, nullptr, nullptr, wY);         // b^T y
      pk.t[6] = term(R_DOT, ncp, Bd, it.L, nullptr, nullptr, wGs);       // clow^T lambda - cupp^T pi + xlow^T gamma - xupp^T phi (signs in wGs)
      pk.t[7] = term(R_DOT, ncp, it.G, it.L, nullptr, nullptr, wG);      // complementarity
      double o[8];
      TRY(reduce(pk, o));
      *rnorm = std::max(std::max(o[0], o[1]), std::max(o[2], o[3]));
      for (int k = 0; k < 4; ++k) last_rparts[k] = o[k];
      *pobj = o[4];
      *dobj = o[5] + o[6];
      *mu_out = n_pairs > 0 ? o[7] / n_pairs : 0.0;
      cur_mu = *mu_out;
      return PIPS_OK;
   }
   double cur_mu = 1.0, free_reg_min = 1e-10, reg_max = 1e-2, reg_eager_max = 1e-2;
   // eager: factorize() adds regularisation until no pivot is reported perturbed (factorize_with_correct_inertia).  lazy: a
   // perturbed static pivot is a rank-one error of the preconditioner, which the outer Krylov solve absorbs in a few
   // iterations - regularisation (which perturbs every pivot) is added only when the outer solve fails to converge
   bool eager_inertia_loop = true;
   int last_pert = 0;
   bool free_reg_follows_mu = true;
   double last_rparts[4] = {0, 0, 0, 0};   // inf-norms of rQ, [rA|rC], rz, [rt|ru|rv|rw] of the last evaluation
   double* wGs = nullptr;   // wG with the sign pattern [+|-|+|-] of the dual objective

   // ---- LinearSystem::factorize with the inertia contract (LinearSystem.C:171-202,295-325) ----------------------------------------
   int perturbed_pivots(int* total) {
      int p_, n_, z_;
      TRY(pips_hip_kkt_root_inertia(kkt, &p_, &n_, &z_));
      *total = z_;
      double leaves = 0.0;
      for (int b = 0; b < N; ++b) {
         TRY(pips_hip_batch_inertia(batch, b, &p_, &n_, &z_));
         leaves += z_;
      }
      if (n_ranks > 1) TRY(reduce_host(&leaves, 1));
      *total += (int)leaves;
      return PIPS_OK;
   }
   int factorize(double reg_start = 0.0) {
      // proximal term of the free variables in the preconditioner: a variable without bounds is a basic variable whose bounds are
      // infinitely far, and the basic variables' own diagonal gamma/v is O(mu) - so the term follows mu downwards (a fixed 1e-6
      // dominates those diagonals by three orders of magnitude at mu = 1e-9 and the outer solve stops converging)
      const double freg = free_reg_follows_mu ? std::min(free_reg, std::max(cur_mu, free_reg_min)) : free_reg;
      hipLaunchKernelGGL(k_diagonals, dim3(egrid(std::max(nx, mz))), dim3(256), 0, stream, lay, it.G, it.L, M, freg, dd, ddp, dyz);
      double reg = reg_start;
      // Escalation that does not lower the number of perturbed pivots only ruins the preconditioner (a pivot can be perturbed for
      // reasons regularisation does not touch): remember the smallest regularisation with the lowest count, stop after two
      // escalations without progress and go back to it.
      int best_pert = INT_MAX, stalled = 0;
      double best_reg = reg;
      auto factor_with = [&](double r, int* pert) -> int {
         if (nleaf > 0)
            hipLaunchKernelGGL(k_leaf_diag, dim3(egrid(nleaf)), dim3(256), 0, stream, nleaf, d_code, ddp, dyz + my, r, dual_reg + r, r, leaf_diag);
         TRY(pips_hip_kkt_set_root_regularization(kkt, r, dual_reg + r));
         if (e_mz0 > 0) hipLaunchKernelGGL(k_shift_diag, dim3(egrid(e_mz0)), dim3(256), 0, stream, dyz + my, dual_reg + r, zd0, e_mz0);
         TRY(pips_hip_kkt_factorize(kkt, leaf_diag, ddp, e_mzl > 0 ? dyz + my + e_mz0 : nullptr));
         ++n_factorize;
         TRY(perturbed_pivots(pert));
         if (verbose_run && (*pert || r > 0.0)) printf("   factorize: regularisation %.1e, %d perturbed pivots\n", r, *pert);
         return PIPS_OK;
      };
      int pert = 0;
      for (int attempt = 0;; ++attempt) {
         TRY(factor_with(reg, &pert));
         if (pert < best_pert) { best_pert = pert; best_reg = reg; stalled = 0; }
         else ++stalled;
         if (pert == 0 || attempt == 4 || stalled >= 2 || !regularize || !eager_inertia_loop || reg >= 0.5 * std::max(reg_eager_max, reg_start)) break;
         reg = reg == 0.0 ? 1e-8 : std::min(reg * 100.0, reg_max);
         ++n_regularised;
      }
      if (pert > best_pert || (pert == best_pert && best_reg < reg && pert > 0)) {
         reg = best_reg;
         TRY(factor_with(reg, &pert));
      }
      last_pert = pert;
      last_reg = reg;
      hipLaunchKernelGGL(k_reg_operator, dim3(egrid(std::max<long long>(nx, (long long)my + mz))), dim3(256), 0, stream, (long long)nx, (long long)my,
                         (long long)mz, free_in_operator ? ddp : dd, dyz, reg, dual_reg, dop_r, dyz_r);
      return PIPS_OK;
   }

   // ---- outer solve on z = [x | y | z] ------------------------------------------------------------------------------------------
   // Test hook (option OUTER_BICG_TEST_PRECOND): distort the preconditioner so that the exits of the outer BiCGStab that a healthy
   // factorisation never takes can be driven on the device - 1: it returns zero (breakdown: r0^T v = 0, LinearSystem.C:640-647),
   // 2: its output is scaled entry-wise by a ramp from +1 to -1 (an operator far from normal: the residual norm climbs, the
   // divergence counter runs up and the best iterate is restored, :741-760).  0 in every production path.
   int test_precond = 0;
   int precond(double* z_) {   // z := solveCompressed(z)
      hipLaunchKernelGGL(k_gather, dim3(egrid(npack)), dim3(256), 0, stream, npack, d_pack, z_, b0, 0);   // b0 and bl are one array
      TRY(pips_hip_kkt_solve_compressed(kkt, b0, bl));
      hipLaunchKernelGGL(k_gather, dim3(egrid(npack)), dim3(256), 0, stream, npack, d_pack, b0, z_, 1);
      if (test_precond) hipLaunchKernelGGL(k_test_distort, dim3(egrid(nxyz)), dim3(256), 0, stream, nxyz, test_precond, z_);
      ++n_precond;
      return PIPS_OK;
   }
   // out = K z with K = [dd J^T; J diag(0, nOmegaInv)]  (LinearSystem::system_mult on the unregularised system)
   // Free variables (no bound: dd_j = 0): optionally the proximal term of the preconditioner is part of the operator too, i.e. the
   // outer solve runs on the primal-regularised system (the reference's choice when OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM is off,
   // LinearSystem.C:505-512 use_regularized_system) - a proximal-point step centred at the current iterate.  Off by default:
   // it made no difference to the 2 % of seeded LPs with free variables that end with status 3 (tools/native_sweep.py).
   // reg_operator: the outer solve runs on the REGULARISED system - primal regularisation on x, dual regularisation on the equality
   // and inequality rows, the terms the factorised matrix carries - i.e. the step is a regularised Newton step.  PIPS-IPM++'s own
   // default (PIPSIPMppOptions.C:293 OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM false; LinearSystem.C:505 use_regularized_system,
   // compute_regularized_system_residuals :806-845); option OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM 0 selects it here for every solve.
   // Default here: the original system first (the reference's known-answer iteration counts are reproduced that way: with
   // dependent equality rows the regularised step of an instance like parallelEqualityRows_B0A2 needs 6 iterations instead
   // of 4), and the regularised system for the rest of an IPM iteration as soon as an outer solve on the original one has been long
   // (> 12 BiCGStab iterations) or inexact: with dependent rows the original K is singular, the null-space component of y is
   // whatever rounding makes of it, and a long Krylov run with a preconditioner whose eigenvalue there is 1 / dual_reg blows it
   // up to 1e12 - hier_approach_4blocks_2by3 ended with status 3 / 4 in 3 % of its runs that way.  The switch costs no
   // factorisation: the factors already belong to the regularised system.
   // The proximal term of free variables is a device of this preconditioner, not one of the reference's regularisation
   // diagonals: it enters the operator only with FREE_VARIABLE_PROXIMAL_IN_OPERATOR.
   const double* dop() const { return reg_operator ? dop_r : (free_in_operator ? ddp : dd); }
   const double* dyzop() const { return reg_operator ? dyz_r : dyz; }
   bool free_in_operator = false;
   bool reg_operator = false, reg_operator_always = false;
   int n_reg_operator = 0;
   int kmult(const double* z_, double* out, const int* pred = nullptr) {
      TRY((spmv<SP_KX>(true, z_ + nx, out, dop(), z_, nullptr, nullptr, pred)));
      TRY((spmv<SP_KYZ>(false, z_, out + nx, dyzop(), z_ + nx, nullptr, nullptr, pred)));
      return PIPS_OK;
   }
   int kresidual(const double* rhs_, const double* z_, double* r, const int* pred = nullptr) {   // r = rhs - K z
      TRY((spmv<SP_RES_X>(true, z_ + nx, r, dop(), z_, rhs_, nullptr, pred)));
      TRY((spmv<SP_RES_YZ>(false, z_, r + nx, dyzop(), z_ + nx, rhs_ + nx, nullptr, pred)));
      return PIPS_OK;
   }
   // device-side reduction for BiCGStab: results to st[B_RED0 ..], summed over the ranks
   int bred(RedPack& pk, const int* pred) {
      hipLaunchKernelGGL(k_multi_reduce, dim3(RED_GRID, pk.n_terms), dim3(256), 0, stream, pk, d_partial, pred);
      hipLaunchKernelGGL(k_multi_reduce_final, dim3(pk.n_terms), dim3(256), 0, stream, pk, RED_GRID, d_partial, d_bst + B_RED0, pred);
      if (n_ranks > 1) TRY(pips_hip_allreduce_sum(comm, d_bst + B_RED0, (size_t)pk.n_terms, stream));
      return PIPS_OK;
   }
   void logic(int phase) { hipLaunchKernelGGL(k_bicg_logic, dim3(1), dim3(1), 0, stream, phase, d_bst, d_pred); }
   int read_state() {
      HIP_TRYH(hipMemcpyAsync(h_bst, d_bst, B_SLOTS * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRYH(hipStreamSynchronize(stream));
      ++n_host_syncs;
      return PIPS_OK;
   }
   // OUTER_SOLVE 2, the reference's default: BiCGStab preconditioned by solveCompressed (LinearSystem.C:550-798) - same half-step
   // structure, convergence tests on the predicted residual confirmed by the true one, best-iterate rollback, divergence and
   // stagnation counters, breakdown tests with PIPSisZero.  Scalars stay on the device; the host reads the state once per
   // iteration (and once for the "skipped" test).
   int bicgstab(const double* b_, double* x_) {
      const long long n = nxyz;
      double init[B_SLOTS] = {0};
      init[B_EPS] = bicg_eps; init[B_MAX_DIV] = bicg_max_div; init[B_MAX_STAG] = bicg_max_stag;
      HIP_TRYH(hipMemcpyAsync(d_bst, init, sizeof(init), hipMemcpyHostToDevice, stream));
      TRY(pips_hip_vec_copy(n, b_, x_, stream));
      TRY(precond(x_));
      TRY(kresidual(b_, x_, w_r));
      RedPack pk;
      pk.n_terms = 2;
      pk.t[0] = term(R_DOT, n, b_, b_, nullptr, nullptr, wXYZ);
      pk.t[1] = term(R_DOT, n, w_r, w_r, nullptr, nullptr, wXYZ);
      TRY(bred(pk, nullptr));
      // target = max(tol ||b||, eps) needs ||b||: set by a tiny logic launch reading red[0]
      hipLaunchKernelGGL(k_bicg_set_target, dim3(1), dim3(1), 0, stream, d_bst, outer_tol);
      logic(0);
      TRY(pips_hip_vec_copy(n, x_, w_best, stream));
      TRY(read_state());
      last_outer_steps = 0;
      auto finish = [&]() {
         const double bn = h_bst[B_BN];
         last_outer_flag = (int)h_bst[B_FLAG];
         last_outer_abs = h_bst[B_RN];
         last_outer_res = bn > 0 ? h_bst[B_RN] / bn : h_bst[B_RN];
      };
      if (h_bst[B_ACTIVE] == 0.0) { finish(); return PIPS_OK; }     // "skipped": the common case (LinearSystem.C:591-600)
      TRY(pips_hip_vec_copy(n, w_r, w_r0, stream));
      hipLaunchKernelGGL(k_scale_by_inv, dim3(egrid(n)), dim3(256), 0, stream, n, d_bst, (int)B_RN, w_r0);
      const int* pA = d_pred + P_ACTIVE;
      int iters = 0;
      for (; iters < bicg_max_iter; ++iters) {
         // ---- first half
         pk.n_terms = 1;
         pk.t[0] = term(R_DOT, n, w_r0, w_r, nullptr, nullptr, wXYZ);
         TRY(bred(pk, pA));
         logic(1);
         hipLaunchKernelGGL(k_bicg_p, dim3(egrid(n)), dim3(256), 0, stream, n, d_bst, d_pred, w_r, w_v, w_p);
         TRY(pips_hip_vec_copy(n, w_p, w_dx, stream));
         TRY(precond(w_dx));
         TRY(kmult(w_dx, w_v, pA));
         pk.n_terms = 3;
         pk.t[0] = term(R_DOT, n, w_r0, w_v, nullptr, nullptr, wXYZ);
         pk.t[1] = term(R_DOT, n, w_dx, w_dx, nullptr, nullptr, wXYZ);
         pk.t[2] = term(R_DOT, n, x_, x_, nullptr, nullptr, wXYZ);
         TRY(bred(pk, pA));
         logic(2);
         hipLaunchKernelGGL(k_bicg_update, dim3(egrid(n)), dim3(256), 0, stream, n, (int)B_ALPHA, d_bst, d_pred, w_dx, w_v, x_, w_r);
         pk.n_terms = 1;
         pk.t[0] = term(R_DOT, n, w_r, w_r, nullptr, nullptr, wXYZ);
         TRY(bred(pk, pA));
         logic(3);
         TRY(kresidual(b_, x_, w_r, d_pred + P_NEED_TRUE));      // actual residual, only if the predicted one passed
         TRY(bred(pk, d_pred + P_NEED_TRUE));
         logic(4);
         TRY(kresidual(b_, w_best, w_dx, d_pred + P_NEED_BEST)); // the guess was bad: re-evaluate the rollback candidate
         pk.t[0] = term(R_DOT, n, w_dx, w_dx, nullptr, nullptr, wXYZ);
         TRY(bred(pk, d_pred + P_NEED_BEST));
         logic(5);
         hipLaunchKernelGGL(k_copy_pred, dim3(egrid(n)), dim3(256), 0, stream, n, d_pred + P_COPY_BEST, x_, w_best);
         // ---- second half
         hipLaunchKernelGGL(k_copy_pred, dim3(egrid(n)), dim3(256), 0, stream, n, pA, w_r, w_dx);
         TRY(precond(w_dx));
         TRY(kmult(w_dx, w_t, pA));
         pk.n_terms = 4;
         pk.t[0] = term(R_DOT, n, w_t, w_t, nullptr, nullptr, wXYZ);
         pk.t[1] = term(R_DOT, n, w_t, w_r, nullptr, nullptr, wXYZ);
         pk.t[2] = term(R_DOT, n, w_dx, w_dx, nullptr, nullptr, wXYZ);
         pk.t[3] = term(R_DOT, n, x_, x_, nullptr, nullptr, wXYZ);
         TRY(bred(pk, pA));
         logic(6);
         hipLaunchKernelGGL(k_bicg_update, dim3(egrid(n)), dim3(256), 0, stream, n, (int)B_OMEGA, d_bst, d_pred, w_dx, w_t, x_, w_r);
         pk.n_terms = 1;
         pk.t[0] = term(R_DOT, n, w_r, w_r, nullptr, nullptr, wXYZ);
         TRY(bred(pk, pA));
         logic(7);
         hipLaunchKernelGGL(k_copy_pred, dim3(egrid(n)), dim3(256), 0, stream, n, d_pred + P_ROLLBACK, w_best, x_);   // diverged: roll back
         TRY(kresidual(b_, x_, w_r, d_pred + P_NEED_TRUE));
         TRY(bred(pk, d_pred + P_NEED_TRUE));
         logic(8);
         TRY(kresidual(b_, w_best, w_dx, d_pred + P_NEED_BEST));
         pk.t[0] = term(R_DOT, n, w_dx, w_dx, nullptr, nullptr, wXYZ);
         TRY(bred(pk, d_pred + P_NEED_BEST));
         logic(9);
         hipLaunchKernelGGL(k_copy_pred, dim3(egrid(n)), dim3(256), 0, stream, n, d_pred + P_COPY_BEST, x_, w_best);
         logic(10);
         hipLaunchKernelGGL(k_copy_pred, dim3(egrid(n)), dim3(256), 0, stream, n, d_pred + P_ROLLBACK, w_best, x_);   // stagnation: best iterate
         TRY(read_state());
         ++n_bicg_iter;
         if (h_bst[B_ACTIVE] == 0.0) { ++iters; break; }
      }
      if (h_bst[B_ACTIVE] != 0.0) h_bst[B_FLAG] = BF_MAX_ITER;
      // Like the reference the iterate of a run that ends at the iteration limit is returned as it stands.  Only after a
      // breakdown (which here also covers non-finite scalars, a guard the reference does not have) the best iterate is preferred
      // when its residual is smaller: x may sit at a half step contaminated by the quantity that broke down
      if (h_bst[B_FLAG] == BF_BREAKDOWN && (h_bst[B_MIN_RN] < h_bst[B_RN] || !(h_bst[B_RN] == h_bst[B_RN]))) {
         TRY(pips_hip_vec_copy(n, w_best, x_, stream));
         h_bst[B_RN] = h_bst[B_MIN_RN];
      }
      last_outer_steps = iters;
      finish();
      return PIPS_OK;
   }
   // OUTER_SOLVE 1 (solveCompressedIterRefin, LinearSystem.C:877-966): x += M^-1 (b - K x) while the residual decreases
   int iter_refine(const double* b_, double* x_) {
      const long long n = nxyz;
      RedPack pk;
      double o[2];
      pk.n_terms = 1;
      pk.t[0] = term(R_DOT, n, b_, b_, nullptr, nullptr, wXYZ);
      TRY(reduce(pk, o));
      const double bn = std::sqrt(o[0]);
      const double target = std::max(bn * outer_tol, 1e-15);
      double best_rn = INFINITY, rn = 0.0;
      TRY(pips_hip_vec_set(n, 0.0, x_, stream));
      TRY(pips_hip_vec_copy(n, b_, w_r, stream));
      last_outer_steps = 0;
      for (int k = 0; k <= outer_max; ++k) {
         TRY(precond(w_r));
         TRY(pips_hip_vec_axpy(n, 1.0, w_r, x_, stream));
         TRY(kresidual(b_, x_, w_r));
         pk.t[0] = term(R_DOT, n, w_r, w_r, nullptr, nullptr, wXYZ);
         TRY(reduce(pk, o));
         rn = std::sqrt(o[0]);
         if (!(rn < best_rn)) { TRY(pips_hip_vec_copy(n, w_best, x_, stream)); break; }
         best_rn = rn;
         last_outer_res = bn > 0 ? rn / bn : rn;
         last_outer_abs = rn;
         TRY(pips_hip_vec_copy(n, x_, w_best, stream));
         if (rn <= target) break;
         ++last_outer_steps;
      }
      last_outer_flag = best_rn <= target ? BF_CONVERGED : BF_MAX_ITER;
      return PIPS_OK;
   }

   // LinearSystem::solve + step.negate() for the current residual set (zero_lin: linear residuals cleared)
   int solve(bool zero_lin, Vars& out) {
      hipLaunchKernelGGL(k_rhs_reduce, dim3(egrid(std::max(std::max(nx, mz), my))), dim3(256), 0, stream, lay, zero_lin ? 1 : 0, rQ, rAC, rz, rG, rL,
                         it.G, it.L, M, dyz, rhs, rs);
      // A pivot whose value is rounding noise can keep the right sign and pass the inertia test; the factors are then useless as
      // a preconditioner and the outer solve does not reach its tolerance.  The reference treats a failed outer solve as
      // numerical trouble of the factorisation; here the system is factorised again with (more) regularisation - which also
      // serves the later solves of the iteration - and the outer solve repeated, at most twice.
      for (int retry = 0;; ++retry) {
         if (outer_mode == 2) TRY(bicgstab(rhs, sol));
         else TRY(iter_refine(rhs, sol));
         const bool reached = last_outer_res <= std::max(1e3 * outer_tol, 1e-7) || last_outer_abs <= 1e-12;
         if (regularize && !reg_operator && (last_outer_steps > 12 || !(last_outer_res <= std::max(10.0 * outer_tol, 1e-9) || last_outer_abs <= 1e-12))) {
            // long or inexact on the original system: same factors, regularised system (see reg_operator)
            if (verbose_run) printf("   outer solve on the original system: %d iterations, rel.res %.1e - repeating it on the regularised system\n",
                                    last_outer_steps, last_outer_res);
            reg_operator = true;
            ++n_reg_operator;
            continue;
         }
         if (!regularize || retry >= 5 || reached || last_reg >= 0.5 * reg_max) break;   // more regularisation than reg_max only ruins the preconditioner
         if (verbose_run) printf("   outer solve stopped at rel.res %.1e: factorising again with regularisation\n", last_outer_res);
         ++n_refactor_outer;
         TRY(factorize(last_reg > 0.0 ? std::min(last_reg * 100.0, reg_max) : 1e-8));
      }
      hipLaunchKernelGGL(k_recover, dim3(egrid(std::max(std::max(nx, mz), my))), dim3(256), 0, stream, lay, zero_lin ? 1 : 0, sol, rs, rG, rL, it.G, it.L,
                         M, dyz, out.P, out.D);
      HIP_TRYH(hipGetLastError());
      return PIPS_OK;
   }

   // ---- step lengths ------------------------------------------------------------------------------------------------------------
   // stepbound_pd (Variables.C:228-263) and mustep_pd of the resulting point in two fused passes
   int step_and_mu(const Vars& d, double* ap, double* ad, double* mu_aff) {
      RedPack pk;
      double o[2];
      pk.n_terms = 2;
      pk.t[0] = term(R_STEPBOUND, ncp, it.G, d.G);
      pk.t[1] = term(R_STEPBOUND, ncp, it.L, d.L);
      TRY(reduce(pk, o));
      *ap = std::min(1.0, o[0]);
      *ad = std::min(1.0, o[1]);
      pk.n_terms = 1;
      pk.t[0] = term(R_DOT_SHIFTED, ncp, it.G, it.L, d.G, d.L, wG, *ap, *ad);
      TRY(reduce(pk, o));
      *mu_aff = n_pairs > 0 ? o[0] / n_pairs : 0.0;
      return PIPS_OK;
   }
   // 11-point search for the corrector weight in [alpha_p alpha_d, 1] that allows the longest steps
   // (calculate_alpha_pd_weight_candidate, InteriorPointMethod.cpp:486-523), one fused device pass over the pair vectors
   int weight_search(double apt, double adt, double* ape, double* ade, double* wp, double* wd) {
      constexpr int NW = 11;
      const double wmin = apt * adt;
      double bounds[2 * NW];
      TRY(pips_hip_vec_weighted_stepbounds(ncp, it.G, st.G, co.G, it.L, st.L, co.L, wmin, NW, bounds, stream));
      ++n_host_syncs;
      if (n_ranks > 1) {
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
   int gfind_blocking(const double* a, const double* da, const double* b2, const double* db, double* out5) {
      TRY(pips_hip_vec_find_blocking(ncp, a, da, b2, db, out5, stream));
      n_host_syncs += 2;
      if (n_ranks == 1) return PIPS_OK;
      std::vector<double> slots(5 * (size_t)n_ranks, 0.0);
      const bool none = !(out5[0] < INFINITY);
      for (int q = 0; q < 5; ++q) slots[5 * rank + q] = (q == 0 && none) ? -1.0 : out5[q];
      TRY(reduce_host(slots.data(), 5 * n_ranks));
      int bestr = -1;
      for (int r = 0; r < n_ranks; ++r)
         if (slots[5 * r] >= 0.0 && (bestr < 0 || slots[5 * r] < slots[5 * bestr])) bestr = r;
      if (bestr < 0) { out5[0] = INFINITY; out5[1] = out5[2] = out5[3] = out5[4] = 0.0; }
      else for (int q = 0; q < 5; ++q) out5[q] = slots[5 * bestr + q];
      return PIPS_OK;
   }
   // Mehrotra's step length heuristic (PrimalDualInteriorPointMethod::mehrotra_step_length, InteriorPointMethod.cpp:745-812)
   int mehrotra_step_length(double* ap, double* ad) {
      const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f), steplength_factor = 0.99999999;
      double pb[5], db[5];
      TRY(gfind_blocking(it.G, st.G, it.L, st.L, pb));
      TRY(gfind_blocking(it.L, st.L, it.G, st.G, db));
      const double amax_p = std::min(1.0, pb[0]), amax_d = std::min(1.0, db[0]);
      RedPack pk;
      double o[1];
      pk.n_terms = 1;
      pk.t[0] = term(R_DOT_SHIFTED, ncp, it.G, it.L, st.G, st.L, wG, amax_p, amax_d);
      TRY(reduce(pk, o));
      const double mufull = o[0] / n_pairs / gamma_a;
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
   void compl_rhs(int mode, double alpha, double ap, double ad, double rmin, double rmax) {
      hipLaunchKernelGGL(k_compl_rhs, dim3(egrid(ncp)), dim3(256), 0, stream, ncp, mode, it.G, it.L, st.G, st.L, M, alpha, ap, ad, rmin, rmax, rL);
   }
   // Gondzio's multiple centrality correctors (gondzio_correction_loop, InteriorPointMethod.cpp:236-358, primal-dual variant)
   int gondzio_loop(double sigma, double mu_now, double* ap, double* ad) {
      const double beta_min = 0.1, beta_max = 10.0, step_factor0 = 0.3, step_factor1 = 1.5, accept = 0.01;
      const double rmin = sigma * mu_now * beta_min, rmax = sigma * mu_now * beta_max;
      int ng = 0;
      while (ng < max_gondzio && (*ap < 1.0 || *ad < 1.0)) {
         const double apt = std::min(1.0, step_factor1 * *ap + step_factor0), adt = std::min(1.0, step_factor1 * *ad + step_factor0);
         compl_rhs(2, 0.0, apt, adt, rmin, rmax);
         TRY(solve(true, co));
         double ape, ade, wp, wd;
         TRY(weight_search(apt, adt, &ape, &ade, &wp, &wd));
         const bool both_one = ape >= 1.0 && ade >= 1.0;
         const bool p_better = ape >= (1.0 + accept) * *ap, d_better = ade >= (1.0 + accept) * *ad;
         if (!both_one && !p_better && !d_better) break;
         if (both_one || p_better) { TRY(pips_hip_vec_axpy(NP, wp, co.P, st.P, stream)); *ap = ape; }
         if (both_one || d_better) { TRY(pips_hip_vec_axpy(ND, wd, co.D, st.D, stream)); *ad = ade; }
         ++ng;
         ++n_gondzio;
         if (both_one) break;
      }
      return PIPS_OK;
   }

   int run(int max_iter, double mutol, double artol, int verbose, double* result) {
      HIP_TRYH(hipSetDevice(device));
      verbose_run = verbose = rank == 0 ? verbose : 0;
      n_gondzio = n_precond = n_bicg_iter = n_host_syncs = 0;
      n_regularised = n_factorize = n_refactor_outer = 0;
      // ---- start point: push_to_interior(sqrt(dnorm)), one affine solve, full step, shift (PIPSIPMppSolver.cpp:36-42, Solver.cpp:19-31)
      const double s0 = std::sqrt(dnorm);
      TRY(pips_hip_vec_set(NP + ND, 0.0, it.base, stream));
      hipLaunchKernelGGL(k_masked_const, dim3(egrid(ncp)), dim3(256), 0, stream, ncp, 0, s0, M, it.G);
      hipLaunchKernelGGL(k_masked_const, dim3(egrid(ncp)), dim3(256), 0, stream, ncp, 0, s0, M, it.L);
      double rnorm, pobj, dobj, m;
      TRY(residuals(&rnorm, &pobj, &dobj, &m));
      compl_rhs(0, 0, 0, 0, 0, 0);
      reg_operator = reg_operator_always;
      n_reg_operator = 0;
      TRY(factorize());
      TRY(solve(false, st));
      TRY(pips_hip_vec_axpy(NP + ND, 1.0, st.base, it.base, stream));
      {
         RedPack pk;
         double o[2];
         pk.n_terms = 2;
         pk.t[0] = term(R_MIN_MASKED, ncp, it.G, nullptr, M);
         pk.t[1] = term(R_MIN_MASKED, ncp, it.L, nullptr, M);
         TRY(reduce(pk, o));
         const double viol = std::max(0.0, std::max(-o[0], -o[1]));
         const double shift = 1e3 + 2.0 * viol;
         hipLaunchKernelGGL(k_masked_const, dim3(egrid(ncp)), dim3(256), 0, stream, ncp, 1, shift, M, it.G);
         hipLaunchKernelGGL(k_masked_const, dim3(egrid(ncp)), dim3(256), 0, stream, ncp, 1, shift, M, it.L);
      }
      int iter = 0, status = 1;  // 1 = max iterations
      trace.clear();
      // Numerical-trouble fallback: the iterate with the best merit max(mu / mutol, ||r|| / (artol dnorm)) is kept and returned
      // with status 3 when the iteration breaks down (NaN, residual blow-up, two stalled steps); the reference answers such
      // situations with its "numerical troubles" logic (InteriorPointMethod.cpp:264-274, PIPSIPMppSolver.cpp:163-185).
      double best_merit = INFINITY, best_rnorm = INFINITY, phi_min = INFINITY;
      int n_stall = 0, n_rstall = 0;
      double prev_rnorm = INFINITY;
      auto merit = [&](double mm, double rr) { return std::max(mm / mutol, rr / (artol * dnorm)); };
      for (; iter < max_iter; ++iter) {
         TRY(residuals(&rnorm, &pobj, &dobj, &m));
         const bool is_nan = !(m == m) || !(rnorm == rnorm) || !(pobj == pobj);
         // a step that throws the residual up by four orders of magnitude counts as a breakdown only late in the run (best iterate
         // within 1e3 of both tolerances): early on the outer solve's tolerance is relative to a right-hand side dominated by the
         // complementarity terms, and an absolute error of 1e-9 |rhs| in the linear rows is harmless and transient
         const bool blown = !is_nan && best_merit < 1e3 && rnorm > 1e4 * std::max(best_rnorm, artol * dnorm);
         n_rstall = (!is_nan && m <= 1e-3 * mutol && rnorm > artol * dnorm && rnorm >= 0.99 * prev_rnorm) ? n_rstall + 1 : 0;
         prev_rnorm = rnorm;
         if ((is_nan || blown || n_stall >= 2 || n_rstall >= 3) && best_merit < INFINITY) {
            if (verbose)
               printf("ipm it %3d  numerical troubles (%s: mu %.3e ||r||inf %.3e), falling back to the best iterate\n", iter,
                      is_nan ? "nan" : (blown ? "residual blow-up" : (n_rstall >= 3 ? "residual stagnates, mu far below its tolerance" : "stalled")), m, rnorm);
            TRY(pips_hip_vec_copy(NP + ND, best.base, it.base, stream));
            TRY(residuals(&rnorm, &pobj, &dobj, &m));
            trace.insert(trace.end(), {m, rnorm, pobj, dobj, 0.0, 0.0, 0.0});
            status = (m <= mutol && rnorm <= artol * dnorm) ? 0 : 3;
            break;
         }
         if (!is_nan && merit(m, rnorm) < best_merit) {
            best_merit = merit(m, rnorm); best_rnorm = rnorm;
            TRY(pips_hip_vec_copy(NP + ND, it.base, best.base, stream));
         }
         trace.insert(trace.end(), {m, rnorm, pobj, dobj, 0.0, 0.0, 0.0});
         if (verbose) {
            printf("ipm it %3d  mu %.3e  ||r||inf %.3e  pobj %.10e  dobj %.10e  (last solve: %d outer its, rel.res %.1e)\n", iter, m, rnorm, pobj,
                   dobj, last_outer_steps, last_outer_res);
            if (verbose > 1) printf("            |rQ| %.2e  |rA,rC| %.2e  |rz| %.2e  |rt,ru,rv,rw| %.2e\n", last_rparts[0], last_rparts[1], last_rparts[2], last_rparts[3]);
            fflush(stdout);
         }
         if (is_nan) { status = 2; break; }
         if (m <= mutol && rnorm <= artol * dnorm) { status = 0; break; }   // PIPSIPMppSolver.cpp:143-149
         {  // "probably infeasible" (PIPSIPMppSolver.cpp:128-170)
            const double phi = (rnorm + std::fabs(pobj - dobj)) / dnorm;
            phi_min = iter == 0 ? phi : std::min(phi_min, phi);
            if (iter >= 10 && phi >= 1e-8 && phi >= 1e4 * phi_min) { status = 4; break; }
         }
         outer_tol = iter <= 3 ? 1e-8 : (iter <= 7 ? 1e-9 : 1e-10);   // InteriorPointMethod.cpp:655-669
         // ---- predictor (affine scaling): complementarity residual = products of the pairs
         compl_rhs(0, 0, 0, 0, 0, 0);
         reg_operator = reg_operator_always;   // every iteration starts on the original system
         TRY(factorize());
         TRY(solve(false, st));
         double ap, ad, maff;
         TRY(step_and_mu(st, &ap, &ad, &maff));
         const double sigma = std::pow(maff / m, 3.0);
         // ---- corrector: linear residuals cleared, r = dG_aff dL_aff - sigma mu  (set_complementarity_residual(step, -sigma mu))
         compl_rhs(1, -sigma * m, 0, 0, 0, 0);
         TRY(solve(true, co));
         double wp, wd;
         TRY(weight_search(ap, ad, &ap, &ad, &wp, &wd));
         TRY(pips_hip_vec_axpy(NP, wp, co.P, st.P, stream));
         TRY(pips_hip_vec_axpy(ND, wd, co.D, st.D, stream));
         TRY(gondzio_loop(sigma, m, &ap, &ad));
         TRY(mehrotra_step_length(&ap, &ad));
         n_stall = (ap < 1e-10 && ad < 1e-10) ? n_stall + 1 : 0;
         { double* row = trace.data() + trace.size() - 7; row[4] = sigma; row[5] = ap; row[6] = ad; }
         TRY(pips_hip_vec_axpy(NP, ap, st.P, it.P, stream));
         TRY(pips_hip_vec_axpy(ND, ad, st.D, it.D, stream));
      }
      if (status == 1) TRY(residuals(&rnorm, &pobj, &dobj, &m));   // the numbers returned describe the iterate returned
      last[0] = pobj; last[1] = iter; last[2] = m; last[3] = rnorm; last[4] = status; last[5] = dobj; last[6] = dnorm;
      if (result)
         for (int i = 0; i < 7; ++i) result[i] = last[i];
      return PIPS_OK;
   }
};

}  // namespace pips

using namespace pips;

// ------------------------------------------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

struct View { int rows = 0, cols = 0; const int* rp = nullptr; const int* ci = nullptr; const double* v = nullptr; };
View view(const pips_csr_view& m) { return View{m.rows, m.cols, m.rowptr, m.colidx, m.val}; }
bool present(const View& m) { return m.rp != nullptr && m.rows > 0; }

int build(Ipm* p, int n_blocks, const pips_ipm_block* blocks, int myl, int mzl, const double* bL, const double* dlow, const double* dupp,
          const double* idlow, const double* idupp, double dual_reg, int device) {
   const pips_ipm_block& root = blocks[0];
   const int N = n_blocks - 1;
   p->N = N; p->n0 = root.n; p->my0 = root.my; p->mz0 = root.mz; p->myl = myl; p->mzl = mzl; p->dual_reg = dual_reg;
   const int n0 = p->n0, my0 = p->my0, mz0 = p->mz0;
   p->ry = my0 + myl; p->rzr = mz0 + mzl;
   std::vector<int> xoff(N + 2, 0), yoff(N + 2, 0), zoff(N + 2, 0);
   std::vector<long long> koff(N + 2, 0);
   xoff[1] = n0; yoff[1] = p->ry; zoff[1] = p->rzr;
   for (int i = 1; i <= N; ++i) {
      const pips_ipm_block& b = blocks[i];
      if (b.n < 0 || b.my < 0 || b.mz < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_general: negative block dimension");
      xoff[i + 1] = xoff[i] + b.n; yoff[i + 1] = yoff[i] + b.my; zoff[i + 1] = zoff[i] + b.mz;
      koff[i + 1] = koff[i] + b.n + b.my + b.mz;
   }
   p->nx = xoff[N + 1]; p->my = yoff[N + 1]; p->mz = zoff[N + 1]; p->nleaf = koff[N + 1];
   const int nx = p->nx, my = p->my, mz = p->mz;
   p->ncp = 2LL * mz + 2LL * nx; p->nxyz = (long long)nx + my + mz;
   p->NP = nx + mz + p->ncp; p->ND = my + mz + p->ncp;
   p->lay = Lay{nx, my, mz, p->ncp};
   // Root inequality rows C0 x0 - s = ...: the reference eliminates them from the root system (-C0^T Omega^-1 C0 on the x0 block,
   // sLinsysRootAug.C:1276-1294) and leaves the rest to dsytrf's Bunch-Kaufman pivoting.  So does the harness now that the dense root
   // pivots (k_tile_diag_bk; pips_hip_kkt_set_root_inequalities switches it on): an active row has Omega^-1 ~ 1e14, the x0 block is
   // then a huge low-rank matrix plus an O(1) rest and a static pivot order reports the cancelled pivots as perturbed at every
   // regularisation (round 2: 2 % of the seeded general LPs ended with status 3 that way, which is why the rows were then kept in
   // the root system as rows of the linking-inequality kind - diagonal nOmegaInv, no cancellation).  That formulation remains for the
   // sparse root, whose factorisation is the leaf engine's (static order), and under PIPS_IPM_ROOT_INEQ_ELIMINATE=0.
   const bool sparse_root_req = getenv("PIPS_IPM_SPARSE_ROOT") && atoi(getenv("PIPS_IPM_SPARSE_ROOT")) != 0;
   const bool eliminate_z0 = getenv("PIPS_IPM_ROOT_INEQ_ELIMINATE") ? atoi(getenv("PIPS_IPM_ROOT_INEQ_ELIMINATE")) != 0 : !sparse_root_req;
   p->e_mz0 = eliminate_z0 ? mz0 : 0;
   p->e_mzl = eliminate_z0 ? mzl : mz0 + mzl;
   const int e_mz0 = p->e_mz0, e_mzl = p->e_mzl;
   const int S = n0 + my0 + myl + e_mzl;
   p->S = S;
   int rc = pips_hip_batch_create(&p->batch, N, S, device, nullptr);
   if (rc) return rc;
   // ---- rows of J = [A; C]: y rows [y0 | ylink | blocks], z rows [z0 | zlink | blocks]
   std::vector<std::vector<std::pair<int, double>>> rows((size_t)my + mz);
   double dn = 0.0;
   auto add_rows = [&](const View& m, int row0, int col0) {
      if (!present(m)) return;
      for (int r = 0; r < m.rows; ++r)
         for (int q = m.rp[r]; q < m.rp[r + 1]; ++q) {
            rows[(size_t)row0 + r].push_back({col0 + m.ci[q], m.v[q]});
            dn = std::max(dn, std::fabs(m.v[q]));
         }
   };
   auto norm_only = [&](const View& m) {
      if (!present(m)) return;
      for (int q = m.rp[0]; q < m.rp[m.rows]; ++q) dn = std::max(dn, std::fabs(m.v[q]));
   };
   // root matrices enter the local J on rank 0 only: replicated rows are summed over the ranks
   if (p->rank == 0) {
      add_rows(view(root.A), 0, 0);                 // A0
      add_rows(view(root.BL), my0, 0);              // F0
      add_rows(view(root.C), my, 0);                // C0
      add_rows(view(root.DL), my + mz0, 0);         // G0
   } else { norm_only(view(root.A)); norm_only(view(root.BL)); norm_only(view(root.C)); norm_only(view(root.DL)); }
   std::vector<std::vector<double>> kvals(N);
   for (int i = 1; i <= N; ++i) {
      const pips_ipm_block& b = blocks[i];
      const View A = view(b.A), B = view(b.B), Cm = view(b.C), D = view(b.D), BL = view(b.BL), DL = view(b.DL);
      if (b.my > 0 && !present(B)) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_general: block %d has equality rows but no B matrix", i);
      add_rows(A, yoff[i], 0);
      add_rows(B, yoff[i], xoff[i]);
      add_rows(Cm, my + zoff[i], 0);
      add_rows(D, my + zoff[i], xoff[i]);
      add_rows(BL, my0, xoff[i]);
      add_rows(DL, my + mz0, xoff[i]);
      // K_i pattern / values and border for the engine
      const int nxi = b.n, myi = b.my, mzi = b.mz, nk = nxi + myi + mzi;
      std::vector<int> empty_rp_y(myi + 1, 0), empty_rp_z(mzi + 1, 0);
      const int* Brp = present(B) ? B.rp : empty_rp_y.data();
      const int* Drp = present(D) ? D.rp : empty_rp_z.data();
      std::vector<int> Krp(nk + 1), dpos(nk);
      rc = pips_kkt_leaf_assemble(nxi, myi, mzi, nullptr, nullptr, nullptr, Brp, B.ci, B.v, Drp, D.ci, D.v, Krp.data(), nullptr, nullptr, nullptr);
      if (rc) return rc;
      std::vector<int> Kci(Krp[nk]);
      kvals[i - 1].assign(Krp[nk], 0.0);
      rc = pips_kkt_leaf_assemble(nxi, myi, mzi, nullptr, nullptr, nullptr, Brp, B.ci, B.v, Drp, D.ci, D.v, Krp.data(), Kci.data(), kvals[i - 1].data(), dpos.data());
      if (rc) return rc;
      std::vector<int> Brd(S + 1);
      auto P3 = [](const View& m) { return m; };
      const View a_ = P3(A), c_ = P3(Cm), f_ = P3(BL);
      View g_ = P3(DL);
      std::vector<int> g_rp;
      if (e_mzl != mzl && present(g_)) {   // root inequality rows ride among the linking inequalities: no leaf part
         g_rp.assign((size_t)e_mzl + 1, 0);
         for (int r = 0; r <= mzl; ++r) g_rp[(size_t)(e_mzl - mzl) + r] = g_.rp[r] - g_.rp[0];
         g_.ci += g_.rp[0]; g_.v += g_.rp[0];
         g_.rp = g_rp.data();
         g_.rows = e_mzl;
      }
      rc = pips_border_assemble(nxi, myi, mzi, n0, my0, myl, e_mzl, nullptr, nullptr, nullptr, present(a_) ? a_.rp : nullptr, a_.ci, a_.v,
                                present(c_) ? c_.rp : nullptr, c_.ci, c_.v, present(f_) ? f_.rp : nullptr, f_.ci, f_.v,
                                present(g_) ? g_.rp : nullptr, g_.ci, g_.v, Brd.data(), nullptr, nullptr);
      if (rc) return rc;
      std::vector<int> Bci(Brd[S]);
      std::vector<double> Bv(Brd[S]);
      rc = pips_border_assemble(nxi, myi, mzi, n0, my0, myl, e_mzl, nullptr, nullptr, nullptr, present(a_) ? a_.rp : nullptr, a_.ci, a_.v,
                                present(c_) ? c_.rp : nullptr, c_.ci, c_.v, present(f_) ? f_.rp : nullptr, f_.ci, f_.v,
                                present(g_) ? g_.rp : nullptr, g_.ci, g_.v, Brd.data(), Bci.data(), Bv.data());
      if (rc) return rc;
      rc = pips_hip_batch_set_block(p->batch, i - 1, nk, nxi, Krp.data(), Kci.data(), Brd.data(), Bci.data(), Bv.data());
      if (rc) return rc;
   }
   // ---- vectors in the flat layout
   std::vector<double> hc(nx, 0.0), hb(my, 0.0), hM(p->ncp, 0.0), hBd(p->ncp, 0.0);
   auto put = [&](const double* src, int n, double* dst) { if (src) for (int k = 0; k < n; ++k) dst[k] = src[k]; };
   const long long oT = 0, oU = mz, oV = 2LL * mz, oW = 2LL * mz + nx;
   for (int i = 0; i <= N; ++i) {
      const pips_ipm_block& b = blocks[i];
      const int x0_ = i == 0 ? 0 : xoff[i], y0_ = i == 0 ? 0 : yoff[i], z0_ = i == 0 ? 0 : zoff[i];
      if (b.n > 0 && (!b.c || !b.ixlow || !b.ixupp || !b.xlow || !b.xupp)) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_general: block %d lacks c / bounds", i);
      put(b.c, b.n, hc.data() + x0_);
      put(b.b, b.my, hb.data() + y0_);
      put(b.ixlow, b.n, hM.data() + oV + x0_); put(b.ixupp, b.n, hM.data() + oW + x0_);
      put(b.xlow, b.n, hBd.data() + oV + x0_); put(b.xupp, b.n, hBd.data() + oW + x0_);
      put(b.iclow, b.mz, hM.data() + oT + z0_); put(b.icupp, b.mz, hM.data() + oU + z0_);
      put(b.clow, b.mz, hBd.data() + oT + z0_); put(b.cupp, b.mz, hBd.data() + oU + z0_);
   }
   put(bL, myl, hb.data() + my0);
   put(idlow, mzl, hM.data() + oT + mz0); put(idupp, mzl, hM.data() + oU + mz0);
   put(dlow, mzl, hBd.data() + oT + mz0); put(dupp, mzl, hBd.data() + oU + mz0);
   for (long long k = 0; k < p->ncp; ++k) {
      if (hM[k] != 0.0 && hM[k] != 1.0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_general: bound indicators must be 0 or 1");
      hBd[k] *= hM[k];   // matchesNonZeroPattern (Problem.cpp:69-79)
      dn = std::max(dn, std::fabs(hBd[k]));
   }
   for (double v : hc) dn = std::max(dn, std::fabs(v));
   for (double v : hb) dn = std::max(dn, std::fabs(v));
   p->dnorm = dn > 0 ? dn : 1.0;
   // weights that count replicated root entries once (iAmSpecial, DistributedVector.C:1293-1303)
   const double wroot = p->rank == 0 ? 1.0 : 0.0;
   std::vector<double> hwG(p->ncp, 1.0), hwGs(p->ncp, 1.0), hwXYZ(p->nxyz, 1.0);
   for (int k = 0; k < p->rzr; ++k) hwG[oT + k] = hwG[oU + k] = wroot;
   for (int k = 0; k < n0; ++k) hwG[oV + k] = hwG[oW + k] = wroot;
   for (long long k = 0; k < p->ncp; ++k) hwGs[k] = hwG[k] * ((k >= oU && k < oV) || k >= oW ? -1.0 : 1.0);
   for (int k = 0; k < n0; ++k) hwXYZ[k] = wroot;
   for (int k = 0; k < p->ry; ++k) hwXYZ[(size_t)nx + k] = wroot;
   for (int k = 0; k < p->rzr; ++k) hwXYZ[(size_t)nx + my + k] = wroot;
   double pairs = 0.0;
   for (long long k = 0; k < p->ncp; ++k) pairs += hwG[k] * hM[k];
   // ---- CSR of J and J^T
   std::vector<int> Jrp((size_t)my + mz + 1, 0), Jtrp((size_t)nx + 1, 0);
   for (size_t r = 0; r < rows.size(); ++r) {
      std::sort(rows[r].begin(), rows[r].end());
      Jrp[r + 1] = Jrp[r] + (int)rows[r].size();
      for (auto& e : rows[r]) ++Jtrp[e.first + 1];
   }
   const int nnz = Jrp[rows.size()];
   std::vector<int> Jci(nnz), Jtci(nnz);
   std::vector<double> Jv(nnz), Jtv(nnz);
   for (int j = 0; j < nx; ++j) Jtrp[j + 1] += Jtrp[j];
   {
      std::vector<int> fill(Jtrp.begin(), Jtrp.end() - 1);
      for (size_t r = 0; r < rows.size(); ++r) {
         int q = Jrp[r];
         for (auto& e : rows[r]) {
            Jci[q] = e.first; Jv[q] = e.second; ++q;
            const int t = fill[e.first]++;
            Jtci[t] = (int)r; Jtv[t] = e.second;
         }
      }
   }
   // ---- engine: analyze, values, root system
   const bool sparse_root = getenv("PIPS_IPM_SPARSE_ROOT") && atoi(getenv("PIPS_IPM_SPARSE_ROOT")) != 0;
   if (sparse_root && p->n_ranks > 1)
      PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create: the sparse root needs the border column sets of all blocks on every rank (pips_hip_kkt_create_sparse); "
                              "the harness passes only its own - use the dense root with several ranks");
   if (sparse_root && (rc = pips_hip_batch_set_schur_mode(p->batch, 1))) return rc;
   if ((rc = pips_hip_batch_analyze(p->batch, 16))) return rc;
   for (int i = 0; i < N; ++i)
      if ((rc = pips_hip_batch_set_values(p->batch, i, kvals[i].data()))) return rc;
   if ((rc = pips_hip_batch_set_refinement_backward_error(p->batch, 2, 1e-15))) return rc;   // PARDISO iparm[7]=2 semantics
   const View A0 = view(root.A), F0 = view(root.BL), C0 = view(root.C);
   View G0 = view(root.DL);
   std::vector<int> g0_rp, g0_ci;
   std::vector<double> g0_v;
   if (e_mzl != mzl) {   // G0' = [C0; G0]
      g0_rp.assign(1, 0);
      const View G0in = G0;
      for (const View* m : {&C0, &G0in}) {
         const int nr = m == &C0 ? mz0 : mzl;
         for (int r = 0; r < nr; ++r) {
            if (present(*m))
               for (int q = m->rp[r]; q < m->rp[r + 1]; ++q) { g0_ci.push_back(m->ci[q]); g0_v.push_back(m->v[q]); }
            g0_rp.push_back((int)g0_ci.size());
         }
      }
      g0_ci.push_back(0); g0_v.push_back(0.0);   // never empty arrays
      G0.rows = e_mzl; G0.cols = n0; G0.rp = g0_rp.data(); G0.ci = g0_ci.data(); G0.v = g0_v.data();
   }
   auto rp = [](const View& m) { return present(m) ? m.rp : nullptr; };
   if (sparse_root)
      rc = pips_hip_kkt_create_sparse(&p->kkt, p->batch, n0, my0, myl, e_mzl, rp(A0), A0.ci, A0.v, rp(F0), F0.ci, F0.v, rp(G0), G0.ci, G0.v, 0, nullptr,
                                      nullptr, p->comm, p->rank, p->n_ranks);
   else
      rc = pips_hip_kkt_create(&p->kkt, p->batch, n0, my0, myl, e_mzl, rp(A0), A0.ci, A0.v, rp(F0), F0.ci, F0.v, rp(G0), G0.ci, G0.v, p->comm, p->rank,
                               p->n_ranks);
   if (rc) return rc;
   // factorize() asks for every inertia right after the factorisation (the inertia loop): nothing would run beside a root on its own stream
   if ((rc = pips_hip_kkt_set_root_stream(p->kkt, 0))) return rc;
   HIP_TRYH(hipGetDevice(&p->device));
   if ((rc = p->alloc(&p->d_red, 64 * (long long)p->n_ranks))) return rc;
   if (p->n_ranks > 1) {
      std::vector<double> slots(p->n_ranks + 1, 0.0);
      slots[p->rank] = p->dnorm; slots[p->n_ranks] = pairs;
      if ((rc = p->reduce_host(slots.data(), p->n_ranks + 1))) return rc;
      for (int r = 0; r < p->n_ranks; ++r) p->dnorm = std::max(p->dnorm, slots[r]);
      pairs = slots[p->n_ranks];
   }
   p->n_pairs = pairs;
   // ---- maps: KKT right-hand sides <- [x|y|z], leaf diagonal codes
   {
      std::vector<long long> pack;
      pack.reserve((size_t)S + e_mz0 + p->nleaf);
      for (int k = 0; k < n0; ++k) pack.push_back(k);
      for (int k = 0; k < my0; ++k) pack.push_back((long long)nx + k);
      for (int k = 0; k < e_mz0; ++k) pack.push_back((long long)nx + my + k);
      for (int k = 0; k < myl; ++k) pack.push_back((long long)nx + my0 + k);
      for (int k = 0; k < e_mzl; ++k) pack.push_back((long long)nx + my + e_mz0 + k);   // [z0 | zlink] is contiguous in the harness' z order
      std::vector<long long> code;
      code.reserve(p->nleaf);
      for (int i = 1; i <= N; ++i) {
         for (int k = xoff[i]; k < xoff[i + 1]; ++k) { pack.push_back(k); code.push_back(k); }
         for (int k = yoff[i]; k < yoff[i + 1]; ++k) { pack.push_back((long long)nx + k); code.push_back(-1); }
         for (int k = zoff[i]; k < zoff[i + 1]; ++k) { pack.push_back((long long)nx + my + k); code.push_back(-2 - (long long)k); }
      }
      p->npack = (long long)pack.size();
      if ((rc = p->up(&p->d_pack, pack)) || (rc = p->up(&p->d_code, code))) return rc;
   }
   if (e_mz0 > 0) {
      if ((rc = pips_hip_kkt_set_root_inequalities(p->kkt, e_mz0, C0.rp, C0.ci, C0.v))) return rc;
   }
   {
      std::vector<int> la, lat;
      for (int r = 0; r < my + mz; ++r) if (Jrp[r + 1] - Jrp[r] > CSR_LONG_ROW) la.push_back(r);
      for (int r = 0; r < nx; ++r) if (Jtrp[r + 1] - Jtrp[r] > CSR_LONG_ROW) lat.push_back(r);
      p->nJ_long = (int)la.size(); p->nJt_long = (int)lat.size();
      p->J_nnz = Jrp[my + mz]; p->Jt_nnz = Jtrp[nx];
      la.push_back(0); lat.push_back(0);
      if ((rc = p->up(&p->J_long, la)) || (rc = p->up(&p->Jt_long, lat))) return rc;
      if ((rc = p->alloc(&p->long_scratch, (long long)std::max<size_t>(std::max(la.size(), lat.size()), 1) * LONG_PARTS))) return rc;
   }
   if ((rc = p->up(&p->J_rp, Jrp)) || (rc = p->up(&p->J_ci, Jci)) || (rc = p->up(&p->J_v, Jv)) || (rc = p->up(&p->Jt_rp, Jtrp)) ||
       (rc = p->up(&p->Jt_ci, Jtci)) || (rc = p->up(&p->Jt_v, Jtv)))
      return rc;
   if ((rc = p->up(&p->c, hc)) || (rc = p->up(&p->bA, hb)) || (rc = p->up(&p->M, hM)) || (rc = p->up(&p->Bd, hBd)) || (rc = p->up(&p->wG, hwG)) ||
       (rc = p->up(&p->wGs, hwGs)) || (rc = p->up(&p->wXYZ, hwXYZ)))
      return rc;
   p->wX = p->wXYZ;
   p->wY = p->wXYZ + nx;
   // ---- state
   double* bases[4];
   for (auto& b : bases)
      if ((rc = p->alloc(&b, p->NP + p->ND))) return rc;
   p->it.bind(bases[0], nx, my, mz, p->ncp); p->st.bind(bases[1], nx, my, mz, p->ncp);
   p->co.bind(bases[2], nx, my, mz, p->ncp); p->best.bind(bases[3], nx, my, mz, p->ncp);
   if ((rc = p->alloc(&p->rQ, nx)) || (rc = p->alloc(&p->rAC, (long long)my + mz)) || (rc = p->alloc(&p->rz, mz)) || (rc = p->alloc(&p->rG, p->ncp)) ||
       (rc = p->alloc(&p->rL, p->ncp)) || (rc = p->alloc(&p->rs, mz)) || (rc = p->alloc(&p->dd, nx)) || (rc = p->alloc(&p->ddp, nx)) || (rc = p->alloc(&p->dop_r, nx)) || (rc = p->alloc(&p->dyz_r, (long long)my + mz)) ||
       (rc = p->alloc(&p->dyz, (long long)my + mz)) || (rc = p->alloc(&p->leaf_diag, p->nleaf)))
      return rc;
   double** zs[] = {&p->rhs, &p->sol, &p->w_r, &p->w_r0, &p->w_best, &p->w_v, &p->w_t, &p->w_p, &p->w_dx, &p->w_tmp};
   for (auto d : zs)
      if ((rc = p->alloc(d, p->nxyz))) return rc;
   if ((rc = p->alloc(&p->b0, (long long)S + e_mz0 + p->nleaf))) return rc;   // [b0 | leaves] in one array (one gather / scatter)
   p->bl = p->b0 + S + e_mz0;
   if ((rc = p->alloc(&p->d_partial, (long long)RED_MAX * RED_GRID)) || (rc = p->alloc(&p->d_out, 2 * RED_MAX)) || (rc = p->alloc(&p->d_bst, B_SLOTS)))
      return rc;
   HIP_TRYH(hipHostMalloc((void**)&p->h_out, 2 * RED_MAX * sizeof(double), hipHostMallocDefault));
   HIP_TRYH(hipHostMalloc((void**)&p->h_bst, B_SLOTS * sizeof(double), hipHostMallocDefault));
   HIP_TRYH(hipMalloc((void**)&p->d_pred, P_COUNT * sizeof(int)));
   p->owned.push_back(p->d_pred);
   HIP_TRYH(hipMemset(p->d_pred, 0, P_COUNT * sizeof(int)));
   if (e_mz0 > 0 && ((rc = p->alloc(&p->zd0, e_mz0)) || (rc = pips_hip_kkt_set_zdiag0_dev(p->kkt, p->zd0)))) return rc;   // nOmegaInv of the root rows, regularised
   return PIPS_OK;
}

}  // namespace

extern "C" {

int pips_ipm_create_general(void** handle, int n_blocks, const pips_ipm_block* blocks, int myl, int mzl, const double* bL, const double* dlow,
                            const double* dupp, const double* idlow, const double* idupp, double dual_reg, int device, void* comm, int rank,
                            int n_ranks) {
   if (!handle || n_blocks < 2 || !blocks || myl < 0 || mzl < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !comm) ||
       (myl > 0 && !bL) || (mzl > 0 && (!dlow || !dupp || !idlow || !idupp)))
      PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create_general: bad arguments");
   auto p = std::make_unique<Ipm>();
   p->comm = comm; p->rank = rank; p->n_ranks = n_ranks;
   int rc = build(p.get(), n_blocks, blocks, myl, mzl, bL, dlow, dupp, idlow, idupp, dual_reg, device);
   if (rc) return rc;
   *handle = p.release();
   return PIPS_OK;
}

int pips_ipm_create(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                    const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                    const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                    const double* F0_val, const double* c, const double* b, double dual_reg, int device) {
   return pips_ipm_create_rank(handle, N, n0, myl, n_i, my_i, W_rowptr, W_colidx, W_val, T_rowptr, T_colidx, T_val, F_rowptr, F_colidx, F_val,
                               F0_rowptr, F0_colidx, F0_val, c, b, dual_reg, device, nullptr, 0, 1);
}

// the generator's class  min c^T x, A x = b, x >= 0  expressed in the general layout: ixlow = 1, xlow = 0, no other bound, no
// inequality rows
int pips_ipm_create_rank(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                         const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                         const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                         const double* F0_val, const double* c, const double* b, double dual_reg, int device, void* comm, int rank,
                         int n_ranks) {
   if (!handle || N <= 0 || n0 < 0 || myl < 0 || !n_i || !my_i || !c || !b) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_create: bad arguments");
   std::vector<pips_ipm_block> blk(N + 1);
   std::memset(blk.data(), 0, blk.size() * sizeof(pips_ipm_block));
   int nmax = n0;
   for (int i = 0; i < N; ++i) nmax = std::max(nmax, n_i[i]);
   std::vector<double> ones(nmax, 1.0), zeros(nmax, 0.0);
   long long wp = 0, tp = 0, fp = 0, wr = 0, tr = 0, fr = 0, xo = n0, yo = myl;
   blk[0].n = n0;
   blk[0].c = c; blk[0].xlow = zeros.data(); blk[0].xupp = zeros.data(); blk[0].ixlow = ones.data(); blk[0].ixupp = zeros.data();
   if (F0_rowptr) blk[0].BL = pips_csr_view{myl, n0, F0_rowptr, F0_colidx, F0_val};
   // block-local row pointers start at 0 but index into the concatenated column / value arrays: shift per block
   std::vector<std::vector<int>> keep;
   for (int i = 0; i < N; ++i) {
      pips_ipm_block& bk = blk[i + 1];
      bk.n = n_i[i]; bk.my = my_i[i];
      bk.c = c + xo; bk.b = b + yo;
      bk.xlow = zeros.data(); bk.xupp = zeros.data(); bk.ixlow = ones.data(); bk.ixupp = zeros.data();
      bk.B = pips_csr_view{my_i[i], n_i[i], W_rowptr + wr, W_colidx + wp, W_val + wp};
      if (T_rowptr) bk.A = pips_csr_view{my_i[i], n0, T_rowptr + tr, T_colidx + tp, T_val + tp};
      if (F_rowptr) bk.BL = pips_csr_view{myl, n_i[i], F_rowptr + fr, F_colidx + fp, F_val + fp};
      wp += W_rowptr[wr + my_i[i]] - W_rowptr[wr]; wr += my_i[i] + 1;
      if (T_rowptr) { tp += T_rowptr[tr + my_i[i]] - T_rowptr[tr]; tr += my_i[i] + 1; }
      if (F_rowptr) { fp += F_rowptr[fr + myl] - F_rowptr[fr]; fr += myl + 1; }
      xo += n_i[i]; yo += my_i[i];
   }
   return pips_ipm_create_general(handle, N + 1, blk.data(), myl, 0, b, nullptr, nullptr, nullptr, nullptr, dual_reg, device, comm, rank, n_ranks);
}

int pips_ipm_solve(void* handle, int max_iter, double mutol, double artol, int verbose, double* result7) {
   Ipm* p = (Ipm*)handle;
   if (!p) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   return p->run(max_iter, mutol, artol, verbose, result7);
}

int pips_ipm_get_solution(void* handle, double* x_host, double* y_host) {
   Ipm* p = (Ipm*)handle;
   if (!p) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (x_host) HIP_TRYH(hipMemcpy(x_host, p->it.x, (size_t)p->nx * sizeof(double), hipMemcpyDeviceToHost));
   if (y_host) HIP_TRYH(hipMemcpy(y_host, p->it.y, (size_t)p->my * sizeof(double), hipMemcpyDeviceToHost));
   return PIPS_OK;
}

int pips_ipm_get_dims(void* handle, long long* dims4) {
   Ipm* p = (Ipm*)handle;
   if (!p || !dims4) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_get_dims: bad arguments");
   dims4[0] = p->nx; dims4[1] = p->my; dims4[2] = p->mz; dims4[3] = (long long)p->n_pairs;
   return PIPS_OK;
}

int pips_ipm_get_iterate(void* handle, double* x, double* s, double* y, double* z, double* t, double* u, double* v, double* w, double* lambda,
                         double* pi, double* gamma, double* phi) {
   Ipm* p = (Ipm*)handle;
   if (!p) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   const int nx = p->nx, my = p->my, mz = p->mz;
   auto get = [&](double* dst, const double* src, long long n) -> int {
      if (dst && n > 0) HIP_TRYH(hipMemcpy(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
      return PIPS_OK;
   };
   TRY(get(x, p->it.x, nx)); TRY(get(s, p->it.s, mz)); TRY(get(y, p->it.y, my)); TRY(get(z, p->it.z, mz));
   TRY(get(t, p->it.G, mz)); TRY(get(u, p->it.G + mz, mz)); TRY(get(v, p->it.G + 2 * mz, nx)); TRY(get(w, p->it.G + 2 * mz + nx, nx));
   TRY(get(lambda, p->it.L, mz)); TRY(get(pi, p->it.L + mz, mz)); TRY(get(gamma, p->it.L + 2 * mz, nx)); TRY(get(phi, p->it.L + 2 * mz + nx, nx));
   return PIPS_OK;
}

int pips_ipm_set_gondzio(void* handle, int max_correctors) {
   Ipm* p = (Ipm*)handle;
   if (!p || max_correctors < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_gondzio: bad arguments");
   p->max_gondzio = max_correctors;
   return PIPS_OK;
}

// legacy entry for the x >= 0 class: entries with mask 0 lose their lower bound (ixlow = 0: free variable)
int pips_ipm_set_free_variables(void* handle, const double* bounded_mask_host) {
   Ipm* p = (Ipm*)handle;
   if (!p || !bounded_mask_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: bad arguments");
   HIP_TRYH(hipSetDevice(p->device));
   double local_pairs = 0.0;
   for (int j = 0; j < p->nx; ++j) {
      if (bounded_mask_host[j] != 0.0 && bounded_mask_host[j] != 1.0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: mask entries must be 0 or 1");
      if (j >= p->n0 || p->rank == 0) local_pairs += bounded_mask_host[j];
   }
   HIP_TRYH(hipMemcpy(p->M + 2LL * p->mz, bounded_mask_host, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice));
   if (p->n_ranks > 1) TRY(p->reduce_host(&local_pairs, 1));
   // pairs of the other three kinds are unchanged (the legacy class has none)
   p->n_pairs = local_pairs;
   if (p->n_pairs <= 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_free_variables: no bounded variable left");
   return PIPS_OK;
}

int pips_ipm_set_option(void* handle, const char* name, double value) {
   Ipm* p = (Ipm*)handle;
   if (!p || !name) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_set_option: bad arguments");
   const std::string key(name);
   // identifiers of the reference's option tables (Options.C:18-73, PIPSIPMppOptions.C:170-264,303-310)
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
   else if (key == "OUTER_BICG_MAX_NORMR_DIVERGENCES") {
      if (value < 0) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_BICG_MAX_NORMR_DIVERGENCES must be >= 0");
      p->bicg_max_div = (int)value;
   }
   else if (key == "OUTER_BICG_EPSILON") {
      if (!(value > 0.0)) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_BICG_EPSILON must be > 0");
      p->bicg_eps = value;
   }
   else if (key == "OUTER_BICG_TEST_PRECOND") {
      if (value != 0.0 && value != 1.0 && value != 2.0) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_BICG_TEST_PRECOND: 0 (off), 1 (zero preconditioner) or 2 (ramp-scaled preconditioner)");
      p->test_precond = (int)value;
   }
   else if (key == "OUTER_BICG_MAX_STAGNATIONS") {
      if (value < 1) PIPS_FAIL(PIPS_ERR_ARG, "OUTER_BICG_MAX_STAGNATIONS must be >= 1");
      p->bicg_max_stag = (int)value;
   }
   else if (key == "REGULARIZATION") p->regularize = value != 0.0;
   else if (key == "FREE_VARIABLE_PROXIMAL_TERM") {   // not a reference identifier: diagonal of the free variables in the preconditioner
      if (!(value > 0.0)) PIPS_FAIL(PIPS_ERR_ARG, "FREE_VARIABLE_PROXIMAL_TERM must be > 0");
      p->free_reg = value;
   }
   else if (key == "OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM") p->reg_operator_always = p->reg_operator = value == 0.0;   // reference identifier (PIPSIPMppOptions.C:293: false)
   else if (key == "FREE_VARIABLE_PROXIMAL_IN_OPERATOR") p->free_in_operator = value != 0.0;
   else if (key == "FREE_VARIABLE_PROXIMAL_FOLLOWS_MU") p->free_reg_follows_mu = value != 0.0;
   else if (key == "FREE_VARIABLE_PROXIMAL_MIN") p->free_reg_min = value;
   else if (key == "REGULARIZATION_MAX") p->reg_max = value;
   else if (key == "REGULARIZATION_EAGER_MAX") p->reg_eager_max = value;
   else if (key == "INERTIA_LOOP") p->eager_inertia_loop = value != 0.0;
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

int pips_ipm_get_stats2(void* handle, long long* stats2) {
   Ipm* p = (Ipm*)handle;
   if (!p || !stats2) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_get_stats2: bad arguments");
   stats2[0] = p->n_bicg_iter; stats2[1] = p->n_host_syncs;
   return PIPS_OK;
}

/* ---- direct entries to the two f-rows, for parity tests: block-angular SpMV and the outer BiCGStab on a given right-hand side ---- */
int pips_ipm_mult(void* handle, int transposed, const double* in_host, double* out_host) {
   Ipm* p = (Ipm*)handle;
   if (!p || !in_host || !out_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_mult: bad arguments");
   HIP_TRYH(hipSetDevice(p->device));
   const long long nin = transposed ? (long long)p->my + p->mz : p->nx, nout = transposed ? p->nx : (long long)p->my + p->mz;
   // scratch: w_dx / w_t are nxyz long; zero epilogue operands make SP_KX / SP_KYZ a plain product
   HIP_TRYH(hipMemcpy(p->w_dx, in_host, (size_t)nin * sizeof(double), hipMemcpyHostToDevice));
   TRY(pips_hip_vec_set(p->nxyz, 0.0, p->w_p, p->stream));
   if (transposed) TRY((p->spmv<SP_KX>(true, p->w_dx, p->w_t, p->w_p, p->w_p, nullptr, nullptr)));
   else TRY((p->spmv<SP_KYZ>(false, p->w_dx, p->w_t, p->w_p, p->w_p, nullptr, nullptr)));
   HIP_TRYH(hipStreamSynchronize(p->stream));
   HIP_TRYH(hipMemcpy(out_host, p->w_t, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost));
   return PIPS_OK;
}

/* solves K sol = rhs for K = [dd J^T; J diag(0, nOmegaInv)] with the diagonals of the given pair vectors: G = [t|u|v|w],
 * L = [lambda|pi|gamma|phi] (host, ncp each; entries outside the masks ignored).  info6 = [flag, iterations, ||r||, ||b||,
 * preconditioner applications, host synchronisations] */
int pips_ipm_outer_solve(void* handle, const double* G_host, const double* L_host, const double* rhs_host, double tol, double* sol_host,
                         double* info6) {
   Ipm* p = (Ipm*)handle;
   if (!p || !G_host || !L_host || !rhs_host || !sol_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_ipm_outer_solve: bad arguments");
   HIP_TRYH(hipSetDevice(p->device));
   HIP_TRYH(hipMemcpy(p->it.G, G_host, (size_t)p->ncp * sizeof(double), hipMemcpyHostToDevice));
   HIP_TRYH(hipMemcpy(p->it.L, L_host, (size_t)p->ncp * sizeof(double), hipMemcpyHostToDevice));
   HIP_TRYH(hipMemcpy(p->rhs, rhs_host, (size_t)p->nxyz * sizeof(double), hipMemcpyHostToDevice));
   p->n_precond = p->n_host_syncs = p->n_bicg_iter = 0;
   TRY(p->factorize());
   p->outer_tol = tol;
   if (p->outer_mode == 2) TRY(p->bicgstab(p->rhs, p->sol)); else TRY(p->iter_refine(p->rhs, p->sol));
   HIP_TRYH(hipStreamSynchronize(p->stream));
   HIP_TRYH(hipMemcpy(sol_host, p->sol, (size_t)p->nxyz * sizeof(double), hipMemcpyDeviceToHost));
   if (info6) {
      info6[0] = p->last_outer_flag; info6[1] = p->last_outer_steps; info6[2] = p->last_outer_abs;
      info6[3] = p->last_outer_res > 0 ? p->last_outer_abs / p->last_outer_res : 0.0;
      info6[4] = (double)p->n_precond; info6[5] = (double)p->n_host_syncs;
   }
   return PIPS_OK;
}

void pips_ipm_destroy(void* handle) { delete (Ipm*)handle; }

}  // extern "C"
