// Flat-arena vector kernels: the DistributedVector<T> / DenseVector<T> operations the IPM uses around the KKT path
// (LinearAlgebra/Distributed/DistributedVector.C:406-460,1160-1340; LinearAlgebra/Dense/DenseVector.cpp:281-516).
// A vector tree {first, children...} is one contiguous device array (root part first, then the leaves of this rank);
// element-wise kernels run over the whole arena in one launch, reductions are two-stage (per-workgroup partials, then
// one workgroup) and return ONE scalar to the host — the reference does one MPI_Allreduce per reduction at the same
// point (DistributedVector.C:421,456,1306).  Replicated root entries are counted once by giving `root_len` entries the
// weight 0 on every rank but the "special" one (iAmSpecial, DistributedVector.C:1293-1303): pass count_root = 0 there.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <sys/syscall.h>
#include <unistd.h>

#include "common.h"
#include "pips_hip.h"

namespace pips {

#define HIP_TRYV(expr)                                                                                   \
   do {                                                                                                  \
      hipError_t _e = (expr);                                                                            \
      if (_e != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
   } while (0)

static inline int vgrid(long long n) {
   long long g = (n + 255) / 256;
   return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

enum VecOp : int {
   OP_AXPY = 0,      // y += a x
   OP_SCALE,         // y *= a
   OP_COPY,          // y = x
   OP_SET,           // y = a
   OP_MUL,           // y *= x             (componentMult)
   OP_DIV,           // y /= x             (componentDiv)
   OP_ADD_PRODUCT,   // y += a x z         (add_product)
   OP_ADD_QUOTIENT,  // y += a x / z  where mask != 0   (add_quotient with index vector)
   OP_DIVIDE_SOME,   // y /= x where mask != 0          (divideSome)
   OP_SELECT,        // y = mask != 0 ? y : 0           (selectNonZeros)
   OP_SAFE_INVERT,   // y = y != 0 ? 1/y : 0            (safe_invert)
   OP_ADD_CONST,     // y += a
   OP_AXPBY,         // y = a x + b y
   OP_GONDZIO,       // y = projection step onto [a, b] (gondzioProjection, DenseVector.cpp:405-420)
};

// one element of operation OP; `v` is y_i on entry where the operation reads it
template <int OP>
__device__ __forceinline__ double vec_elem(double v, double a, double b, double x, double z, bool on) {
   switch (OP) {
      case OP_AXPY: return v + a * x;
      case OP_SCALE: return v * a;
      case OP_COPY: return x;
      case OP_SET: return a;
      case OP_MUL: return v * x;
      case OP_DIV: return v / x;
      case OP_ADD_PRODUCT: return v + a * x * z;
      case OP_ADD_QUOTIENT: return on ? v + a * x / z : v;
      case OP_DIVIDE_SOME: return on ? v / x : v;
      case OP_SELECT: return on ? v : 0.0;
      case OP_SAFE_INVERT: return v != 0.0 ? 1.0 / v : 0.0;
      case OP_ADD_CONST: return v + a;
      case OP_AXPBY: return a * x + b * v;
      default: { double t = v < a ? a - v : (v > b ? b - v : 0.0); return t < -b ? -b : t; }   // OP_GONDZIO
   }
}
template <int OP> constexpr bool vec_reads_y() { return OP != OP_COPY && OP != OP_SET; }
template <int OP> constexpr bool vec_reads_x() { return OP == OP_AXPY || OP == OP_COPY || OP == OP_MUL || OP == OP_DIV || OP == OP_ADD_PRODUCT || OP == OP_ADD_QUOTIENT || OP == OP_DIVIDE_SOME || OP == OP_AXPBY; }
template <int OP> constexpr bool vec_reads_z() { return OP == OP_ADD_PRODUCT || OP == OP_ADD_QUOTIENT; }
template <int OP> constexpr bool vec_uses_mask() { return OP == OP_ADD_QUOTIENT || OP == OP_DIVIDE_SOME || OP == OP_SELECT; }

// Element-wise kernels are pure streaming: one specialisation per operation (no switch in the loop, operands an operation does
// not use are never read - y itself for SET / COPY), 16-byte accesses (two doubles per lane and step) when every pointer is
// 16-byte aligned, a scalar tail.
typedef double double2v __attribute__((ext_vector_type(2)));
template <int OP, bool VEC2>
__global__ __launch_bounds__(256) void k_vec_op(long long n, double a, double b, const double* __restrict__ x, const double* __restrict__ z,
                                                const double* __restrict__ mask, double* __restrict__ y) {
   const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
   if (VEC2) {
      const long long n2 = n / 2;
      for (long long i = tid; i < n2; i += stride) {
         double2v yv = {0.0, 0.0}, xv = {0.0, 0.0}, zv = {0.0, 0.0}, mv = {1.0, 1.0};
         if (vec_reads_y<OP>()) yv = ((const double2v*)y)[i];
         if (vec_reads_x<OP>()) xv = ((const double2v*)x)[i];
         if (vec_reads_z<OP>()) zv = ((const double2v*)z)[i];
         if (vec_uses_mask<OP>() && mask) mv = ((const double2v*)mask)[i];
         double2v r;
         r.x = vec_elem<OP>(yv.x, a, b, xv.x, zv.x, mv.x != 0.0);
         r.y = vec_elem<OP>(yv.y, a, b, xv.y, zv.y, mv.y != 0.0);
         ((double2v*)y)[i] = r;
      }
      if (tid == 0 && (n & 1)) {
         const long long i = n - 1;
         y[i] = vec_elem<OP>(vec_reads_y<OP>() ? y[i] : 0.0, a, b, vec_reads_x<OP>() ? x[i] : 0.0, vec_reads_z<OP>() ? z[i] : 0.0,
                             !(vec_uses_mask<OP>() && mask) || mask[i] != 0.0);
      }
   } else {
      for (long long i = tid; i < n; i += stride)
         y[i] = vec_elem<OP>(vec_reads_y<OP>() ? y[i] : 0.0, a, b, vec_reads_x<OP>() ? x[i] : 0.0, vec_reads_z<OP>() ? z[i] : 0.0,
                             !(vec_uses_mask<OP>() && mask) || mask[i] != 0.0);
   }
}

enum RedOp : int {
   RED_DOT = 0,     // sum x y
   RED_SUM_ABS,     // sum |x|            (one_norm)
   RED_MAX_ABS,     // max |x|            (inf_norm)
   RED_SUMSQ_SCALED,// sum (x/s)^2        (two_norm = s sqrt(.), s = inf_norm; DistributedVector.C:424-437)
   RED_MIN,         // min x
   RED_STEPBOUND,   // min over {i : dx_i < 0, mask_i != 0} of -x_i/dx_i      (fraction_to_boundary / stepbound, Variables.C:191-225)
   RED_DOT_SHIFTED, // sum (x + a dx)(y + b dy)                                (mustep_pd, Variables.C:109)
};

__device__ __forceinline__ double red_combine(int op, double u, double v) {
   switch (op) {
      case RED_MAX_ABS: return fmax(u, v);
      case RED_MIN:
      case RED_STEPBOUND: return fmin(u, v);
      default: return u + v;
   }
}

__device__ __forceinline__ double red_identity(int op) {
   switch (op) {
      case RED_MIN:
      case RED_STEPBOUND: return INFINITY;
      default: return 0.0;
   }
}

// skip: the first `skip` entries (replicated root part) are ignored on non-special ranks
__global__ void k_vec_reduce(int op, long long n, long long skip, double a, double b, const double* __restrict__ x,
                             const double* __restrict__ y, const double* __restrict__ dx, const double* __restrict__ dy,
                             const double* __restrict__ mask, double* __restrict__ partial) {
   double acc = red_identity(op);
   for (long long i = skip + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      double t;
      switch (op) {
         case RED_DOT: t = x[i] * y[i]; break;
         case RED_SUM_ABS: t = fabs(x[i]); break;
         case RED_MAX_ABS: t = fabs(x[i]); break;
         case RED_SUMSQ_SCALED: { const double q = x[i] * a; t = q * q; break; }
         case RED_MIN: t = x[i]; break;
         case RED_STEPBOUND: t = (dx[i] < 0.0 && (mask == nullptr || mask[i] != 0.0)) ? -x[i] / dx[i] : INFINITY; break;
         default: t = (x[i] + a * dx[i]) * (y[i] + b * dy[i]); break;
      }
      acc = red_combine(op, acc, t);
   }
   __shared__ double red[256];
   red[threadIdx.x] = acc;
   __syncthreads();
   for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = red_combine(op, red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void k_vec_reduce_final(int op, int n, const double* __restrict__ partial, double* __restrict__ out) {
   double acc = red_identity(op);
   for (int i = threadIdx.x; i < n; i += blockDim.x) acc = red_combine(op, acc, partial[i]);
   __shared__ double red[256];
   red[threadIdx.x] = acc;
   __syncthreads();
   for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = red_combine(op, red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0) out[0] = red[0];
}

// find_blocking (Variables::find_blocking, Variables.C:227-308): index of the entry that attains the minimum ratio -x_i / dx_i
// over dx_i < 0 (smallest index on ties), then the values of x, dx and of the paired vectors y, dy there
__global__ void k_find_index(long long n, const double* __restrict__ x, const double* __restrict__ dx, const double* __restrict__ target,
                             unsigned long long* __restrict__ idx) {
   const double t = target[0];
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      if (dx[i] < 0.0 && -x[i] / dx[i] == t) atomicMin(idx, (unsigned long long)i);
}

__global__ void k_gather_blocking(const unsigned long long* __restrict__ idx, const double* __restrict__ x, const double* __restrict__ dx,
                                  const double* __restrict__ y, const double* __restrict__ dy, double* __restrict__ out) {
   const unsigned long long i = idx[0];
   if (i == ~0ULL) { out[1] = out[2] = out[3] = out[4] = 0.0; return; }
   out[1] = x[i]; out[2] = dx[i]; out[3] = y[i]; out[4] = dy[i];
}

// Step bounds of nw blended directions at once: for w_k = min(1, wmin + (1 - wmin) k / (nw - 1)) the largest alpha with
// x + alpha (dx + w_k cx) >= 0, for two vector triples (the 11-point corrector weight search of the IPM, InteriorPointMethod.cpp:
// 486-523: 22 reductions and 44 vector passes in one kernel).  out holds 2 nw doubles preset to +inf; the ratios are non-negative,
// so the order of their bit patterns is the order of the numbers and an integer atomicMin finishes the reduction.
constexpr int WS_MAX = 16;
__global__ __launch_bounds__(256) void k_weighted_stepbounds(long long n, const double* __restrict__ x, const double* __restrict__ dx,
                                                            const double* __restrict__ cx, const double* __restrict__ y,
                                                            const double* __restrict__ dy, const double* __restrict__ cy, double wmin,
                                                            int nw, double* __restrict__ out) {
   double bx[WS_MAX], by[WS_MAX];
#pragma unroll
   for (int k = 0; k < WS_MAX; ++k) bx[k] = by[k] = INFINITY;
   const double step = nw > 1 ? (1.0 - wmin) / (nw - 1) : 0.0;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      const double xi = x[i], dxi = dx[i], cxi = cx[i], yi = y[i], dyi = dy[i], cyi = cy[i];
#pragma unroll
      for (int k = 0; k < WS_MAX; ++k) {
         if (k < nw) {
            const double w = fmin(1.0, wmin + step * k);
            const double sx = dxi + w * cxi, sy = dyi + w * cyi;
            if (sx < 0.0) bx[k] = fmin(bx[k], -xi / sx);
            if (sy < 0.0) by[k] = fmin(by[k], -yi / sy);
         }
      }
   }
   __shared__ double red[256];
   for (int k = 0; k < 2 * nw; ++k) {
      double t = k < nw ? INFINITY : INFINITY;
#pragma unroll
      for (int q = 0; q < WS_MAX; ++q) {   // register arrays need constant indices
         if (q == (k < nw ? k : k - nw)) t = k < nw ? bx[q] : by[q];
      }
      red[threadIdx.x] = t;
      __syncthreads();
      for (int sft = 128; sft > 0; sft >>= 1) {
         if ((int)threadIdx.x < sft) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + sft]);
         __syncthreads();
      }
      if (threadIdx.x == 0 && red[0] < INFINITY) atomicMin((unsigned long long*)(out + k), (unsigned long long)__double_as_longlong(red[0]));
      __syncthreads();
   }
}

static double red_identity_host(int op) { return (op == RED_MIN || op == RED_STEPBOUND) ? INFINITY : 0.0; }

// Reduction workspace: one per (thread, device).  The entry points take device pointers only, so the workspace of the device
// that is current at the call is used - a workspace allocated on another device would be written across the fabric (or
// fault without peer access).  Freed when a worker thread ends.
struct VecWorkspace {
   double* d_partial = nullptr;  // 2048 partials + 8 result slots
   double* h_out = nullptr;      // pinned
};
constexpr int VEC_MAX_DEVICES = 64;
struct VecWorkspaces {
   VecWorkspace ws[VEC_MAX_DEVICES];
   ~VecWorkspaces() {
      // the main thread's copy goes away at process exit, when the HIP runtime may already be shutting down: leave it to the OS
      if (getpid() == (pid_t)syscall(SYS_gettid)) return;
      int cur = 0;
      const bool have = hipGetDevice(&cur) == hipSuccess;
      for (int d = 0; d < VEC_MAX_DEVICES; ++d) {
         if (!ws[d].d_partial) continue;
         if (hipSetDevice(d) == hipSuccess) {
            (void)hipFree(ws[d].d_partial);
            (void)hipHostFree(ws[d].h_out);
         }
      }
      if (have) (void)hipSetDevice(cur);
   }
};
static thread_local VecWorkspaces g_wss;
static thread_local VecWorkspace* g_cur = nullptr;
struct WsRef {   // g_ws.member resolves to the current device's workspace (set by g_ws.init())
   int init() {
      int dev = 0;
      HIP_TRYV(hipGetDevice(&dev));
      if (dev < 0 || dev >= VEC_MAX_DEVICES) PIPS_FAIL(PIPS_ERR_ARG, "vector layer: device index %d out of range", dev);
      VecWorkspace& w = g_wss.ws[dev];
      if (!w.d_partial) {
         HIP_TRYV(hipMalloc((void**)&w.d_partial, (2048 + 8) * sizeof(double)));
         HIP_TRYV(hipHostMalloc((void**)&w.h_out, 8 * sizeof(double), hipHostMallocDefault));
      }
      g_cur = &w;
      d_partial = w.d_partial;
      h_out = w.h_out;
      return PIPS_OK;
   }
   double* d_partial = nullptr;
   double* h_out = nullptr;
};
static thread_local WsRef g_ws;

template <int OP>
static void vec_launch(long long n, double a, double b, const double* x, const double* z, const double* mask, double* y, hipStream_t s) {
   auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
   if (al(x) && al(z) && al(mask) && al(y) && n >= 2)
      hipLaunchKernelGGL((k_vec_op<OP, true>), dim3(vgrid((n + 1) / 2)), dim3(256), 0, s, n, a, b, x, z, mask, y);
   else
      hipLaunchKernelGGL((k_vec_op<OP, false>), dim3(vgrid(n)), dim3(256), 0, s, n, a, b, x, z, mask, y);
}

int vec_apply(int op, long long n, double a, double b, const double* x, const double* z, const double* mask, double* y,
              hipStream_t s) {
   if (n <= 0) return PIPS_OK;
   switch (op) {
#define PIPS_VEC_CASE(OP) case OP: vec_launch<OP>(n, a, b, x, z, mask, y, s); break;
      PIPS_VEC_CASE(OP_AXPY) PIPS_VEC_CASE(OP_SCALE) PIPS_VEC_CASE(OP_COPY) PIPS_VEC_CASE(OP_SET) PIPS_VEC_CASE(OP_MUL) PIPS_VEC_CASE(OP_DIV)
      PIPS_VEC_CASE(OP_ADD_PRODUCT) PIPS_VEC_CASE(OP_ADD_QUOTIENT) PIPS_VEC_CASE(OP_DIVIDE_SOME) PIPS_VEC_CASE(OP_SELECT)
      PIPS_VEC_CASE(OP_SAFE_INVERT) PIPS_VEC_CASE(OP_ADD_CONST) PIPS_VEC_CASE(OP_AXPBY) PIPS_VEC_CASE(OP_GONDZIO)
#undef PIPS_VEC_CASE
      default: PIPS_FAIL(PIPS_ERR_ARG, "vec_apply: unknown operation %d", op);
   }
   HIP_TRYV(hipGetLastError());
   return PIPS_OK;
}

int vec_reduce(int op, long long n, long long skip, double a, double b, const double* x, const double* y, const double* dx,
               const double* dy, const double* mask, double* result, hipStream_t s) {
   int rc = g_ws.init();
   if (rc) return rc;
   if (n - skip <= 0) {
      *result = red_identity_host(op);
      return PIPS_OK;
   }
   const int g = vgrid(n - skip);
   hipLaunchKernelGGL(k_vec_reduce, dim3(g), dim3(256), 0, s, op, n, skip, a, b, x, y, dx, dy, mask, g_ws.d_partial);
   hipLaunchKernelGGL(k_vec_reduce_final, dim3(1), dim3(256), 0, s, op, g, g_ws.d_partial, g_ws.d_partial + 2048);
   HIP_TRYV(hipMemcpyAsync(g_ws.h_out, g_ws.d_partial + 2048, sizeof(double), hipMemcpyDeviceToHost, s));
   HIP_TRYV(hipStreamSynchronize(s));
   *result = g_ws.h_out[0];
   return PIPS_OK;
}

}  // namespace pips

using namespace pips;

extern "C" {

int pips_hip_vec_axpy(long long n, double a, const double* x_dev, double* y_dev, void* stream) {
   return vec_apply(OP_AXPY, n, a, 0, x_dev, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_axpby(long long n, double a, const double* x_dev, double b, double* y_dev, void* stream) {
   return vec_apply(OP_AXPBY, n, a, b, x_dev, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_scale(long long n, double a, double* y_dev, void* stream) {
   return vec_apply(OP_SCALE, n, a, 0, nullptr, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_copy(long long n, const double* x_dev, double* y_dev, void* stream) {
   return vec_apply(OP_COPY, n, 0, 0, x_dev, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_set(long long n, double a, double* y_dev, void* stream) {
   return vec_apply(OP_SET, n, a, 0, nullptr, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_add_const(long long n, double a, double* y_dev, void* stream) {
   return vec_apply(OP_ADD_CONST, n, a, 0, nullptr, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_mul(long long n, const double* x_dev, double* y_dev, void* stream) {
   return vec_apply(OP_MUL, n, 0, 0, x_dev, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_div(long long n, const double* x_dev, double* y_dev, void* stream) {
   return vec_apply(OP_DIV, n, 0, 0, x_dev, nullptr, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_add_product(long long n, double a, const double* x_dev, const double* z_dev, double* y_dev, void* stream) {
   return vec_apply(OP_ADD_PRODUCT, n, a, 0, x_dev, z_dev, nullptr, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_add_quotient(long long n, double a, const double* x_dev, const double* z_dev, const double* mask_dev,
                              double* y_dev, void* stream) {
   return vec_apply(OP_ADD_QUOTIENT, n, a, 0, x_dev, z_dev, mask_dev, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_divide_some(long long n, const double* x_dev, const double* mask_dev, double* y_dev, void* stream) {
   return vec_apply(OP_DIVIDE_SOME, n, 0, 0, x_dev, nullptr, mask_dev, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_select_nonzeros(long long n, const double* mask_dev, double* y_dev, void* stream) {
   return vec_apply(OP_SELECT, n, 0, 0, nullptr, nullptr, mask_dev, y_dev, (hipStream_t)stream);
}
int pips_hip_vec_safe_invert(long long n, double* y_dev, void* stream) {
   return vec_apply(OP_SAFE_INVERT, n, 0, 0, nullptr, nullptr, nullptr, y_dev, (hipStream_t)stream);
}

int pips_hip_vec_gondzio_projection(long long n, double rmin, double rmax, double* y_dev, void* stream) {
   return vec_apply(OP_GONDZIO, n, rmin, rmax, nullptr, nullptr, nullptr, y_dev, (hipStream_t)stream);
}

int pips_hip_vec_dot(long long n, long long skip_root, const double* x_dev, const double* y_dev, double* result, void* stream) {
   return vec_reduce(RED_DOT, n, skip_root, 0, 0, x_dev, y_dev, nullptr, nullptr, nullptr, result, (hipStream_t)stream);
}
int pips_hip_vec_one_norm(long long n, long long skip_root, const double* x_dev, double* result, void* stream) {
   return vec_reduce(RED_SUM_ABS, n, skip_root, 0, 0, x_dev, nullptr, nullptr, nullptr, nullptr, result, (hipStream_t)stream);
}
int pips_hip_vec_inf_norm(long long n, const double* x_dev, double* result, void* stream) {
   return vec_reduce(RED_MAX_ABS, n, 0, 0, 0, x_dev, nullptr, nullptr, nullptr, nullptr, result, (hipStream_t)stream);
}
int pips_hip_vec_min(long long n, const double* x_dev, double* result, void* stream) {
   return vec_reduce(RED_MIN, n, 0, 0, 0, x_dev, nullptr, nullptr, nullptr, nullptr, result, (hipStream_t)stream);
}
/* two_norm = s * sqrt(sum (x/s)^2), s = inf_norm (DistributedVector.C:424-437): scale_inv = 1/s from a previous inf_norm
 * (combined across ranks by the caller); returns the local sum of squares of the scaled entries */
int pips_hip_vec_sumsq_scaled(long long n, long long skip_root, double scale_inv, const double* x_dev, double* result, void* stream) {
   return vec_reduce(RED_SUMSQ_SCALED, n, skip_root, scale_inv, 0, x_dev, nullptr, nullptr, nullptr, nullptr, result, (hipStream_t)stream);
}
/* largest alpha in (0, +inf] with x + alpha dx >= 0 on the masked entries (min ratio test) */
int pips_hip_vec_stepbound(long long n, const double* x_dev, const double* dx_dev, const double* mask_dev, double* result, void* stream) {
   return vec_reduce(RED_STEPBOUND, n, 0, 0, 0, x_dev, nullptr, dx_dev, nullptr, mask_dev, result, (hipStream_t)stream);
}
/* out5 = [min ratio -x_i/dx_i over dx_i < 0 (inf if none), x_b, dx_b, y_b, dy_b at the blocking index b]
 * (Variables::find_blocking, Variables.C:227-308, for one pair of complementary vectors) */
int pips_hip_vec_find_blocking(long long n, const double* x_dev, const double* dx_dev, const double* y_dev, const double* dy_dev,
                               double* out5, void* stream) {
   hipStream_t s = (hipStream_t)stream;
   double ratio;
   int rc = vec_reduce(RED_STEPBOUND, n, 0, 0, 0, x_dev, nullptr, dx_dev, nullptr, nullptr, &ratio, s);
   if (rc) return rc;
   out5[0] = ratio;
   out5[1] = out5[2] = out5[3] = out5[4] = 0.0;
   if (n <= 0 || !(ratio < INFINITY)) return PIPS_OK;
   // slot layout of the workspace: [2048] the minimum (still there from vec_reduce), [2049] the index, [2050..2054] the result
   unsigned long long* d_idx = (unsigned long long*)(g_ws.d_partial + 2049);
   HIP_TRYV(hipMemsetAsync(d_idx, 0xff, sizeof(unsigned long long), s));
   hipLaunchKernelGGL(k_find_index, dim3(vgrid(n)), dim3(256), 0, s, n, x_dev, dx_dev, g_ws.d_partial + 2048, d_idx);
   hipLaunchKernelGGL(k_gather_blocking, dim3(1), dim3(1), 0, s, d_idx, x_dev, dx_dev, y_dev, dy_dev, g_ws.d_partial + 2050);
   HIP_TRYV(hipMemcpyAsync(g_ws.h_out, g_ws.d_partial + 2050, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
   HIP_TRYV(hipStreamSynchronize(s));
   for (int i = 1; i < 5; ++i) out5[i] = g_ws.h_out[i];
   return PIPS_OK;
}
int pips_hip_vec_weighted_stepbounds(long long n, const double* x_dev, const double* dx_dev, const double* cx_dev, const double* y_dev,
                                     const double* dy_dev, const double* cy_dev, double wmin, int nw, double* out2nw, void* stream) {
   if (nw < 1 || nw > WS_MAX || !out2nw) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_vec_weighted_stepbounds: 1 <= nw <= 16");
   hipStream_t s = (hipStream_t)stream;
   int rc = g_ws.init();
   if (rc) return rc;
   for (int k = 0; k < 2 * nw; ++k) out2nw[k] = INFINITY;
   if (n <= 0) return PIPS_OK;
   double* d_out = g_ws.d_partial;   // the first 2 nw slots of the reduction workspace
   HIP_TRYV(hipMemcpyAsync(d_out, out2nw, 2 * nw * sizeof(double), hipMemcpyHostToDevice, s));
   hipLaunchKernelGGL(k_weighted_stepbounds, dim3(vgrid(n) < 1024 ? vgrid(n) : 1024), dim3(256), 0, s, n, x_dev, dx_dev, cx_dev, y_dev, dy_dev,
                      cy_dev, wmin, nw, d_out);
   HIP_TRYV(hipMemcpyAsync(out2nw, d_out, 2 * nw * sizeof(double), hipMemcpyDeviceToHost, s));
   HIP_TRYV(hipStreamSynchronize(s));
   return PIPS_OK;
}
/* sum (x + a dx)(y + b dy)  (complementarity after a trial step) */
int pips_hip_vec_dot_shifted(long long n, long long skip_root, const double* x_dev, double a, const double* dx_dev,
                             const double* y_dev, double b, const double* dy_dev, double* result, void* stream) {
   return vec_reduce(RED_DOT_SHIFTED, n, skip_root, a, b, x_dev, y_dev, dx_dev, dy_dev, nullptr, result, (hipStream_t)stream);
}

}  // extern "C"
