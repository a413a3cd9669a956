// Calibration of FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) for the access widths the sparse kernels use: the guide
// (MI355X_MICROARCH.md, HBM section) gives the factor only for 16-byte-per-lane streaming accesses and asks to calibrate
// other widths on a known byte count.  Each kernel moves exactly BYTES bytes of a buffer larger than the Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ void calib_read8(const double* __restrict__ p, long long n, double* out) {
   double s = 0;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[i];
   if (s == 1.2345e300) out[0] = s;
}
__global__ void calib_read16(const double2_t* __restrict__ p, long long n2, double* out) {
   double s = 0;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) { double2_t v = p[i]; s += v.x + v.y; }
   if (s == 1.2345e300) out[0] = s;
}
__global__ void calib_gather8(const double* __restrict__ p, const int* __restrict__ idx, long long n, double* out) {   // 8-byte reads through an index vector
   double s = 0;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[idx[i]];
   if (s == 1.2345e300) out[0] = s;
}
__global__ void calib_write8(double* __restrict__ p, long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 1.0;
}
__global__ void calib_write16(double2_t* __restrict__ p, long long n2) {
   const double2_t v = {1.0, 2.0};
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void calib_atomic8(double* __restrict__ p, long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      __hip_atomic_fetch_add(p + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the access pattern of the solve sweeps (k_head_fwd_chain / k_head_bwd_chain / k_leaf_fwd_gather): a wave reads ONE contiguous piece of a few
// kilobytes (a compact panel of ~33 x 16 doubles = 4224 bytes; a row of the border-row arena) at 8 bytes per lane, the pieces themselves
// at unrelated places of a large arena.  start[] in doubles: multiples of 16 (128-byte aligned) or of 2 (16-byte aligned: what the arena
// offsets guarantee).  Round-4 verdict, weak 3: does the x 2 of streaming reads apply to these kernels?
template <int CHUNK>
__global__ void calib_chunk8(const double* __restrict__ p, const int* __restrict__ start, long long nchunks, double* out) {
   double s = 0;
   const int lane = threadIdx.x & 63;
   for (long long c = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6; c < nchunks; c += ((long long)gridDim.x * blockDim.x) >> 6) {
      const double* q = p + (long long)start[c] * 2;
      for (int i = lane; i < CHUNK; i += 64) s += q[i];
   }
   if (s == 1.2345e300) out[0] = s;
}

int main() {
   const long long n = 1LL << 28;   // 2 GiB of doubles
   double *p, *out; int* idx;
   CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, n / 8 * 4));
   CK(hipMemset(p, 0, n * 8));
   {  // a permutation-like index vector with stride 97 (distinct cache lines for neighbouring lanes)
      int* h = (int*)malloc(n / 8 * 4);
      for (long long i = 0; i < n / 8; ++i) h[i] = (int)((i * 97LL) % (n / 8)) * 8;
      CK(hipMemcpy(idx, h, n / 8 * 4, hipMemcpyHostToDevice));
      free(h);
   }
   hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, p, n, out);
   hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const double2_t*)p, n / 2, out);
   hipLaunchKernelGGL(calib_gather8, dim3(4096), dim3(256), 0, 0, p, idx, n / 8, out);
   hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, p, n);
   hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, (double2_t*)p, n / 2);
   hipLaunchKernelGGL(calib_atomic8, dim3(4096), dim3(256), 0, 0, p, n);
   {  // 2^19 pieces of 528 doubles (2.2 GB read) out of the 2 GiB buffer, starts drawn by a multiplicative hash
      const long long nch = 1LL << 19;
      int *h = (int*)malloc(nch * 4), *st;
      CK(hipMalloc(&st, nch * 4));
      for (int align : {8, 1}) {      // units of 2 doubles: 8 -> 128-byte aligned starts, 1 -> 16-byte aligned
         for (long long i = 0; i < nch; ++i) h[i] = (int)(((unsigned long long)(i * 2654435761ULL) % (unsigned long long)((n - 1024) / 2 / align)) * align);
         CK(hipMemcpy(st, h, nch * 4, hipMemcpyHostToDevice));
         if (align == 8) hipLaunchKernelGGL(calib_chunk8<528>, dim3(4096), dim3(256), 0, 0, p, st, nch, out);
         else hipLaunchKernelGGL(calib_chunk8<529>, dim3(4096), dim3(256), 0, 0, p, st, nch, out);     // (another instantiation = another kernel name in the counter file)
         CK(hipDeviceSynchronize());
      }
      free(h);
   }
   CK(hipDeviceSynchronize());
   printf("calib_chunk8<528>: %lld bytes in 128-byte aligned pieces of 4224; calib_chunk8<529>: %lld bytes in 16-byte aligned pieces of 4232\n", (1LL << 19) * 528 * 8, (1LL << 19) * 529 * 8);
   printf("bytes moved per kernel: read8 %lld read16 %lld gather8 %lld (+ %lld index bytes; 64-byte sectors touched: %lld bytes) write8 %lld write16 %lld atomic8 %lld\n",
          n * 8, n * 8, n / 8 * 8, n / 8 * 4, n / 8 * 64, n * 8, n * 8, n * 8);
   return 0;
}
