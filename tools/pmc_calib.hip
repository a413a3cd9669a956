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
   CK(hipDeviceSynchronize());
   printf("bytes moved per kernel: read8 %lld read16 %lld gather8 %lld (+ %lld index bytes; 64-byte sectors touched: %lld bytes) write8 %lld write16 %lld atomic8 %lld\n",
          n * 8, n * 8, n / 8 * 8, n / 8 * 4, n / 8 * 64, n * 8, n * 8, n * 8);
   return 0;
}
