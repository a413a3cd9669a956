// Development micro-benchmark 3: MFMA 4x4x4 f64 with the GEMM's operand pattern (64 accumulators, 4 x 16 operand grid), registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NW>
__global__ __launch_bounds__(256, NW) void k_pat(double* out, int iters) {
   double acc[4][16], fr[4], fc[16];
   for (int i = 0; i < 4; ++i) fr[i] = threadIdx.x * 1e-3 + i;
   for (int c = 0; c < 16; ++c) fc[c] = threadIdx.x * 2e-3 + c;
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 16; ++c) acc[i][c] = 0;
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
         for (int i = 0; i < 4; ++i) acc[i][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], fr[i], acc[i][c], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fr[i] += 1e-9;
   }
   double s = 0;
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 16; ++c) s += acc[i][c];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same but the inner order keeps the SAME accumulator column group for 4 consecutive MFMAs vs striding
template <int NW>
__global__ __launch_bounds__(256, NW) void k_pat2(double* out, int iters) {
   double acc[4][16], fr[4], fc[16];
   for (int i = 0; i < 4; ++i) fr[i] = threadIdx.x * 1e-3 + i;
   for (int c = 0; c < 16; ++c) fc[c] = threadIdx.x * 2e-3 + c;
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 16; ++c) acc[i][c] = 0;
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int c = 0; c < 16; ++c) acc[i][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], fr[i], acc[i][c], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fr[i] += 1e-9;
   }
   double s = 0;
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 16; ++c) s += acc[i][c];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   double* out; CK(hipMalloc(&out, 512 * 4096 * sizeof(double)));
   float ms; const int iters = 4000;
#define RUN(name, wg, ...) do { __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
   CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-40s %8.3f ms  %7.2f TFLOP/s\n", name, ms, (double)(wg) * 4 * iters * 64 * 512.0 / ms / 1e9); } while (0)
   RUN("pattern c-outer, 256 WG (1 wave/SIMD)", 256, hipLaunchKernelGGL(k_pat<1>, dim3(256), dim3(256), 0, 0, out, iters));
   RUN("pattern c-outer, 512 WG (2 waves/SIMD)", 512, hipLaunchKernelGGL(k_pat<2>, dim3(512), dim3(256), 0, 0, out, iters));
   RUN("pattern i-outer, 256 WG (1 wave/SIMD)", 256, hipLaunchKernelGGL(k_pat2<1>, dim3(256), dim3(256), 0, 0, out, iters));
   RUN("pattern i-outer, 512 WG (2 waves/SIMD)", 512, hipLaunchKernelGGL(k_pat2<2>, dim3(512), dim3(256), 0, 0, out, iters));
   return 0;
}
