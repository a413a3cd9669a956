// Development micro-benchmark 4: shader clock under different FP64 loads on gfx950 (is the GEMM's ~1.95 GHz a power cap?)
//  (a) 4x4x4 MFMA from registers only, (b) the same plus LDS fragment reads in the GEMM's pattern, (c) LDS reads only
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int WITH_MFMA, int WITH_LDS>
__global__ __launch_bounds__(512, 4) void k_load(double* out, int iters, unsigned long long* clk) {
   __shared__ double As[16 * 144], Bs[16 * 144];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   for (int i = tid; i < 16 * 144; i += 512) { As[i] = i * 1e-3; Bs[i] = i * 2e-3; }
   __syncthreads();
   double acc[4][8];
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 8; ++c) acc[i][c] = 0.0;
   const double* Ab = As + (lane >> 4) * 144 + (wave & 1) * 64 + (lane & 15);
   const double* Bb = Bs + (lane >> 4) * 144 + (wave >> 1) * 32 + (lane & 3);
   double fr[4] = {1.0 + lane, 2.0, 3.0, 4.0}, fc[8] = {1, 2, 3, 4, 5, 6, 7, 8.0 + lane};
   const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
         if (WITH_LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fr[i] = Ab[(4 * q) * 144 + i * 16];
#pragma unroll
            for (int c = 0; c < 8; ++c) fc[c] = Bb[(4 * q) * 144 + c * 4];
         }
         if (WITH_MFMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
               for (int c = 0; c < 8; ++c) acc[i][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], fr[i], acc[i][c], 0, 0, 0);
         } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
               for (int c = 0; c < 8; ++c) asm volatile("" :: "v"(fr[i]), "v"(fc[c]));
         }
      }
   }
   const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
   double s = 0;
   for (int i = 0; i < 4; ++i) for (int c = 0; c < 8; ++c) s += acc[i][c];
   out[blockIdx.x * 512 + tid] = s + fr[0] + fc[0];
   if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   const int wg = 512, iters = 8000;
   double* out; unsigned long long* clk; static unsigned long long h[2 * 512];
   CK(hipMalloc(&out, wg * 512 * sizeof(double))); CK(hipMalloc(&clk, sizeof(h)));
#define RUN(name, M, L) do { float ms; \
   hipLaunchKernelGGL((k_load<M, L>), dim3(wg), dim3(512), 0, 0, out, iters, clk); CK(hipDeviceSynchronize()); \
   CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_load<M, L>), dim3(wg), dim3(512), 0, 0, out, iters, clk); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
   CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost)); \
   double cyc = 0, tick = 0; for (int i = 0; i < wg; ++i) { cyc += h[2 * i]; tick += h[2 * i + 1]; } \
   printf("%-28s %8.3f ms  %7.2f TFLOP/s  clock %.0f MHz  cycles/MFMA/SIMD %.2f\n", name, ms, M ? (double)wg * 8 * iters * 128 * 512.0 / ms / 1e9 : 0.0, cyc / tick * 100.0, \
          cyc / wg / ((double)iters * 128 * 4)); } while (0)
   RUN("mfma only", 1, 0);
   RUN("mfma + lds fragment reads", 1, 1);
   RUN("lds fragment reads only", 0, 1);
   RUN("mfma only (again)", 1, 0);
   return 0;
}
