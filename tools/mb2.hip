// Development micro-benchmark 2: FP64 issue rates on gfx950 — MFMA 16x16x4 vs 4x4x4, VALU v_fma_f64, mixed, and the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double* out, int iters, unsigned long long* clk) {
   double4_t acc[NACC];
   for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
   double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
   unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
   }
   unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
   double s = 0;
   for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters) {
   double acc[8];
   for (int i = 0; i < 8; ++i) acc[i] = 0;
   double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
   }
   double s = 0;
   for (int i = 0; i < 8; ++i) s += acc[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
   double acc[16];
   for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
   double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
   }
   double s = 0;
   for (int i = 0; i < 16; ++i) s += acc[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0,1 issue MFMA, waves 2..: VALU FMA (same CU, different SIMDs get both kinds when 2 WGs are resident)
__global__ __launch_bounds__(512) void k_mixed(double* out, int iters, int fma_per_mfma) {
   const int wave = threadIdx.x >> 6;
   double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
   if (wave < 4) {
      double4_t acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
         for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      double s = 0;
      for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
      out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   } else {
      double acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
      for (int it = 0; it < iters * fma_per_mfma; ++it) {
#pragma unroll
         for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
      }
      double s = 0;
      for (int i = 0; i < 16; ++i) s += acc[i];
      out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   }
}

int main() {
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   double* out; unsigned long long* clk; unsigned long long hclk[2];
   CK(hipMalloc(&out, 512 * 4096 * sizeof(double))); CK(hipMalloc(&clk, 16));
   float ms;
   const int iters = 20000;
#define RUN(name, flops, ...) do { __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
   CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", name, ms, (flops) / ms / 1e9); } while (0)
   for (int wg : {256, 512, 1024}) {
      char nm[64];
      snprintf(nm, 64, "mfma16x16x4 4acc  %d WG", wg);
      RUN(nm, (double)wg * 4 * iters * 4 * 2048.0, hipLaunchKernelGGL(k_mfma16<4>, dim3(wg), dim3(256), 0, 0, out, iters, clk));
      snprintf(nm, 64, "mfma16x16x4 16acc %d WG", wg);
      RUN(nm, (double)wg * 4 * iters * 16 * 2048.0, hipLaunchKernelGGL(k_mfma16<16>, dim3(wg), dim3(256), 0, 0, out, iters, clk));
      CK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
      printf("   shader clock during run: %.0f MHz (cycles per MFMA per wave: %.1f)\n", (double)hclk[0] / hclk[1] * 100.0, (double)hclk[0] / (iters * 16.0));
      snprintf(nm, 64, "mfma4x4x4 8acc    %d WG", wg);
      RUN(nm, (double)wg * 4 * iters * 8 * 512.0, hipLaunchKernelGGL(k_mfma4, dim3(wg), dim3(256), 0, 0, out, iters));
      snprintf(nm, 64, "v_fma_f64 16acc   %d WG", wg);
      RUN(nm, (double)wg * 4 * iters * 16 * 128.0, hipLaunchKernelGGL(k_fma, dim3(wg), dim3(256), 0, 0, out, iters));
   }
   for (int f : {1, 2, 4}) {
      char nm[64];
      snprintf(nm, 64, "mixed 4 mfma waves + 4 fma waves x%d, 512 WG", f);
      const double fl = 512.0 * (4.0 * iters * 8 * 2048.0 + 4.0 * iters * f * 16 * 128.0);
      RUN(nm, fl, hipLaunchKernelGGL(k_mixed, dim3(512), dim3(512), 0, 0, out, iters, f));
   }
   // ---- the ceiling of the update kernel: v_mfma_f64_4x4x4_4b issued from registers (no LDS, no memory) on every SIMD, SUSTAINED.  The
   //      78.6 TFLOP/s of the specification assumes 2.4 GHz; under FP64 matrix load the chip settles at a power-limited clock.  Forty
   //      back-to-back launches of ~25 ms each (one second of load): rate per launch, and the shader clock of the last one.
   {
      const int wg = 1024, it2 = 250000;     // 4 waves per SIMD, ~25 ms per launch at the full rate
      double first = 0, last = 0, lo = 1e30;
      for (int rep = 0; rep < 40; ++rep) {
         CK(hipEventRecord(e0));
         hipLaunchKernelGGL(k_mfma4, dim3(wg), dim3(256), 0, 0, out, it2);
         CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
         CK(hipEventElapsedTime(&ms, e0, e1));
         const double tf = (double)wg * 4 * it2 * 8 * 512.0 / ms / 1e9;
         if (rep == 0) first = tf;
         last = tf; lo = tf < lo ? tf : lo;
         if (rep % 8 == 0 || rep == 39) printf("sustained mfma4x4x4, launch %2d: %7.3f ms  %6.2f TFLOP/s\n", rep, ms, tf);
      }
      hipLaunchKernelGGL(k_mfma16<4>, dim3(1024), dim3(256), 0, 0, out, 50000, clk);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
      printf("sustained mfma4x4x4 from registers: first launch %.2f, last %.2f, lowest %.2f TFLOP/s; shader clock right after: %.0f MHz\n", first, last, lo,
             (double)hclk[0] / hclk[1] * 100.0);
   }
   return 0;
}
