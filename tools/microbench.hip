// Development micro-benchmarks (not part of the product): (1) peak rate of v_mfma_f64_16x16x4_f64 from registers,
// (2) steady-state rate of k_tile_gemm<0> on one huge synthetic tail.  Build: make -C tools ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../include/pips_hip.h"
#include "../pips-ipmpp_amd/csrc/common.h"
#include "../pips-ipmpp_amd/csrc/kernels.hip.h"
using namespace pips;

__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters) {
   double4_t acc[8];
   for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0, 0, 0, 0};
   double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
   }
   double s = 0;
   for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   double* out;
   CK(hipMalloc(&out, 256 * 2048 * sizeof(double)));
   for (int wg : {256, 512, 1024, 2048}) {
      const int iters = 20000;
      hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, out, 100);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = (double)wg * 4 * iters * 8 * 2048.0;
      printf("mfma_f64 peak: %4d WGs x 4 waves: %.2f TFLOP/s (%.3f ms)\n", wg, fl / ms / 1e9, ms);
   }
   // ---- tile GEMM steady state: one block, K = (nt-1)*128, `rows` distinct tile rows below
   const int nt = argc > 1 ? atoi(argv[1]) : 32;
   const int rows = argc > 2 ? atoi(argv[2]) : 2048;
   BlkDesc bd{};
   bd.m_pad = nt * TILE; bd.m = bd.m_pad; bd.nb = 0; bd.nb_pad = rows * TILE; bd.ldT = bd.m_pad + bd.nb_pad;
   bd.ntc = nt; bd.ntr = nt + rows; bd.T = 0; bd.dt_off = 0; bd.winv_off = 0; bd.bmap_off = 0;
   const size_t tsz = (size_t)bd.ldT * bd.m_pad;
   printf("tail panel %.2f GB, K = %d, %d tiles\n", tsz * 8 / 1e9, (nt - 1) * TILE, rows);
   double *T, *dt; BlkDesc* dbd; TileTask* dtk;
   CK(hipMalloc(&T, tsz * sizeof(double))); CK(hipMalloc(&dt, bd.m_pad * sizeof(double)));
   std::vector<double> h(tsz); for (size_t i = 0; i < tsz; ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
   CK(hipMemcpy(T, h.data(), tsz * sizeof(double), hipMemcpyHostToDevice));
   std::vector<double> hd(bd.m_pad, 1.0); CK(hipMemcpy(dt, hd.data(), bd.m_pad * sizeof(double), hipMemcpyHostToDevice));
   CK(hipMalloc(&dbd, sizeof(BlkDesc))); CK(hipMemcpy(dbd, &bd, sizeof(BlkDesc), hipMemcpyHostToDevice));
   std::vector<TileTask> tk; for (int r = 0; r < rows; ++r) tk.push_back({0, nt + r, nt - 1, 0});
   CK(hipMalloc(&dtk, tk.size() * sizeof(TileTask))); CK(hipMemcpy(dtk, tk.data(), tk.size() * sizeof(TileTask), hipMemcpyHostToDevice));
   double* dbg = nullptr;
   double* U;   // scaled copy of the tail rows (B operand)
   CK(hipMalloc(&U, (size_t)bd.m_pad * bd.m_pad * sizeof(double)));
   CK(hipMemcpy(U, T, (size_t)bd.m_pad * bd.m_pad * sizeof(double), hipMemcpyDeviceToDevice));
   // (a) every tile reads its OWN 128 rows of the A panel from HBM (what a left-looking column launch does: the rows of L left of the
   //     column are read once per launch) - and (b) the same number of tiles, the same arithmetic, but only `distinct` different tile rows, so that
   //     the A panel comes out of the caches: what does the update kernel gain when its HBM traffic goes away?  (The C tiles of (b) collide:
   //     irrelevant for the timing.)  Several launches back to back, as a factorisation issues them.
   for (int distinct : {rows, 64, 8}) {
      std::vector<TileTask> t2; for (int r = 0; r < rows; ++r) t2.push_back({0, nt + (r % distinct), nt - 1, 0});
      CK(hipMemcpy(dtk, t2.data(), t2.size() * sizeof(TileTask), hipMemcpyHostToDevice));
      float best = 1e30f, last = 0;
      for (int rep = 0; rep < 12; ++rep) {
         CK(hipEventRecord(e0));
         hipLaunchKernelGGL(k_tile_gemm<0>, dim3((unsigned)tk.size()), dim3(512), 0, 0, dtk, (int)tk.size(), dbd, T, dt, (const double*)nullptr, (const int*)nullptr, dbg, 0, (const int*)nullptr, U);
         CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
         float ms; CK(hipEventElapsedTime(&ms, e0, e1));
         best = ms < best ? ms : best; last = ms;
      }
      const double fl = (double)tk.size() * 2.0 * TILE * TILE * (nt - 1) * TILE;
      printf("k_tile_gemm<0>: %zu tiles K=%d, %4d distinct tile rows (A panel %.2f GB): best %.3f ms %.2f TFLOP/s, 12th launch %.3f ms %.2f TFLOP/s\n", tk.size(), (nt - 1) * TILE, distinct,
             (double)std::min(distinct, rows) * TILE * (nt - 1) * TILE * 8 / 1e9, best, fl / best / 1e9, last, fl / last / 1e9);
   }
   return 0;
}
