// Development micro-benchmark: k_tile_diag alone on n independent diagonal tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pips-ipmpp_amd/csrc/kernels.hip.h"
using namespace pips;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
   const int nt = 64;
   BlkDesc bd{};
   bd.m_pad = nt * TILE; bd.m = bd.m_pad; bd.ldT = bd.m_pad; bd.ntc = bd.ntr = nt; bd.n_head = 0; bd.n = bd.m;
   bd.thr_rel = 1e-13; bd.repl_rel = 1e-8; bd.repl_abs = 1e-8;
   const size_t tsz = (size_t)bd.ldT * bd.m_pad;
   double *T, *dt, *wi, *pref; BlkDesc* dbd; TileTask* dtk; signed char* ps; long long* pso; int* inert;
   CK(hipMalloc(&T, tsz * 8)); CK(hipMalloc(&dt, bd.m_pad * 8)); CK(hipMalloc(&wi, (size_t)nt * TILE * TILE * 8)); CK(hipMalloc(&pref, bd.m_pad * 8));
   CK(hipMalloc(&ps, bd.m_pad)); CK(hipMalloc(&pso, 8)); CK(hipMalloc(&inert, 12));
   std::vector<double> h(tsz, 0.0);
   for (int t = 0; t < nt; ++t)
      for (int c = 0; c < TILE; ++c)
         for (int r = c; r < TILE; ++r) {
            const size_t idx = (size_t)(t * TILE + r) + (size_t)(t * TILE + c) * bd.ldT;
            h[idx] = r == c ? 200.0 : ((r * 131 + c * 71) % 97) * 0.01;
         }
   CK(hipMemcpy(T, h.data(), tsz * 8, hipMemcpyHostToDevice));
   std::vector<double> hp(bd.m_pad, 200.0); CK(hipMemcpy(pref, hp.data(), bd.m_pad * 8, hipMemcpyHostToDevice));
   std::vector<signed char> hs(bd.m_pad, 1); CK(hipMemcpy(ps, hs.data(), bd.m_pad, hipMemcpyHostToDevice));
   long long z = 0; CK(hipMemcpy(pso, &z, 8, hipMemcpyHostToDevice)); CK(hipMemset(inert, 0, 12));
   CK(hipMalloc(&dbd, sizeof(BlkDesc))); CK(hipMemcpy(dbd, &bd, sizeof(BlkDesc), hipMemcpyHostToDevice));
   std::vector<TileTask> tk; for (int t = 0; t < nt; ++t) tk.push_back({0, t, t, 0});
   CK(hipMalloc(&dtk, tk.size() * sizeof(TileTask))); CK(hipMemcpy(dtk, tk.data(), tk.size() * sizeof(TileTask), hipMemcpyHostToDevice));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int n : {1, 8, 64, 64}) {
      CK(hipMemcpy(T, h.data(), tsz * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_tile_diag, dim3(n), dim3(256), 0, 0, dtk, dbd, T, dt, wi, ps, pso, inert, pref);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("k_tile_diag %d tiles: %.1f us\n", n, ms * 1e3);
   }
   return 0;
}
