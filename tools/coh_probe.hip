// Does a line that a compute unit has read with an agent-scope access (sc1: the way the single-launch factorisations read and write the
// tiles they accumulate) stay in its XCD's L2, so that a later PLAIN load of the same address could hit the old value after another XCD
// has written the line through?  The question behind "may the final L of a tile live where the tile was accumulated" (DESIGN.md 4.2a).
// Pairs of workgroups on different XCDs (neighbouring workgroup ids): the reader touches a fresh piece of memory (zeros) in mode
//   0: plain loads (control: stale values EXPECTED if the L2 keeps lines across the flag)      1: agent-scope loads
//   2: agent-scope LDS-DMA (global_load_lds_dwordx4 sc1)                                          3: agent-scope load + agent-scope store back (RMW)
// then the writer stores ones (agent scope: written through) and raises a flag; the reader reads the piece again with PLAIN loads
// (reread 0) or plain LDS-DMA (reread 1) and counts what is not one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int PIECE = 2048;   // doubles per (pair, round): 16 KB = 128 lines of 128 bytes
__device__ __forceinline__ void glds16(const double* g, double* l, bool sc1) {
   const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)l);
   if (sc1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(g), "s"(lds) : "memory", "m0");
   else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void wait_ge(int* f, int v) {
   if (threadIdx.x == 0) while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(2);
   __syncthreads();
}
__device__ __forceinline__ void publish(int* f, int v) {
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_probe(double* X, int* flags, int rounds, int mode, int reread, unsigned long long* stale, unsigned long long* first_nonzero, int* xcd_of) {
   __shared__ double buf[PIECE];
   const int pair = blockIdx.x >> 1, role = blockIdx.x & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   if (tid == 0) xcd_of[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
   int* fa = flags + 2 * pair;
   int* fb = fa + 1;
   unsigned long long bad = 0, nz = 0;
   for (int r = 0; r < rounds; ++r) {
      double* p = X + ((long long)pair * rounds + r) * PIECE;
      if (role == 0) {
         double s = 0.0;
         if (mode == 0) for (int i = tid; i < PIECE; i += 256) s += p[i];
         else if (mode == 1) for (int i = tid; i < PIECE; i += 256) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         else if (mode == 2) {
            for (int c = wave; c < PIECE / 128; c += 4) glds16(p + c * 128 + 2 * lane, buf + c * 128, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int i = tid; i < PIECE; i += 256) s += buf[i];
         } else
            for (int i = tid; i < PIECE; i += 256) {
               const double v = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               s += v;
               __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
         if (s != 0.0) ++nz;
         publish(fa, r + 1);
         wait_ge(fb, r + 1);
         if (reread == 0) { for (int i = tid; i < PIECE; i += 256) bad += p[i] != 1.0; }
         else {
            __syncthreads();
            for (int c = wave; c < PIECE / 128; c += 4) glds16(p + c * 128 + 2 * lane, buf + c * 128, false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int i = tid; i < PIECE; i += 256) bad += buf[i] != 1.0;
            __syncthreads();
         }
      } else {
         wait_ge(fa, r + 1);
         for (int i = tid; i < PIECE; i += 256) __hip_atomic_store(p + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         publish(fb, r + 1);
      }
   }
   if (bad) atomicAdd(stale, bad);
   if (nz) atomicAdd(first_nonzero, nz);
}
int main(int argc, char** argv) {
   const int pairs = 128, rounds = argc > 1 ? atoi(argv[1]) : 64;
   double* X; int* flags; unsigned long long* cnt; int* xcd;
   const size_t n = (size_t)pairs * rounds * PIECE;
   CK(hipMalloc((void**)&X, n * sizeof(double)));
   CK(hipMalloc((void**)&flags, 2 * pairs * sizeof(int)));
   CK(hipMalloc((void**)&cnt, 2 * sizeof(unsigned long long)));
   CK(hipMalloc((void**)&xcd, 2 * pairs * sizeof(int)));
   for (int reread = 0; reread < 2; ++reread)
      for (int mode = 0; mode < 4; ++mode) {
         CK(hipMemset(X, 0, n * sizeof(double)));
         CK(hipMemset(flags, 0, 2 * pairs * sizeof(int)));
         CK(hipMemset(cnt, 0, 2 * sizeof(unsigned long long)));
         CK(hipDeviceSynchronize());
         hipLaunchKernelGGL(k_probe, dim3(2 * pairs), dim3(256), 0, 0, X, flags, rounds, mode, reread, cnt, cnt + 1, xcd);
         CK(hipDeviceSynchronize());
         unsigned long long h[2]; int hx[2 * pairs];
         CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
         CK(hipMemcpy(hx, xcd, sizeof(hx), hipMemcpyDeviceToHost));
         int cross = 0;
         for (int p = 0; p < pairs; ++p) cross += hx[2 * p] != hx[2 * p + 1];
         static const char* mn[4] = {"plain loads (control)", "agent-scope loads", "agent-scope LDS-DMA", "agent-scope load + store"};
         printf("first touch: %-26s second read: %-13s stale values %llu of %zu (pairs on two XCDs: %d of %d; non-zero first reads %llu)\n", mn[mode],
                reread ? "plain LDS-DMA" : "plain loads", h[0], n, cross, pairs, h[1]);
      }
   return 0;
}
