// Dense root LDL^T (DeSymIndefSolver::matrixChanged, DeSymIndefSolver.C:56-118: dsytrf) as ONE dependency-driven launch.
//
// Every workgroup draws a ticket and takes the task with that number from a list the host built once per Schur dimension
// (rootplan.cpp: a list schedule of the tile DAG).  Three kinds of task on 128 x 128 tiles, left-looking:
//   UPD (i, j, k0, k1)  C(i,j) -= L(i, k0:k1) U(j, k0:k1)^T     the K range as deep as the schedule found it (every tile takes its K steps
//                                                               in ascending order, cut at the same places in every factorisation)
//   DIAG (j)            C(j,j) = L D L^T, Winv_j = D^-1 L^-1    (static pivot order: the fused path's Schur complement is quasi-definite)
//   TRSM (i, j)         L(i,j) = C(i,j) Winv_j^T, U(i,j) = L(i,j) D_j
// and three kinds of flag in device memory: prog[i][j] = tile columns applied to C(i,j); rowdone[i] = L(i, k), U(i, k) are final for
// k < rowdone[i]; dready[j].  A task waits only for tasks EARLIER in the list, and a workgroup holds its ticket before it waits, so the
// workgroup with the smallest unfinished ticket never waits for one that has not started: no deadlock whatever else shares the device;
// waits are bounded all the same (poll limit -> error word, like the solve sweeps).
//
// Coherence across the eight XCDs (each has its own L2, memory-side cache behind them).  L (d_R), U (d_U), Winv and d are WRITE-ONCE
// inside a launch and are read only after their flag: no L2 can hold an older copy of such a line (a launch begins with every L2
// invalidated), so the readers use plain loads / LDS-DMA and keep their L2 hits; the writer stores them with agent scope (written
// through) and raises the flag once its stores are acknowledged.  The accumulating tiles C live in a scratch array of their own (d_C) and are read and written with
// agent-scope accesses only, which go past the L2s.  No acquire fence anywhere: an invalidate would empty the L2 under the other 63
// workgroups of the XCD every microsecond.
#pragma once
#include "kernels.hip.h"

namespace pips {

struct RootArgs {
   const TileTask* tasks;   // blk = kind (0 UPD, 1 TRSM, 2 DIAG), ti, tj, pad = k0 | k1 << 16: the bulk list, then the chain's list
   int n_tasks, n_bulk, ntc, ld;   // n_tasks = all of them: [0, n_bulk) bulk, the rest the chain's
   double *C, *R, *U, *winv, *dtail;
   const double* pref;
   const signed char* psign;
   int* inertia;
   int* ctl;                // [0] ticket of the bulk list, [1] error word, [2] ticket of the chain's list, [3] the chain's compute unit (key + 1)
   int *prog, *rowdone, *dready;
   const BlkDesc* blk;      // thr_rel / repl_rel / repl_abs / m of the one block
   long long poll_limit;
   int diag_blocked;        // 1: root_diag_blocked, 0: the 128-barrier algorithm (A/B)
   long long* trace;        // diagnostics (PIPS_HIP_ROOT_TRACE): per ticket the 100 MHz clock at the draw, after the waits, at the end
};
constexpr int ROOT_UPD = 0, ROOT_TRSM = 1, ROOT_DIAG = 2;

__device__ __forceinline__ void glds16_sc1(const double* gptr_lane, double* lds_base) {   // LDS-DMA past the L2 (agent scope)
   const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_base;
   asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(gptr_lane), "s"(lds) : "memory", "m0");
}

// thread 0 polls until *f >= want, everybody learns the outcome
__device__ __forceinline__ bool root_wait_ge(const int* f, int want, long long poll_limit, int* sh_ok) {
   if (threadIdx.x == 0) {
      long long spins = 0;
      int ok = 1;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
         __builtin_amdgcn_s_sleep(2);
         if (++spins > poll_limit) { ok = 0; break; }
      }
      *sh_ok = ok;
   }
   __syncthreads();
   const bool ok = *sh_ok != 0;
   __syncthreads();
   return ok;
}
// Every result of a task is stored with root_store: agent scope, i.e. written through to memory.  A release fence instead would write
// back ALL dirty lines of the XCD's L2 - the output tiles of sixty other workgroups - before the flag may go up: traced as 20 - 50 us
// between the end of a diagonal tile and the start of the trsm that polls its flag, on a loaded chip.  With written-through stores the
// flag only has to wait for this workgroup's own stores to be acknowledged.
__device__ __forceinline__ void root_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void root_publish(int* flag, int value) {
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the MFMA main loop of tile_gemm_body (kernels.hip.h) on explicit panels: acc += A(0:128, 0:K) B(0:128, 0:K)^T, both column-major
template <bool A_SC1>
__device__ __forceinline__ void root_mainloop(GemmShared& sh, const double* Ap, long long lda, const double* Bp, long long ldb, int K,
                                              double (&acc)[4][8], int lane, int wave, int wr, int wc) {
   auto& As = sh.As;
   auto& Bs = sh.Bs;
   const int nst = K / KB;
   const double* Al = Ap + 2 * lane;
   const double* Bl = Bp + 2 * lane;
   auto issue = [&](int st, int buf) {
#pragma unroll
      for (int q = 0; q < KB / 8; ++q) {
         const int k = wave + 8 * q;
         if (A_SC1) glds16_sc1(Al + (long long)(st * KB + k) * lda, &As[buf][k * LDSW]);
         else glds16(Al + (long long)(st * KB + k) * lda, &As[buf][k * LDSW]);
         glds16(Bl + (long long)(st * KB + k) * ldb, &Bs[buf][k * LDSW]);
      }
   };
   if (nst > 0) issue(0, 0);
   const int rlane = wr * 64 + (lane & 15);
   const int clane = wc * 32 + (lane & 3);
   for (int st = 0; st < nst; ++st) {
      const int buf = st & 1;
      dma_wait();
      __syncthreads();
      const double* Ab = As[buf] + (lane >> 4) * LDSW + rlane;
      const double* Bb = Bs[buf] + (lane >> 4) * LDSW + clane;
#pragma unroll
      for (int q = 0; q < KB / 4; ++q) {
         if (q == KB / 16 && st + 1 < nst) issue(st + 1, buf ^ 1);
         double fr[4], fc[8];
#pragma unroll
         for (int i = 0; i < 4; ++i) fr[i] = Ab[(4 * q) * LDSW + i * 16];
#pragma unroll
         for (int c = 0; c < 8; ++c) fc[c] = Bb[(4 * q) * LDSW + c * 4];
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[i][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], fr[i], acc[i][c], 0, 0, 0);
      }
   }
}

// ---- DIAG: the algorithm of k_tile_diag (kernels.hip.h) on 512 threads: thread (tx, ty) of a 16 x 32 grid owns A(tx + 16 a, ty + 32 b)
// and the same entries of X = L^-1 for the 20 blocks (a, b) that reach the lower triangle (a >= 2 b): 40 doubles per thread, inside the
// 128 registers the update role leaves a wave.
struct RootDiagShared {
   double colk[2][TILE];
   double xrow[2][TILE];
   double dk[TILE];
   double prs[TILE];
   int sgn[TILE];
};
__device__ __forceinline__ constexpr int LI2(int a, int b) { return b == 0 ? a : (b == 1 ? 6 + a : (b == 2 ? 10 + a : 12 + a)); }

template <int KBLK>
__device__ __forceinline__ void root_diag_block(double (&A)[20], double (&X)[20], RootDiagShared& sh, const BlkDesc& bd, int tx, int ty, int tid,
                                                int gk0, int3& cnt) {
   constexpr int kb = KBLK >> 1;   // the 32-column block the 16 pivots of this call lie in
#pragma unroll 1
   for (int kt = 0; kt < 16; ++kt) {
      const int k = KBLK * 16 + kt, buf = kt & 1, kt32 = (KBLK & 1) * 16 + kt;
      if (ty == kt32) {
#pragma unroll
         for (int a = KBLK; a < 8; ++a) sh.colk[buf][tx + 16 * a] = A[LI2(a, kb)];
      }
      if (tx == kt) {
#pragma unroll
         for (int b = 0; b <= kb; ++b) sh.xrow[buf][ty + 32 * b] = X[LI2(KBLK, b)];
      }
      __syncthreads();
      const double piv = sh.colk[buf][k], pr = sh.prs[k];
      const int sg = sh.sgn[k];
      double ci[8], cj[4], xr[4];
#pragma unroll
      for (int a = KBLK; a < 8; ++a) ci[a] = sh.colk[buf][tx + 16 * a];
#pragma unroll
      for (int b = kb; b < 4; ++b) cj[b] = sh.colk[buf][ty + 32 * b];
#pragma unroll
      for (int b = 0; b <= kb; ++b) xr[b] = sh.xrow[buf][ty + 32 * b];
      bool pert;
      const double d = fix_pivot(piv, sg, pr, bd.thr_rel, bd.repl_rel, bd.repl_abs, pert);
      if (gk0 + k < bd.m) { cnt.z += pert; cnt.x += (!pert && d > 0); cnt.y += (!pert && !(d > 0)); }
      if (tid == 0) sh.dk[k] = d;
      const double dinv = pivot_rcp(d);
      double li[8];
      li[KBLK] = tx > kt ? ci[KBLK] * dinv : 0.0;
#pragma unroll
      for (int a = KBLK + 1; a < 8; ++a) li[a] = ci[a] * dinv;
      // A(i, j) -= l_ik a_jk for j > k
      {
         const double ajk = ty + 32 * kb > k ? cj[kb] : 0.0;
#pragma unroll
         for (int a = (KBLK > 2 * kb ? KBLK : 2 * kb); a < 8; ++a) A[LI2(a, kb)] -= li[a] * ajk;
      }
#pragma unroll
      for (int b = kb + 1; b < 4; ++b) {
#pragma unroll
         for (int a = 2 * b; a < 8; ++a) A[LI2(a, b)] -= li[a] * cj[b];
      }
      // X(i, c) -= l_ik X(k, c) for c < k; X(i, k) = -l_ik
#pragma unroll
      for (int b = 0; b < kb; ++b) {
#pragma unroll
         for (int a = KBLK; a < 8; ++a) X[LI2(a, b)] -= li[a] * xr[b];
      }
      {
         const double xkc = ty < kt32 ? xr[kb] : (ty == kt32 ? 1.0 : 0.0);
#pragma unroll
         for (int a = KBLK; a < 8; ++a) X[LI2(a, kb)] -= li[a] * xkc;
      }
   }
}

__device__ __forceinline__ void root_diag_role(const RootArgs& a, RootDiagShared& sh, int j) {
   __builtin_amdgcn_s_setprio(3);
   const BlkDesc bd = a.blk[0];
   const int tid = threadIdx.x, ld = a.ld;
   const int tx = tid & 15, ty = tid >> 4;
   const double* Cin = a.C + (long long)j * TILE + (long long)j * TILE * ld;
   if (tid < TILE) {
      sh.prs[tid] = a.pref[j * TILE + tid];
      sh.sgn[tid] = j * TILE + tid < bd.m ? (int)a.psign[j * TILE + tid] : 1;
   }
   double A[20], X[20];
#pragma unroll
   for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int aa = 2 * b; aa < 8; ++aa) {
         const int i = tx + 16 * aa, c = ty + 32 * b;
         A[LI2(aa, b)] = i >= c ? __hip_atomic_load(Cin + i + (long long)c * ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
         X[LI2(aa, b)] = 0.0;
      }
   __syncthreads();
   const int gk0 = j * TILE;
   int3 cnt = make_int3(0, 0, 0);
   root_diag_block<0>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<1>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<2>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<3>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<4>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<5>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<6>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   root_diag_block<7>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   const double* dk = sh.dk;
   __syncthreads();
   double* Lout = a.R + (long long)j * TILE + (long long)j * TILE * ld;
   double* W = a.winv + (long long)j * TILE * TILE;
#pragma unroll
   for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int aa = 0; aa < 8; ++aa) {
         const int i = tx + 16 * aa, c = ty + 32 * b;
         if (aa < 2 * b) { root_store(W + i + (long long)c * TILE, 0.0); continue; }
         const double av = A[LI2(aa < 2 * b ? 2 * b : aa, b)], xv = X[LI2(aa < 2 * b ? 2 * b : aa, b)];
         if (i > c) root_store(Lout + i + (long long)c * ld, av / dk[c]);
         else if (i == c) root_store(Lout + i + (long long)c * ld, dk[c]);
         const double x = i == c ? 1.0 : (i > c ? xv : 0.0);
         root_store(W + i + (long long)c * TILE, x / dk[i]);
      }
   if (tid < TILE) root_store(a.dtail + j * TILE + tid, dk[tid]);
   if (tid == 0) {
      if (cnt.x) atomicAdd(&a.inertia[0], cnt.x);
      if (cnt.y) atomicAdd(&a.inertia[1], cnt.y);
      if (cnt.z) atomicAdd(&a.inertia[2], cnt.z);
   }
   __builtin_amdgcn_s_setprio(0);
}


// ---- DIAG, blocked: the 128 x 128 tile as 4 x 4 sub-blocks of 32.  The 128 pivots of k_tile_diag are 128 workgroup barriers with all
// waves in every step (80 us alone, 150 - 190 us beside the update tiles of a loaded chip: the chain every column waits for, traced with
// PIPS_HIP_ROOT_TRACE).  Here ONE wave factorises a 32 x 32 diagonal sub-block without any barrier - lanes 0..31 hold the rows of A,
// lanes 32..63 the columns of X = L^-1, the pivot column travels through one LDS line per step - and everything else is 32 x 32 x 32
// products on the matrix pipe shared by the eight waves (wave w owns the output columns 4 w .. 4 w + 3 of every sub-block, two
// accumulators per sub-block), all of A living in accumulators until a sub-block is needed as an operand:
//   step b:  A(b,b) -> LDS;  wave 0: A(b,b) = L D L^T, W_bb = D^-1 L^-1;  L(i,b) = A(i,b) W_bb^T;  A(i,c) -= L(i,b) D_b L(c,b)^T;
//            row b of the inverse:  X(b,j) = -X(b,b) sum_{k=j}^{b-1} L(b,k) X(k,j),  whose k = j term Y(b,j) = L(b,j) X(j,j) was formed at
//            step j, while X(j,j) was at hand, and waited in the accumulators that A(b,j) left.
// The inverse is kept transposed (Xt(j,i) = X(i,j)^T, in the LDS slot of L(i,j), which nobody needs after row i): every product then has
// both operands in the [k][row] image the fragments are read from.  Eight LDS sub-blocks: six below the diagonal, two for exchange.
constexpr int DB = 32, DLD = 33;
struct RootDiagShared2 {
   double blk[8][DB * DLD];
   double park[10][64];    // wave 0's waiting accumulators while it factorises a diagonal sub-block (the factor loop wants the registers)
   double dk[TILE], dki[TILE], prs[TILE];
   int sgn[TILE];
   int cnt[4];             // accepted positive / negative / perturbed pivots of the tile
};
__device__ __forceinline__ constexpr int OB(int i, int j) { return j == 0 ? i - 1 : (j == 1 ? i + 1 : 5); }              // LDS slot of sub-block (i, j), i > j
__device__ __forceinline__ constexpr int AB(int i, int j) { return j == 0 ? i : (j == 1 ? 3 + i : (j == 2 ? 5 + i : 9)); }   // accumulator pair of (i, j), i >= j

// acc[h] += scale * sum_k Aop[k][16 h + (lane & 15)] * Bop[k][4 w + (lane & 3)] (* sc[k]): result element (row 16 h + (lane & 15), column 4 w + (lane >> 4))
template <bool SCALE>
__device__ __forceinline__ void blk_gemm(double (&acc)[2], const double* __restrict__ Aop, const double* __restrict__ Bop, const double* __restrict__ sc,
                                         double scale, int lane, int w) {
   const double* Ab = Aop + (lane >> 4) * DLD + (lane & 15);
   const double* Bb = Bop + (lane >> 4) * DLD + 4 * w + (lane & 3);
#pragma unroll
   for (int q = 0; q < DB / 4; ++q) {
      const double f0 = Ab[4 * q * DLD], f1 = Ab[4 * q * DLD + 16];
      double fc = Bb[4 * q * DLD] * scale;
      if (SCALE) fc *= sc[4 * q + (lane >> 4)];
      acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc, f0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc, f1, acc[1], 0, 0, 0);
   }
}
// an accumulator pair into an LDS sub-block, column-major, every column scaled by cs[column] if given
__device__ __forceinline__ void blk_dump(double* __restrict__ B, const double (&acc)[2], int lane, int w, const double* __restrict__ cs = nullptr) {
   const int col = 4 * w + (lane >> 4);
   const double f = cs ? cs[col] : 1.0;
   B[col * DLD + (lane & 15)] = acc[0] * f;
   B[col * DLD + 16 + (lane & 15)] = acc[1] * f;
}


// One wave factorises the 32 x 32 sub-block in S1 (column-major, lower triangle): lanes 0..31 hold the rows of A, lanes 32..63 the columns
// of X = L^-1 (identity at the start), 32 registers each, and the array SHIFTS by one place per step - entry 0 is always the one of the
// current pivot column, so the loop over the pivots is a loop (no unrolling, constant register indices).  Step k: every lane stores its
// entry 0 - rows: A(r, k) into column k of S1, which thereby becomes L D column by column; columns of X: X(k, c), final by now, into row k
// of S2 - and the FMAs read column k back as the broadcast line s (the LDS round trip is off the dependent path: the pivot itself
// travels by v_readlane); pivot rule of fix_pivot with everything that does not depend on the pivot prepared per lane beforehand and
// 1 / pivot started before the rule is evaluated; multiplier m (rows: A(r, k) / d below the pivot; columns of X: X(k, c) / d, which is 0
// right of the pivot and 1 / d on it); v[c - 1] = v[c] - m s_{k + c}.  No barrier: one wave.  Not inlined: one copy of the loop, with
// registers of its own (inlined four times the last copy kept v[] in scratch memory: 118 us instead of 25).
// Afterwards: S1 = W_bb = D^-1 X (column-major), S2 = X (row-major, zeros above the diagonal), dk / dki, the L tile's sub-block in memory.
__device__ __forceinline__ double root_readlane_f64(double x, int l) {
   return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// out = in + nm * s as ONE three-operand instruction: the compiler's choice for v[c - 1] = v[c] - m s is a register move plus the
// two-operand v_fmac_f64 - 30 moves per pivot on the one wave everything waits for
__device__ __forceinline__ double fma3(double nm, double sv, double in) {
   double out;
   asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(nm), "v"(sv), "v"(in));
   return out;
}
// pivots 8 P .. 8 P + 7 of root_diag_wave: 32 - 8 P places of v are alive
template <int P>
__device__ __forceinline__ void root_diag_wave_steps(double (&v)[DB], double* __restrict__ mine, const double* __restrict__ S1, bool lo, int r) {
   constexpr int NC = DB - 8 * P;   // v[0 .. NC)
#pragma unroll 1
   for (int k = 8 * P; k < 8 * P + 8; ++k) {
      mine[k * DLD] = v[0];
      const double piv = root_readlane_f64(v[0], k);
      const double dinv = pivot_rcp(piv);
      const double nm = (lo && r <= k) ? 0.0 : -(v[0] * dinv);
      // (the line beyond row 31 is whatever follows in LDS: it only reaches places of v that are spent)
      const double* s = S1 + k * DLD + k;
      double sa[8], sb[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) sa[c] = s[c];
      if (NC > 8) {
#pragma unroll
         for (int c = 0; c < 8; ++c) sb[c] = s[8 + c];
      }
#pragma unroll
      for (int c = 1; c < 8; ++c) v[c - 1] = fma3(nm, sa[c], v[c]);
      if (NC > 16) {
#pragma unroll
         for (int c = 0; c < 8; ++c) sa[c] = s[16 + c];
      }
      if (NC > 8) {
#pragma unroll
         for (int c = 8; c < 16; ++c) v[c - 1] = fma3(nm, sb[c - 8], v[c]);
      }
      if (NC > 24) {
#pragma unroll
         for (int c = 0; c < 8; ++c) sb[c] = s[24 + c];
      }
      if (NC > 16) {
#pragma unroll
         for (int c = 16; c < 24; ++c) v[c - 1] = fma3(nm, sa[c - 16], v[c]);
      }
      if (NC > 24) {
#pragma unroll
         for (int c = 24; c < 32; ++c) v[c - 1] = fma3(nm, sb[c - 24], v[c]);
      }
   }
}
// (WHO: one out-of-line instance per calling kernel - k_root_ldl 0, k_tail_ldl 1: with a single caller the compiler knows which shared
// object shp is; shared between the two kernels this function, on the root's chain, made S = 8000 17 % slower)
template <int WHO>
__device__ __noinline__ void root_diag_wave(RootDiagShared2* shp, int b, int rows_left, double thr_rel, double repl_rel, double repl_abs,
                                            double* __restrict__ Lout, int ld) {
   RootDiagShared2& sh = *shp;
   double* S1 = sh.blk[6];
   double* S2 = sh.blk[7];
   const int lane = threadIdx.x & 63;
   const bool lo = lane < 32;
   const int r = lane & 31;
   double v[DB];
#pragma unroll
   for (int c = 0; c < DB; ++c) {
      const double s1 = S1[c * DLD + r];
      v[c] = lo ? s1 : (c == r ? 1.0 : 0.0);
   }
   double* mine = lo ? S1 + r : S2 + r;   // where this lane's entry 0 goes at step k: + k DLD
   // The pivots are taken as they come: the rule of fix_pivot is applied to all 32 of them at once after the loop (lane r looks at
   // pivot r), and a sub-block with a pivot the rule rejects makes the whole tile start over in the careful kernel (root_diag_role).
   // On the dependent path of a step: two v_readlane, the reciprocal, one multiply.
   // (four loops of eight pivots: after 8 p pivots only 32 - 8 p places of v are alive - 31, 23, 15, 7 multiply-adds per pivot instead of 31)
   root_diag_wave_steps<0>(v, mine, S1, lo, r);
   root_diag_wave_steps<1>(v, mine, S1, lo, r);
   root_diag_wave_steps<2>(v, mine, S1, lo, r);
   root_diag_wave_steps<3>(v, mine, S1, lo, r);
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   // pivot r as it was taken (the diagonal of L D), the rule, the counts
   {
      const double d = S1[r * DLD + r], pref = sh.prs[32 * b + r];
      const int sg = sh.sgn[32 * b + r];
      const double sd = sg > 0 ? 1.0 : (sg < 0 ? -1.0 : (d < 0.0 ? -1.0 : 1.0));
      const bool ok = sd * d > thr_rel * pref;   // false for NaN
      const unsigned long long all = __ballot(1), good = __ballot(ok), pos = __ballot(ok && d > 0.0 && r < rows_left && lo), neg = __ballot(ok && !(d > 0.0) && r < rows_left && lo);
      if (lo) { sh.dk[32 * b + r] = d; sh.dki[32 * b + r] = pivot_rcp(d); }
      if (lane == 0) {
         if (good != all) sh.cnt[3] = 1;
         sh.cnt[0] += __popcll(pos);
         sh.cnt[1] += __popcll(neg);
      }
   }
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   // the L sub-block of the tile in memory: L(r, c) = (L D)(r, c) / d_c below the diagonal, d on it
#pragma unroll 4
   for (int c = lane >> 5; c < DB; c += 2) {
      const double x = S1[c * DLD + r];
      if (c <= r) root_store(Lout + r + (long long)c * ld, c == r ? sh.dk[32 * b + c] : x * sh.dki[32 * b + c]);
   }
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   // S1 = W_bb: W(r, c) = X(r, c) / d_r
   const double dir = sh.dki[32 * b + r];
#pragma unroll 4
   for (int c = lane >> 5; c < DB; c += 2) S1[c * DLD + r] = S2[r * DLD + c] * dir;
}

template <int WHO = 0>
__device__ __forceinline__ bool root_diag_blocked(const RootArgs& a, RootDiagShared2& sh, int j) {
   __builtin_amdgcn_s_setprio(3);
   const BlkDesc bd = a.blk[0];
   const int tid = threadIdx.x, ld = a.ld, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
   const double* Cin = a.C + (long long)j * TILE + (long long)j * TILE * ld;
   double* Lout = a.R + (long long)j * TILE + (long long)j * TILE * ld;
   double* W = a.winv + (long long)j * TILE * TILE;
   if (tid < TILE) {
      sh.prs[tid] = a.pref[j * TILE + tid];
      sh.sgn[tid] = j * TILE + tid < bd.m ? (int)a.psign[j * TILE + tid] : 1;
   }
   // this lane's two elements of every sub-block: rows er, er + 16, column ec.  Diagonal sub-blocks wait in accumulators, the ones below
   // the diagonal in their LDS slots (the trailing updates add to them there: every element has one owner)
   const int er = lane & 15, ec = 4 * w + (lane >> 4);
   double ad[4][2];
#pragma unroll
   for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int I = J; I < 4; ++I) {
         double t[2];
#pragma unroll
         for (int h = 0; h < 2; ++h)
            t[h] = __hip_atomic_load(Cin + (32 * I + 16 * h + er) + (long long)(32 * J + ec) * ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         if (I == J) { ad[I][0] = t[0]; ad[I][1] = t[1]; }
         else blk_dump(sh.blk[OB(I, J)], t, lane, w);
      }
   // zero the sub-blocks of Winv above the diagonal
   for (int idx = tid; idx < 6 * DB * DB; idx += 512) {
      const int q = idx >> 10, e = idx & 1023, I = q < 3 ? 0 : (q < 5 ? 1 : 2), J = q < 3 ? q + 1 : (q < 5 ? q - 1 : 3);
      root_store(W + (32 * I + (e & 31)) + (long long)(32 * J + (e >> 5)) * TILE, 0.0);
   }
   if (tid < 4) sh.cnt[tid] = 0;
   long long* stamp = a.trace ? a.trace + 3 * (long long)a.n_tasks + 32 * (long long)j : nullptr;   // diagnostics: phase clocks of this tile
   if (stamp && tid == 0) stamp[0] = wall_clock64();
   double* S1 = sh.blk[6];
   double* S2 = sh.blk[7];
   double ys[6][2];   // Y^T(i,j) = (L(i,j) X(j,j))^T, formed at step j, the start of S^T(i,j) of row i
#pragma unroll
   for (int b = 0; b < 4; ++b) {
      // ---- B0: the diagonal sub-block becomes an operand
      blk_dump(S1, ad[b], lane, w);
      __syncthreads();
      // ---- B1: wave 0 factorises it; meanwhile row b of the inverse: S^T(b,jj) = Y^T(b,jj) + sum_{jj < k < b} (L(b,k) X(k,jj))^T
      if (w == 0) {
         int np = 0;
#pragma unroll
         for (int c = b + 1; c < 4; ++c) { sh.park[np][lane] = ad[c][0]; sh.park[np + 1][lane] = ad[c][1]; np += 2; }
#pragma unroll
         for (int jj = 0; jj < b; ++jj)
#pragma unroll
            for (int i = b; i < 4; ++i) { sh.park[np][lane] = ys[OB(i, jj)][0]; sh.park[np + 1][lane] = ys[OB(i, jj)][1]; np += 2; }
         if (stamp && tid == 0) stamp[1 + 6 * b] = wall_clock64();
         root_diag_wave<WHO>(&sh, b, bd.m - j * TILE - 32 * b, bd.thr_rel, bd.repl_rel, bd.repl_abs, Lout + 32 * b + (long long)(32 * b) * ld, ld);
         if (stamp && tid == 0) stamp[2 + 6 * b] = wall_clock64();
         np = 0;
#pragma unroll
         for (int c = b + 1; c < 4; ++c) { ad[c][0] = sh.park[np][lane]; ad[c][1] = sh.park[np + 1][lane]; np += 2; }
#pragma unroll
         for (int jj = 0; jj < b; ++jj)
#pragma unroll
            for (int i = b; i < 4; ++i) { ys[OB(i, jj)][0] = sh.park[np][lane]; ys[OB(i, jj)][1] = sh.park[np + 1][lane]; np += 2; }
      }
      if (b >= 2) {
#pragma unroll
         for (int jj = 0; jj < b - 1; ++jj)
#pragma unroll
            for (int k = jj + 1; k < b; ++k) blk_gemm<false>(ys[OB(b, jj)], sh.blk[OB(k, jj)], sh.blk[OB(b, k)], nullptr, 1.0, lane, w);
      }
      __syncthreads();
      if (sh.cnt[3]) { __builtin_amdgcn_s_setprio(0); return false; }   // a pivot the rule rejects: the tile starts over in root_diag_role
      if (stamp && tid == 0) stamp[3 + 6 * b] = wall_clock64();
      // ---- B2: panel L(i,b) = A(i,b) W_bb^T into registers; the sums S^T(b,jj) become operands (in the slot of L(b,jj), which is spent)
      double pl[3][2];
#pragma unroll
      for (int i = b + 1; i < 4; ++i) {
         pl[i - 1][0] = pl[i - 1][1] = 0.0;
         blk_gemm<false>(pl[i - 1], sh.blk[OB(i, b)], S1, nullptr, 1.0, lane, w);
      }
      __syncthreads();
      if (stamp && tid == 0) stamp[4 + 6 * b] = wall_clock64();
      // ---- B3: the panel into LDS and into the L tile; S^T(b,jj) into LDS; W(b,b) into the Winv tile
#pragma unroll
      for (int i = b + 1; i < 4; ++i) {
         blk_dump(sh.blk[OB(i, b)], pl[i - 1], lane, w);
#pragma unroll
         for (int h = 0; h < 2; ++h) root_store(Lout + (32 * i + 16 * h + er) + (long long)(32 * b + ec) * ld, pl[i - 1][h]);
      }
#pragma unroll
      for (int jj = 0; jj < b; ++jj) blk_dump(sh.blk[OB(b, jj)], ys[OB(b, jj)], lane, w);
      for (int e = tid; e < DB * DB; e += 512) root_store(W + (32 * b + (e & 31)) + (long long)(32 * b + (e >> 5)) * TILE, S1[(e >> 5) * DLD + (e & 31)]);
      __syncthreads();
      if (stamp && tid == 0) stamp[5 + 6 * b] = wall_clock64();
      // ---- B4: trailing update; Y^T(i,b) for the rows below; W(b,jj)^T = -S^T(b,jj) W_bb^T
#pragma unroll
      for (int c = b + 1; c < 4; ++c)
#pragma unroll
         for (int i = c; i < 4; ++i) {
            if (i == c) blk_gemm<true>(ad[c], sh.blk[OB(i, b)], sh.blk[OB(c, b)], sh.dk + 32 * b, -1.0, lane, w);
            else {
               double* T = sh.blk[OB(i, c)] + ec * DLD + er;
               double t[2] = {T[0], T[16]};
               blk_gemm<true>(t, sh.blk[OB(i, b)], sh.blk[OB(c, b)], sh.dk + 32 * b, -1.0, lane, w);
               T[0] = t[0]; T[16] = t[1];
            }
         }
#pragma unroll
      for (int i = b + 1; i < 4; ++i) {
         ys[OB(i, b)][0] = ys[OB(i, b)][1] = 0.0;
         blk_gemm<false>(ys[OB(i, b)], S2, sh.blk[OB(i, b)], nullptr, 1.0, lane, w);
      }
      double wt[3][2];
#pragma unroll
      for (int jj = 0; jj < b; ++jj) {
         wt[jj][0] = wt[jj][1] = 0.0;
         blk_gemm<false>(wt[jj], sh.blk[OB(b, jj)], S1, nullptr, -1.0, lane, w);
      }
      __syncthreads();
      if (stamp && tid == 0) stamp[6 + 6 * b] = wall_clock64();
      // ---- B5: row b of the inverse into the Winv tile and, transposed and scaled (Xt = d_r W^T), into LDS for the rows below
#pragma unroll
      for (int jj = 0; jj < b; ++jj) {
         // element (row c = 16 h + er, column r = ec) of W(b,jj)^T
#pragma unroll
         for (int h = 0; h < 2; ++h) root_store(W + (32 * b + ec) + (long long)(32 * jj + 16 * h + er) * TILE, wt[jj][h]);
         if (b < 3) blk_dump(sh.blk[OB(b, jj)], wt[jj], lane, w, sh.dk + 32 * b);
      }
      // (the next step's first barrier separates these writes from their readers)
   }
   __syncthreads();
   if (stamp && tid == 0) stamp[25] = wall_clock64();
   if (tid < TILE) root_store(a.dtail + j * TILE + tid, sh.dk[tid]);
   if (tid == 0) {   // (wave 0 counted)
      if (sh.cnt[0]) atomicAdd(&a.inertia[0], sh.cnt[0]);
      if (sh.cnt[1]) atomicAdd(&a.inertia[1], sh.cnt[1]);
   }
   __builtin_amdgcn_s_setprio(0);
   return true;
}

union RootShared {
   GemmShared g;
   RootDiagShared d;
   RootDiagShared2 d2;
};

// one task of the launch (index t into the joint list); every thread of the workgroup calls it with the same t
__device__ __noinline__ void root_task(const RootArgs& a, RootShared& sh, int& s_ok, int t) {
   if (a.trace && threadIdx.x == 0) a.trace[3 * (long long)t] = wall_clock64();
   const TileTask task = a.tasks[t];
   const int kind = task.blk, ti = task.ti, tj = task.tj, ntc = a.ntc, ld = a.ld;
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int wr = wave & 1, wc = wave >> 1;
   bool ok = true;
   if (kind == ROOT_DIAG) {
      ok = root_wait_ge(a.prog + (long long)tj * ntc + tj, tj, a.poll_limit, &s_ok);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 1] = wall_clock64();
      if (ok) {
         bool done = false;
         if (a.diag_blocked) done = root_diag_blocked(a, sh.d2, tj);
         if (!done) {
            __syncthreads();
            root_diag_role(a, sh.d, tj);
         }
      }
      else if (tid == 0) a.ctl[1] = 1;
      root_publish(a.dready + tj, 1);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 2] = wall_clock64();
      return;
   }
   double acc[4][8];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[i][c] = 0.0;
   if (kind == ROOT_UPD) {
      const int k0 = task.pad & 0xffff, k1 = task.pad >> 16;
      ok = root_wait_ge(a.rowdone + ti, k1, a.poll_limit, &s_ok);
      if (ok && ti != tj) ok = root_wait_ge(a.rowdone + tj, k1, a.poll_limit, &s_ok);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 1] = wall_clock64();
      if (ok) {
         root_mainloop<false>(sh.g, a.R + (long long)ti * TILE + (long long)k0 * TILE * ld, ld, a.U + (long long)tj * TILE + (long long)k0 * TILE * ld, ld,
                              (k1 - k0) * TILE, acc, lane, wave, wr, wc);
         ok = root_wait_ge(a.prog + (long long)ti * ntc + tj, k0, a.poll_limit, &s_ok);   // the update before this one has stored the tile
      }
      if (ok) {
         double* c0 = a.C + (long long)ti * TILE + wr * 64 + (lane & 15) + ((long long)tj * TILE + wc * 32 + (lane >> 4)) * ld;
#pragma unroll
         for (int h = 0; h < 4; ++h) {
            double cv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
               for (int c = 0; c < 2; ++c) cv[i][c] = __hip_atomic_load(c0 + i * 16 + (long long)((2 * h + c) * 4) * ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
               for (int c = 0; c < 2; ++c)
                  __hip_atomic_store(c0 + i * 16 + (long long)((2 * h + c) * 4) * ld, cv[i][c] - acc[i][2 * h + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
         }
      } else if (tid == 0) a.ctl[1] = 1;
      root_publish(a.prog + (long long)ti * ntc + tj, k1);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 2] = wall_clock64();
      return;
   }
   // TRSM
   ok = root_wait_ge(a.prog + (long long)ti * ntc + tj, tj, a.poll_limit, &s_ok);
   if (ok) ok = root_wait_ge(a.dready + tj, 1, a.poll_limit, &s_ok);
   if (a.trace && tid == 0) a.trace[3 * (long long)t + 1] = wall_clock64();
   if (ok) {
      if (ti == tj + 1) __builtin_amdgcn_s_setprio(2);   // the tile the next diagonal tile waits for
      root_mainloop<true>(sh.g, a.C + (long long)ti * TILE + (long long)tj * TILE * ld, ld, a.winv + (long long)tj * TILE * TILE, TILE, TILE, acc, lane, wave,
                          wr, wc);
      const int col0 = tj * TILE + wc * 32 + (lane >> 4), row0 = ti * TILE + wr * 64 + (lane & 15);
      double dsc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) dsc[c] = a.dtail[col0 + 4 * c];
      double* l0 = a.R + row0 + (long long)col0 * ld;
      double* u0 = a.U + row0 + (long long)col0 * ld;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int c = 0; c < 8; ++c) root_store(l0 + i * 16 + (long long)(c * 4) * ld, acc[i][c]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int c = 0; c < 8; ++c) root_store(u0 + i * 16 + (long long)(c * 4) * ld, acc[i][c] * dsc[c]);
   } else if (tid == 0) a.ctl[1] = 1;
   root_publish(a.rowdone + ti, tj + 1);
   if (a.trace && tid == 0) a.trace[3 * (long long)t + 2] = wall_clock64();
}

// the next task of a list (the workgroup waits inside root_task for what the task needs), -1 when the list is empty
__device__ __forceinline__ int root_draw(int base, int n, int* ticket) {
   if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) return -1;
   const int c = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   return c < n ? base + c : -1;
}

__global__ __launch_bounds__(512, 4) void k_root_ldl(RootArgs a) {
   __shared__ RootShared sh;
   __shared__ int s_t, s_ok, s_mine;
   // Two lists, each in the order of the schedule (rootplan.cpp).  The chain of the diagonal tiles - DIAG j -> TRSM (j + 1, j) -> last
   // update of C(j + 1, j + 1) -> DIAG j + 1 - is strictly sequential and every column waits for it; beside the matrix-pipe waves of an
   // update tile its 128 dependent pivots take 2.4 x as long (traced).  So the compute unit the launch's first workgroup lands on belongs
   // to the chain: a workgroup that starts there STAYS and draws from the chain's list until it is empty (one of the two works, the other
   // waits for its turn: nothing competes for the unit's issue slots); every other workgroup takes one task of the bulk list and leaves;
   // whoever finds its own list empty helps with the other.
   // No cycle of waiting workgroups: both lists are subsequences of ONE topological order and each is drawn in order.  Let x be the first
   // task of that order which is not finished; everything it waits for is.  If x is drawn it runs.  If not, it is the next ticket of its
   // list and every drawn task of that list is earlier, hence finished: on the chain's list its two resident workgroups are free to draw
   // it (they do not depend on the dispatcher sending more workgroups - the first version did, and hung once the chain unit's XCD had
   // used up its share of the grid); on the bulk list the slots those tasks held are free.
   if (threadIdx.x == 0) {
      int mine = 0;
      if (a.n_tasks > a.n_bulk) {
         const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_ID */), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20 /* XCC_ID[3:0] */);
         const int key = 1 + (int)(((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));   // XCC, shader engine / array, compute unit
         int seen = __hip_atomic_load(a.ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         if (seen == 0) {
            int expected = 0;
            seen = __hip_atomic_compare_exchange_strong(a.ctl + 3, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? key : expected;
         }
         mine = seen == key;
      }
      s_mine = mine;
   }
   __syncthreads();
   const bool mine = s_mine != 0;
   const int n_chain = a.n_tasks - a.n_bulk;
   for (;;) {
      if (threadIdx.x == 0) {
         int t = mine ? root_draw(a.n_bulk, n_chain, a.ctl + 2) : root_draw(0, a.n_bulk, a.ctl);
         if (t < 0) t = mine ? root_draw(0, a.n_bulk, a.ctl) : root_draw(a.n_bulk, n_chain, a.ctl + 2);
         s_t = t;
      }
      __syncthreads();
      const int t = s_t;
      if (t < 0) return;
      root_task(a, sh, s_ok, t);
      if (!mine) return;
      __syncthreads();
   }
}

}  // namespace pips
