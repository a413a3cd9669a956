// The dense tails of the leaf blocks (the trailing part of every K_i with its border rows below: PardisoSchurSolver's partial factorisation,
// sLinsysLeafSchurSlave) factorised by ONE dependency-driven launch: the tasks of rootkernel.hip.h - UPD / DIAG / TRSM on 128 x 128 tiles,
// flags instead of launch boundaries - over all blocks of the batch at once.
//
// The launch-per-step driver (tail_factor) walks the tile columns: per column a deep update launch, the diagonal tiles on a side stream,
// a trsm launch - 34 + 44 launches per configs[1] factorisation, each of which drains before the next starts, the trsm launches
// memory-bound and alone on the device.  Here the SAME tasks with the same K ranges (TailPlan: groups of four tile columns, the tiles
// of a tile row side by side) form one list in the driver's order; a workgroup draws the next task of its XCD's share of the list and
// waits only for the tiles that task reads.  The default for batches of up to 16 blocks (a leaf handle, the sparse root's one-block engine),
// where the chain of launches per tile column is the factorisation; large batches keep the column launches, which are within 4 % of
// their slot-time bound.
//
// Where the tiles live.  A tile is accumulated with agent-scope loads and stores - and an agent-scope STORE leaves its line in the
// writer's L2 (tools/coh_probe: a plain load on that XCD afterwards returns the old value in 91 % of the cases, after agent-scope
// LOADS alone in none).  The finished L(i, j), read with plain loads by every later update, must therefore not lie where the tile was
// accumulated: the tail panel is assembled and accumulated in a scratch region behind the panels (BlkDesc::T_in) and trsm / the
// diagonal role write the final tile - once - into the panel itself (BlkDesc::T), where the solves and the Schur product find it.
//
// 512 resident workgroups walk eight lists (one per XCD, a task's tile row decides its list: the rows of L it reads stay in one L2).
// No cycle of waiting workgroups: the lists are subsequences of one topological order, each drawn in order; the first unfinished task of
// that order is either running or the next ticket of its list, and the workgroups of that list's XCD (workgroups 0..7 serve lists 0..7
// whatever XCD they landed on) hold only earlier - finished - tasks.  Waits are bounded all the same (poll limit -> error word).
#pragma once
#include "rootkernel.hip.h"

namespace pips {

struct TailLdlArgs {
   const TileTask* tasks;       // eight lists back to back; .blk = block | kind << 24 (ROOT_UPD / ROOT_TRSM / ROOT_DIAG), .pad = K range of an update
   int xoff[9];                 // list x = tasks[xoff[x] .. xoff[x + 1])
   const BlkDesc* blks;
   double *arena, *uarena, *winv, *dtail;
   const double* pref;
   const signed char* psign;
   const long long* psign_off;
   int* inertia;
   int* ctl;                    // [0..7] the lists' tickets, [8] pad, [9] error word
   int* flags;                  // per block (flag_off): prog[ntr][ntc] | rowdone[ntr] | dready[ntc]; copied from a template before every launch
   const long long* flag_off;
   long long poll_limit;
   int diag_blocked;
   int n_tasks;
   long long* trace;            // diagnostics (PIPS_HIP_TAIL_TRACE): per task the 100 MHz clock at the draw, after the waits, at the end
};

// The roles of rootkernel.hip.h's root_task for one block's tail - a copy with what a tail needs on top: U has a leading dimension of its own
// and no rows for the border tile rows, the trsm of a row commit in column order, the first wait that gave up is recorded.  (A copy, not a
// shared function: the root's chain is sensitive to the code around it - with the two sharing one body, out of line or as a template, the
// root measured 4 - 17 % slower, tools A/B on one box.)
__device__ __forceinline__ void tail_do(const RootArgs& a, RootShared& sh, int& s_ok, int t, int kind, int ti, int tj, int pad, int ldu, int* fail) {
   if (a.trace && threadIdx.x == 0) a.trace[3 * (long long)t] = wall_clock64();
   const int ntc = a.ntc, ld = a.ld;
   auto gave_up = [&]() { a.ctl[1] = 1; if (atomicCAS(fail, 0, 1) == 0) { fail[1] = kind; fail[2] = ti; fail[3] = tj; fail[4] = pad; } };
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int wr = wave & 1, wc = wave >> 1;
   bool ok = true;
   if (kind == ROOT_DIAG) {
      ok = root_wait_ge(a.prog + (long long)tj * ntc + tj, tj, a.poll_limit, &s_ok);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 1] = wall_clock64();
      if (ok) {
         bool done = false;
         if (a.diag_blocked) done = root_diag_blocked<1>(a, sh.d2, tj);
         if (!done) {
            __syncthreads();
            root_diag_role(a, sh.d, tj);
         }
      }
      else if (tid == 0) gave_up();
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
      const int k0 = pad & 0xffff, k1 = pad >> 16;
      ok = root_wait_ge(a.rowdone + ti, k1, a.poll_limit, &s_ok);
      if (ok && ti != tj) ok = root_wait_ge(a.rowdone + tj, k1, a.poll_limit, &s_ok);
      if (a.trace && tid == 0) a.trace[3 * (long long)t + 1] = wall_clock64();
      if (ok) {
         root_mainloop<false>(sh.g, a.R + (long long)ti * TILE + (long long)k0 * TILE * ld, ld, a.U + (long long)tj * TILE + (long long)k0 * TILE * ldu, ldu,
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
      } else if (tid == 0) gave_up();
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
      double* u0 = a.U + row0 + (long long)col0 * ldu;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int c = 0; c < 8; ++c) root_store(l0 + i * 16 + (long long)(c * 4) * ld, acc[i][c]);
      if (ti < ntc) {   // (the tile rows of the border below the square have no rows in U)
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) root_store(u0 + i * 16 + (long long)(c * 4) * ldu, acc[i][c] * dsc[c]);
      }
   } else if (tid == 0) gave_up();
   // rowdone[ti] = tj + 1 says "every L(ti, k), k <= tj, is final": the trsm of a row commit in column order.  In the root each one depends on
   // its predecessor anyway (its tile took an update with that column); under a tile envelope neighbouring columns of a row may have nothing
   // to do with each other, finish in any order - and the later store would take the flag back.  pad = the first tile column of the row
   // (its envelope); the wait is for a task earlier in the list, like every other one.
   if (tj > pad) (void)root_wait_ge(a.rowdone + ti, tj, a.poll_limit, &s_ok);
   root_publish(a.rowdone + ti, tj + 1);
   if (a.trace && tid == 0) a.trace[3 * (long long)t + 2] = wall_clock64();
}


// one task of the launch: the block's view of the arguments, then the roles
__device__ __noinline__ void tail_task(const TailLdlArgs& a, RootShared& sh, int& s_ok, int t) {
   const TileTask task = a.tasks[t];
   const int blk = task.blk & 0xffffff, kind = task.blk >> 24;
   const BlkDesc* bdp = a.blks + blk;
   RootArgs v{};
   v.ntc = bdp->ntc; v.ld = bdp->ldT;
   v.C = a.arena + bdp->T_in; v.R = a.arena + bdp->T; v.U = a.uarena + bdp->U;
   v.winv = a.winv + bdp->winv_off; v.dtail = a.dtail + bdp->dt_off;
   v.pref = a.pref + bdp->xw_off + bdp->n_head;
   v.psign = a.psign + a.psign_off[blk] + bdp->n_head;
   v.inertia = a.inertia + 3 * blk;
   v.ctl = a.ctl + 8;            // (the roles raise ctl[1])
   int* f = a.flags + a.flag_off[blk];
   v.prog = f; v.rowdone = f + (long long)bdp->ntr * bdp->ntc; v.dready = v.rowdone + bdp->ntr;
   v.blk = bdp; v.poll_limit = a.poll_limit; v.diag_blocked = a.diag_blocked; v.trace = a.trace; v.n_tasks = a.n_tasks;
   tail_do(v, sh, s_ok, t, kind, task.ti, task.tj, task.pad, bdp->m_pad, a.ctl + 10);   // (ctl[10] taken, [11..14] kind, ti, tj, K range, [15] block + 1)
   if (threadIdx.x == 0 && a.ctl[10] == 1 && a.ctl[15] == 0 && a.ctl[11] == kind && a.ctl[12] == task.ti && a.ctl[13] == task.tj) a.ctl[15] = blk + 1;
}

__global__ __launch_bounds__(512, 4) void k_tail_ldl(TailLdlArgs a) {
   __shared__ RootShared sh;
   __shared__ int s_t, s_ok;
   int mine = 0;
   if (threadIdx.x == 0) mine = blockIdx.x < 8 ? (int)blockIdx.x : (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20 /* XCC_ID[3:0] */) & 7u);
   for (;;) {
      if (threadIdx.x == 0) {
         int t = -1;
         for (int q = 0; q < 8 && t < 0; ++q) {
            const int x = (mine + q) & 7;
            t = root_draw(a.xoff[x], a.xoff[x + 1] - a.xoff[x], a.ctl + x);
         }
         s_t = t;
      }
      __syncthreads();
      const int t = s_t;
      if (t < 0) return;
      tail_task(a, sh, s_ok, t);
      __syncthreads();
   }
}

}  // namespace pips
