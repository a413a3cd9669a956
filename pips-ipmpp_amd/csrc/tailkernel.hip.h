// The dense tails of the leaf blocks (the trailing part of every K_i with its border rows below: PardisoSchurSolver's partial factorisation,
// sLinsysLeafSchurSlave) factorised by ONE dependency-driven launch: the tasks of rootkernel.hip.h - UPD / DIAG / TRSM on 128 x 128 tiles,
// flags instead of launch boundaries - over all blocks of the batch at once.
//
// The launch-per-step driver (tail_factor) walks the tile columns: per column a deep update launch, the diagonal tiles on a side stream,
// a trsm launch - 34 + 44 launches per configs[1] factorisation, each of which drains before the next starts, the trsm launches
// memory-bound and alone on the device.  Here the SAME tasks with the same K ranges (TailPlan: groups of four tile columns, the tiles
// of a tile row side by side) form one list in the driver's order; a workgroup draws the next task of its XCD's share of the list and
// waits only for the tiles that task reads.
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
      const TileTask task = a.tasks[t];
      const int blk = task.blk & 0xffffff, kind = task.blk >> 24;
      const BlkDesc* bdp = a.blks + blk;
      RootArgs v{};
      v.ntc = bdp->ntc; v.ld = bdp->ldT; v.ldu = bdp->m_pad;
      v.C = a.arena + bdp->T_in; v.R = a.arena + bdp->T; v.U = a.uarena + bdp->U;
      v.winv = a.winv + bdp->winv_off; v.dtail = a.dtail + bdp->dt_off;
      v.pref = a.pref + bdp->xw_off + bdp->n_head;
      v.psign = a.psign + a.psign_off[blk] + bdp->n_head;
      v.inertia = a.inertia + 3 * blk;
      v.ctl = a.ctl + 8;            // (root_do raises ctl[1])
      int* f = a.flags + a.flag_off[blk];
      v.prog = f; v.rowdone = f + (long long)bdp->ntr * bdp->ntc; v.dready = v.rowdone + bdp->ntr;
      v.fail = a.ctl + 10;          // ([10] taken, [11..14] kind, ti, tj, K range, [15] block)
      v.blk = bdp; v.poll_limit = a.poll_limit; v.diag_blocked = a.diag_blocked; v.trace = a.trace; v.n_tasks = a.n_tasks;
      root_do(v, sh, s_ok, kind, task.ti, task.tj, task.pad, t);
      if (threadIdx.x == 0 && a.ctl[10] == 1 && a.ctl[15] == 0 && a.ctl[11] == kind && a.ctl[12] == task.ti && a.ctl[13] == task.tj) a.ctl[15] = blk + 1;
      __syncthreads();
   }
}

}  // namespace pips
