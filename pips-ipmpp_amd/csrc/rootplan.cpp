// Task list of the single-launch dense root (rootkernel.hip.h: k_root_ldl): a list schedule of the tile DAG of a left-looking tiled
// LDL^T, built once per Schur dimension on the host.  The reference hands the whole matrix to dsytrf (DeSymIndefSolver.C:56-118);
// here the factorisation is ntc (ntc + 1) / 2 tiles whose updates are cut into K ranges by THIS schedule: a discrete-event simulation
// of `workers` workgroup slots with a cost model picks, whenever a slot is free, the most urgent task that is ready
//   1. the diagonal tile of the chain column           (DIAG j: everything waits for it)
//   2. the triangular solves of finished columns       (TRSM (i, j), columns left to right, rows top down)
//   3. updates, nearest column first                   (UPD (i, j, k0, k1): with every column that is final by now - as deep as it gets -
//                                                       provided that finishes the tile, is at least qmin tile columns deep, or the tile
//                                                       is one the chain needs next)
// and the order in which tasks START is the ticket order of the launch.  Tiles far right of the chain are therefore touched rarely and
// deeply (K of a thousand and more: the update kernel's efficient regime), tiles next to it promptly; the chain of diagonal tiles gets
// a slot the moment it is ready because every task of the launch has the same footprint.  A task's dependencies have all FINISHED in
// the simulation before it starts, so they precede it in the list: the launch cannot deadlock, and where the model is off a workgroup
// polls a little.  tools/root_schedule_sim.py is the prototype this restates.
#include <algorithm>
#include <queue>
#include <vector>

#include "common.h"

namespace pips {

int build_root_plan(int ntc, const RootPlanParams& p, std::vector<int>& tasks, std::vector<int>& chain_tasks, double* makespan_us) {
   tasks.clear();
   chain_tasks.clear();
   if (ntc <= 0) return PIPS_OK;
   const int W = std::max(p.workers, 1);
   auto at = [ntc](int i, int j) { return (size_t)i * ntc + j; };
   std::vector<int> prog((size_t)ntc * ntc, 0), rowdone(ntc, 0);
   std::vector<char> busy((size_t)ntc * ntc, 0), queued((size_t)ntc * ntc, 0), trsm_done((size_t)ntc * ntc, 0), dready(ntc, 0);
   int chain = 0;
   struct Cand { int prio, key, j, i; };   // key: the column the task is ordered by
   auto worse = [](const Cand& a, const Cand& b) { return a.prio != b.prio ? a.prio > b.prio : (a.key != b.key ? a.key > b.key : (a.j != b.j ? a.j > b.j : a.i > b.i)); };
   typedef std::priority_queue<Cand, std::vector<Cand>, decltype(worse)> Heap;
   Heap heaps[3] = {Heap(worse), Heap(worse), Heap(worse)};   // 0 the chain's, 2 everything else (1 unused)
   struct Event { double t; long long seq; int kind, i, j, k1, cls; };
   auto later = [](const Event& a, const Event& b) { return a.t != b.t ? a.t > b.t : a.seq > b.seq; };
   std::priority_queue<Event, std::vector<Event>, decltype(later)> events(later);

   // what tile (i, j) could start now: -1 nothing, else the kind; k1 of an update
   auto startable = [&](int i, int j, int& k1) -> int {
      if (busy[at(i, j)]) return -1;
      if (prog[at(i, j)] == j) {
         if (i == j) return dready[j] ? -1 : (j == chain ? 2 : -1);
         return (dready[j] && !trsm_done[at(i, j)]) ? 1 : -1;
      }
      const int a = std::min({rowdone[i], rowdone[j], j}), q = a - prog[at(i, j)];
      if (q <= 0) return -1;
      const bool urgent = j <= chain + p.urgent && i <= j + p.urgent;
      if (a == j || q >= p.qmin || urgent) { k1 = std::min(a, prog[at(i, j)] + p.max_depth); return 0; }
      return -1;
   };
   // Two classes, two lists (each in the order its tasks start here):
   //  0 the chain's own: DIAG (j), TRSM (j + 1, j), the update that completes C(j + 1, j + 1) - strictly sequential, on the compute unit
   //    the launch keeps for them (chain_width > 1 adds the trsm / completing updates of the next diagonals: measured slower, the unit's
   //    second workgroup then multiplies beside the pivots);
   //  2 everything else.
   // (Also built and measured, S = 16 000, against 29.6 ms with these two: an "urgent" list for the tiles next to the chain that any
   //  arriving workgroup takes the moment its head is ready, 31.8 ms - in order, so a head that is not ready blocks the ones behind it; and
   //  a second unit of its own for the next diagonals, 29.5 - 30.1 ms - the waiting only moves one diagonal further out; and a head start
   //  of 128 - 400 places in the bulk list for the trsm / completing updates next to the chain: no change, the chain then waits for the
   //  tile's update before that one.  Both lists taken out of order, only what is ready (64 lanes look at 64 tasks each from a low-water
   //  mark on, compare-and-swap on a flag per task): 116 ms - the ready deep updates lie thousands of places behind the mark, out of any
   //  window's reach, while in-order draws let them start early and wait.  What the chain waits for where the chip is saturated is the
   //  trsm -> update pipeline of the tile rows it needs next: two bulk tasks per column and row, each some 90 us in line behind deep
   //  updates before it is drawn, so no row advances faster than a column per 250 us and the chain follows.  (A trsm that goes on, in
   //  the same workgroup, to apply its column to the row's next tile - one turn instead of two - applies to one tile row per column
   //  only: no change.)  Removed.)
   auto task_class = [&](int kind, int i, int j, int k1) {
      if (p.chain_slots > 0 && (kind == 2 || (kind == 1 && i - j <= p.chain_width) || (kind == 0 && k1 == j && i - j < p.chain_width))) return 0;
      return 2;
   };
   auto consider = [&](int i, int j) {
      if (queued[at(i, j)]) return;
      int k1 = 0;
      const int kind = startable(i, j, k1);
      if (kind < 0) return;
      queued[at(i, j)] = 1;
      // Updates are taken nearest column first - but the tiles next to the diagonal as if their column were `boost` columns nearer.  Nearest
      // first makes a column's tiles take their one deep update (everything that is final by then: 50 columns, 1.5 ms at S = 16 000) some
      // seven columns before the chain arrives - traced: the deep update of tile (60, 59) ended 50 us before DIAG 59 began, the seven
      // columns that had become final meanwhile followed as ONE task that could only start when the last of them was, and the chain waited
      // 200 us for it.  The few tiles the chain needs first in a column take their deep update earlier and then keep up column by column.
      const int jeff = kind == 0 && i - j <= p.boost_width ? std::max(j - p.boost, 0) : j;
      heaps[task_class(kind, i, j, kind == 0 ? k1 : j)].push(Cand{kind == 2 ? 0 : (kind == 1 ? 1 : 2), jeff, j, i});
   };

   double t = 0.0;
   long long seq = 0;
   int free_slots = W, free_chain = std::max(p.chain_slots, 0), diag_done = 0;
   consider(0, 0);
   while (diag_done < ntc) {
      for (int pass = 0; pass < 3; ++pass) {
         Heap& heap = heaps[pass];
         while (!heap.empty() && (pass == 0 ? free_chain > 0 : free_slots > 0)) {
            const Cand c = heap.top();
            heap.pop();
            queued[at(c.i, c.j)] = 0;
            int k1 = 0;
            const int kind = startable(c.i, c.j, k1);
            if (kind < 0) continue;
            if (kind != 0) k1 = c.j;
            const int k0 = prog[at(c.i, c.j)];
            const int cls = task_class(kind, c.i, c.j, k1);
            if (cls != pass) {   // an update that has changed its class since it was queued (it now completes its tile, the chain has moved): the other heap's
               queued[at(c.i, c.j)] = 1;
               heaps[cls].push(c);
               if (cls < pass) { pass = cls - 1; break; }   // (an earlier pass has work again)
               continue;
            }
            const bool alone = cls == 0 || W - free_slots < W / 2;   // fewer workgroups than compute units: a tile has its matrix pipe to itself
            const double dur = kind == 2 ? p.t_diag : (kind == 1 ? (alone ? p.t_trsm_alone : p.t_trsm) : p.t0 + (k1 - k0) * (alone ? p.t_step_alone : p.t_step));
            busy[at(c.i, c.j)] = 1;
            std::vector<int>& out = cls == 0 ? chain_tasks : tasks;
            out.push_back(kind);
            out.push_back(c.i);
            out.push_back(c.j);
            out.push_back(k0 | (k1 << 16));
            events.push(Event{t + dur, seq++, kind, c.i, c.j, k1, cls});
            if (cls == 0) --free_chain; else --free_slots;
         }
      }
      if (events.empty()) PIPS_FAIL(PIPS_ERR_STATE, "root plan: the schedule stalled at column %d of %d", chain, ntc);
      const Event e = events.top();
      events.pop();
      t = e.t;
      if (e.cls == 0) ++free_chain; else ++free_slots;
      busy[at(e.i, e.j)] = 0;
      if (e.kind == 2) {
         dready[e.j] = 1;
         ++diag_done;
         chain = e.j + 1;
         for (int i = e.j + 1; i < ntc; ++i) consider(i, e.j);
         for (int j = chain; j <= std::min(chain + p.urgent, ntc - 1); ++j)
            for (int i = j; i <= std::min(j + p.urgent, ntc - 1); ++i) consider(i, j);
      } else if (e.kind == 1) {
         trsm_done[at(e.i, e.j)] = 1;
         rowdone[e.i] = e.j + 1;
         for (int j = e.j + 1; j <= e.i; ++j) consider(e.i, j);       // tiles of row i
         for (int i = e.i; i < ntc; ++i) consider(i, e.i);            // tiles of column i (U(i, .) is their B operand)
      } else {
         prog[at(e.i, e.j)] = e.k1;
         consider(e.i, e.j);
      }
   }
   if (makespan_us) {
      while (!events.empty()) { t = std::max(t, events.top().t); events.pop(); }
      *makespan_us = t;
   }
   return PIPS_OK;
}

}  // namespace pips

// C entry for the tests (no device involved): tasks as (kind, i, j, k0 | k1 << 16) quadruples
extern "C" int pips_root_plan_build(int ntc, int workers, int qmin, int urgent, int chain_slots, int* out, long long cap, long long* n_tasks,
                                    long long* n_chain_tasks, double* makespan_us) {
   pips::RootPlanParams p;
   if (workers > 0) p.workers = workers;
   if (qmin > 0) p.qmin = qmin;
   if (urgent >= 0) p.urgent = urgent;
   if (chain_slots >= 0) p.chain_slots = chain_slots;
   std::vector<int> tasks, chain;
   const int rc = pips::build_root_plan(ntc, p, tasks, chain, makespan_us);
   if (rc) return rc;
   if (n_tasks) *n_tasks = (long long)tasks.size() / 4;
   if (n_chain_tasks) *n_chain_tasks = (long long)chain.size() / 4;
   if (out) {   // the bulk list, then the chain's list
      if ((long long)(tasks.size() + chain.size()) > cap) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_root_plan_build: %zu ints needed, %lld given", tasks.size() + chain.size(), cap);
      std::copy(tasks.begin(), tasks.end(), out);
      std::copy(chain.begin(), chain.end(), out + tasks.size());
   }
   return pips::PIPS_OK;
}
