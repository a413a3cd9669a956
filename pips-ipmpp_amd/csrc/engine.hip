// Device engine of the MI355X KKT backend: batched numeric factorisation / solves for all leaf blocks of one GPU,
// the dense root solver, and the C ABI declared in include/pips_hip.h.
//
// Reference call sites replaced (see include/pips_hip.h for the per-entry citations):
//   DistributedLeafLinearSystem::factor2 -> solver->matrixChanged()          (DistributedLeafLinearSystem.C:74-86)
//   addTermToSchurComplBlocked + addBiTLeftKiBiRightToResBlockedParallelSolvers (DistributedLinearSystem.C:766-1175)
//   DeSymIndefSolver::matrixChanged/solve                                      (DeSymIndefSolver.C:56-129)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "kernels.hip.h"
#include "rootkernel.hip.h"
#include "tailkernel.hip.h"
#include "pips_hip.h"

namespace pips {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
std::atomic<long long> g_host_waits{0};
// per call site: a small open-addressed table keyed by (file, line) - the literals' addresses are stable, a site is entered once
struct WaitSite { std::atomic<const char*> file{nullptr}; std::atomic<int> line{0}; std::atomic<long long> n{0}; };
static WaitSite g_wait_sites[128];
void note_host_wait(const char* file, int line) {
   g_host_waits.fetch_add(1, std::memory_order_relaxed);
   size_t h = ((size_t)(uintptr_t)file / 8 + (size_t)line * 31) % 128;
   for (int probe = 0; probe < 128; ++probe, h = (h + 1) % 128) {
      WaitSite& w = g_wait_sites[h];
      const char* f = w.file.load(std::memory_order_acquire);
      if (!f) {
         const char* expect = nullptr;
         if (w.file.compare_exchange_strong(expect, file)) { w.line.store(line); f = file; }
         else f = expect;
      }
      if (f == file) {
         while (w.line.load() == 0) {}   // (the entering thread stores the line right after the file)
         if (w.line.load() == line) { w.n.fetch_add(1, std::memory_order_relaxed); return; }
      }
   }
}
const char* last_error() { return g_last_error.c_str(); }

#define HIP_TRY(expr)                                                                              \
   do {                                                                                            \
      hipError_t _e = (expr);                                                                      \
      if (_e != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
   } while (0)

template <class T>
static int dev_upload(T** dptr, const std::vector<T>& h, hipStream_t) {
   *dptr = nullptr;
   const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
   HIP_TRY(hipMalloc((void**)dptr, bytes));
   if (!h.empty()) HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
   return PIPS_OK;
}

// environment switches: value of an integer switch, dflt when it is not set
static int env_int(const char* name, int dflt) {
   const char* v = getenv(name);
   return v ? atoi(v) : dflt;
}

static inline int grid_for(long long n, int block, int cap = 4096) {
   long long g = (n + block - 1) / block;
   if (g < 1) g = 1;
   if (g > cap) g = cap;
   return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------
// tiled dense LDL^T driver shared by the leaf tails and the dense root
// ---------------------------------------------------------------------------------------------------------------
struct TaskList { long long off = 0; int cnt = 0; };

struct TailPlan {
   int ntc_max = 0;
   std::vector<TaskList> upd, upd_diag, diag, trsm, fwd, bwd, trail, trail_next;
   TaskList schur;
   TileTask* d_tasks = nullptr;
   std::vector<TileTask> h_tasks;   // host copy (the single-launch factorisation re-lists them: TailSingle::build)

   // panel == 0: pure left-looking (tile column j is updated once, with everything to its left: minimal traffic on C,
   //   one task per tile of the column - right when many blocks share every launch).
   // panel == P > 0: blocked right-looking with panels of P tile columns (the dense root, a single block: a left-looking
   //   column launch has at most ntr workgroups with a K as deep as the matrix; here the column update only reaches back
   //   to the start of its panel and each finished panel is applied to the whole trailing matrix in one launch of
   //   (ntr - p1)^2 / 2 tiles with K = P * TILE).  The K range rides in TileTask::pad = k0 | k1 << 16 (tile columns).
   // split_diag (left-looking batches): the update of the diagonal tile of column j gets a task list of its own (upd_diag),
   // so that the driver can factorise that tile on a side stream while the rest of the column is still being updated.
   // first: per block the tile-row envelope of its tail (BlockSym::tile_first); tiles left of it hold structural zeros and
   // get no task, update depths start at the envelope (a banded tail costs band^2 per column instead of column^2)
   int build(const std::vector<BlkDesc>& blks, int panel = 0, bool lookahead = false, bool split_diag = false,
             const std::vector<const std::vector<int>*>* first = nullptr, bool diag_ahead = false, int pair2 = 1) {
      std::vector<TileTask> all;
      ntc_max = 0;
      for (auto& b : blks) ntc_max = std::max(ntc_max, b.ntc);
      upd.assign(ntc_max, {});
      upd_diag.assign(ntc_max, {});
      diag.assign(ntc_max, {});
      trsm.assign(ntc_max, {});
      fwd.assign(ntc_max, {});
      bwd.assign(ntc_max, {});
      trail.assign(ntc_max, {});
      trail_next.assign(ntc_max, {});
      const int nblk = (int)blks.size();
      auto begin = [&](TaskList& l) { l.off = (long long)all.size(); };
      auto end = [&](TaskList& l) { l.cnt = (int)((long long)all.size() - l.off); };
      for (int j = 0; j < ntc_max; ++j) {
         const int p0 = panel > 0 ? j / panel * panel : 0;   // first tile column of j's panel
         auto fst = [&](int b, int t) { return first ? (*(*first)[b])[t] : 0; };
         // diag_ahead: the diagonal tile of column j+1 takes its update with the columns < j inside the launch of column j
         // (same depth as the other tiles of that launch); the list of its own only holds the last step, with column j-1.
         // The side stream then has a K = TILE product and the tile factorisation on the path to trsm(j), not a product as
         // deep as the matrix done by one workgroup per block.
         begin(upd_diag[j]);
         if (j > p0 && split_diag)
            for (int b = 0; b < nblk; ++b)
               if (blks[b].ntc > j) {
                  const int k0 = std::max(p0, fst(b, j));
                  if (k0 < j) all.push_back({b, j, j, (diag_ahead ? std::max(k0, j - 1) : k0) | (j << 16)});
               }
         end(upd_diag[j]);
         begin(upd[j]);
         // pair2 = P (left-looking batches with the diagonal tiles ahead): the tile columns of a group g .. g + P - 1 share the launch of column
         // g for everything left of the group - the tiles of a tile row read the same rows of L, side by side in the task list (one XCD, one
         // after the other) - and column g + q takes the rest, the q columns of its own group, in the launch of its own (q tiles deep).  The
         // diagonal tiles up to the next group's first ride along the same way, so that the short launches hold no deep task.
         const int P = (pair2 > 1 && panel == 0 && split_diag && diag_ahead) ? pair2 : 1;
         const int g = j / P * P, q = j - g;
         if (j > p0 && split_diag && diag_ahead)
            for (int b = 0; b < nblk; ++b)
               if (blks[b].ntc > j + 1) {
                  const int k0 = std::max(p0, fst(b, j + 1)), lo = q > 0 ? std::max(k0, g) : k0;
                  if (lo < j) all.push_back({b, j + 1, j + 1, lo | (j << 16)});
               }
         if (P > 1 && q == 0 && j > p0)
            for (int b = 0; b < nblk; ++b)
               for (int d = j + 2; d <= j + P && d < blks[b].ntc; ++d) {
                  const int k0 = std::max(p0, fst(b, d));
                  if (k0 < j) all.push_back({b, d, d, k0 | (j << 16)});
               }
         if (j > p0)
            for (int b = 0; b < nblk; ++b)
               if (blks[b].ntc > j)
                  for (int ti = split_diag ? j + 1 : j; ti < blks[b].ntr; ++ti) {
                     if (j >= fst(b, ti)) {                                         // (else outside the envelope: stays zero)
                        const int k0 = std::max({p0, fst(b, ti), fst(b, j)}), lo = q > 0 ? std::max(k0, g) : k0;
                        if (lo < j) all.push_back({b, ti, j, lo | (j << 16)});
                     }
                     if (P > 1 && q == 0)
                        for (int c = j + 1; c < j + P && c < blks[b].ntc && c < ti; ++c)
                           if (c >= fst(b, ti)) {
                              const int k0 = std::max({p0, fst(b, ti), fst(b, c)});
                              if (k0 < j) all.push_back({b, ti, c, k0 | (j << 16)});
                           }
                  }
         end(upd[j]);
         begin(diag[j]);
         for (int b = 0; b < nblk; ++b)
            if (blks[b].ntc > j) all.push_back({b, j, j, 0});
         end(diag[j]);
         begin(trsm[j]);
         for (int b = 0; b < nblk; ++b)
            if (blks[b].ntc > j)
               for (int ti = j + 1; ti < blks[b].ntr; ++ti)
                  if (j >= fst(b, ti)) all.push_back({b, ti, j, 0});
         end(trsm[j]);
         begin(fwd[j]);
         for (int b = 0; b < nblk; ++b)
            if (blks[b].ntc > j)
               for (int ti = j; ti < blks[b].ntc; ++ti)
                  if (ti == j || (j >= 1 && j - 1 >= fst(b, ti))) all.push_back({b, ti, j, 0});   // L(ti, j-1) inside the envelope
         end(fwd[j]);
         begin(bwd[j]);
         for (int b = 0; b < nblk; ++b)
            if (blks[b].ntc > j)
               for (int tj = 0; tj <= j; ++tj)
                  if (tj == j || (j + 1 < blks[b].ntc && tj >= fst(b, j + 1))) all.push_back({b, tj, j, 0});   // L(j+1, tj)
         end(bwd[j]);
         // panel [p0, j] complete: apply it to every tile right of it.  The next tile column is listed separately
         // (trail_next) so that the driver can finish it first and overlap the next diagonal tile with the rest.
         if (panel > 0 && (j + 1) % panel == 0) {
            begin(trail_next[j]);
            if (lookahead)
               for (int b = 0; b < nblk; ++b)
                  if (j + 1 < blks[b].ntc)
                     for (int ti = j + 1; ti < blks[b].ntr; ++ti) all.push_back({b, ti, j + 1, p0 | ((j + 1) << 16)});
            end(trail_next[j]);
            begin(trail[j]);
            for (int b = 0; b < nblk; ++b)
               for (int tj = j + (lookahead ? 2 : 1); tj < blks[b].ntc; ++tj)
                  for (int ti = tj; ti < blks[b].ntr; ++ti) all.push_back({b, ti, tj, p0 | ((j + 1) << 16)});
            end(trail[j]);
         }
      }
      begin(schur);
      for (int b = 0; b < nblk; ++b)
         if (blks[b].ntc > 0 && blks[b].nb > 0) {
            const int nt = blks[b].nb_pad / TILE;
            for (int ti = 0; ti < nt; ++ti)
               for (int tj = 0; tj <= ti; ++tj) all.push_back({b, ti, tj, 0});
         }
      end(schur);
      h_tasks = all;
      return dev_upload(&d_tasks, all, nullptr);
   }
   void release() {
      if (d_tasks) (void)hipFree(d_tasks);
      d_tasks = nullptr;
   }
};

// Single-launch tail sweeps (kernels.hip.h k_tail_rows_fwd / _bwd): task list (block, tile row) sorted by row, ticket and flag
// storage.  PIPS_HIP_SWEEP_LAUNCHES=1 keeps the launch-per-tile-column kernels.
struct SweepRt {
   TileTask* d_tasks = nullptr;
   int n_tasks = 0;
   int* d_ints = nullptr;          // [0 .. 2 NRHS) ticket / finished per right-hand side, then the error word, then the flags
   long long n_flags = 0;
   long long* d_flag_off = nullptr;
   int* d_tfirst = nullptr;
   long long* d_tfirst_off = nullptr;
   int epoch = 0;
   bool enabled = false;
   long long poll_limit = SWEEP_POLL_LIMIT;
   // A wait that gave up poisoned its output with NaN and raised the error word; the host learns of it at its next synchronisation
   // point (batch sync, inertia queries, host-side solves): the word is read, cleared and turned into an error return.
   int take_error(const char* who) {
      if (!d_ints) return PIPS_OK;
      int w = 0;
      HIP_TRY(hipMemcpy(&w, d_ints + 2 * SWEEP_NRHS_MAX, sizeof(int), hipMemcpyDeviceToHost));
      if (!w) return PIPS_OK;
      HIP_TRY(hipMemset(d_ints + 2 * SWEEP_NRHS_MAX, 0, sizeof(int)));
      PIPS_FAIL(PIPS_ERR_HIP, "%s: a single-launch solve sweep gave up waiting for a piece of the solution after %lld polls (its output is NaN); "
                              "another process holding the device for seconds can cause that - PIPS_HIP_SWEEP_LAUNCHES=1 selects the launch-per-column sweeps",
                who, poll_limit);
   }
   int build(const std::vector<BlkDesc>& blks, const std::vector<const std::vector<int>*>* first) {
      release();
      if (const char* pl = getenv("PIPS_HIP_SWEEP_POLL_LIMIT")) poll_limit = atoll(pl);
      const int nblk = (int)blks.size();
      int ntc_max = 0;
      for (auto& b : blks) ntc_max = std::max(ntc_max, b.ntc);
      std::vector<TileTask> tasks;
      for (int i = 0; i < ntc_max; ++i)
         for (int b = 0; b < nblk; ++b)
            if (i < blks[b].ntc) tasks.push_back({b, i, 0, 0});
      n_tasks = (int)tasks.size();
      std::vector<long long> foff(nblk), tfoff(nblk);
      std::vector<int> tf;
      long long nf = 0;
      for (int b = 0; b < nblk; ++b) {
         foff[b] = nf;
         nf += blks[b].ntc;
         tfoff[b] = (long long)tf.size();
         if (first)   // tail rows and, behind them, the border tile rows (border-backward sweep)
            for (int i = 0; i < blks[b].ntr; ++i) tf.push_back(i < blks[b].ntc ? std::min(i, (*(*first)[b])[i]) : (*(*first)[b])[i]);
      }
      int rc;
      if ((rc = dev_upload(&d_tasks, tasks, nullptr))) return rc;
      if ((rc = dev_upload(&d_flag_off, foff, nullptr))) return rc;
      if (first) {
         if ((rc = dev_upload(&d_tfirst, tf, nullptr))) return rc;
         if ((rc = dev_upload(&d_tfirst_off, tfoff, nullptr))) return rc;
      }
      n_flags = std::max<long long>(nf, 1);
      const size_t ints = (size_t)(2 * SWEEP_NRHS_MAX + 2 + n_flags * SWEEP_NRHS_MAX);
      HIP_TRY(hipMalloc((void**)&d_ints, ints * sizeof(int)));
      HIP_TRY(hipMemset(d_ints, 0, ints * sizeof(int)));
      epoch = 0;
      enabled = n_tasks > 0 && !getenv("PIPS_HIP_SWEEP_LAUNCHES");
      return PIPS_OK;
   }
   SweepArgs args(long long xw_stride, hipStream_t st) {
      int* ep = d_ints + 2 * SWEEP_NRHS_MAX + 1;
      hipLaunchKernelGGL(k_sweep_bump, dim3(1), dim3(1), 0, st, ep);
      return SweepArgs{d_tasks, n_tasks, d_ints, d_ints + 2 * SWEEP_NRHS_MAX + 2, d_flag_off, d_tfirst, d_tfirst_off, ep,
                       d_ints + 2 * SWEEP_NRHS_MAX, n_flags, xw_stride, poll_limit};
   }
   void release() {
      for (void* p : {(void*)d_tasks, (void*)d_ints, (void*)d_flag_off, (void*)d_tfirst, (void*)d_tfirst_off})
         if (p) (void)hipFree(p);
      d_tasks = nullptr; d_ints = nullptr; d_flag_off = nullptr; d_tfirst = nullptr; d_tfirst_off = nullptr;
      n_tasks = 0;
      enabled = false;
   }
};

struct PhaseTimer {
   bool on = false;
   struct Rec { hipEvent_t a, b; int phase; };
   std::vector<Rec> recs;
   std::vector<hipEvent_t> pool;
   size_t used = 0;
   static constexpr int NPHASE = 16;
   double ms[NPHASE] = {0};
   long long cnt[NPHASE] = {0};
   hipEvent_t get() {
      if (used == pool.size()) {
         hipEvent_t e;
         (void)hipEventCreate(&e);
         pool.push_back(e);
      }
      return pool[used++];
   }
   void reset() { recs.clear(); used = 0; }
   void begin(hipStream_t s, int phase) {
      if (!on) return;
      Rec r{get(), get(), phase};
      (void)hipEventRecord(r.a, s);
      recs.push_back(r);
   }
   void end(hipStream_t s) {
      if (!on) return;
      (void)hipEventRecord(recs.back().b, s);
   }
   // a record that stays open across others: begin_i returns its index (-1 when off)
   int begin_i(hipStream_t s, int phase) {
      if (!on) return -1;
      begin(s, phase);
      return (int)recs.size() - 1;
   }
   void end_i(int idx, hipStream_t s) {
      if (on && idx >= 0) (void)hipEventRecord(recs[idx].b, s);
   }
   void collect() {
      for (int i = 0; i < NPHASE; ++i) { ms[i] = 0; cnt[i] = 0; }
      for (auto& r : recs) {
         float t = 0;
         if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.phase] += t; ++cnt[r.phase]; }
      }
   }
   ~PhaseTimer() { for (auto e : pool) (void)hipEventDestroy(e); }
};

// Host side of k_tail_ldl (tailkernel.hip.h): the tasks of the column launches of a TailPlan as ONE list in the driver's order, dealt to
// eight lists by tile row; the flags and the template they start from.
struct TailSingle {
   TailLdlArgs args{};
   TileTask* d_tasks = nullptr;
   int* d_flags = nullptr;
   int* d_flags_init = nullptr;
   long long* d_flag_off = nullptr;
   int* d_ctl = nullptr;
   long long n_flags = 0;
   int n_tasks = 0, n_blocks = 0;
   void release() {
      for (void* q : {(void*)d_tasks, (void*)d_flags, (void*)d_flags_init, (void*)d_flag_off, (void*)d_ctl}) if (q) (void)hipFree(q);
      d_tasks = nullptr; d_flags = d_flags_init = d_ctl = nullptr; d_flag_off = nullptr;
   }
   int build(const TailPlan& p, const std::vector<BlkDesc>& blks, hipStream_t stream) {
      release();
      const int nblk = (int)blks.size();
      n_blocks = nblk;
      std::vector<long long> foff(nblk + 1, 0);
      for (int b = 0; b < nblk; ++b) foff[b + 1] = foff[b] + (long long)blks[b].ntr * blks[b].ntc + blks[b].ntr + blks[b].ntc;
      n_flags = foff[nblk];
      // the template: prog of a tile = the first K its first update starts from (an update waits for prog >= its k0: the one before it has
      // stored the tile), = its tile column where it takes no update at all (trsm / the diagonal role wait for prog >= tile column)
      std::vector<int> init((size_t)std::max<long long>(n_flags, 1), 0);
      for (int b = 0; b < nblk; ++b)
         for (int ti = 0; ti < blks[b].ntr; ++ti)
            for (int tj = 0; tj < blks[b].ntc; ++tj) init[foff[b] + (long long)ti * blks[b].ntc + tj] = -1;
      std::vector<TileTask> order;
      auto take = [&](const TaskList& l, int kind) {
         for (long long q = l.off; q < l.off + l.cnt; ++q) {
            TileTask t = p.h_tasks[q];
            if (t.blk < 0) continue;
            if (kind == ROOT_UPD) {
               int& pr = init[foff[t.blk] + (long long)t.ti * blks[t.blk].ntc + t.tj];
               if (pr < 0) pr = t.pad & 0xffff;
            } else
               t.pad = 0;
            t.blk |= kind << 24;
            order.push_back(t);
         }
      };
      // first tile column of every tile row (its envelope): the trsm of a row commit in column order from there (root_do)
      std::vector<int> row_first((size_t)std::max<long long>(n_flags, 1), 1 << 20);
      for (int j = 0; j < p.ntc_max; ++j)
         for (long long q = p.trsm[j].off; q < p.trsm[j].off + p.trsm[j].cnt; ++q) {
            const TileTask& t = p.h_tasks[q];
            if (t.blk >= 0) { int& f = row_first[foff[t.blk] + t.ti]; f = std::min(f, t.tj); }
         }
      for (int j = 0; j < p.ntc_max; ++j) {
         take(p.upd_diag[j], ROOT_UPD);
         take(p.diag[j], ROOT_DIAG);
         take(p.upd[j], ROOT_UPD);
         const size_t before = order.size();
         take(p.trsm[j], ROOT_TRSM);
         for (size_t q = before; q < order.size(); ++q) order[q].pad = row_first[foff[order[q].blk & 0xffffff] + order[q].ti];
      }
      for (int b = 0; b < nblk; ++b)
         for (int ti = 0; ti < blks[b].ntr; ++ti)
            for (int tj = 0; tj < blks[b].ntc; ++tj) { int& pr = init[foff[b] + (long long)ti * blks[b].ntc + tj]; if (pr < 0) pr = tj; }
      // eight lists: runs of tasks of one (block, tile row) go to the lists round-robin, every list keeps the order
      std::vector<std::vector<TileTask>> lists(8);
      int run = -1, last_b = -2, last_i = -2;
      for (const TileTask& t : order) {
         const int b = t.blk & 0xffffff;
         if (b != last_b || t.ti != last_i) { ++run; last_b = b; last_i = t.ti; }
         lists[run & 7].push_back(t);
      }
      std::vector<TileTask> all;
      for (int x = 0; x < 8; ++x) { args.xoff[x] = (int)all.size(); all.insert(all.end(), lists[x].begin(), lists[x].end()); }
      args.xoff[8] = (int)all.size();
      n_tasks = (int)all.size();
      if (all.empty()) all.push_back({0, 0, 0, 0});
      int rc;
      if ((rc = dev_upload(&d_tasks, all, stream)) || (rc = dev_upload(&d_flags_init, init, stream)) || (rc = dev_upload(&d_flag_off, foff, stream))) return rc;
      HIP_TRY(hipMalloc((void**)&d_flags, init.size() * sizeof(int)));
      HIP_TRY(hipMalloc((void**)&d_ctl, 16 * sizeof(int)));
      HIP_TRY(hipMemsetAsync(d_ctl, 0, 16 * sizeof(int), stream));
      args.tasks = d_tasks; args.flags = d_flags; args.flag_off = d_flag_off; args.ctl = d_ctl;
      return PIPS_OK;
   }
};

struct TailCtx {
   const BlkDesc* d_blks;
   const TailPlan* plan;
   double* d_arena;
   double* d_dtail;
   double* d_winv;
   const signed char* d_psign;
   const long long* d_psign_off;
   const int* d_bmap;
   int* d_inertia;
   hipStream_t stream;
   PhaseTimer* timer;
   const double* d_pref;
   hipStream_t side = nullptr;          // second stream for the lookahead of the right-looking root factorisation
   hipEvent_t ev_panel = nullptr;       // main -> side: panel j is final (trsm done)
   hipEvent_t ev_rest = nullptr;        // side -> main: trailing update of panel j is done
   bool is_root = false;                // dense root: same update kernel under its own name (k_tile_gemm<3>)
   const int* d_sctab = nullptr;        // sparse Schur complement: per-block position tables (kernels.hip.h sc_entry)
   double* d_uarena = nullptr;          // scaled copies U = L D of the tail rows (BlkDesc::U), B operand of the updates
   // Schur SYRK in row-panel groups (several ranks): group p holds the tile rows whose first Schur row lies in panel p; after
   // its launch ev_sc[p] is recorded and rows of panel p are final on this rank (Engine::set_sc_panels)
   const std::vector<TaskList>* sc_groups = nullptr;
   const TileTask* d_sc_tasks = nullptr;
   const std::vector<hipEvent_t>* ev_sc = nullptr;
   // k_tile_gemm_bal (tasks drawn from per-XCD counters, an eighth more workgroups than tasks): pool of 8-counter slots, zeroed at the
   // start of a factorisation; every such launch takes the next slot
   int* d_ctr_pool = nullptr;
   int* ctr_cursor = nullptr;
   // deterministic Schur accumulation (Engine::set_det_groups): the blocks are cut into at most eight contiguous groups with a
   // buffer each; round k of the SYRK handles the k-th block of every group, so a launch never has two workgroups on the same
   // entry of a buffer and the blocks of a group arrive in their order; k_reduce_groups then adds the buffers in a fixed tree
   const std::vector<TaskList>* det_rounds = nullptr;
   const TileTask* d_det_tasks = nullptr;
   double* d_gbuf = nullptr;
   long long gstride = 0;
   const int* d_blk_group = nullptr;
   int n_groups = 0, first_slot = 0;
   long long sc_len = 0;                // sparse Schur complement: length of its value array (one group buffer)
   bool det_defer_reduce = false;       // several ranks: the group buffers are added over ALL ranks' groups in one fixed tree by the caller
   SweepRt* sweep = nullptr;            // single-launch solve sweeps
   bool bunch_kaufman = false;          // diagonal tiles with 1 x 1 / 2 x 2 pivoting (k_tile_diag_bk) instead of the static pivot order
   int *d_pert_cnt = nullptr, *d_pert_list = nullptr;   // ... and where they record the indices no pivot was found for inside the tile
   const double* bk_orig = nullptr;                     // the matrix being factorised as the caller holds it (partner search), its layout, the order
   int bk_orig_ld = 0, bk_orig_rowmajor = 0;
   const int* d_bk_perm = nullptr;
   int bk_isolate = 0;
   const TailSingle* single = nullptr;   // the leaf tails as one dependency-driven launch (tailkernel.hip.h) instead of the column loop
};
constexpr int GEMM_CTR_SLOTS = 4096;
constexpr int GEMM_BAL_MIN_TASKS = 1024;   // below two full rounds of the chip a static one-task-per-workgroup launch does as well

static int tail_factor(const TailCtx& c, double* SC, int ldSC) {
   const TailPlan& p = *c.plan;
   auto gemm_diag_tiles = [&](const TaskList& l, hipStream_t st) {
      hipLaunchKernelGGL(k_tile_gemm<4>, dim3((l.cnt + 7) / 8 * 8), dim3(512), 0, st, p.d_tasks + l.off, l.cnt, c.d_blks, c.d_arena,
                         c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
   };
   auto gemm0 = [&](const TaskList& l, hipStream_t st) {
      // leaf tails: launches of two rounds of the chip or more draw their tiles from per-XCD counters with an eighth more workgroups than
      // tiles (k_tile_gemm_bal); smaller ones one tile per workgroup.  (A persistent work-stealing variant, static shares, the root's bulk
      // in chunks or as a persistent launch with fewer workgroups than the chip holds: measured, no gain - docs/HISTORY_r4.md.)
      if (!c.is_root && c.d_ctr_pool && l.cnt >= GEMM_BAL_MIN_TASKS && *c.ctr_cursor < GEMM_CTR_SLOTS) {
         int* ctr = c.d_ctr_pool + 8 * (*c.ctr_cursor)++;
         const int wgs = ((l.cnt + 7) / 8 * 8) / 8 * 9;
         hipLaunchKernelGGL(k_tile_gemm_bal<0>, dim3((wgs + 7) / 8 * 8), dim3(512), 0, st, p.d_tasks + l.off, l.cnt, c.d_blks, c.d_arena, c.d_dtail,
                            c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena, ctr);
         return;
      }
      if (c.is_root)
         // The bulk update of the trailing matrix runs on the side stream beside the diagonal-tile chain of the next column.  Stream
         // priorities only order the launches the command processor has not started yet: once a launch of thousands of workgroups is
         // running, the single workgroup of k_tile_diag gets no slot until that launch drains (rocprofv3 trace at S = 16000: the diagonal
         // kernel "takes" 270 - 1140 us beside the bulk and 70 us alone - it is waiting): the lookahead overlaps only the drain.
         hipLaunchKernelGGL(k_tile_gemm<3>, dim3((l.cnt + 7) / 8 * 8), dim3(512), 0, st, p.d_tasks + l.off, l.cnt, c.d_blks, c.d_arena,
                            c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
      else
         hipLaunchKernelGGL(k_tile_gemm<0>, dim3((l.cnt + 7) / 8 * 8), dim3(512), 0, st, p.d_tasks + l.off, l.cnt, c.d_blks, c.d_arena,
                            c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
   };
   // Lookahead bookkeeping (right-looking modes with a side stream): while the side stream applies a finished panel to
   // the tile columns >= side_from, the main stream may only write columns left of that.
   bool side_busy = false;
   int side_from = 0;
   auto main_writes = [&](int col) -> int {
      if (side_busy && col >= side_from) {
         HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_rest, 0));
         side_busy = false;
      }
      return PIPS_OK;
   };
   int rc;
   if (c.single && c.single->n_tasks > 0) {
      const TailSingle& ts = *c.single;
      if (c.timer) c.timer->begin(c.stream, 2);
      HIP_TRY(hipMemcpyAsync(ts.d_flags, ts.d_flags_init, (size_t)ts.n_flags * sizeof(int), hipMemcpyDeviceToDevice, c.stream));
      HIP_TRY(hipMemsetAsync(ts.d_ctl, 0, 9 * sizeof(int), c.stream));   // (the tickets; the error word stays until it is read)
      TailLdlArgs ta = ts.args;
      ta.n_tasks = ts.n_tasks;
      const char* trace_file = getenv("PIPS_HIP_TAIL_TRACE");   // diagnostics: per-task clocks of this launch into a file (tools/tail_trace.py)
      long long* d_trace = nullptr;
      const size_t n_trace = (size_t)3 * ts.n_tasks + 32 * (size_t)(p.ntc_max + 1);
      if (trace_file) {
         HIP_TRY(hipMalloc((void**)&d_trace, n_trace * sizeof(long long)));
         HIP_TRY(hipMemsetAsync(d_trace, 0, n_trace * sizeof(long long), c.stream));
         ta.trace = d_trace;
      }
      // two workgroups per compute unit; a single block has fewer tasks ready than that and its diagonal tiles - the chain - run 3 x faster on a
      // compute unit of their own: one workgroup per unit there (a leaf handle's factorisation 6.9 -> 6.6 ms, level 1.5 1.03 -> 0.98 s per unit;
      // 8 blocks: 17.7 -> 17.9 ms, so only there)
      hipLaunchKernelGGL(k_tail_ldl, dim3(ts.n_blocks <= 2 ? 256 : 512), dim3(512), 0, c.stream, ta);
      if (c.timer) c.timer->end(c.stream);
      if (trace_file) {
         std::vector<long long> h(n_trace);
         std::vector<TileTask> ht((size_t)ts.n_tasks);
         HIP_TRY(hipStreamSynchronize(c.stream));
         HIP_TRY(hipMemcpy(h.data(), d_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
         HIP_TRY(hipMemcpy(ht.data(), ts.d_tasks, ht.size() * sizeof(TileTask), hipMemcpyDeviceToHost));
         (void)hipFree(d_trace);
         if (FILE* f = fopen(trace_file, "w")) {   // ticket, list, kind, block, ti, tj, K range, the three clocks
            for (int t = 0; t < ts.n_tasks; ++t) {
               int x = 0;
               while (x < 7 && t >= ts.args.xoff[x + 1]) ++x;
               fprintf(f, "%d %d %d %d %d %d %d %lld %lld %lld\n", t, x, ht[t].blk >> 24, ht[t].blk & 0xffffff, ht[t].ti, ht[t].tj, ht[t].pad, h[3 * (size_t)t], h[3 * (size_t)t + 1],
                       h[3 * (size_t)t + 2]);
            }
            fclose(f);
         }
      }
   }
   for (int j = 0; !c.single && j < p.ntc_max; ++j) {
      if ((rc = main_writes(j))) return rc;
      const bool ahead = c.side && p.upd_diag[j].cnt > 0;   // left-looking batch: diagonal tiles first, factorised on the side stream
      if (ahead) {
         // side stream: update the diagonal tiles of column j (deep K, one workgroup per block) and factorise them, while
         // the main stream updates the rest of the column
         HIP_TRY(hipEventRecord(c.ev_panel, c.stream));
         HIP_TRY(hipStreamWaitEvent(c.side, c.ev_panel, 0));
         gemm_diag_tiles(p.upd_diag[j], c.side);
         if (c.bunch_kaufman)
            hipLaunchKernelGGL(k_tile_diag_bk, dim3(p.diag[j].cnt), dim3(256), 0, c.side, p.d_tasks + p.diag[j].off, c.d_blks, c.d_arena, c.d_dtail,
                               c.d_winv, c.d_inertia, c.d_pert_cnt, c.d_pert_list, c.bk_orig, c.bk_orig_ld, c.bk_orig_rowmajor, c.d_bk_perm, c.bk_isolate);
         else
         hipLaunchKernelGGL(k_tile_diag, dim3(p.diag[j].cnt), dim3(256), 0, c.side, p.d_tasks + p.diag[j].off,
                            c.d_blks, c.d_arena, c.d_dtail, c.d_winv, c.d_psign, c.d_psign_off, c.d_inertia, c.d_pref);
         HIP_TRY(hipEventRecord(c.ev_rest, c.side));
      } else if (p.upd_diag[j].cnt > 0) {
         gemm_diag_tiles(p.upd_diag[j], c.stream);
      }
      if (p.upd[j].cnt > 0) {
         if (c.timer) c.timer->begin(c.stream, 2);
         gemm0(p.upd[j], c.stream);
         if (c.timer) c.timer->end(c.stream);
      }
      if (ahead) {
         HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_rest, 0));   // trsm needs Winv_j and d_j
      } else {
         if (c.timer) c.timer->begin(c.stream, 3);
         if (c.bunch_kaufman)
            hipLaunchKernelGGL(k_tile_diag_bk, dim3(p.diag[j].cnt), dim3(256), 0, c.stream, p.d_tasks + p.diag[j].off, c.d_blks, c.d_arena, c.d_dtail,
                               c.d_winv, c.d_inertia, c.d_pert_cnt, c.d_pert_list, c.bk_orig, c.bk_orig_ld, c.bk_orig_rowmajor, c.d_bk_perm, c.bk_isolate);
         else
         hipLaunchKernelGGL(k_tile_diag, dim3(p.diag[j].cnt), dim3(256), 0, c.stream, p.d_tasks + p.diag[j].off,
                            c.d_blks, c.d_arena, c.d_dtail, c.d_winv, c.d_psign, c.d_psign_off, c.d_inertia, c.d_pref);
         if (c.timer) c.timer->end(c.stream);
      }
      if (p.trsm[j].cnt > 0) {
         if (c.timer) c.timer->begin(c.stream, 4);
         hipLaunchKernelGGL(k_tile_gemm<1>, dim3((p.trsm[j].cnt + 7) / 8 * 8), dim3(512), 0, c.stream, p.d_tasks + p.trsm[j].off, p.trsm[j].cnt,
                            c.d_blks, c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
         if (c.timer) c.timer->end(c.stream);
         // Bunch-Kaufman root that will be looked at (DenseLdl::check_pivots): multipliers beyond what a whole-column search allows?
         if (c.bunch_kaufman && c.is_root && c.d_pert_cnt)
            hipLaunchKernelGGL(k_bk_growth, dim3(TILE), dim3(256), 0, c.stream, c.d_blks, c.d_arena, j, 1e8, c.d_pert_cnt, c.d_pert_list);
      }
      // right-looking modes: trailing update with the panel that ends at column j.  With a side stream the bulk of it
      // (tile columns >= j+2) runs there, so that the main stream can finish column j+1 and go on with its diagonal tile
      // and trsm meanwhile (lookahead of one column).
      if (p.trail_next[j].cnt > 0 || p.trail[j].cnt > 0) {
         if ((rc = main_writes(j + 1))) return rc;
         const bool split = c.side && p.trail[j].cnt > 0;
         if (split) {
            HIP_TRY(hipEventRecord(c.ev_panel, c.stream));   // panel final: its trsm is queued before this point
            HIP_TRY(hipStreamWaitEvent(c.side, c.ev_panel, 0));
            gemm0(p.trail[j], c.side);                       // queued behind the bulk of the previous panel
            HIP_TRY(hipEventRecord(c.ev_rest, c.side));
            side_busy = true;
            side_from = j + 2;
         }
         if (c.timer) c.timer->begin(c.stream, 2);
         if (p.trail_next[j].cnt > 0) gemm0(p.trail_next[j], c.stream);
         if (!split && p.trail[j].cnt > 0) gemm0(p.trail[j], c.stream);
         if (c.timer) c.timer->end(c.stream);
      }
   }
   if ((rc = main_writes(1 << 30))) return rc;   // join the side stream
   if (SC && c.det_rounds) {
      if (c.timer) c.timer->begin(c.stream, 5);
      for (const TaskList& l : *c.det_rounds)   // the group buffers were zeroed (and took the head's contributions) in Engine::factor
         if (l.cnt > 0)
            hipLaunchKernelGGL(k_tile_gemm<2>, dim3((l.cnt + 7) / 8 * 8), dim3(512), 0, c.stream, c.d_det_tasks + l.off, l.cnt, c.d_blks, c.d_arena,
                               c.d_dtail, c.d_winv, c.d_bmap, SC, ldSC, c.d_sctab, c.d_uarena, c.d_gbuf, c.gstride, c.d_blk_group);
      const int S_ = ldSC;
      if (c.det_defer_reduce) {
      } else if (c.d_sctab)   // sparse Schur complement: the value array as one column
         hipLaunchKernelGGL(k_reduce_groups, dim3((unsigned)std::max(1LL, std::min(1024LL, (c.sc_len + 255) / 256)), 1), dim3(256), 0, c.stream, SC, 0, (int)c.sc_len,
                            c.d_gbuf, c.gstride, c.n_groups, c.first_slot);
      else
         hipLaunchKernelGGL(k_reduce_groups, dim3(std::max(1, std::min(64, (S_ + 255) / 256)), S_), dim3(256), 0, c.stream, SC, ldSC, S_, c.d_gbuf,
                            c.gstride, c.n_groups, c.first_slot);
      if (c.timer) c.timer->end(c.stream);
   } else if (SC && c.sc_groups && !c.d_sctab) {
      if (c.timer) c.timer->begin(c.stream, 5);
      for (size_t g = 0; g < c.sc_groups->size(); ++g) {
         const TaskList& l = (*c.sc_groups)[g];
         if (l.cnt > 0)
            hipLaunchKernelGGL(k_tile_gemm<2>, dim3((l.cnt + 7) / 8 * 8), dim3(512), 0, c.stream, c.d_sc_tasks + l.off, l.cnt, c.d_blks, c.d_arena,
                               c.d_dtail, c.d_winv, c.d_bmap, SC, ldSC, c.d_sctab, c.d_uarena);
         HIP_TRY(hipEventRecord((*c.ev_sc)[g], c.stream));
      }
      if (c.timer) c.timer->end(c.stream);
   } else if (SC && p.schur.cnt > 0 && c.d_ctr_pool && !c.d_sctab && *c.ctr_cursor < GEMM_CTR_SLOTS) {
      if (c.timer) c.timer->begin(c.stream, 5);
      const int wgs = ((p.schur.cnt + 7) / 8 * 8) / 8 * 9;
      hipLaunchKernelGGL(k_tile_gemm_bal<2>, dim3((wgs + 7) / 8 * 8), dim3(512), 0, c.stream, p.d_tasks + p.schur.off, p.schur.cnt, c.d_blks,
                         c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, SC, ldSC, c.d_sctab, c.d_uarena, c.d_ctr_pool + 8 * (*c.ctr_cursor)++);
      if (c.timer) c.timer->end(c.stream);
   } else if (SC && p.schur.cnt > 0) {
      if (c.timer) c.timer->begin(c.stream, 5);
      hipLaunchKernelGGL(k_tile_gemm<2>, dim3((p.schur.cnt + 7) / 8 * 8), dim3(512), 0, c.stream, p.d_tasks + p.schur.off, p.schur.cnt, c.d_blks,
                         c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, SC, ldSC, c.d_sctab, c.d_uarena);
      if (c.timer) c.timer->end(c.stream);
   }
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

static int tail_fwd(const TailCtx& c, double* xw, int nrhs = 1, long long xw_stride = 0) {
   const TailPlan& p = *c.plan;
   if (c.sweep && c.sweep->enabled && nrhs <= SWEEP_NRHS_MAX) {
      hipLaunchKernelGGL(k_tail_rows_fwd, dim3(c.sweep->n_tasks, nrhs), dim3(256), 0, c.stream, c.sweep->args(xw_stride, c.stream), c.d_blks, c.d_arena, c.d_dtail,
                         c.d_winv, xw);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
   for (int j = 0; j < p.ntc_max; ++j)
      if (p.fwd[j].cnt > 0)
         hipLaunchKernelGGL(k_tail_fwd, dim3(p.fwd[j].cnt, nrhs), dim3(256), 0, c.stream, p.d_tasks + p.fwd[j].off, c.d_blks,
                            c.d_arena, c.d_dtail, c.d_winv, xw, j, xw_stride);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

static int tail_bwd(const TailCtx& c, double* xw, int nrhs = 1, long long xw_stride = 0, int border = 0) {
   const TailPlan& p = *c.plan;
   if (p.ntc_max == 0) return PIPS_OK;   // no block has a dense tail
   if (border && !(c.sweep && c.sweep->enabled)) PIPS_FAIL(PIPS_ERR_STATE, "border-backward sweep needs the single-launch tail sweeps");
   if (c.sweep && c.sweep->enabled && nrhs <= SWEEP_NRHS_MAX) {
      hipLaunchKernelGGL(k_tail_rows_bwd, dim3(c.sweep->n_tasks, nrhs), dim3(256), 0, c.stream, c.sweep->args(xw_stride, c.stream), c.d_blks, c.d_arena, c.d_dtail,
                         c.d_winv, xw, border);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
   for (int i = p.ntc_max - 1; i >= 0; --i)
      if (p.bwd[i].cnt > 0)
         hipLaunchKernelGGL(k_tail_bwd, dim3(p.bwd[i].cnt, nrhs), dim3(256), 0, c.stream, p.d_tasks + p.bwd[i].off, c.d_blks,
                            c.d_arena, c.d_dtail, c.d_winv, xw, i, xw_stride);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// batched leaf engine
// ---------------------------------------------------------------------------------------------------------------
struct BlockInput {
   int n = 0, n_primal = -1;
   std::vector<int> krow, kcol;
   std::vector<int> btrow, btcol;  // S+1 / nnz ; empty if no border
   std::vector<double> btval;
};

// The head solve sweeps use the register-lean "chain" kernels (k_head_fwd_chain / k_head_bwd_chain: all loads up front, partial sums in
// registers, one LDS transpose) at every launch size (round 2 drew a line at 1024 waves; with supernodes capped at 16 columns they win
// everywhere: configs[3] share, leaf solve 13.5 -> 11.6 ms).  Deterministic mode takes k_head_fwd (it writes slots) forward, the chain
// kernel backward.
struct LevelRange {
   int simple_begin, simple_cnt, small_begin, small_cnt, large_begin, large_cnt;
   int small_lds = 0, large_lds = 0;   // doubles of LDS the widest L21 panel of the class needs (r * (w | 1)), capped at the kernel's capacity
};

// Multifrontal head: one launch per (level, front class); class = (workgroup size, width bound) of k_front.
struct MfLaunch { int level, cls, begin, cnt, lds_doubles; };
// doubles of the update matrix a front keeps: all r columns, or (border split) those of its rb rows of K
// rows a front holds below its pivot block: all, or (fronts on the rows of K only, BlockSym::mf_konly) its rows of K
static inline int mf_rows(const BlockSym& bs, const HeadSupernode& s) { return bs.mf_konly ? s.rb : s.r; }
static inline long long mf_unp(const BlockSym& bs, const HeadSupernode& s) {
   const long long uc = bs.mf_split ? s.rb : s.r, r = mf_rows(bs, s);
   return uc * r - uc * (uc - 1) / 2;
}
static inline int mf_class(int w, long long nf, long long unp, long long lds_budget) {
   const long long r = nf - w, pw = std::max<long long>(w * nf - (long long)w * (w - 1) / 2, (long long)w * ((r + 3) / 4 * 4));
   if (pw + unp + 8 > lds_budget) return 6 + (nf <= 256 ? 0 : 1);   // update matrix stays in device memory
   // more than one wave: 256 threads - the phases around the pivots are spread over them (128 threads up to 128 rows - twice the fronts
   // per compute unit where the registers set the limit - measured 15.3 against 15.0 ms on the 256-block chain: docs/HISTORY_r4.md)
   return (nf <= 64 ? 0 : 2) + 3 * (w <= 16 ? 0 : 1);
}

// Tile geometry + the cost-model / amalgamation knobs (environment overrides are for tuning runs only).
static void apply_tuning(AnalyzeOptions& opt) {
   opt.tile = TILE;
   // Supernode width cap.  16 instead of the kernels' limit of 32: on the time-coupled family (tools/config3_probe.py, 64 x 50 000)
   // narrower supernodes carry fewer explicit zeros (nnz(L) 164 M -> 151 M), halve the dependent pivot chain of a front and the
   // registers its rows take; factorisation 13.4 -> 12.5 ms, solveCompressed 9.3 -> 8.9 ms.  Random sparsity (config 2) has no
   // supernodes wider than one column in the head.
   opt.max_sn_width = 16;
   if (const char* sw = getenv("PIPS_HIP_SN_WIDTH")) opt.max_sn_width = std::max(1, std::min(HEAD_WMAX, atoi(sw)));
   if (const char* rz = getenv("PIPS_HIP_RELAX_ZEROS")) opt.relax_zeros = atof(rz);   // share of explicit zeros per panel
   if (const char* ndd = getenv("PIPS_HIP_ND_DEPTH")) opt.nd_depth = atoi(ndd);        // dissection levels (0 = off)
   if (const char* ndm = getenv("PIPS_HIP_ND_MIN")) opt.nd_min_size = atoi(ndm);      // smallest segment that is still dissected
   if (const char* ml = getenv("PIPS_HIP_MF_LDS")) opt.mf_lds_doubles = atoll(ml);   // LDS budget of a front in doubles (tests: small values force the device-memory variant)
   if (const char* sp = getenv("PIPS_HIP_MF_SPLIT")) opt.mf_split_nb_max = atoi(sp) == 0 ? 0 : std::min(176, std::max(atoi(sp), 2));   // border split: 0 = off, else the largest nb
}

struct Engine {
   int device = 0;
   hipStream_t stream = nullptr;
   int nblk = 0, S = 0;
   bool analyzed = false, factored = false;
   long long analysis_gen = 0;   // bumped by every analyze(): device buffers of the previous analysis are gone (captured graphs are stale)
   int refine_steps = 1;       // maximum number of iterative-refinement steps per solve
   double refine_tol = 0.0;    // > 0: adaptive refinement, stop as soon as the error measure of every block is <= tol
   int refine_mode = 0;        // 0: ||r_b||inf / ||rhs_b||inf ; 1: normwise backward error ||r_b||inf / (max|K_b| ||x_b||inf + ||rhs_b||inf)
   double last_refine_measure = 0.0;
   double* d_norms = nullptr;  // 3 * nblk: ||r||, ||rhs||, ||x|| per block
   double* h_norms = nullptr;  // pinned
   int last_refine_steps = 0;
   double thr_rel = 1e-13, repl_rel = 1e-8;
   AnalyzeOptions opt;
   std::vector<BlockInput> in;
   std::vector<BlockSym> sym;
   std::vector<BlkDesc> h_blks;
   std::vector<long long> kptr;     // nblk+1 offsets into kval
   std::vector<long long> x_off;    // nblk+1 offsets into flat vectors
   std::vector<LevelRange> levels;
   std::vector<LevelRange> levels_top;   // the spine's levels, for the multi-vector sweeps (which are level-scheduled throughout)
   int *d_frowptr = nullptr, *d_fcol = nullptr, *d_fsrc = nullptr;   // full row structure of K for the refinement residual
   long long* d_flong = nullptr;   // its rows longer than FULL_LONG_ROW
   int n_flong = 0;
   int* d_sctab = nullptr;    // sparse Schur complement (set_sc_tables): position tables, BlkDesc::sctab_off
   int schur_mode = 0;        // requested: 0 auto, 1 augmented partial factorisation, 2 blocked solves (reference K4-K6)
   int schur_mode_eff = 1;    // what analyze() settled on
   std::vector<int> schur_cols;   // non-empty Schur columns (any block), ascending
   int *d_schur_cols = nullptr, *d_schur_slot = nullptr;
   // fronts on the rows of K only (BlockSym::mf_konly): records and lists of k_border_rows / k_border_tail
   int* d_kb_rec = nullptr;
   long long* d_kb_off = nullptr;       // per supernode (sorted id): offset of its record, -1 none
   int* d_kb_list = nullptr;            // fronts with border rows, level after level
   std::vector<int> kb_level_off;       // offsets into d_kb_list per level (size levels + 1)
   std::vector<int> kb_level_pairs, kb_level_lds;   // per level: most pairs of one front, bytes of the largest border-row block (LDS of k_border_rows)
   int* d_kb_tail = nullptr;
   int n_kb_tail = 0;
   int sn_width = 0;           // > 0: supernode width cap of this engine instead of the tuned default (the sparse root: a single block, every level is latency)
   bool mf = false;            // multifrontal head (k_front): update matrices go from child to parent front, no FP64 atomics in the head
   std::vector<MfLaunch> mf_launches;
   double* d_mfU = nullptr;    // update matrices of the fronts
   double* d_mfLV = nullptr;   // d and l of the simple leaves below fronts, front by front
   int *d_roots = nullptr, *d_root_off = nullptr;   // fronts without a head parent, per block (k_root_assemble)
   BbBatch* d_bb_batches = nullptr;                 // border split (k_border_schur): batches of supernodes with border rows, block after block
   BbMeta* d_bb_meta = nullptr;
   int *d_bb_off = nullptr, *d_bb_pos = nullptr;    // batches of block b: [d_bb_off[b], d_bb_off[b + 1]); compressed border ids of the staged rows
   long long bb_doubles = 0;                        // doubles of the border-row arena (behind the panels inside d_arena)
   double* d_bb_out = nullptr;                      // deterministic mode: the blocks' border x border triangles before they join their groups
   std::vector<int> h_bb_off_keep;
   int n_bb = 0, bb_stage = 3072, bb_nbmax = 0, bb_poscap = 0;
   bool bb_two_per_cu = false;   // k_border_schur: staging area sized for two workgroups per compute unit (bb_plan_size)
   int* d_bb_round_blk = nullptr;                   // deterministic mode: the blocks of round k of k_border_schur
   std::vector<int> bb_round_off;
   std::vector<int> h_root_off_keep;
   int n_roots = 0;
   int* d_round_blk = nullptr;                       // deterministic mode: the blocks of round k at [round_off[k], round_off[k + 1])
   std::vector<int> round_off;
   int* d_mfint = nullptr;     // front records (common.h)
   long long mfU_total = 0;
   int spine_total = 0, n_levels_all = 0;   // supernodes handled by the per-block spine kernels; tree height before the cut
   int *d_spine = nullptr, *d_spine_off = nullptr;
   long long n_total = 0, nnzK_total = 0, nnzB_total = 0, arena_total = 0, xw_total = 0, bt_rows_total = 0;
   int nsn_total = 0;
   TailPlan plan;
   PhaseTimer timer;

   double* d_uarena = nullptr;   // U = L D copies of the tails (see k_tile_gemm)
   long long uarena_total = 0;
   double *d_arena = nullptr, *d_kval = nullptr, *d_bval = nullptr, *d_winv = nullptr, *d_dtail = nullptr, *d_xw = nullptr;
   double *d_rhs = nullptr, *d_res = nullptr, *d_stage = nullptr, *d_pref = nullptr;
   long long *d_kdst = nullptr, *d_bdst = nullptr, *d_kdiag = nullptr, *d_kptr = nullptr, *d_psign_off = nullptr,
             *d_perm_off = nullptr, *d_rowbase = nullptr, *d_bt_xoff = nullptr;
   SnDesc* d_sns = nullptr;
   BlkDesc* d_blks = nullptr;
   // Forward substitution of the simple leaves as a gather (k_leaf_fwd_gather): the leaves' L entries by TARGET row - row list,
   // row pointers, source (the leaf's position in the work vector), values (written in this order by k_head_factor_simple through
   // d_lf_pos, which is indexed like d_rowidx).  No atomics, fixed order of the sums.
   int *d_lf_rows = nullptr, *d_lf_ptr = nullptr, *d_lf_src = nullptr, *d_lf_pos = nullptr;
   double* d_lf_val = nullptr;
   long long lf_rows = 0, lf_entries = 0;
   LeafDesc* d_leafdesc = nullptr;   // compact records of the level-0 simple leaves, in the order of d_sns (k_leaf_bwd)
   int* d_lb_list = nullptr;         // the simple leaves that own border rows (k_leaf_border)
   int n_lb = 0, nb_pad_max = 0;
   int head_wcap = HEAD_WMAX;   // widest head supernode of this analysis (picks the register-lean variants of the chain kernels)
   int *d_rowidx = nullptr, *d_upd = nullptr, *d_sncol = nullptr, *d_bmap = nullptr, *d_perm = nullptr, *d_inertia = nullptr, *d_nprimal = nullptr;
   int *d_br_rowptr = nullptr, *d_br_sc = nullptr, *d_br_src = nullptr;   // the border by leaf row (k_border_mult_rows)
   int *d_krowptr = nullptr, *d_kcolidx = nullptr, *d_bt_rowptr = nullptr, *d_bt_colidx = nullptr, *d_bt_rowsc = nullptr;
   signed char* d_psign = nullptr;
   std::vector<int> h_inertia;
   std::vector<double> h_amax;   // max|K_b| of the current factorisation (backward-error refinement criterion)

   ~Engine() {
      release();
      if (side) (void)hipStreamDestroy(side);
      if (ev_diag_in) (void)hipEventDestroy(ev_diag_in);
      if (ev_diag_out) (void)hipEventDestroy(ev_diag_out);
   }
   void release() {
      if (d_uarena) (void)hipFree(d_uarena);
      d_uarena = nullptr;
      // the row-panel groups of the Schur SYRK belong to the analysis they were cut for (set_sc_panels)
      for (auto ev : ev_sc) (void)hipEventDestroy(ev);
      ev_sc.clear(); sc_groups.clear(); sc_row_begin.clear();
      if (d_sc_tasks) (void)hipFree(d_sc_tasks);
      d_sc_tasks = nullptr;
      for (void* q : {(void*)d_lf_rows, (void*)d_lf_ptr, (void*)d_lf_src, (void*)d_lf_pos, (void*)d_lf_val})
         if (q) (void)hipFree(q);
      d_lf_rows = d_lf_ptr = d_lf_src = d_lf_pos = nullptr; d_lf_val = nullptr;
      for (void* q : {(void*)d_br_rowptr, (void*)d_br_sc, (void*)d_br_src})
         if (q) (void)hipFree(q);
      d_br_rowptr = d_br_sc = d_br_src = nullptr;
      if (d_leafdesc) (void)hipFree(d_leafdesc);
      d_leafdesc = nullptr;
      lf_rows = 0; lf_entries = 0;
      if (d_mfU) (void)hipFree(d_mfU);
      if (d_mfLV) (void)hipFree(d_mfLV);
      for (void* q : {(void*)d_bb_batches, (void*)d_bb_meta, (void*)d_bb_off, (void*)d_bb_pos, (void*)d_bb_round_blk, (void*)d_bb_out})
         if (q) (void)hipFree(q);
      d_bb_out = nullptr;
      d_bb_batches = nullptr; d_bb_meta = nullptr; d_bb_off = nullptr; d_bb_pos = nullptr; d_bb_round_blk = nullptr; n_bb = 0; bb_doubles = 0;
      for (void* q : {(void*)d_roots, (void*)d_root_off, (void*)d_round_blk})
         if (q) (void)hipFree(q);
      d_roots = d_root_off = d_round_blk = nullptr;
      n_roots = 0;
      if (d_mfint) (void)hipFree(d_mfint);
      d_mfU = nullptr; d_mfLV = nullptr; d_mfint = nullptr;
      mf_launches.clear();
      if (d_gemm_ctr) (void)hipFree(d_gemm_ctr);
      d_gemm_ctr = nullptr;
      for (auto& g : g_levels) g.release();
      for (auto& g : gv_levels) g.release();
      g_levels.clear(); gv_levels.clear();
      g_tail.release(); g_sc.release(); gv_tail.release(); g_btm.release(); g_bm.release();
      if (d_bt_tmp) (void)hipFree(d_bt_tmp);
      d_bt_tmp = nullptr;
      for (void* p : {(void*)d_det_tasks, (void*)d_gbuf, (void*)d_blk_group, (void*)d_gvec, (void*)d_tvec})
         if (p) (void)hipFree(p);
      d_det_tasks = nullptr; d_gbuf = nullptr; d_blk_group = nullptr; d_gvec = d_tvec = nullptr;
      g_btm_grp.release(); g_sc_grp.release(); g_bslot_grp.release();
      for (void* q : {(void*)d_bg_ent, (void*)d_bg_ptr, (void*)d_bg_idx, (void*)d_bg_val, (void*)d_bg_slot}) if (q) (void)hipFree(q);
      d_bg_ent = nullptr; d_bg_ptr = d_bg_slot = nullptr; d_bg_idx = nullptr; d_bg_val = nullptr; n_bg_targets = n_bg_ent = 0; det_aug_ready = false;
      if (d_slot_val) (void)hipFree(d_slot_val);
      if (d_vslot_val) (void)hipFree(d_vslot_val);
      if (d_mvslot) (void)hipFree(d_mvslot);
      d_slot_val = d_vslot_val = d_mvslot = nullptr;
      void* ptrs[] = {d_arena, d_kval, d_bval, d_winv, d_dtail, d_xw, d_rhs, d_res, d_stage, d_pref, d_norms, d_kdst, d_bdst, d_kdiag, d_kptr,
                      d_psign_off, d_perm_off, d_rowbase, d_bt_xoff, d_sns, d_blks, d_rowidx, d_upd, d_sncol, d_bmap, d_perm, d_spine, d_spine_off, d_schur_cols, d_schur_slot, d_sctab, d_frowptr, d_fcol, d_fsrc, d_flong,
                      d_inertia, d_nprimal, d_krowptr, d_kcolidx, d_bt_rowptr, d_bt_colidx, d_bt_rowsc, d_psign};
      for (void* p : ptrs)
         if (p) (void)hipFree(p);
      d_arena = d_kval = d_bval = d_winv = d_dtail = d_xw = d_rhs = d_res = d_stage = d_pref = d_norms = nullptr;
      if (h_norms) (void)hipHostFree(h_norms);
      h_norms = nullptr;
      if (h_inertia_pin) (void)hipHostFree(h_inertia_pin);
      h_inertia_pin = nullptr;
      if (ev_inertia) (void)hipEventDestroy(ev_inertia);
      ev_inertia = nullptr;
      inertia_in_flight = inertia_on_host = false;
      for (double** q : {&d_mx_xw, &d_mx_rhs, &d_mx_res, &d_hostx, &d_hostpack, &d_mmeasure}) { if (*q) (void)hipFree(*q); *q = nullptr; }
      if (d_midx) (void)hipFree(d_midx);
      d_midx = nullptr;
      if (h_mmeasure) (void)hipHostFree(h_mmeasure);
      h_mmeasure = nullptr;
      mx_cap = 0;
      hostx_cap = hostpack_cap = 0;
      d_kdst = d_bdst = d_kdiag = d_kptr = d_psign_off = d_perm_off = d_rowbase = d_bt_xoff = nullptr;
      d_sns = nullptr; d_blks = nullptr;
      d_nprimal = nullptr;
      d_spine = d_spine_off = d_schur_cols = d_schur_slot = d_sctab = nullptr;
      for (void* q : {(void*)d_kb_rec, (void*)d_kb_off, (void*)d_kb_list, (void*)d_kb_tail}) if (q) (void)hipFree(q);
      d_kb_rec = nullptr; d_kb_off = nullptr; d_kb_list = nullptr; d_kb_tail = nullptr; n_kb_tail = 0; kb_level_off.clear(); kb_level_pairs.clear(); kb_level_lds.clear();
      d_frowptr = d_fcol = d_fsrc = nullptr; d_flong = nullptr;
      d_rowidx = d_upd = d_sncol = d_bmap = d_perm = d_inertia = d_krowptr = d_kcolidx = d_bt_rowptr = d_bt_colidx = d_bt_rowsc = nullptr;
      d_psign = nullptr;
      plan.release();
      sweep.release();
      tsingle.release();
   }

   SweepRt sweep;
   TailSingle tsingle;                               // the tails as one dependency-driven launch (tailkernel.hip.h)
   bool tail_single = false;                         // ... decided at analyze time: the tails are then assembled in a scratch region (BlkDesc::T_in)
   long long tail_scratch = 0;                       // doubles of that region, behind the panels and the border-row arena
   hipStream_t side = nullptr;                       // diagonal tiles of the tail are factorised here, beside the column update
   hipEvent_t ev_diag_in = nullptr, ev_diag_out = nullptr;
   // ---- deterministic mode (pips_hip_batch_set_deterministic): no FP64 atomics on the path.  Every scattered contribution of
   // the head owns a slot; at analyze time the kernels run once in recording mode, the host groups the slots by target (per
   // elimination-tree level, then the tail, then the Schur complement) and the factorisation gathers them in that fixed order.
   struct GatherList {
      long long n_targets = 0, n_slots = 0;
      long long *d_tgt = nullptr, *d_off = nullptr, *d_slots = nullptr;
      void release() {
         for (void* p : {(void*)d_tgt, (void*)d_off, (void*)d_slots})
            if (p) (void)hipFree(p);
         d_tgt = d_off = d_slots = nullptr;
         n_targets = n_slots = 0;
      }
   };
   struct SlotEntry { long long target, slot; };
   bool deterministic = false;
   // The slot / gather scheme of the head belongs to deterministic mode (without it, round 2 measured the head phase of config 2 at 5.0 ms
   // against 5.5 ms with FP64 atomics - not worth 16 bytes of device memory per contribution and the longer analysis; the switch that
   // selected it alone is gone).  Beyond HEAD_SLOTS_MAX contributions deterministic mode refuses.
   bool head_slots = false;
   bool slot_solves = false;    // single-RHS forward substitution through slots outside deterministic mode too (measured: no gain)
   static constexpr long long HEAD_SLOTS_MAX = 400LL * 1000 * 1000;
   long long slots_total = 0, vslots_total = 0;
   double *d_slot_val = nullptr, *d_vslot_val = nullptr;
   int last_multi_path = 0;      // how the last solve(nrhs) went: 0 one sweep per right-hand side, 1 interleaved panels, 2 interleaved panels with the slot / gather forward substitution
   double* d_mvslot = nullptr;   // deterministic mode, several right-hand sides: the forward substitution's slots for one panel of MQ (solve_multi)
   std::vector<GatherList> g_levels, gv_levels;   // targets inside the head, per level (factorisation / forward substitution)
   GatherList g_tail, g_sc, gv_tail;
   std::vector<SlotEntry> sc_e_keep;
   std::vector<int> sc_blk_keep;
   GatherList g_sc_grp;            // Schur targets of the head inside the group buffers
   std::vector<int> h_bt_rowsc_keep, h_bt_rownnz_keep, h_bt_rowblk_keep;
   GatherList g_btm_grp;           // border rows per (group, Schur column): Br^T z summed group-wise, then in the fixed tree
   double *d_gvec = nullptr, *d_tvec = nullptr;
   std::vector<TaskList> det_rounds;
   TileTask* d_det_tasks = nullptr;
   double* d_gbuf = nullptr;
   int* d_blk_group = nullptr;
   int det_n_groups = 0, det_first_slot = 0;
   // Several ranks whose count divides eight: every rank's group buffers travel to every rank (an all-reduce in which the others hold
   // zeros at these slots: exact) and ALL eight slots are added in the one fixed tree of k_reduce_groups on every rank - the sums
   // associate the same way for 1, 2, 4 and 8 ranks (pips_hip_kkt_factorize / the deterministic Lsolve), at eight times the bytes of
   // the plain reduction of the Schur complement
   bool det_global = false;
   int det_slots = 8;                 // group slots of this rank among the global eight (8 / n_ranks where that divides)
   bool defer_group_reduce = false;   // set by pips_hip_kkt_factorize around its factor() call: it adds the groups of ALL ranks in one tree itself;
                                      // a direct pips_hip_batch_factor on the same batch reduces its own groups as on one rank
   // length of one group buffer: S x S (dense Schur complement) or the value array of the sparse one (set_sc_tables)
   long long sc_len = 0;
   long long det_gstride() const { return sc_len > 0 ? sc_len : (long long)S * S; }
   // groups for the deterministic Schur accumulation: the global problem has eight group slots; this rank (rank of n_ranks, blocks
   // sharded contiguously and evenly) fills 8 / n_ranks of them (all eight when n_ranks does not divide 8).  What holds across rank
   // counts: 1 and 2 ranks give equal bits; 4 and 8 ranks are reproducible run to run only - the all-reduce, not the fixed tree,
   // associates the ranks' partial sums
   int set_det_groups(int rank, int n_ranks) {
      if (!analyzed || !deterministic) return PIPS_OK;
      const int slots = (n_ranks >= 1 && n_ranks <= 8 && 8 % n_ranks == 0) ? 8 / n_ranks : 8;
      det_first_slot = slots == 8 ? 0 : rank * slots;
      det_global = n_ranks > 1 && slots < 8;
      det_slots = slots;
      const int gs = std::max(1, (nblk + slots - 1) / slots);     // blocks per group
      det_n_groups = (nblk + gs - 1) / gs;
      std::vector<int> grp(std::max(nblk, 1), 0);
      for (int b = 0; b < nblk; ++b) grp[b] = b / gs;
      std::vector<std::vector<TileTask>> rounds(gs);
      for (int b = 0; b < nblk; ++b) {
         if (h_blks[b].ntc <= 0 || h_blks[b].nb <= 0) continue;
         const int nt = h_blks[b].nb_pad / TILE;
         for (int ti = 0; ti < nt; ++ti)
            for (int tj = 0; tj <= ti; ++tj) rounds[b % gs].push_back({b, ti, tj, 0});
      }
      if (mf) {   // root fronts of round k = those of the k-th block of every group: one block per group and launch
         std::vector<int> rb;
         round_off.assign(1, 0);
         for (int k = 0; k < gs; ++k) {
            for (int b = k; b < nblk; b += gs)
               if (h_root_off_keep[b + 1] > h_root_off_keep[b]) rb.push_back(b);
            round_off.push_back((int)rb.size());
         }
         if (d_round_blk) { (void)hipFree(d_round_blk); d_round_blk = nullptr; }
         if (rb.empty()) rb.push_back(0);
         int rcb = dev_upload(&d_round_blk, rb, stream);
         if (rcb) return rcb;
         // ... and the same rounds over the blocks whose border x border part k_border_schur forms
         std::vector<int> bbr;
         bb_round_off.assign(1, 0);
         for (int k = 0; k < gs; ++k) {
            for (int b = k; b < nblk; b += gs)
               if (n_bb > 0 && h_bb_off_keep[b + 1] > h_bb_off_keep[b]) bbr.push_back(b);
            bb_round_off.push_back((int)bbr.size());
         }
         if (d_bb_round_blk) { (void)hipFree(d_bb_round_blk); d_bb_round_blk = nullptr; }
         if (bbr.empty()) bbr.push_back(0);
         if ((rcb = dev_upload(&d_bb_round_blk, bbr, stream))) return rcb;
      }
      det_rounds.clear();
      std::vector<TileTask> all;
      for (int k = 0; k < gs; ++k) {
         TaskList l;
         l.off = (long long)all.size(); l.cnt = (int)rounds[k].size();
         all.insert(all.end(), rounds[k].begin(), rounds[k].end());
         det_rounds.push_back(l);
      }
      if (all.empty()) all.push_back({-1, 0, 0, 0});
      for (void* p : {(void*)d_det_tasks, (void*)d_gbuf, (void*)d_blk_group})
         if (p) (void)hipFree(p);
      d_det_tasks = nullptr; d_gbuf = nullptr; d_blk_group = nullptr;
      int rc;
      if ((rc = dev_upload(&d_det_tasks, all, stream)) || (rc = dev_upload(&d_blk_group, grp, stream))) return rc;
      // (the group buffers themselves are allocated by the first factorisation that has a Schur complement to fill: a system with the
      //  sparse root never needs the det_n_groups x S x S doubles of the dense layout this is first called with)
      {  // the head's own Schur contributions join their block's group buffer
         g_sc_grp.release();
         std::vector<SlotEntry> ent(sc_e_keep.size());
         for (size_t i = 0; i < sc_e_keep.size(); ++i) ent[i] = {(long long)grp[sc_blk_keep[i]] * det_gstride() + sc_e_keep[i].target, sc_e_keep[i].slot};
         if ((rc = upload_gather(ent, g_sc_grp))) return rc;
      }
      // Br^T z in the same group order: rows of group g go to slot g of an 8 x S scratch matrix
      g_btm_grp.release();
      if (d_gvec) (void)hipFree(d_gvec);
      if (d_tvec) (void)hipFree(d_tvec);
      d_gvec = d_tvec = nullptr;
      if (S > 0) {
         std::vector<SlotEntry> ent;
         for (long long i = 0; i < bt_rows_total; ++i)
            if (h_bt_rownnz_keep[(size_t)i] > 0) ent.push_back({(long long)grp[h_bt_rowblk_keep[(size_t)i]] * S + h_bt_rowsc_keep[(size_t)i], i});
         if ((rc = upload_gather(ent, g_btm_grp))) return rc;
         HIP_TRY(hipMalloc((void**)&d_gvec, (size_t)8 * S * sizeof(double)));
         HIP_TRY(hipMalloc((void**)&d_tvec, (size_t)S * sizeof(double)));
         if ((rc = build_det_aug(grp))) return rc;
      }
      if (!det_aug_ready) aug_sweeps_ok = false;   // (deterministic mode takes the sweeps of the augmented factor only through forward_augmented_det)
      return PIPS_OK;
   }
   // Deterministic forward sweep of the augmented factor (forward_augmented_det).  Per (block, border id): the head supernodes that hold that
   // border row, in the order of the supernode array - entry = where its w factors L_b(a, 0 .. w-1) lie (the border-row arena of a front
   // under the border split, else the panel) and the first of its w columns in the work vector.  And the gather of the blocks' border slots
   // into the group slots of d_gvec.
   int build_det_aug(const std::vector<int>& grp) {
      det_aug_ready = false;
      for (void* q : {(void*)d_bg_ent, (void*)d_bg_ptr, (void*)d_bg_idx, (void*)d_bg_val, (void*)d_bg_slot}) if (q) (void)hipFree(q);
      d_bg_ent = nullptr; d_bg_ptr = d_bg_slot = nullptr; d_bg_idx = nullptr; d_bg_val = nullptr; n_bg_targets = n_bg_ent = 0;
      g_bslot_grp.release();
      if (!aug_sweeps_ok || h_sns_keep.empty()) return PIPS_OK;
      std::vector<long long> tbase(nblk + 1, 0);
      for (int b = 0; b < nblk; ++b) tbase[b + 1] = tbase[b] + h_blks[b].nb;
      const long long nt = tbase[nblk];
      if (nt == 0) return PIPS_OK;
      std::vector<long long> cnt((size_t)nt + 1, 0);
      auto for_rows = [&](auto&& fn) {
         for (size_t i = 0; i < h_sns_keep.size(); ++i) {
            const SnDesc& sn = h_sns_keep[i];
            if (sn.rb >= sn.r) continue;
            const BlockSym& bs = sym[sn.blk];
            const int loc = bs.sn_of_col[(size_t)sn.c0];
            const int* rows = bs.rowidx.data() + bs.sn[loc].rows;
            for (int a = sn.rb; a < sn.r; ++a) fn(sn, tbase[sn.blk] + (rows[a] - bs.n), a);
         }
      };
      for_rows([&](const SnDesc&, long long t, int) { ++cnt[(size_t)t + 1]; });
      for (long long t = 0; t < nt; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
      // entries in the order of the supernode array (a supernode's border rows side by side: the threads of k_border_rowdot_det read them
      // coalesced); per target the indices of its entries, ascending
      const long long n_ent = cnt[(size_t)nt];
      if (n_ent >= (1LL << 31) || xw_total >= (1LL << 32)) return PIPS_OK;   // (index widths of the lists; the refined path stays)
      std::vector<BgEntry> ent((size_t)n_ent);
      std::vector<int> idx((size_t)n_ent);
      std::vector<long long> fill(cnt.begin(), cnt.end() - 1);
      bool ok = true;
      long long e_next = 0;
      for_rows([&](const SnDesc& sn, long long t, int a) {
         BgEntry e;
         const bool in_panel = sn.ld >= sn.w + sn.r;
         const int stride = in_panel ? sn.ld : ((sn.r - sn.rb + 3) & ~3);
         if (!in_panel && sn.bb < 0) ok = false;
         if (stride > 65535 || sn.w > 65535) ok = false;
         e.off = in_panel ? sn.panel + sn.w + a : sn.bb + (a - sn.rb);
         e.y = (unsigned)(h_blks[sn.blk].xw_off + sn.c0); e.stride = (unsigned short)stride; e.w = (unsigned short)sn.w;
         idx[(size_t)fill[(size_t)t]++] = (int)e_next;
         ent[(size_t)e_next++] = e;
      });
      if (!ok) return PIPS_OK;   // (a layout this sweep does not read: the refined path stays)
      std::vector<long long> slot((size_t)nt);
      std::vector<SlotEntry> col;
      for (int b = 0; b < nblk; ++b)
         for (int g = 0; g < h_blks[b].nb; ++g) {
            slot[(size_t)(tbase[b] + g)] = h_blks[b].xw_off + h_blks[b].n_head + h_blks[b].m_pad + g;
            col.push_back({(long long)grp[b] * S + sym[b].bmap[(size_t)g], slot[(size_t)(tbase[b] + g)]});
         }
      int rc;
      if ((rc = dev_upload(&d_bg_ent, ent, stream)) || (rc = dev_upload(&d_bg_ptr, cnt, stream)) || (rc = dev_upload(&d_bg_idx, idx, stream)) ||
          (rc = dev_upload(&d_bg_slot, slot, stream)) || (rc = upload_gather(col, g_bslot_grp)))
         return rc;
      HIP_TRY(hipMalloc((void**)&d_bg_val, (size_t)std::max<long long>(n_ent, 1) * sizeof(double)));
      n_bg_ent = n_ent;
      n_bg_targets = nt;
      det_aug_ready = true;
      return PIPS_OK;
   }
   GatherList g_btm, g_bm;        // border products: rows per Schur column, entries per leaf row
   // deterministic mode, forward sweep of the augmented factor: the border slots of the work vector are gathered, target by target, from the
   // border rows of the head supernodes (k_border_gather_det: no atomics, fixed order) and then group-wise into d_gvec like Br^T z
   BgEntry* d_bg_ent = nullptr;
   long long *d_bg_ptr = nullptr, *d_bg_slot = nullptr;
   int* d_bg_idx = nullptr;
   double* d_bg_val = nullptr;
   long long n_bg_targets = 0, n_bg_ent = 0;
   GatherList g_bslot_grp;
   bool det_aug_ready = false;
   double* d_bt_tmp = nullptr;
   int* d_gemm_ctr = nullptr;     // counter slots of the persistent update kernel
   int gemm_ctr_cursor = 0;
   // ---- Schur SYRK in row-panel groups, so that a multi-rank root can reduce panel p while the leaves still compute p + 1 ..
   std::vector<TaskList> sc_groups;
   std::vector<int> sc_row_begin;          // panel p = Schur rows [sc_row_begin[p], sc_row_begin[p + 1])
   std::vector<hipEvent_t> ev_sc;
   TileTask* d_sc_tasks = nullptr;
   int set_sc_panels(int n_panels) {
      if (!analyzed) PIPS_FAIL(PIPS_ERR_STATE, "set_sc_panels: analyze first");
      for (auto ev : ev_sc) (void)hipEventDestroy(ev);
      ev_sc.clear(); sc_groups.clear(); sc_row_begin.clear();
      if (d_sc_tasks) { (void)hipFree(d_sc_tasks); d_sc_tasks = nullptr; }
      n_panels = std::min(n_panels, std::max(1, S / (2 * TILE)));
      if (n_panels <= 1 || schur_mode_eff != 1) return PIPS_OK;
      for (int q = 0; q <= n_panels; ++q) sc_row_begin.push_back(q == n_panels ? S : (int)((long long)S * q / n_panels) / TILE * TILE);
      std::vector<std::vector<TileTask>> g(n_panels);
      for (int b = 0; b < nblk; ++b) {
         if (h_blks[b].ntc <= 0 || h_blks[b].nb <= 0) continue;
         const int nt = h_blks[b].nb_pad / TILE;
         for (int ti = 0; ti < nt; ++ti) {
            // a tile row belongs to the panel of its FIRST Schur row: every tile row that touches panel q then sits in a group <= q
            // (compressed border ids ascend like the Schur ids), so panel q is final once group q has run
            const int first = sym[b].bmap[std::min(ti * TILE, h_blks[b].nb - 1)];
            int q = (int)(std::upper_bound(sc_row_begin.begin(), sc_row_begin.end(), first) - sc_row_begin.begin()) - 1;
            q = std::max(0, std::min(q, n_panels - 1));
            for (int tj = 0; tj <= ti; ++tj) g[q].push_back({b, ti, tj, 0});
         }
      }
      std::vector<TileTask> all;
      for (int q = 0; q < n_panels; ++q) {
         TaskList l;
         l.off = (long long)all.size(); l.cnt = (int)g[q].size();
         all.insert(all.end(), g[q].begin(), g[q].end());
         sc_groups.push_back(l);
         hipEvent_t ev;
         HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
         ev_sc.push_back(ev);
      }
      if (all.empty()) all.push_back({-1, 0, 0, 0});
      return dev_upload(&d_sc_tasks, all, stream);
   }
   TailCtx ctx() {
      TailCtx c{d_blks, &plan, d_arena, d_dtail, d_winv, d_psign, d_psign_off, d_bmap, d_inertia, stream,
                timer.on ? &timer : nullptr, d_pref, side, ev_diag_in, ev_diag_out, false, d_sctab, d_uarena};
      if (!sc_groups.empty()) { c.sc_groups = &sc_groups; c.d_sc_tasks = d_sc_tasks; c.ev_sc = &ev_sc; }
      c.d_ctr_pool = d_gemm_ctr; c.ctr_cursor = &gemm_ctr_cursor;
      if (deterministic && d_gbuf) {
         c.det_rounds = &det_rounds; c.d_det_tasks = d_det_tasks; c.d_gbuf = d_gbuf; c.gstride = det_gstride(); c.sc_len = sc_len; c.d_blk_group = d_blk_group;
         c.n_groups = det_n_groups; c.first_slot = det_first_slot;
         c.det_defer_reduce = det_global && defer_group_reduce;   // (only the kkt paths finish the reduction themselves)
      }
      c.sweep = &sweep;
      if (tail_single) {
         tsingle.args.blks = d_blks; tsingle.args.arena = d_arena; tsingle.args.uarena = d_uarena; tsingle.args.winv = d_winv; tsingle.args.dtail = d_dtail;
         tsingle.args.pref = d_pref; tsingle.args.psign = d_psign; tsingle.args.psign_off = d_psign_off; tsingle.args.inertia = d_inertia;
         c.single = &tsingle;
      }
      return c;
   }

   int analyze_host(int n_threads, bool with_border = true) {
      sym.assign(nblk, BlockSym());
      std::vector<int> rc(nblk, 0);
      std::vector<std::string> msgs(nblk);
      n_threads = std::max(1, std::min(n_threads, nblk));
      auto work = [&](int t) {
         for (int b = t; b < nblk; b += n_threads) {
            CsrPattern K{in[b].n, in[b].n, in[b].krow.data(), in[b].kcol.data()};
            CsrPattern B{0, in[b].n, nullptr, nullptr};
            if (with_border && !in[b].btrow.empty()) B = CsrPattern{S, in[b].n, in[b].btrow.data(), in[b].btcol.data()};
            rc[b] = analyze_block(K, B, in[b].n_primal, opt, sym[b]);
            if (rc[b]) msgs[b] = last_error();
         }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
      work(0);
      for (auto& t : th) t.join();
      for (int b = 0; b < nblk; ++b)
         if (rc[b]) PIPS_FAIL(rc[b], "block %d: %s", b, msgs[b].c_str());
      return PIPS_OK;
   }

   // Schur contribution by the augmented partial factorisation (border rows ride in every panel: dense work, right when
   // the factor is dense anyway) or by blocked solves with the plain factor (the reference's way: 4 nnz(L) flops per border
   // column, right when L is sparse and L^-1 Br would fill in).  Estimated from the bordered symbolic analysis.
   bool blocked_solves_cheaper() const {
      double t_aug = 0.0, l_bytes = 0.0;
      int max_levels = 0, max_ntc = 0;
      std::vector<char> used(S, 0);
      for (int b = 0; b < nblk; ++b) {
         const BlockSym& s = sym[b];
         double pairs = 0.0, border_entries = 0.0;
         for (const HeadSupernode& sn : s.sn) {
            const double rbd = sn.r - sn.rb;
            // every scattered pair costs a w-long dot product besides its atomic (calibrated: 8e-11 s at w = 1, 1e-9 s at w = 25)
            pairs += (rbd * (sn.r - rbd) + 0.5 * rbd * rbd) * std::max(1.0, 0.5 * sn.w);
            border_entries += rbd * sn.w;
         }
         t_aug += opt.head_cost * pairs + s.flops_border / opt.mfma_rate;
         l_bytes += 8.0 * ((double)s.nnzL - border_entries);
         max_levels = std::max(max_levels, s.n_levels);
         max_ntc = std::max(max_ntc, s.m_pad / TILE);
         for (int c : s.bmap) used[c] = 1;
      }
      double ncols = 0;
      for (char u : used) ncols += u;
      const double launches = 2.0 * (max_levels + 2 * max_ntc) + 8;
      // multi-RHS sweeps over a sparse factor run at ~0.8 TB/s effective (measured, tools/banded_schur_probe.py)
      const double t_sol = ncols * 2.0 * l_bytes / 0.8e12 + std::ceil(ncols / 32.0) * launches * 12e-6;
      return t_sol < t_aug;
   }

   int analyze(int n_threads) {
      for (int b = 0; b < nblk; ++b)
         if (in[b].n <= 0) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_analyze: block %d was never set", b);
      apply_tuning(opt);
      if (sn_width > 0 && !getenv("PIPS_HIP_SN_WIDTH")) opt.max_sn_width = std::min(HEAD_WMAX, sn_width);
      bool any_border = false;
      for (int b = 0; b < nblk; ++b) any_border = any_border || !in[b].btrow.empty();
      {  // the border split (compact front panels) only exists with the multifrontal head
         if (env_int("PIPS_HIP_MF", 1) == 0) opt.mf_split_nb_max = 0;
      }
      // fronts on the rows of K only where the border split applies (PIPS_HIP_MF_KONLY=1; off by default: on the configs[3] share the fronts fall
      // from 9.7 to 4.0 ms, forming the border rows afterwards costs 8.0 - DESIGN.md 4.1c): not in deterministic mode (the border rows of the dense
      // tail take their head contributions with atomics, k_border_tail) and with supernodes of at most 16 columns (k_border_rows<., 16>)
      opt.mf_konly = opt.mf_split_nb_max > 0 && !deterministic && opt.max_sn_width <= 16 && env_int("PIPS_HIP_MF_KONLY", 0) != 0;
      int rc = analyze_host(n_threads, schur_mode != 2);
      if (rc) return rc;
      schur_mode_eff = (schur_mode == 2 && any_border) ? 2 : 1;
      if (schur_mode == 0 && any_border && blocked_solves_cheaper()) {
         schur_mode_eff = 2;
         if ((rc = analyze_host(n_threads, false))) return rc;
      }
      HIP_TRY(hipSetDevice(device));
      release();
      {  // multifrontal head: every block's fronts must fit the LDS; the slot machinery of deterministic mode records scatters
         mf = env_int("PIPS_HIP_MF", 1) != 0;
         const bool mf_wanted = mf;
         auto fronts_fit = [&]() {
            bool ok = mf_wanted;
            for (int b = 0; b < nblk && ok; ++b) {
               ok = sym[b].mf_ok;
               // fronts with very many leaves below them: the staged leaf data must fit beside the front
               for (size_t l = 0; l < sym[b].sn.size() && ok; ++l) {
                  if (sym[b].mf_meta[l] < 0) continue;
                  const HeadSupernode& s = sym[b].sn[l];
                  const int* H = sym[b].mf_int.data() + sym[b].mf_meta[l];
                  const long long fr = mf_rows(sym[b], s), nf = s.w + fr, pw = std::max<long long>(s.w * nf - (long long)s.w * (s.w - 1) / 2, (long long)s.w * ((fr + 3) / 4 * 4));
                  const long long packed = pw + mf_unp(sym[b], s) + 8, panel = pw + 8;
                  const long long extra = H[5] + (H[6] + H[3] + 1) / 2 + 2;
                  if ((packed > opt.mf_lds_doubles ? panel : packed) + extra > 20352) ok = false;   // 159 KB of the 160
               }
            }
            return ok;
         };
         mf = fronts_fit();
         bool any_split = false;
         for (int b = 0; b < nblk; ++b) any_split = any_split || sym[b].mf_split;
         // ... or k_border_schur's triangle + staged batch + row positions exceed the LDS (nb close to the cap under wide fronts whose
         // below-rows are nearly all border rows): the same formula the launch uses, evaluated here so that such an input is analysed with
         // whole update matrices instead of failing in every factor()
         const bool bb_too_big = mf && any_split && !bb_fits(bb_plan_size());
         if ((!mf && any_split) || bb_too_big) {   // every block back to full panels
            opt.mf_split_nb_max = 0;
            if ((rc = analyze_host(n_threads, schur_mode_eff != 2))) return rc;
            if (bb_too_big) mf = fronts_fit();   // (the fronts grew by their border columns: they must still fit)
         }
      }

      // ---- offsets
      h_blks.assign(nblk, BlkDesc());
      kptr.assign(nblk + 1, 0);
      x_off.assign(nblk + 1, 0);
      std::vector<long long> bptr(nblk + 1, 0), rows_base(nblk + 1, 0), sn_base(nblk + 1, 0), bmap_off(nblk + 1, 0),
         upd_base(nblk + 1, 0), mfU_base(nblk + 1, 0), mfint_base(nblk + 1, 0), mfLV_base(nblk + 1, 0);
      long long arena = 0, xw = 0, winv = 0, dt = 0, sncol = 0, uar = 0;
      for (int b = 0; b < nblk; ++b) {
         const BlockSym& s = sym[b];
         BlkDesc& d = h_blks[b];
         d.arena_off = arena;
         d.T = arena + s.T_off;
         d.sncol_off = sncol;
         d.xw_off = xw;
         d.x_off = x_off[b];
         d.bmap_off = bmap_off[b];
         d.winv_off = winv;
         d.dt_off = dt;
         d.n = s.n; d.n_head = s.n_head; d.m = s.m; d.m_pad = s.m_pad; d.nb = s.nb; d.nb_pad = s.nb_pad; d.ldT = s.ldT;
         d.ntc = s.m_pad / TILE;
         d.ntr = s.m > 0 ? s.ldT / TILE : 0;
         d.mf_split = (mf && s.mf_split) ? (s.mf_konly ? 2 : 1) : 0;   // 2: fronts on the rows of K only (k_border_rows forms their border rows)
         d.U = uar;
         uar += (long long)s.m_pad * s.m_pad;
         d.thr_rel = 0; d.repl_rel = 1e-8; d.repl_abs = 1;
         arena += s.arena;
         xw += s.n_head + s.m_pad + s.nb_pad;   // [head | padded tail | border rows (border-backward sweep only)]
         winv += (long long)d.ntc * TILE * TILE;
         dt += s.m_pad;
         sncol += s.n_head;
         kptr[b + 1] = kptr[b] + (long long)in[b].kcol.size();
         bptr[b + 1] = bptr[b] + (long long)in[b].btcol.size();
         x_off[b + 1] = x_off[b] + s.n;
         rows_base[b + 1] = rows_base[b] + (long long)s.rowidx.size();
         upd_base[b + 1] = upd_base[b] + (long long)s.upd.size();
         mfU_base[b + 1] = mfU_base[b] + (mf ? s.mf_U_total : 0);
         mfint_base[b + 1] = mfint_base[b] + (mf ? (long long)s.mf_int.size() : 0);
         mfLV_base[b + 1] = mfLV_base[b] + (mf ? s.mf_LV_total : 0);
         d.lv_off = mfLV_base[b];
         d.k_off = kptr[b]; d.b_off = bptr[b];
         sn_base[b + 1] = sn_base[b] + (long long)s.sn.size();
         bmap_off[b + 1] = bmap_off[b] + s.nb;
      }
      arena_total = arena; uarena_total = uar; xw_total = xw; n_total = x_off[nblk]; nnzK_total = kptr[nblk]; nnzB_total = bptr[nblk];
      // the concatenated CSR copies of K (refinement residual: both triangles) and of the borders are indexed with int32
      if (2 * nnzK_total > (long long)INT32_MAX || nnzB_total > (long long)INT32_MAX || n_total > (long long)INT32_MAX)
         PIPS_FAIL(PIPS_ERR_ARG, "batch too large for the 32-bit index arrays of one rank: sum nnz(K) %lld (limit 2^30), sum nnz(border) %lld, sum n %lld - "
                                 "use more ranks or fewer blocks per batch", nnzK_total, nnzB_total, n_total);
      nsn_total = (int)sn_base[nblk];

      // ---- supernodes sorted by (level, size class); multifrontal head: by (level, kernel variant, LDS need)
      struct Key { int level, cls, lds, blk, loc; };
      std::vector<Key> keys;
      keys.reserve(nsn_total);
      int nlev = 0;
      auto is_simple = [](const HeadSupernode& s) { return s.w == 1 && s.r <= SIMPLE_RMAX && s.level == 0; };
      // multifrontal: a level with few fronts is latency, not throughput - all its (LDS-resident) fronts go into ONE launch of the
      // largest variant any of them needs instead of one launch per variant
      constexpr int MF_MERGE_MAX = 1024;
      std::vector<int> lev_cnt, lev_b, lev_w;
      if (mf)
         for (int b = 0; b < nblk; ++b)
            for (const HeadSupernode& s : sym[b].sn) {
               if (is_simple(s)) continue;
               if ((int)lev_cnt.size() <= s.level) { lev_cnt.resize(s.level + 1, 0); lev_b.resize(s.level + 1, 0); lev_w.resize(s.level + 1, 0); }
               const int c = mf_class(s.w, s.w + mf_rows(sym[b], s), mf_unp(sym[b], s), opt.mf_lds_doubles);
               ++lev_cnt[s.level];
               if (c < 6) { lev_b[s.level] = std::max(lev_b[s.level], c % 3); lev_w[s.level] = std::max(lev_w[s.level], c / 3); }
            }
      for (int b = 0; b < nblk; ++b)
         for (int l = 0; l < (int)sym[b].sn.size(); ++l) {
            const HeadSupernode& s = sym[b].sn[l];
            // class 0: "simple leaf" (w = 1, r <= 16, level 0) -> one thread each;
            // class 1: small (one wave); class 2: large (256 threads)
            int cls = (s.w <= 8 && s.r <= 64) ? 1 : 2, lds = 0;
            if (mf && !is_simple(s)) {
               const long long fr = mf_rows(sym[b], s);
               int c = mf_class(s.w, s.w + fr, mf_unp(sym[b], s), opt.mf_lds_doubles);
               if (c < 6 && lev_cnt[s.level] <= MF_MERGE_MAX) c = lev_b[s.level] + 3 * lev_w[s.level];
               cls = 1 + c;
               // LDS of a front: the packed front (or its panel columns) + 8 doubles of slack, the leaves' values, and as ints the
               // children's position lists and the leaf part of the record (common.h "Front record")
               const long long nf = s.w + fr;
               const int* H = sym[b].mf_int.data() + sym[b].mf_meta[l];
               const long long pw = std::max<long long>(s.w * nf - (long long)s.w * (s.w - 1) / 2, (long long)s.w * ((fr + 3) / 4 * 4));   // packed panel / aligned L21 copy
               lds = (int)(pw + (c >= 6 ? 0 : mf_unp(sym[b], s)) + 8 + H[5] + (H[6] + H[3] + 1) / 2 + 2);
            }
            if (is_simple(s)) cls = 0;
            keys.push_back({s.level, cls, lds, b, l});
            nlev = std::max(nlev, s.level + 1);
         }
      std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
         return a.level != b.level ? a.level < b.level : (a.cls != b.cls ? a.cls < b.cls : a.lds < b.lds);
      });
      if (getenv("PIPS_HIP_DUMP_LEVELS")) {   // development aid: shape of the head, level by level
         std::vector<long long> cnt(nlev * 3, 0), rmax(nlev, 0), wsum(nlev, 0), pairs(nlev, 0);
         for (const Key& k : keys) {
            const HeadSupernode& s = sym[k.blk].sn[k.loc];
            ++cnt[k.level * 3 + std::min(k.cls, 2)];
            rmax[k.level] = std::max<long long>(rmax[k.level], s.r);
            wsum[k.level] += s.w;
            pairs[k.level] += (long long)s.r * (s.r + 1) / 2;
         }
         for (int l = 0; l < nlev; ++l)
            fprintf(stderr, "level %3d: simple %lld small %lld large %lld  columns %lld  max r %lld  update pairs %lld\n", l, cnt[3 * l], cnt[3 * l + 1],
                    cnt[3 * l + 2], wsum[l], rmax[l], pairs[l]);
      }
      // ---- spine: the top levels that hold at most two supernodes of every block (chain-like trees of time-coupled
      //      blocks).  One launch per level would be pure latency there; they go to the per-block spine kernels instead.
      n_levels_all = nlev;
      int lstar = nlev;
      {
         std::vector<int> width(nlev, 0);   // max over blocks of the supernode count per level
         std::vector<int> cnt(nlev);
         for (int b = 0; b < nblk; ++b) {
            std::fill(cnt.begin(), cnt.end(), 0);
            for (const HeadSupernode& s : sym[b].sn) ++cnt[s.level];
            for (int l = 0; l < nlev; ++l) width[l] = std::max(width[l], cnt[l]);
         }
         while (lstar > 0 && width[lstar - 1] <= 2) --lstar;
         const char* env = getenv("PIPS_HIP_SPINE");
         if (nlev - lstar < 8 || (env && atoi(env) == 0) || deterministic || mf) lstar = nlev;   // the spine kernels hand over through atomics
      }
      std::vector<SnDesc> h_sns(nsn_total);
      std::vector<std::vector<int>> roots_of(nblk);   // multifrontal head: fronts without a head parent, ascending
      long long slots_acc = 0, vslots_acc = 0;
      std::vector<std::vector<int>> sorted_id(nblk);
      for (int b = 0; b < nblk; ++b) sorted_id[b].resize(sym[b].sn.size());
      levels.assign(lstar, LevelRange{0, 0, 0, 0, 0, 0, 0, 0});
      levels_top.assign(nlev - lstar, LevelRange{0, 0, 0, 0, 0, 0, 0, 0});
      head_wcap = 1;
      for (int i = 0; i < nsn_total; ++i) {
         const Key& k = keys[i];
         const HeadSupernode& s = sym[k.blk].sn[k.loc];
         head_wcap = std::max(head_wcap, s.w);
         h_sns[i] = SnDesc{h_blks[k.blk].arena_off + s.panel, rows_base[k.blk] + s.rows, upd_base[k.blk] + s.upd, s.w, s.r, s.c0, k.blk,
                           s.n_useg, s.rb, s.ld, 0, slots_acc, vslots_acc, -1, -1, -1};
         if (mf && k.cls > 0 && s.r > 0 && sym[k.blk].sn_parent[k.loc] < 0) roots_of[k.blk].push_back(i);
         if (mf) {
            const BlockSym& bs = sym[k.blk];
            if (bs.mf_U[k.loc] >= 0) h_sns[i].U = (k.cls == 0 ? mfLV_base[k.blk] : mfU_base[k.blk]) + bs.mf_U[k.loc];
            if (bs.mf_meta[k.loc] >= 0) h_sns[i].mf = mfint_base[k.blk] + bs.mf_meta[k.loc];
         }
         // factorisation slots: every scattering supernode; multifrontal head: only the simple leaves without a front above them scatter
         // (the fronts hand their update matrices on, k_root_assemble adds the last ones in a fixed order)
         if (!mf || (k.cls == 0 && s.n_useg == 0)) slots_acc += (long long)s.r * (s.r + 1) / 2;
         vslots_acc += s.r;
         sorted_id[k.blk][k.loc] = i;
         LevelRange& L = k.level >= lstar ? levels_top[k.level - lstar] : levels[k.level];
         if (k.cls == 0) { if (L.simple_cnt++ == 0) L.simple_begin = i; }
         else if (k.cls == 1 || mf) { if (L.small_cnt++ == 0) L.small_begin = i; }   // multifrontal: one contiguous range of fronts per level
         else { if (L.large_cnt++ == 0) L.large_begin = i; }
         if (mf && k.cls > 0) {
            // one launch per (level, variant, LDS bucket): the dynamic LDS of a launch is that of its largest front, and it decides how
            // many fronts share a compute unit
            bool open = mf_launches.empty() || mf_launches.back().level != k.level || mf_launches.back().cls != k.cls - 1;
            if (!open) {
               const MfLaunch& m = mf_launches.back();
               const int first_lds = keys[m.begin].lds;
               {
                  // how many fronts of the variant share a compute unit: the LDS decides up to the limit the registers set (123 VGPRs: four
                  // waves per SIMD - four workgroups of 256 threads, sixteen of 64); a bucket = one such class, since inside a class a
                  // smaller front gains nothing from a launch of its own and across a boundary every front of the launch loses a slot
                  auto cls_of = [&](int lds_doubles) {
                     const int c = k.cls - 1, kmax = c == 0 ? 16 : c == 3 ? 12 : c == 1 ? 8 : c == 4 ? 6 : c == 2 ? 4 : 3;   // (154 VGPRs for the 32-wide variants)
                     return std::min(kmax, (int)(163840 / ((long long)lds_doubles * 8 + 1024)));
                  };
                  if (m.cnt >= 256 && cls_of(k.lds) < cls_of(first_lds)) open = true;
               }
            }
            if (open) mf_launches.push_back({k.level, k.cls - 1, i, 0, 0});
            ++mf_launches.back().cnt;
            mf_launches.back().lds_doubles = std::max(mf_launches.back().lds_doubles, k.lds);
         }
         const long long need = (long long)s.r * (s.w | 1);
         if (k.cls == 1) L.small_lds = (int)std::max<long long>(L.small_lds, std::min<long long>(need, 640));
         else if (k.cls == 2) L.large_lds = (int)std::max<long long>(L.large_lds, std::min<long long>(need, 6144));
      }
      if (mf) {
         std::vector<int> h_roots, h_root_off(nblk + 1, 0);
         for (int b = 0; b < nblk; ++b) {
            h_roots.insert(h_roots.end(), roots_of[b].begin(), roots_of[b].end());
            h_root_off[b + 1] = (int)h_roots.size();
         }
         n_roots = (int)h_roots.size();
         h_root_off_keep = h_root_off;
         if ((rc = dev_upload(&d_roots, h_roots, stream)) || (rc = dev_upload(&d_root_off, h_root_off, stream))) return rc;
         // border split: the supernodes whose border rows k_border_schur multiplies out - the fronts, and the simple leaves below a front
         // (a leaf without a front above it scatters its whole rank-one update itself), ascending; their border rows live a second
         // time in the border-row arena (per supernode w x rp doubles + w pivots, padded to even), cut into batches of up to BB_GMAX
         // supernodes / bb_stage doubles that the kernel stages as one contiguous piece
         std::vector<BbMeta> h_meta;
         std::vector<BbBatch> h_batch;
         std::vector<int> h_bbpos;
         h_bb_off_keep.assign(nblk + 1, 0);
         bb_stage = 3072; bb_nbmax = 0; bb_poscap = 0;
         // staging area: as much of the LDS as the packed triangle of the widest border leaves (a batch is one barrier pair and one request
         // latency whatever it holds; the supernodes of the upper levels take 2000+ doubles each), at most 6144 doubles, at least the largest
         // single supernode (bb_plan_size)
         { const BbPlanSize z = bb_plan_size(); bb_stage = z.stage; bb_two_per_cu = z.two_per_cu; }
         long long bb_total = arena_total;   // the border-row arena lives behind the panels in the same allocation (offsets like SnDesc::panel)
         for (int b = 0; b < nblk; ++b) {
            const BlockSym& bs = sym[b];
            if (bs.mf_split) {
               bb_nbmax = std::max(bb_nbmax, bs.nb);
               BbBatch cur{0, 0, 0, 0, 0, 0, 0, 0};
               auto flush = [&]() { if (cur.cnt > 0) { h_batch.push_back(cur); bb_poscap = std::max(bb_poscap, cur.npos); } cur = BbBatch{0, 0, 0, 0, 0, 0, 0, 0}; };
               for (int l = 0; l < (int)bs.sn.size(); ++l) {
                  const HeadSupernode& hs = bs.sn[l];
                  if (hs.rb >= hs.r || (is_simple(hs) && bs.sn_parent[l] < 0)) continue;
                  const int nbj = hs.r - hs.rb, rp = (nbj + 3) & ~3, sz = hs.w * rp + ((hs.w + 1) & ~1);
                  if (cur.cnt == BB_GMAX || cur.ndoubles + sz > bb_stage) flush();
                  if (cur.cnt == 0) { cur.src = bb_total; cur.pos = (long long)h_bbpos.size(); cur.first = (int)h_meta.size(); }
                  h_meta.push_back(BbMeta{cur.ndoubles, cur.npos, hs.w, nbj, cur.ntiles, 0, 0, 0});
                  h_sns[sorted_id[b][l]].bb = bb_total;
                  for (int a = hs.rb; a < hs.r; ++a) h_bbpos.push_back(bs.rowidx[hs.rows + a] - bs.n);
                  ++cur.cnt; cur.ndoubles += sz; cur.ntiles += bb_tile_count(rp); cur.npos += nbj;
                  bb_total += sz;
               }
               flush();
            }
            h_bb_off_keep[b + 1] = (int)h_batch.size();
         }
         n_bb = (int)h_batch.size();
         if (n_bb > 0) {
            if ((rc = dev_upload(&d_bb_batches, h_batch, stream)) || (rc = dev_upload(&d_bb_meta, h_meta, stream)) ||
                (rc = dev_upload(&d_bb_pos, h_bbpos, stream)) || (rc = dev_upload(&d_bb_off, h_bb_off_keep, stream))) return rc;
         }
         bb_doubles = bb_total - arena_total;
      }
      {  // gather-form metadata of the blocks whose fronts hold the rows of K only
         std::vector<int> h_rec, h_list, h_tail;
         std::vector<long long> h_off((size_t)std::max(nsn_total, 1), -1);
         std::vector<std::vector<int>> by_level(levels.size());
         kb_level_pairs.assign(levels.size(), 0); kb_level_lds.assign(levels.size(), 0);
         bool any = false;
         for (int b = 0; b < nblk && mf; ++b) {
            const BlockSym& bs = sym[b];
            if (!bs.mf_konly) continue;
            any = true;
            for (int l = 0; l < (int)bs.sn.size(); ++l) {
               if (bs.kb_off[l] < 0) continue;
               const int* R = bs.kb_rec.data() + bs.kb_off[l];
               const int np = R[0], ne = R[1];
               h_off[sorted_id[b][l]] = (long long)h_rec.size();
               h_rec.push_back(np); h_rec.push_back(ne);
               for (int q = 0; q < np; ++q) { h_rec.push_back(sorted_id[b][R[2 + 2 * q]]); h_rec.push_back(R[3 + 2 * q]); }
               h_rec.insert(h_rec.end(), R + 2 + 2 * np, R + 2 + 2 * np + 2 * ne);
               const HeadSupernode& sj = bs.sn[l];
               if (sj.level >= (int)by_level.size()) PIPS_FAIL(PIPS_ERR_STATE, "analyze: internal error, level of a front with border rows");
               by_level[sj.level].push_back(sorted_id[b][l]);
               kb_level_pairs[sj.level] = std::max(kb_level_pairs[sj.level], np);
               kb_level_lds[sj.level] = std::max(kb_level_lds[sj.level], (int)(sj.w * ((sj.r - sj.rb + 3) / 4 * 4) * sizeof(double)));
            }
            for (size_t q = 0; q + 1 < bs.kb_tail.size(); q += 2) { h_tail.push_back(sorted_id[b][bs.kb_tail[q]]); h_tail.push_back(bs.kb_tail[q + 1]); }
         }
         kb_level_off.assign(levels.size() + 1, 0);
         for (size_t l = 0; l < by_level.size(); ++l) { h_list.insert(h_list.end(), by_level[l].begin(), by_level[l].end()); kb_level_off[l + 1] = (int)h_list.size(); }
         n_kb_tail = (int)(h_tail.size() / 2);
         if (any) {
            if (h_rec.empty()) h_rec.push_back(0);
            if (h_list.empty()) h_list.push_back(0);
            if (h_tail.empty()) h_tail.push_back(0);
            if ((rc = dev_upload(&d_kb_rec, h_rec, stream)) || (rc = dev_upload(&d_kb_off, h_off, stream)) || (rc = dev_upload(&d_kb_list, h_list, stream)) ||
                (rc = dev_upload(&d_kb_tail, h_tail, stream))) return rc;
         }
      }
      // spine lists: per block, ascending local index = postorder (children before parents)
      std::vector<int> h_spine, h_spine_off(nblk + 1, 0);
      for (int b = 0; b < nblk; ++b) {
         for (int l = 0; l < (int)sym[b].sn.size(); ++l)
            if (sym[b].sn[l].level >= lstar) h_spine.push_back(sorted_id[b][l]);
         h_spine_off[b + 1] = (int)h_spine.size();
      }
      spine_total = (int)h_spine.size();
      // ---- the tails as one launch (tailkernel.hip.h; the default for batches of up to 16 blocks, slower than the column launches for the
      // large ones: DESIGN.md 4.2a): multifrontal head (every producer of the tail panel goes by BlkDesc::T_in), no deterministic mode (its slot records hold
      // panel addresses), room for a second copy of the tail panels
      {
         long long scratch = 0;
         for (int b = 0; b < nblk; ++b) scratch += sym[b].arena - sym[b].T_off;
         size_t free_b = 0, total_b = 0;
         (void)hipMemGetInfo(&free_b, &total_b);
         const double need = 8.0 * (double)(arena_total + bb_doubles + scratch + uarena_total) + 4e9;
         // few blocks: the column launches are a chain of ~4 launches per tile column whatever the batch holds, and the one launch wins
         // (leaf factorisation of configs[1] blocks, profiles/r6_tail_single_by_blocks.txt: 1 block 8.41 -> 7.48 ms, 4: 13.2 -> 12.1,
         // 8: 19.3 -> 17.8, 16: 31.1 -> 30.4; from 24 blocks on it loses: 43.0 -> 43.7, 32: 54.2 -> 56.4, 64: 101 -> 110).
         // PIPS_HIP_TAIL_SINGLE=0 / 1 forces one side.
         const int want_single = env_int("PIPS_HIP_TAIL_SINGLE", nblk <= 16 ? 1 : 0);
         tail_single = mf && !deterministic && scratch > 0 && want_single != 0 && need < (double)free_b;
         tail_scratch = tail_single ? scratch : 0;
         long long at = arena_total + bb_doubles;
         for (int b = 0; b < nblk; ++b) {
            h_blks[b].T_in = tail_single ? at : h_blks[b].T;
            at += sym[b].arena - sym[b].T_off;
         }
      }
      // ---- concatenated index arrays
      std::vector<int> h_rowidx, h_sncol, h_bmap, h_perm, h_upd;
      h_upd.reserve(upd_base[nblk]);
      std::vector<signed char> h_psign;
      std::vector<long long> h_psign_off(nblk), h_perm_off(nblk), h_kdst(nnzK_total), h_bdst(nnzB_total), h_kdiag(n_total),
         h_rowbase(n_total);
      std::vector<int> h_krowptr(n_total + 1), h_kcolidx(nnzK_total);
      h_rowidx.reserve(rows_base[nblk]);
      h_sncol.reserve(sncol);
      for (int b = 0; b < nblk; ++b) {
         const BlockSym& s = sym[b];
         h_rowidx.insert(h_rowidx.end(), s.rowidx.begin(), s.rowidx.end());
         h_upd.insert(h_upd.end(), s.upd.begin(), s.upd.end());
         for (int c = 0; c < s.n_head; ++c) h_sncol.push_back(sorted_id[b][s.sn_of_col[c]]);
         h_bmap.insert(h_bmap.end(), s.bmap.begin(), s.bmap.end());
         h_psign_off[b] = (long long)h_psign.size();
         h_psign.insert(h_psign.end(), s.psign.begin(), s.psign.end());
         h_perm_off[b] = (long long)h_perm.size();
         h_perm.insert(h_perm.end(), s.perm.begin(), s.perm.end());
         // multifrontal head: the fronts read their panel entries from the value arrays (k_front), nobody reads them from the arena
         // (entries of the tail panel land where the tail is assembled: BlkDesc::T_in)
         auto dst = [&](long long rel) { return rel >= s.T_off ? h_blks[b].T_in + (rel - s.T_off) : h_blks[b].arena_off + rel; };
         for (size_t p = 0; p < s.a_dst.size(); ++p) h_kdst[kptr[b] + p] = (mf && s.a_front[p]) ? -1 : dst(s.a_dst[p]);
         for (size_t p = 0; p < s.b_dst.size(); ++p) h_bdst[bptr[b] + p] = (mf && s.b_front[p]) ? -1 : dst(s.b_dst[p]);
         for (int i = 0; i < s.n; ++i) {
            long long dp = -1;
            for (int p = in[b].krow[i]; p < in[b].krow[i + 1]; ++p)
               if (in[b].kcol[p] == i) dp = kptr[b] + p;
            if (dp < 0) PIPS_FAIL(PIPS_ERR_ARG, "block %d row %d has no explicit diagonal entry (create_kkt always stores one)", b, i);
            h_kdiag[x_off[b] + i] = dp;
            h_rowbase[x_off[b] + i] = x_off[b];
            h_krowptr[x_off[b] + i] = (int)(kptr[b] + in[b].krow[i]);
         }
         std::copy(in[b].kcol.begin(), in[b].kcol.end(), h_kcolidx.begin() + kptr[b]);
      }
      h_krowptr[n_total] = (int)nnzK_total;
      if (mf) {
         std::vector<int> h_mfint((size_t)mfint_base[nblk]);
         for (int b = 0; b < nblk; ++b) {
            const BlockSym& s = sym[b];
            std::copy(s.mf_int.begin(), s.mf_int.end(), h_mfint.begin() + mfint_base[b]);
            for (int64_t pos : s.mf_fix) h_mfint[(size_t)(mfint_base[b] + pos)] = sorted_id[b][s.mf_int[(size_t)pos]];
         }
         mfU_total = mfU_base[nblk];
         if ((rc = dev_upload(&d_mfint, h_mfint, stream))) return rc;
         HIP_TRY(hipMalloc((void**)&d_mfU, (size_t)std::max<long long>(mfU_total, 1) * sizeof(double)));
         HIP_TRY(hipMalloc((void**)&d_mfLV, (size_t)std::max<long long>(mfLV_base[nblk], 1) * sizeof(double)));
      }
      {  // both triangles, row by row: entry (i, j) of the lower CSR also appears in row j as (j, i)
         std::vector<int> frp(n_total + 1, 0);
         for (int b = 0; b < nblk; ++b)
            for (int i = 0; i < sym[b].n; ++i)
               for (int p = in[b].krow[i]; p < in[b].krow[i + 1]; ++p) {
                  const int j = in[b].kcol[p];
                  ++frp[x_off[b] + i + 1];
                  if (j != i) ++frp[x_off[b] + j + 1];
               }
         for (long long r = 0; r < n_total; ++r) frp[r + 1] += frp[r];
         std::vector<int> fcol(frp[n_total]), fsrc(frp[n_total]), fill(frp.begin(), frp.end() - 1);
         for (int b = 0; b < nblk; ++b)
            for (int i = 0; i < sym[b].n; ++i)
               for (int p = in[b].krow[i]; p < in[b].krow[i + 1]; ++p) {
                  const int j = in[b].kcol[p], src = (int)(kptr[b] + p);
                  int q = fill[x_off[b] + i]++;
                  fcol[q] = j; fsrc[q] = src;
                  if (j != i) { q = fill[x_off[b] + j]++; fcol[q] = i; fsrc[q] = src; }
               }
         std::vector<long long> flong;
         for (long long r = 0; r < n_total; ++r)
            if (frp[r + 1] - frp[r] > FULL_LONG_ROW) flong.push_back(r);
         n_flong = (int)flong.size();
         if ((rc = dev_upload(&d_frowptr, frp, stream)) || (rc = dev_upload(&d_fcol, fcol, stream)) || (rc = dev_upload(&d_fsrc, fsrc, stream)) ||
             (rc = dev_upload(&d_flong, flong, stream)))
            return rc;
      }
      // border CSR, global
      std::vector<int> h_bt_rowptr, h_bt_colidx(nnzB_total), h_bt_rowsc;
      std::vector<long long> h_bt_xoff;
      std::vector<double> h_bval(nnzB_total);
      h_bt_rowptr.push_back(0);
      for (int b = 0; b < nblk; ++b) {
         if (in[b].btrow.empty()) continue;
         for (int s2 = 0; s2 < S; ++s2) {
            h_bt_rowptr.push_back((int)(bptr[b] + in[b].btrow[s2 + 1]));
            h_bt_rowsc.push_back(s2);
            h_bt_xoff.push_back(x_off[b]);
         }
         std::copy(in[b].btcol.begin(), in[b].btcol.end(), h_bt_colidx.begin() + bptr[b]);
         std::copy(in[b].btval.begin(), in[b].btval.end(), h_bval.begin() + bptr[b]);
      }
      bt_rows_total = (long long)h_bt_rowsc.size();
      h_bt_rowsc_keep = h_bt_rowsc;
      h_bt_rownnz_keep.assign((size_t)bt_rows_total, 0);
      for (long long i = 0; i < bt_rows_total; ++i) h_bt_rownnz_keep[(size_t)i] = h_bt_rowptr[i + 1] - h_bt_rowptr[i];
      h_bt_rowblk_keep.clear();
      for (int b = 0; b < nblk; ++b)
         if (!in[b].btrow.empty()) h_bt_rowblk_keep.insert(h_bt_rowblk_keep.end(), (size_t)S, b);
      if (deterministic && bt_rows_total > 0) {   // gather lists of the border products (see k_border_rowdot)
         std::vector<SlotEntry> by_sc, by_entry;
         for (long long i = 0; i < bt_rows_total; ++i) {
            if (h_bt_rowptr[i + 1] > h_bt_rowptr[i]) by_sc.push_back({(long long)h_bt_rowsc[i], i});
            for (int q = h_bt_rowptr[i]; q < h_bt_rowptr[i + 1]; ++q) by_entry.push_back({h_bt_xoff[i] + h_bt_colidx[q], (long long)q});
         }
         g_btm.release(); g_bm.release();
         if ((rc = upload_gather(by_sc, g_btm)) || (rc = upload_gather(by_entry, g_bm))) return rc;
         if (d_bt_tmp) (void)hipFree(d_bt_tmp);
         HIP_TRY(hipMalloc((void**)&d_bt_tmp, (size_t)std::max<long long>(std::max(bt_rows_total, nnzB_total), 1) * sizeof(double)));
      }
      {  // non-empty Schur columns over all blocks (the reference skips empty border columns, :870-874)
         std::vector<char> used(std::max(S, 1), 0);
         for (int b = 0; b < nblk; ++b)
            if (!in[b].btrow.empty())
               for (int s2 = 0; s2 < S; ++s2)
                  if (in[b].btrow[s2 + 1] > in[b].btrow[s2]) used[s2] = 1;
         schur_cols.clear();
         std::vector<int> slot(std::max(S, 1), -1);
         for (int s2 = 0; s2 < S; ++s2)
            if (used[s2]) { slot[s2] = (int)schur_cols.size(); schur_cols.push_back(s2); }
         if ((rc = dev_upload(&d_schur_cols, schur_cols, stream))) return rc;
         if ((rc = dev_upload(&d_schur_slot, slot, stream))) return rc;
      }

      // ---- device allocation / upload
      HIP_TRY(hipMalloc((void**)&d_arena, std::max<long long>(arena_total + bb_doubles + tail_scratch, 1) * sizeof(double)));
      if (tail_single) HIP_TRY(hipMemsetAsync(d_arena, 0, (size_t)arena_total * sizeof(double), stream));   // (tiles of the panels outside the envelopes are never written: they read as zero)
      if (bb_doubles > 0) HIP_TRY(hipMemsetAsync(d_arena + arena_total, 0, (size_t)bb_doubles * sizeof(double), stream));   // (the padding rows of the border-row arena stay zero)
      HIP_TRY(hipMalloc((void**)&d_uarena, std::max<long long>(uarena_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_kval, std::max<long long>(nnzK_total, 1) * sizeof(double)));
      HIP_TRY(hipMemset(d_kval, 0, std::max<long long>(nnzK_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_winv, std::max<long long>(winv, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_dtail, std::max<long long>(dt, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_xw, std::max<long long>(xw_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_pref, std::max<long long>(xw_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_rhs, std::max<long long>(n_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_res, std::max<long long>(n_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_stage, std::max<long long>(n_total, 1) * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_norms, (size_t)3 * nblk * sizeof(double)));
      HIP_TRY(hipHostMalloc((void**)&h_norms, (size_t)3 * nblk * sizeof(double), hipHostMallocDefault));
      HIP_TRY(hipMalloc((void**)&d_inertia, (size_t)3 * nblk * sizeof(int)));
      HIP_TRY(hipMemset(d_inertia, 0, (size_t)3 * nblk * sizeof(int)));
      if ((rc = dev_upload(&d_bval, h_bval, stream))) return rc;
      if ((rc = dev_upload(&d_kdst, h_kdst, stream))) return rc;
      if ((rc = dev_upload(&d_bdst, h_bdst, stream))) return rc;
      if ((rc = dev_upload(&d_kdiag, h_kdiag, stream))) return rc;
      if ((rc = dev_upload(&d_kptr, kptr, stream))) return rc;
      if ((rc = dev_upload(&d_psign_off, h_psign_off, stream))) return rc;
      if ((rc = dev_upload(&d_perm_off, h_perm_off, stream))) return rc;
      if ((rc = dev_upload(&d_rowbase, h_rowbase, stream))) return rc;
      if ((rc = dev_upload(&d_bt_xoff, h_bt_xoff, stream))) return rc;
      if ((rc = dev_upload(&d_sns, h_sns, stream))) return rc;
      if ((rc = dev_upload(&d_blks, h_blks, stream))) return rc;
      if ((rc = dev_upload(&d_rowidx, h_rowidx, stream))) return rc;
      {  // simple leaves that own border rows (sweeps of the augmented factor: k_leaf_border), and the widest padded border
         std::vector<int> lb;
         if (!levels.empty())
            for (int i = levels[0].simple_begin; i < levels[0].simple_begin + levels[0].simple_cnt; ++i)
               if (h_sns[i].rb < h_sns[i].r) lb.push_back(i);
         n_lb = (int)lb.size();
         if (d_lb_list) { (void)hipFree(d_lb_list); d_lb_list = nullptr; }
         if (n_lb > 0 && (rc = dev_upload(&d_lb_list, lb, stream))) return rc;
         nb_pad_max = 0;
         for (int b = 0; b < nblk; ++b) nb_pad_max = std::max(nb_pad_max, h_blks[b].nb_pad);
      }
      // ---- the simple leaves' L entries by target row (forward substitution as a gather, see d_lf_rows)
      {
         const LevelRange* L0 = levels.empty() ? nullptr : &levels[0];
         if (L0 && L0->simple_cnt > 0 && xw_total < (1LL << 31) && h_rowidx.size() < (1ull << 31)) {
            std::vector<LeafDesc> h_leaf((size_t)L0->simple_cnt);
            for (int i = 0; i < L0->simple_cnt; ++i) {
               const SnDesc& sn = h_sns[L0->simple_begin + i];
               const BlkDesc& bd = h_blks[sn.blk];
               int r_in = 0;
               while (r_in < sn.r && h_rowidx[sn.rows + r_in] < bd.n) ++r_in;
               h_leaf[i] = LeafDesc{sn.panel, (int)sn.rows, (int)bd.xw_off, sn.c0, r_in};
            }
            if ((rc = dev_upload(&d_leafdesc, h_leaf, stream))) return rc;
            std::vector<int> cnt((size_t)xw_total + 1, 0);
            long long nent = 0;
            for (int i = L0->simple_begin; i < L0->simple_begin + L0->simple_cnt; ++i) {
               const SnDesc& sn = h_sns[i];
               const BlkDesc& bd = h_blks[sn.blk];
               for (int a = 0; a < sn.r; ++a) {
                  const int ra = h_rowidx[sn.rows + a];
                  if (ra >= bd.n) break;
                  ++cnt[bd.xw_off + ra];
                  ++nent;
               }
            }
            if (nent > 0 && nent < (1LL << 31)) {
               std::vector<int> h_rows, h_ptr(1, 0), h_src((size_t)nent), h_pos(h_rowidx.size(), -1);
               std::vector<int> slot((size_t)xw_total, -1);   // target row -> its index in the compact list
               for (long long t = 0; t < xw_total; ++t)
                  if (cnt[t] > 0) { slot[t] = (int)h_rows.size(); h_rows.push_back((int)t); h_ptr.push_back(h_ptr.back() + cnt[t]); }
               std::vector<int> fill(h_ptr.begin(), h_ptr.end() - 1);
               for (int i = L0->simple_begin; i < L0->simple_begin + L0->simple_cnt; ++i) {   // ascending leaves: the order of every sum
                  const SnDesc& sn = h_sns[i];
                  const BlkDesc& bd = h_blks[sn.blk];
                  for (int a = 0; a < sn.r; ++a) {
                     const int ra = h_rowidx[sn.rows + a];
                     if (ra >= bd.n) break;
                     const int q = fill[slot[bd.xw_off + ra]]++;
                     h_src[q] = (int)(bd.xw_off + sn.c0);
                     h_pos[sn.rows + a] = q;
                  }
               }
               lf_rows = (long long)h_rows.size(); lf_entries = nent;
               if ((rc = dev_upload(&d_lf_rows, h_rows, stream))) return rc;
               if ((rc = dev_upload(&d_lf_ptr, h_ptr, stream))) return rc;
               if ((rc = dev_upload(&d_lf_src, h_src, stream))) return rc;
               if ((rc = dev_upload(&d_lf_pos, h_pos, stream))) return rc;
               HIP_TRY(hipMalloc((void**)&d_lf_val, (size_t)nent * sizeof(double)));
            }
         }
      }
      if ((rc = dev_upload(&d_upd, h_upd, stream))) return rc;
      if ((rc = dev_upload(&d_spine, h_spine, stream))) return rc;
      if ((rc = dev_upload(&d_spine_off, h_spine_off, stream))) return rc;
      if ((rc = dev_upload(&d_sncol, h_sncol, stream))) return rc;
      if ((rc = dev_upload(&d_bmap, h_bmap, stream))) return rc;
      if ((rc = dev_upload(&d_perm, h_perm, stream))) return rc;
      if ((rc = dev_upload(&d_psign, h_psign, stream))) return rc;
      if ((rc = dev_upload(&d_krowptr, h_krowptr, stream))) return rc;
      if ((rc = dev_upload(&d_kcolidx, h_kcolidx, stream))) return rc;
      if ((rc = dev_upload(&d_bt_rowptr, h_bt_rowptr, stream))) return rc;
      if ((rc = dev_upload(&d_bt_colidx, h_bt_colidx, stream))) return rc;
      if ((rc = dev_upload(&d_bt_rowsc, h_bt_rowsc, stream))) return rc;
      // ---- the border by LEAF row (t += alpha Br x0 as a gather, k_border_mult_rows): row pointers over the flat leaf space, Schur
      //      column and position in d_bval of every entry
      if (bt_rows_total > 0 && nnzB_total > 0 && nnzB_total < (1LL << 31)) {
         std::vector<int> rp((size_t)n_total + 1, 0);
         for (long long r = 0; r < bt_rows_total; ++r)
            for (int p = h_bt_rowptr[r]; p < h_bt_rowptr[r + 1]; ++p) ++rp[h_bt_xoff[r] + h_bt_colidx[p] + 1];
         for (long long i = 0; i < n_total; ++i) rp[i + 1] += rp[i];
         std::vector<int> sc((size_t)nnzB_total), src((size_t)nnzB_total), fill(rp.begin(), rp.end() - 1);
         for (long long r = 0; r < bt_rows_total; ++r)   // ascending (block, Schur column): the order of every row's sum
            for (int p = h_bt_rowptr[r]; p < h_bt_rowptr[r + 1]; ++p) {
               const int q = fill[h_bt_xoff[r] + h_bt_colidx[p]]++;
               sc[q] = h_bt_rowsc[r]; src[q] = p;
            }
         if ((rc = dev_upload(&d_br_rowptr, rp, stream)) || (rc = dev_upload(&d_br_sc, sc, stream)) || (rc = dev_upload(&d_br_src, src, stream)))
            return rc;
      }
      {
         std::vector<int> np(nblk);
         for (int b = 0; b < nblk; ++b) np[b] = in[b].n_primal;
         if ((rc = dev_upload(&d_nprimal, np, stream))) return rc;
      }
      // the diagonal tile of a column runs ahead on the side stream, its update first; tile rows start at their envelope (variants of the
      // tail factorisation without either: docs/HISTORY_r1_r2.md)
      std::vector<const std::vector<int>*> firsts(nblk);
      for (int b = 0; b < nblk; ++b) firsts[b] = &sym[b].tile_first;
      // tile columns per launch of the left-looking tail update (TailPlan::build, pair2).  configs[1], ms per unit / ms of update, two boxes:
      // 1 column 125.0 - 126.8 / 73.8 - 74.7; 2: 123.6 - 124.0 / 72.4; 3: 122.0 / 70.8; 4: 120.0 - 121.7 / 70.3 - 71.0; 6: 120.5 / 70.3; 8: 120.8 / 70.9
      const int pair2 = 4;   // (1: a column per launch; the switch that chose it went with round 6)
      if ((rc = plan.build(h_blks, 0, false, true, &firsts, true, pair2))) return rc;
      if ((rc = sweep.build(h_blks, &firsts))) return rc;
      if (tail_single) {
         if ((rc = tsingle.build(plan, h_blks, stream))) return rc;
         tsingle.args.poll_limit = env_int("PIPS_HIP_ROOT_POLL_LIMIT", 400000) > 0 ? (long long)env_int("PIPS_HIP_ROOT_POLL_LIMIT", 400000) * 50 : 0;   // (a deep update runs a millisecond)
         tsingle.args.diag_blocked = env_int("PIPS_HIP_ROOT_DIAG_BARRIERS", 0) ? 0 : 1;
      }
      {
         // border-backward sweep: worth it where the border rows of the factor (what it reads on top of a backward sweep) are no
         // more than what the forward sweep it saves would read, with a margin for the chain and the launches it also saves
         double fwd_entries = 0.0, border_entries = 0.0;
         for (int b = 0; b < nblk; ++b) {
            const BlockSym& sb = sym[b];
            fwd_entries += 0.5 * (double)sb.m_pad * sb.m_pad;
            border_entries += (double)sb.nb_pad * sb.m_pad;
            for (const HeadSupernode& sn : sb.sn) {
               const int nbord = sn.r - sn.rb;
               fwd_entries += (double)sn.w * (sn.r - nbord) + 0.5 * sn.w * sn.w;
               border_entries += (double)sn.w * nbord;
            }
         }
         // (round 4: 3 x instead of 1.25 x - with compact front panels and the border-row arena the sweep reads the border rows as one piece per supernode;
         //  on the configs[3] share, ratio 1.23, the witness pass of a factorisation drops from two refined solves to one + this sweep)
         border_backward_ok = schur_mode_eff == 1 && nnzB_total > 0 && (sweep.enabled || plan.ntc_max == 0) && !deterministic && border_entries <= 3.0 * fwd_entries;
         if (const char* bb = getenv("PIPS_HIP_BORDER_BACKWARD"))
            border_backward_ok = atoi(bb) != 0 && schur_mode_eff == 1 && nnzB_total > 0 && sweep.enabled && !deterministic;
         // Both halves of solveCompressed from the augmented factor (forward_augmented / backward_augmented): one forward and one backward
         // sweep that also read the border rows, instead of two full solves (two sweeps each, a residual check each, two border
         // products) - pays as long as the border rows are not several times what a sweep reads anyway
         const bool aug_paths = schur_mode_eff == 1 && nnzB_total > 0 && (sweep.enabled || plan.ntc_max == 0) && spine_total == 0 &&
                                (deterministic || !head_slots);   // (deterministic mode: forward_augmented_det, if set_det_groups can build its lists)
         aug_sweeps_ok = aug_paths && border_entries <= 3.0 * fwd_entries;
         if (const char* as = getenv("PIPS_HIP_AUG_SWEEPS")) aug_sweeps_ok = atoi(as) != 0 && aug_paths;
      }
      if (!side) {
         // highest priority: the few workgroups of the diagonal chain must not queue behind the thousands of the column update
         int prio_lo = 0, prio_hi = 0;
         HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
         HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_hi));
         HIP_TRY(hipEventCreateWithFlags(&ev_diag_in, hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&ev_diag_out, hipEventDisableTiming));
      }
      h_inertia.assign(3 * nblk, 0);
      slots_total = slots_acc; vslots_total = vslots_acc;
      h_sns_keep = h_sns;
      analyzed = true;
      ++analysis_gen;
      factored = false;
      head_slots = false;
      const bool want_slots = deterministic;
      if (want_slots && slots_total + vslots_total <= HEAD_SLOTS_MAX) {
         if ((rc = build_deterministic(n_threads))) return rc;
         head_slots = true;
      } else if (deterministic)
         PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode: %lld head contributions exceed the slot budget of %lld", slots_total + vslots_total, HEAD_SLOTS_MAX);
      if (deterministic && (rc = set_det_groups(0, 1))) return rc;
      return PIPS_OK;
   }

   std::vector<SnDesc> h_sns_keep;
   ScatterCtx sx_atomic() const { return ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}; }
   void launch_head_level(const LevelRange& L, double* SC, int ldSC, const ScatterCtx& sx) {
      if (L.simple_cnt > 0)
         hipLaunchKernelGGL(k_head_factor_simple, dim3((L.simple_cnt + 255) / 256), dim3(256), 0, stream, d_sns, L.simple_begin,
                            L.simple_cnt, d_blks, d_rowidx, d_upd, d_psign, d_psign_off, d_bmap, d_arena, SC, ldSC, d_inertia, d_pref, d_sctab, sx,
                            mf ? 1 : 0, d_mfLV, d_lf_pos, d_lf_val, d_arena);
      if (mf) return;
      if (L.small_cnt > 0)
         hipLaunchKernelGGL((k_head_factor<64, 8, 640>), dim3(L.small_cnt), dim3(64), (size_t)std::max(L.small_lds, 1) * sizeof(double), stream, d_sns,
                            L.small_begin, d_blks, d_rowidx, d_upd, d_psign, d_psign_off, d_bmap, d_arena, SC, ldSC, d_inertia, d_pref, d_sctab, sx,
                            L.small_lds);
      if (L.large_cnt > 0)   // the L21 cache is sized to the widest panel of this launch: small panels, many workgroups per compute unit
         hipLaunchKernelGGL((k_head_factor<256, 32, 6144>), dim3(L.large_cnt), dim3(256), (size_t)std::max(L.large_lds, 1) * sizeof(double), stream, d_sns,
                            L.large_begin, d_blks, d_rowidx, d_upd, d_psign, d_psign_off, d_bmap, d_arena, SC, ldSC,
                            d_inertia, d_pref, d_sctab, sx, L.large_lds);
   }
   template <int BLOCK, int WMAX, bool UG = false>
   int launch_front(const MfLaunch& m, double* SC, int ldSC) {
      const size_t lds = (size_t)m.lds_doubles * sizeof(double);
      if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_front<BLOCK, WMAX, UG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_front<BLOCK, WMAX, UG>), dim3(m.cnt), dim3(BLOCK), lds, stream, d_sns, m.begin, d_blks, d_rowidx, d_mfint, d_psign,
                         d_psign_off, d_bmap, d_arena, d_mfU, SC, ldSC, d_inertia, d_pref, d_sctab, d_mfdbg, d_mfLV, d_kval, d_bval, deterministic ? 1 : 0, d_arena);
      return PIPS_OK;
   }
   // ---- k_border_schur's LDS need, from the symbolic analysis alone (the same rules as the batching loop of analyze()): staging area,
   //      most row positions of a batch, widest border.  Evaluated at analyze time: a block set whose border rows do not fit (nb near 176
   //      under wide fronts that are nearly all border rows) goes back to whole update matrices there instead of failing in every factor()
   struct BbPlanSize { int stage = 3072, poscap = 0, nbmax = 0; bool two_per_cu = false; };
   BbPlanSize bb_plan_size() const {
      BbPlanSize z;
      for (int b = 0; b < nblk; ++b) if (sym[b].mf_split) z.nbmax = std::max(z.nbmax, sym[b].nb);
      const long long tri = ((long long)z.nbmax * (z.nbmax + 1) / 2 + 1) & ~1LL;
      const long long room = 19200 - tri - 4 * 512 / 2 - 64;   // (positions: up to 4 * 512 ints; supernode records)
      z.stage = (int)std::max<long long>(3072, std::min<long long>(6144, room)) & ~1;
      // two workgroups on a compute unit where half the LDS leaves a staging area of 3072 doubles or more: the walk is a chain of
      // barriers and request latencies per batch, a second workgroup fills them (configs[3] shape, nb = 103: k_border_schur 3.3 -> 2.1 ms
      // with 3072 - 4096 doubles and two workgroups per block; 2048 doubles and two or three: 2.7 - 3.2 ms)
      const long long room2 = 9600 - tri - 4 * 512 / 2 - 64;
      if (room2 >= 3072) { z.stage = (int)std::min<long long>(4096, room2) & ~1; z.two_per_cu = true; }
      for (int b = 0; b < nblk; ++b) {
         const BlockSym& bs = sym[b];
         if (!bs.mf_split) continue;
         for (const HeadSupernode& hs : bs.sn)
            if (hs.rb < hs.r) {
               const int need = hs.w * ((hs.r - hs.rb + 3) / 4 * 4) + ((hs.w + 1) & ~1);
               if (need > z.stage) { z.stage = need; z.two_per_cu = false; }   // (a supernode beyond the half-LDS area: back to one workgroup's rule)
            }
      }
      if (!z.two_per_cu) z.stage = std::max<int>(z.stage, (int)std::max<long long>(3072, std::min<long long>(6144, room)) & ~1);
      for (int b = 0; b < nblk; ++b) {
         const BlockSym& bs = sym[b];
         if (!bs.mf_split) continue;
         int cnt = 0, nd = 0, np = 0;
         for (int l = 0; l < (int)bs.sn.size(); ++l) {
            const HeadSupernode& hs = bs.sn[l];
            const bool simple = hs.w == 1 && hs.r <= SIMPLE_RMAX && hs.level == 0;    // (analyze()'s is_simple)
            if (hs.rb >= hs.r || (simple && bs.sn_parent[l] < 0)) continue;
            const int nbj = hs.r - hs.rb, rp = (nbj + 3) & ~3, sz = hs.w * rp + ((hs.w + 1) & ~1);
            if (cnt == BB_GMAX || nd + sz > z.stage) { z.poscap = std::max(z.poscap, np); cnt = nd = np = 0; }
            ++cnt; nd += sz; np += nbj;
         }
         z.poscap = std::max(z.poscap, np);
      }
      return z;
   }
   static size_t bb_lds_bytes(const BbPlanSize& z) {
      const long long ncp = ((long long)z.nbmax * (z.nbmax + 1) / 2 + 1) & ~1LL;
      return (size_t)(ncp + z.stage) * sizeof(double) + (size_t)((z.poscap + 3) & ~3) * sizeof(int) + BB_GMAX * sizeof(BbMeta);
   }
   static bool bb_fits(const BbPlanSize& z) { return bb_lds_bytes(z) <= 160 * 1024 && z.poscap <= 4 * 512 && z.stage <= 2 * 6 * 512; }
   int launch_border_schur(double* SC, int ldSC) {
      // LDS: packed nb x nb triangle + one staged batch + its rows' positions + the batch's supernode records
      constexpr int BLK = 512;
      const long long ncp = ((long long)bb_nbmax * (bb_nbmax + 1) / 2 + 1) & ~1LL;
      // (the triangle shared by 2 / 3 / 4 workgroups by column ranges, so that several fit a compute unit, measured 14.97 / 17.7 / 18.3 ms
      // against 14.94 on the 256-block chain's head: every part stages every batch - docs/HISTORY_r4.md; one workgroup holds it whole)
      constexpr int parts = 1;
      const long long c_cap = ncp;
      BbPlanSize z; z.stage = bb_stage; z.poscap = bb_poscap; z.nbmax = bb_nbmax;
      const size_t lds = bb_lds_bytes(z);
      if (!bb_fits(z))   // (analyze() has checked the same formula and taken the split off where it does not hold: an assertion)
         PIPS_FAIL(PIPS_ERR_STATE, "k_border_schur: %zu bytes of LDS for nb = %d (batch of %d doubles, %d rows)", lds, bb_nbmax, bb_stage, bb_poscap);
      // a block's batches are walked by `split` workgroups (each with its own accumulator): enough of them to fill the chip
      const int split = std::max(1, std::min(32, (bb_two_per_cu ? 512 : 256) / std::max(nblk, 1)));
      auto go = [&](auto kern, int cnt, const int* list, double* gb, long long gs, const int* grp, int sp, int ordered, double* out = nullptr,
                    long long out_stride = 0) -> int {
         if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
         hipLaunchKernelGGL(kern, dim3(cnt, sp, parts), dim3(BLK), lds, stream, list, d_bb_off, d_bb_batches, d_bb_meta, d_bb_pos, d_blks, d_bmap, d_arena, SC, ldSC,
                            d_sctab, gb, gs, grp, bb_stage, bb_poscap, ordered, out, out_stride, (int)c_cap);
         return PIPS_OK;
      };
      const bool small = bb_stage <= 2 * 3 * BLK;
      int rc = PIPS_OK;
      if (deterministic && d_gbuf && d_bb_round_blk) {
         // every block at once (each supernode list in its fixed order) into a triangle of its own, then the blocks of a group in order: a walk
         // takes as long for 8 workgroups as for 256, so one launch per round of blocks made the share's factorisation 230 ms
         const long long tri = (long long)ncp;
         if (!d_bb_out) HIP_TRY(hipMalloc((void**)&d_bb_out, (size_t)std::max(nblk, 1) * tri * sizeof(double)));
         rc = small ? go(k_border_schur<BLK, 3>, nblk, (const int*)nullptr, (double*)nullptr, 0LL, (const int*)nullptr, 1, 1, d_bb_out, tri)
                    : go(k_border_schur<BLK, 6>, nblk, (const int*)nullptr, (double*)nullptr, 0LL, (const int*)nullptr, 1, 1, d_bb_out, tri);
         for (size_t k = 0; k + 1 < bb_round_off.size() && !rc; ++k) {
            const int cnt = bb_round_off[k + 1] - bb_round_off[k];
            if (cnt > 0)
               hipLaunchKernelGGL(k_border_schur_add, dim3(cnt, 8), dim3(256), 0, stream, d_bb_round_blk + bb_round_off[k], d_blks, d_bmap, d_bb_out, tri, ldSC, d_sctab,
                                  d_gbuf, det_gstride(), d_blk_group);
         }
      } else
         rc = small ? go(k_border_schur<BLK, 3>, nblk, (const int*)nullptr, (double*)nullptr, 0LL, (const int*)nullptr, split, deterministic ? 1 : 0)
                    : go(k_border_schur<BLK, 6>, nblk, (const int*)nullptr, (double*)nullptr, 0LL, (const int*)nullptr, split, deterministic ? 1 : 0);
      return rc;
   }
   long long* d_mfdbg = nullptr;   // PIPS_HIP_MF_CLOCKS: phase stamps of every front (8 per supernode), dumped after the factorisation
   int dump_front_clocks() {
      std::vector<long long> h((size_t)nsn_total * 8);
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipMemcpy(h.data(), d_mfdbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      static const char* names[7] = {"panel+zero", "stage ints", "leaf values + children", "leaf columns", "pivots", "update", "write-out"};
      for (const MfLaunch& m : mf_launches) {
         double ph[8] = {0}; long long nfs = 0, ws = 0;
         for (int i = m.begin; i < m.begin + m.cnt; ++i) {
            for (int q = 0; q < 7; ++q) ph[q] += (double)(h[(size_t)i * 8 + q + 1] - h[(size_t)i * 8 + q]);
            nfs += h_sns_keep[i].w + h_sns_keep[i].r; ws += h_sns_keep[i].w;
         }
         fprintf(stderr, "[mf clocks] level %2d variant %d fronts %6d lds %6d  avg nf %5.1f w %4.1f | ", m.level, m.cls, m.cnt, m.lds_doubles * 8, (double)nfs / m.cnt, (double)ws / m.cnt);
         for (int q = 0; q < 7; ++q) fprintf(stderr, "%s %.1f%s", names[q], ph[q] / m.cnt * 0.01, q == 6 ? " us\n" : ", ");
      }
      return PIPS_OK;
   }
   // the border rows of the fronts of one level that hold the rows of K only (k_border_rows): a wave per pair - one wave per front where the
   // fronts of the level have a few pairs, four or eight where some have many
   void launch_border_rows(size_t l, hipStream_t st) {
      const int cnt = kb_level_off[l + 1] - kb_level_off[l];
      if (cnt == 0) return;
      const int* lst = d_kb_list + kb_level_off[l];
      if (kb_level_pairs[l] <= 3 || (cnt >= 16384 && kb_level_pairs[l] <= 8))   // (one wave for every level of >= 4096 fronts: head 17.5 against 16.6 ms)
         hipLaunchKernelGGL((k_border_rows<64, 16>), dim3(cnt), dim3(64), kb_level_lds[l], st, lst, d_sns, d_blks, d_rowidx, d_kb_rec, d_kb_off, d_arena, d_arena, d_bval);
      else if (cnt <= 4096 && kb_level_pairs[l] > 8)
         hipLaunchKernelGGL((k_border_rows<512, 16>), dim3(cnt), dim3(512), kb_level_lds[l], st, lst, d_sns, d_blks, d_rowidx, d_kb_rec, d_kb_off, d_arena, d_arena, d_bval);
      else
         hipLaunchKernelGGL((k_border_rows<256, 16>), dim3(cnt), dim3(256), kb_level_lds[l], st, lst, d_sns, d_blks, d_rowidx, d_kb_rec, d_kb_off, d_arena, d_arena, d_bval);
   }
   int launch_fronts(int level, double* SC, int ldSC) {
      int rc = PIPS_OK;
      for (const MfLaunch& m : mf_launches) {
         if (m.level != level) continue;
         switch (m.cls) {
            case 0: rc = launch_front<64, 16>(m, SC, ldSC); break;
            case 1: rc = launch_front<128, 16>(m, SC, ldSC); break;
            case 2: rc = launch_front<256, 16>(m, SC, ldSC); break;
            case 3: rc = launch_front<64, 32>(m, SC, ldSC); break;
            case 4: rc = launch_front<128, 32>(m, SC, ldSC); break;
            case 5: rc = launch_front<256, 32>(m, SC, ldSC); break;
            case 6: rc = launch_front<256, 32, true>(m, SC, ldSC); break;
            default: rc = launch_front<512, 32, true>(m, SC, ldSC); break;
         }
         if (rc) return rc;
      }
      return PIPS_OK;
   }
   void gather(const GatherList& g, const double* vals, double* target) {
      if (g.n_targets > 0)
         hipLaunchKernelGGL(k_gather_slots, dim3(grid_for(g.n_targets, 256)), dim3(256), 0, stream, g.n_targets, g.d_tgt, g.d_off, g.d_slots, vals, target);
   }

   // the recorded contributions (target, slot) are sorted, then turned into the CSR "target -> its slots"
   int upload_gather(std::vector<SlotEntry>& e, GatherList& g) {
      std::sort(e.begin(), e.end(), [](const SlotEntry& a, const SlotEntry& b) { return a.target != b.target ? a.target < b.target : a.slot < b.slot; });
      std::vector<long long> tgt, off, sl(e.size());
      for (size_t i = 0; i < e.size(); ++i) {
         if (i == 0 || e[i].target != e[i - 1].target) { tgt.push_back(e[i].target); off.push_back((long long)i); }
         sl[i] = e[i].slot;
      }
      off.push_back((long long)e.size());
      g.n_targets = (long long)tgt.size(); g.n_slots = (long long)e.size();
      if (tgt.empty()) return PIPS_OK;
      int rc;
      if ((rc = dev_upload(&g.d_tgt, tgt, stream)) || (rc = dev_upload(&g.d_off, off, stream)) || (rc = dev_upload(&g.d_slots, sl, stream))) return rc;
      return PIPS_OK;
   }
   int build_deterministic(int n_threads) {
      if (schur_mode_eff != 1) PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode needs Schur mode 1 (augmented factorisation)");
      if (spine_total > 0) PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode: spine kernels must be off");
      const int nlev = (int)levels.size();
      g_levels.assign(nlev, GatherList());
      gv_levels.assign(nlev, GatherList());
      int rc = PIPS_OK;
      // ---- factorisation scatter: record where every slot goes (structure only; the numbers this pass produces are discarded)
      std::vector<long long> rec((size_t)std::max<long long>(slots_total, 1), -1);
      if (slots_total > 0) {
         long long* d_rec = nullptr;
         HIP_TRY(hipMalloc((void**)&d_rec, (size_t)slots_total * sizeof(long long)));
         HIP_TRY(hipMemsetAsync(d_rec, 0xff, (size_t)slots_total * sizeof(long long), stream));
         double* fakeSC = d_arena;   // only offsets relative to it are formed (S > 0 blocks have border rows)
         const ScatterCtx sx{1, d_rec, nullptr, d_arena, fakeSC};
         for (const LevelRange& L : levels) launch_head_level(L, fakeSC, S, sx);
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipStreamSynchronize(stream));
         HIP_TRY(hipMemcpy(rec.data(), d_rec, (size_t)slots_total * sizeof(long long), hipMemcpyDeviceToHost));
         (void)hipFree(d_rec);
         HIP_TRY(hipMalloc((void**)&d_slot_val, (size_t)slots_total * sizeof(double)));
      }
      // classify: Schur complement / tail of block b / head panel of a supernode at level l (targets of different blocks are disjoint)
      std::vector<std::vector<SlotEntry>> per_level(nlev);
      std::vector<SlotEntry> tail_e, sc_e;
      sc_blk_keep.clear();
      std::vector<std::vector<std::pair<long long, int>>> panel_level(nblk);   // (panel offset inside the block arena, level), ascending
      for (int b = 0; b < nblk; ++b) {
         for (const HeadSupernode& hs : sym[b].sn) panel_level[b].push_back({hs.panel, hs.level});
         std::sort(panel_level[b].begin(), panel_level[b].end());
      }
      for (int i = 0; i < nsn_total; ++i) {
         const SnDesc& sn = h_sns_keep[i];
         if (mf && !(sn.mf < 0 && sn.n_useg == 0)) continue;   // multifrontal head: only the leaves without a front above them own slots
         const long long cnt = (long long)sn.r * (sn.r + 1) / 2;
         const BlkDesc& bd = h_blks[sn.blk];
         for (long long q = 0; q < cnt; ++q) {
            const long long t = rec[(size_t)(sn.slot + q)];
            if (t < 0) continue;
            if (t & SCATTER_SC_FLAG) { sc_e.push_back({t & ~SCATTER_SC_FLAG, sn.slot + q}); sc_blk_keep.push_back(sn.blk); continue; }
            if (t >= bd.T) { tail_e.push_back({t, sn.slot + q}); continue; }
            const long long rel = t - bd.arena_off;
            auto& pl = panel_level[sn.blk];
            auto it = std::upper_bound(pl.begin(), pl.end(), std::make_pair(rel, INT32_MAX));
            if (it == pl.begin()) PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode: a recorded target lies outside every head panel");
            per_level[std::prev(it)->second].push_back({t, sn.slot + q});
         }
      }
      std::vector<long long>().swap(rec);
      for (int l = 0; l < nlev; ++l)
         if ((rc = upload_gather(per_level[l], g_levels[l]))) return rc;
      sc_e_keep = sc_e;   // (unsorted, parallel to sc_blk_keep) for the group-wise variant of deterministic mode
      if ((rc = upload_gather(tail_e, g_tail)) || (rc = upload_gather(sc_e, g_sc))) return rc;
      // ---- forward-substitution scatter: same recording, targets are entries of the permuted work vector
      if (vslots_total > 0) {
         long long* d_rec = nullptr;
         HIP_TRY(hipMalloc((void**)&d_rec, (size_t)vslots_total * sizeof(long long)));
         HIP_TRY(hipMemsetAsync(d_rec, 0xff, (size_t)vslots_total * sizeof(long long), stream));
         const ScatterCtx sxv{1, d_rec, nullptr, d_xw, nullptr};
         for (const LevelRange& L : levels) {
            // (the simple leaves own no slots where their forward substitution is the gather by target row: fixed order by construction)
            if (L.simple_cnt > 0 && lf_rows == 0)
               hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin,
                                  L.simple_cnt, d_blks, d_rowidx, d_arena, d_xw, 0LL, 0, sxv);
            const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
            const int cnt = L.small_cnt + L.large_cnt;
            if (cnt > 0) hipLaunchKernelGGL(k_head_fwd, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin, d_blks, d_rowidx, d_arena, d_xw, 0LL, sxv);
         }
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipStreamSynchronize(stream));
         std::vector<long long> vrec((size_t)vslots_total);
         HIP_TRY(hipMemcpy(vrec.data(), d_rec, (size_t)vslots_total * sizeof(long long), hipMemcpyDeviceToHost));
         (void)hipFree(d_rec);
         HIP_TRY(hipMalloc((void**)&d_vslot_val, (size_t)vslots_total * sizeof(double)));
         std::vector<std::vector<SlotEntry>> v_level(nlev);
         std::vector<SlotEntry> v_tail;
         for (int i = 0; i < nsn_total; ++i) {
            const SnDesc& sn = h_sns_keep[i];
            const BlkDesc& bd = h_blks[sn.blk];
            for (int a = 0; a < sn.r; ++a) {
               const long long t = vrec[(size_t)(sn.vslot + a)];
               if (t < 0) continue;
               const long long col = t - bd.xw_off;
               if (col >= bd.n_head) { v_tail.push_back({t, sn.vslot + a}); continue; }
               const int loc = sym[sn.blk].sn_of_col[(size_t)col];
               v_level[sym[sn.blk].sn[loc].level].push_back({t, sn.vslot + a});
            }
         }
         for (int l = 0; l < nlev; ++l)
            if ((rc = upload_gather(v_level[l], gv_levels[l]))) return rc;
         if ((rc = upload_gather(v_tail, gv_tail))) return rc;
      }
      (void)n_threads;
      return PIPS_OK;
   }

   // Sparse Schur complement: tab holds, block after block, the nb x nb table "position of entry (la, lb), la >= lb, of this
   // block's contribution inside the value array of SC's lower-triangular CSR"; factor(values, 0) then accumulates there.
   int set_sc_tables(const std::vector<int>& tab, const std::vector<long long>& off, long long sc_nnz) {
      if (!analyzed) PIPS_FAIL(PIPS_ERR_STATE, "set_sc_tables: analyze first");
      if (schur_mode_eff != 1) PIPS_FAIL(PIPS_ERR_STATE, "a sparse Schur complement needs Schur mode 1 (set it before analyze)");
      HIP_TRY(hipSetDevice(device));
      if (d_sctab) { (void)hipFree(d_sctab); d_sctab = nullptr; }
      int rc = dev_upload(&d_sctab, tab, stream);
      if (rc) return rc;
      for (int b = 0; b < nblk; ++b) h_blks[b].sctab_off = off[b];
      HIP_TRY(hipMemcpy(d_blks, h_blks.data(), (size_t)nblk * sizeof(BlkDesc), hipMemcpyHostToDevice));
      sc_len = sc_nnz;
      if (head_slots) {   // the Schur targets of the head moved into the CSR value array: record again
         for (auto& g : g_levels) g.release();
         for (auto& g : gv_levels) g.release();
         g_tail.release(); g_sc.release(); gv_tail.release();
         if (d_slot_val) (void)hipFree(d_slot_val);
         if (d_vslot_val) (void)hipFree(d_vslot_val);
         if (d_mvslot) (void)hipFree(d_mvslot);
         d_slot_val = d_vslot_val = d_mvslot = nullptr;
         if ((rc = build_deterministic(1))) return rc;
      }
      return PIPS_OK;
   }

   int factor(double* SC, int ldSC) {
      if (!analyzed) PIPS_FAIL(PIPS_ERR_STATE, "factor called before analyze");
      HIP_TRY(hipSetDevice(device));
      if (deterministic && SC && S > 0 && !d_gbuf && d_blk_group)
         HIP_TRY(hipMalloc((void**)&d_gbuf, (size_t)det_n_groups * det_gstride() * sizeof(double)));
      timer.reset();
      if (timer.on) timer.begin(stream, 6);
      if (timer.on) timer.begin(stream, 0);
      hipLaunchKernelGGL(k_block_absmax_init, dim3((nblk + 255) / 256), dim3(256), 0, stream, d_blks, nblk);
      hipLaunchKernelGGL(k_block_absmax, dim3(std::max(1, std::min(64, (int)(nnzK_total / nblk / 4096))), nblk), dim3(256), 0, stream, d_kval,
                         d_kptr, d_blks);
      hipLaunchKernelGGL(k_block_absmax_finish, dim3((nblk + 255) / 256), dim3(256), 0, stream, d_blks, nblk, thr_rel, repl_rel);
      // multifrontal head: no panel of the head is read before it is written (simple leaves: every entry of the panel is an entry of K or
      // of the border and comes with k_scatter; fronts: assembled in LDS, written out whole) - only the tails are cleared
      hipLaunchKernelGGL(k_arena_clear, dim3(256, nblk), dim3(256), 0, stream, d_blks, d_arena, mf ? 1 : 0);
      HIP_TRY(hipMemsetAsync(d_inertia, 0, (size_t)3 * nblk * sizeof(int), stream));
      {
         if (!d_gemm_ctr) HIP_TRY(hipMalloc((void**)&d_gemm_ctr, (size_t)GEMM_CTR_SLOTS * 8 * sizeof(int)));
         HIP_TRY(hipMemsetAsync(d_gemm_ctr, 0, (size_t)GEMM_CTR_SLOTS * 8 * sizeof(int), stream));
         gemm_ctr_cursor = 0;
      }
      if (nnzK_total > 0)
         hipLaunchKernelGGL(k_scatter, dim3(grid_for(nnzK_total, 256)), dim3(256), 0, stream, d_kdst, d_kval, d_arena, nnzK_total);
      if (nnzB_total > 0 && schur_mode_eff == 1)
         hipLaunchKernelGGL(k_scatter, dim3(grid_for(nnzB_total, 256)), dim3(256), 0, stream, d_bdst, d_bval, d_arena, nnzB_total);
      hipLaunchKernelGGL(k_tail_pad_diag, dim3(nblk), dim3(128), 0, stream, d_blks, d_arena, nblk);
      hipLaunchKernelGGL(k_pref_rows, dim3(32, nblk), dim3(256), 0, stream, d_blks, d_kval, d_kdiag, d_res, d_nprimal, d_krowptr, d_kcolidx);   // (d_res: scratch of the solves)
      hipLaunchKernelGGL(k_pref_init, dim3(32, nblk), dim3(256), 0, stream, d_blks, d_perm, d_perm_off, (const double*)d_res, d_pref);
      if (timer.on) timer.end(stream);
      // the whole-factor record (phase 6) was pushed first; close it at the end
      const size_t total_rec = 0;
      const ScatterCtx sx = head_slots ? ScatterCtx{2, nullptr, d_slot_val, d_arena, SC} : sx_atomic();
      for (size_t li = 0; li < levels.size(); ++li) {
         if (timer.on) timer.begin(stream, 1);
         if (head_slots) gather(g_levels[li], d_slot_val, d_arena);   // contributions of the lower levels, in fixed order
         launch_head_level(levels[li], SC, ldSC, sx);
         if (mf) { const int frc = launch_fronts((int)li, SC, ldSC); if (frc) return frc; }
         if (timer.on) timer.end(stream);
      }
      if (head_slots) {
         if (timer.on) timer.begin(stream, 1);
         gather(g_tail, d_slot_val, d_arena);
         if (SC && deterministic && d_gbuf) {
            HIP_TRY(hipMemsetAsync(d_gbuf, 0, (size_t)det_n_groups * det_gstride() * sizeof(double), stream));
            gather(g_sc_grp, d_slot_val, d_gbuf);
         } else if (SC) gather(g_sc, d_slot_val, SC);
         if (timer.on) timer.end(stream);
      }
      if (mf && n_roots > 0) {   // the last update matrices: into the tail and the Schur complement, front by front
         if (timer.on) timer.begin(stream, 1);
         // about 2048 workgroups in the launch (what is resident at once): half of a block's chunks on the tail's columns, half on the border's
         const int asm_half = std::max(1, std::min(32, 1024 / std::max(nblk, 1)));
         if (deterministic && d_gbuf && SC && d_round_blk) {
            for (size_t k = 0; k + 1 < round_off.size(); ++k) {
               const int cnt = round_off[k + 1] - round_off[k];
               if (cnt > 0)
                  hipLaunchKernelGGL(k_root_assemble, dim3(cnt, 2 * asm_half), dim3(256), 0, stream, d_round_blk + round_off[k], d_root_off, d_roots, d_sns, d_blks, d_rowidx,
                                     d_bmap, d_arena, d_mfU, SC, ldSC, d_sctab, d_gbuf, det_gstride(), d_blk_group, asm_half, asm_half);
            }
         } else
            hipLaunchKernelGGL(k_root_assemble, dim3(nblk, 2 * asm_half), dim3(256), 0, stream, (const int*)nullptr, d_root_off, d_roots, d_sns, d_blks, d_rowidx, d_bmap,
                               d_arena, d_mfU, SC, ldSC, d_sctab, (double*)nullptr, 0LL, (const int*)nullptr, asm_half, asm_half);
         if (timer.on) timer.end(stream);
      }
      if (mf && d_kb_list) {   // fronts on the rows of K only: their border rows now, level by level, from the finished panels; then the head's part
                               // of the dense tail's border rows.  (Forming level l on a second stream beside the fronts above it, which need
                               // nothing of it, was measured: no gain, 17.4 against 16.7 ms of head on the configs[3] share.)
         if (timer.on) timer.begin(stream, 1);
         for (size_t l = 0; l + 1 < kb_level_off.size(); ++l) launch_border_rows(l, stream);
         if (n_kb_tail > 0) hipLaunchKernelGGL(k_border_tail<16>, dim3(n_kb_tail), dim3(64), 0, stream, d_kb_tail, d_sns, d_blks, d_rowidx, d_arena);
         if (timer.on) timer.end(stream);
      }
      if (mf && n_bb > 0 && SC) {   // border split: the border x border part of every block's Schur contribution, from the finished panels
         if (timer.on) timer.begin(stream, 1);
         const int rc_bb = launch_border_schur(SC, ldSC);
         if (rc_bb) return rc_bb;
         if (timer.on) timer.end(stream);
      }
      if (spine_total > 0) {
         if (timer.on) timer.begin(stream, 1);
         hipLaunchKernelGGL((k_head_factor_spine<256, 32, 6144>), dim3(nblk), dim3(256), 0, stream, d_spine, d_spine_off, d_sns, d_blks,
                            d_rowidx, d_upd, d_psign, d_psign_off, d_bmap, d_arena, SC, ldSC, d_inertia, d_pref, d_sctab);
         if (timer.on) timer.end(stream);
      }
      if (mf && getenv("PIPS_HIP_MF_CLOCKS")) {
         if (!d_mfdbg) HIP_TRY(hipMalloc((void**)&d_mfdbg, (size_t)std::max(nsn_total, 1) * 8 * sizeof(long long)));
         else { int drc = dump_front_clocks(); if (drc) return drc; }
      }
      hipLaunchKernelGGL(k_pref_tail, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_arena, d_pref, 0);
      HIP_TRY(hipGetLastError());
      TailCtx c = ctx();
      int rc = tail_factor(c, SC, ldSC);
      if (rc) return rc;
      factored = true;
      perturbed_cache = -1;
      if (SC && schur_mode_eff == 2 && !schur_cols.empty()) {
         if (timer.on) timer.begin(stream, 5);
         rc = schur_by_solves(SC, ldSC);
         if (timer.on) timer.end(stream);
         if (rc) return rc;
      }
      if (timer.on) (void)hipEventRecord(timer.recs[total_rec].b, stream);
      h_amax.clear();
      if (!h_inertia_pin) {
         HIP_TRY(hipHostMalloc((void**)&h_inertia_pin, (size_t)(3 * nblk + 1) * sizeof(int), hipHostMallocDefault));
         h_inertia_pin[3 * nblk] = 0;
         HIP_TRY(hipEventCreateWithFlags(&ev_inertia, hipEventDisableTiming));
      }
      HIP_TRY(hipMemcpyAsync(h_inertia_pin, d_inertia, (size_t)3 * nblk * sizeof(int), hipMemcpyDeviceToHost, stream));
      if (tail_single)   // (the error word of the single-launch tail factorisation travels with the counters)
         HIP_TRY(hipMemcpyAsync(h_inertia_pin + 3 * nblk, tsingle.d_ctl + 9, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipEventRecord(ev_inertia, stream));
      inertia_in_flight = true;
      inertia_on_host = false;
      return PIPS_OK;
   }

   // SC -= sum_b Br_b^T K_b^-1 Br_b, chunk by chunk of 32 border columns: densify, solve (all blocks at once), multiply back
   int schur_by_solves(double* SC, int ldSC) {
      int rc = ensure_multi_buffers();
      if (rc) return rc;
      const int bs = 32, ncols = (int)schur_cols.size();
      for (int c0 = 0; c0 < ncols; c0 += bs) {
         const int nr = std::min(bs, ncols - c0);
         HIP_TRY(hipMemsetAsync(d_mx_rhs, 0, (size_t)nr * n_total * sizeof(double), stream));
         hipLaunchKernelGGL(k_border_rows_to_dense, dim3(grid_for(bt_rows_total, 256)), dim3(256), 0, stream, d_bt_rowptr, d_bt_colidx,
                            d_bval, d_bt_rowsc, d_bt_xoff, d_schur_slot, c0, nr, d_mx_rhs, n_total, bt_rows_total);
         if ((rc = use_multi(nr) ? solve_once_multi(d_mx_rhs, nr, n_total, d_mx_xw) : solve_once(d_mx_rhs, nr, n_total, d_mx_xw))) return rc;
         hipLaunchKernelGGL(k_border_tmult_chunk, dim3(grid_for(bt_rows_total, 256, 1024), nr), dim3(256), 0, stream, d_bt_rowptr,
                            d_bt_colidx, d_bval, d_bt_rowsc, d_bt_xoff, d_schur_cols + c0, nr, d_mx_rhs, n_total, SC, ldSC,
                            bt_rows_total);
      }
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // workgroups per block of the vector norms: slices of about 16 K rows, the launch kept near the number resident at once
   int absmax_chunks() const { return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(64, 2048 / std::max(nblk, 1) + 1), n_total / std::max(nblk, 1) / 16384 + 1)); }
   // nrhs right-hand sides at x_dev + r * x_stride (flat over all blocks each); work vectors at xw + r * xw_total
   int solve_once(double* x_dev, int nrhs = 1, long long x_stride = 0, double* xw = nullptr) {
      if (!xw) xw = d_xw;
      const long long xws = nrhs > 1 ? xw_total : 0;
      const dim3 pg(64, nblk, nrhs);
      timer.begin(stream, 7);
      hipLaunchKernelGGL(k_permute_in, pg, dim3(256), 0, stream, d_blks, d_perm, d_perm_off, x_dev, x_stride, xw, xws);
      timer.end(stream);
      timer.begin(stream, 8);
      if (deterministic && nrhs != 1) PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode solves one right-hand side at a time");
      if (head_slots && nrhs == 1 && (deterministic || slot_solves)) {
         // forward substitution without atomics: the contributions go to their slots, every level first gathers what the lower
         // levels left for its own columns, the tail rows are gathered before the dense sweep
         const ScatterCtx sxv{2, nullptr, d_vslot_val, xw, nullptr};
         for (size_t li = 0; li < levels.size(); ++li) {
            const LevelRange& L = levels[li];
            gather(gv_levels[li], d_vslot_val, xw);
            if (L.simple_cnt > 0 && lf_rows > 0)
               hipLaunchKernelGGL(k_leaf_fwd_gather, dim3((unsigned)((lf_rows + 255) / 256), 1), dim3(256), 0, stream, d_lf_rows, d_lf_ptr, d_lf_src,
                                  d_lf_val, xw, 0LL, (int)lf_rows);
            else if (L.simple_cnt > 0)
               hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin,
                                  L.simple_cnt, d_blks, d_rowidx, d_arena, xw, 0LL, 0, sxv);
            const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
            const int cnt = L.small_cnt + L.large_cnt;
            if (cnt > 0) hipLaunchKernelGGL(k_head_fwd, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin, d_blks, d_rowidx, d_arena, xw, 0LL, sxv);
         }
         gather(gv_tail, d_vslot_val, xw);
      } else
      for (const LevelRange& L : levels) {
         if (L.simple_cnt > 0 && lf_rows > 0)   // the leaves' columns are final as they stand: every target row collects its sum
            hipLaunchKernelGGL(k_leaf_fwd_gather, dim3((unsigned)((lf_rows + 255) / 256), nrhs), dim3(256), 0, stream, d_lf_rows, d_lf_ptr, d_lf_src,
                               d_lf_val, xw, xws, (int)lf_rows);
         else if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, nrhs), dim3(256), 0, stream, d_sns, L.simple_begin,
                               L.simple_cnt, d_blks, d_rowidx, d_arena, xw, xws, 0);
         // small and large supernodes of one level are contiguous in d_sns
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0)
            hipLaunchKernelGGL(head_wcap <= 16 ? k_head_fwd_chain<16> : k_head_fwd_chain<HEAD_WMAX>, dim3(cnt, nrhs), dim3(64), 0, stream, d_sns, begin,
                               d_blks, d_rowidx, d_arena, xw, xws, 0);
      }
      if (spine_total > 0)
         hipLaunchKernelGGL(k_head_solve_spine, dim3(nblk, nrhs), dim3(64), 0, stream, d_spine, d_spine_off, d_sns, d_blks, d_rowidx,
                            d_arena, xw, xws, 0);
      timer.end(stream);
      timer.begin(stream, 9);
      TailCtx c = ctx();
      c.timer = nullptr;   // (the tail's own phase records belong to the factorisation)
      int rc = tail_fwd(c, xw, nrhs, xws);
      if (rc) return rc;
      // D^-1 of the head columns: fused into the backward kernels of the common path (chain kernels + thread-per-leaf kernel: they
      // read the diagonal's cache line anyway; the separate pass reads 88 bytes of descriptor per supernode - 12.9 M of them on the
      // configs[3] share); the other paths (spine kernels, deterministic mode, k_head_bwd) keep the pass
      const bool fused_d = spine_total == 0;
      if (nsn_total > 0 && !fused_d)
         hipLaunchKernelGGL(k_head_dscale, dim3(grid_for(nsn_total, 256), nrhs), dim3(256), 0, stream, d_sns, nsn_total, d_blks,
                            d_arena, xw, xws, 0);
      rc = tail_bwd(c, xw, nrhs, xws);
      if (rc) return rc;
      timer.end(stream);
      timer.begin(stream, 10);
      if (spine_total > 0)
         hipLaunchKernelGGL(k_head_solve_spine, dim3(nblk, nrhs), dim3(64), 0, stream, d_spine, d_spine_off, d_sns, d_blks, d_rowidx,
                            d_arena, xw, xws, 1);
      for (int l = (int)levels.size() - 1; l >= 0; --l) {
         const LevelRange& L = levels[l];
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0)
            hipLaunchKernelGGL(head_wcap <= 16 ? k_head_bwd_chain<16> : k_head_bwd_chain<HEAD_WMAX>, dim3(cnt, nrhs), dim3(64), 0, stream, d_sns, begin,
                               d_blks, d_rowidx, d_arena, xw, xws, 0, fused_d ? 1 : 0);
         if (L.simple_cnt > 0 && d_leafdesc)
            hipLaunchKernelGGL(k_leaf_bwd, dim3((L.simple_cnt + 255) / 256, nrhs), dim3(256), 0, stream, d_leafdesc, L.simple_cnt, d_rowidx, d_arena,
                               xw, xws, fused_d ? 1 : 0);
         else if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, nrhs), dim3(256), 0, stream, d_sns, L.simple_begin,
                               L.simple_cnt, d_blks, d_rowidx, d_arena, xw, xws, 1, sx_atomic(), 0, 0, fused_d ? 1 : 0);
      }
      timer.end(stream);
      timer.begin(stream, 7);
      hipLaunchKernelGGL(k_permute_out, pg, dim3(256), 0, stream, d_blks, d_perm, d_perm_off, x_dev, x_stride, xw, xws);
      timer.end(stream);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // The interleaved multi-vector sweep reads every entry of L once per panel of 32 right-hand sides instead of once per right-hand side
   // and multiplies on the matrix pipe; the per-right-hand-side sweeps (grid.y = right-hand side) re-read L each time.  Round 2 measured
   // one config-2 block, 256 rhs: 11 ms separate / 43 ms interleaved - with a scalar multiply-add loop in the tile kernels; with the
   // matrix-pipe tiles and all panels in every launch the interleaved sweep wins from a few right-hand sides on.
   bool use_multi(int nr) const {
      if (const char* f = getenv("PIPS_HIP_MULTI")) return atoi(f) != 0 && nr >= 2;
      return nr >= 8;
   }

   // nr right-hand sides at X + q * x_stride in one interleaved sweep (kernels.hip.h "multi-vector solves"): panels of MQ, every launch
   // takes all of them (grid.y / grid.z); xm holds ceil(nr / MQ) * MQ * xw_total doubles
   int solve_once_multi(double* X, int nr, long long x_stride, double* xm) {
      const int np = (nr + MQ - 1) / MQ;
      const long long ps = (long long)MQ * xw_total;
      const dim3 pg(64, nblk, np);
      hipLaunchKernelGGL(k_mpermute, pg, dim3(256), 0, stream, d_blks, d_perm, d_perm_off, X, x_stride, nr, xm, 0, ps);
      auto head = [&](const LevelRange& L, int backward) {
         const int cnt = L.simple_cnt + L.small_cnt + L.large_cnt;   // contiguous: sorted by (level, class)
         if (cnt == 0) return;
         const int begin = L.simple_cnt > 0 ? L.simple_begin : (L.small_cnt > 0 ? L.small_begin : L.large_begin);
         hipLaunchKernelGGL(k_mhead, dim3((cnt + 3) / 4, np), dim3(256), 0, stream, d_sns, begin, cnt, d_blks, d_rowidx, d_arena, xm, backward, ps);
      };
      if (deterministic) {
         // forward substitution without atomics, like solve_once's: every contribution of a supernode goes to its slot (MQ right-hand sides
         // wide: d_mvslot, one panel at a time), every level first takes what the lower levels left for its columns, in the recorded order;
         // the simple leaves by target row where that list exists.  One panel per call (the caller cuts the right-hand sides into panels).
         if (np != 1 || !d_mvslot) PIPS_FAIL(PIPS_ERR_STATE, "deterministic mode: several right-hand sides go panel by panel");
         auto mgather = [&](const GatherList& g) {
            if (g.n_targets > 0)
               hipLaunchKernelGGL(k_mgather_slots, dim3(grid_for(g.n_targets * 32, 256)), dim3(256), 0, stream, g.n_targets, g.d_tgt, g.d_off, g.d_slots, d_mvslot, xm);
         };
         for (size_t li = 0; li < levels.size(); ++li) {
            const LevelRange& L = levels[li];
            mgather(gv_levels[li]);
            if (L.simple_cnt > 0 && lf_rows > 0)
               hipLaunchKernelGGL(k_mleaf_fwd_gather, dim3(grid_for(lf_rows * 32, 256)), dim3(256), 0, stream, d_lf_rows, d_lf_ptr, d_lf_src, d_lf_val, xm, (int)lf_rows);
            else if (L.simple_cnt > 0)
               hipLaunchKernelGGL(k_mhead, dim3((L.simple_cnt + 3) / 4, 1), dim3(256), 0, stream, d_sns, L.simple_begin, L.simple_cnt, d_blks, d_rowidx, d_arena, xm, 0, ps,
                                  d_mvslot);
            const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
            const int cnt = L.small_cnt + L.large_cnt;
            if (cnt > 0)
               hipLaunchKernelGGL(k_mhead, dim3((cnt + 3) / 4, 1), dim3(256), 0, stream, d_sns, begin, cnt, d_blks, d_rowidx, d_arena, xm, 0, ps, d_mvslot);
         }
         mgather(gv_tail);
      } else {
      for (const LevelRange& L : levels) head(L, 0);
      for (const LevelRange& L : levels_top) head(L, 0);
      }
      const TailPlan& p = plan;
      const bool rows = sweep.enabled && np * 4 <= SWEEP_NRHS_MAX && p.ntc_max > 0;   // the tail sweeps as one launch per direction (tickets, flags)
      // few tile rows per step (a single leaf): a workgroup takes a quarter or a half of a panel's right-hand sides, so that the chain of a pass -
      // one workgroup's tile products per step - is shorter; many blocks: whole panels, L read once per 32 right-hand sides.  Every workgroup
      // of the launch must be resident for the slices to advance side by side (two per compute unit: the whole tile sits in registers), so the
      // finest slicing that keeps the launch within 512 workgroups is taken
      const long long wg = (long long)sweep.n_tasks * np;
      const int forced = env_int("PIPS_HIP_MULTI", 1);   // (2: whole panels whatever the size, 4: half panels - tests)
      const int sl = forced == 2 ? 1 : forced == 4 ? 2 : wg * 4 <= 512 ? 4 : wg * 2 <= 512 ? 2 : 1;
      auto tail_rows = [&](int backward) {
         const SweepArgs sa = sweep.args(0, stream);
         if (sl == 4 && !backward) hipLaunchKernelGGL(k_mtail_rows_fwd<2>, dim3(sweep.n_tasks, np * 4), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
         else if (sl == 4) hipLaunchKernelGGL(k_mtail_rows_bwd<2>, dim3(sweep.n_tasks, np * 4), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
         else if (sl == 2 && !backward) hipLaunchKernelGGL(k_mtail_rows_fwd<4>, dim3(sweep.n_tasks, np * 2), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
         else if (sl == 2) hipLaunchKernelGGL(k_mtail_rows_bwd<4>, dim3(sweep.n_tasks, np * 2), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
         else if (!backward) hipLaunchKernelGGL(k_mtail_rows_fwd<8>, dim3(sweep.n_tasks, np), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
         else hipLaunchKernelGGL(k_mtail_rows_bwd<8>, dim3(sweep.n_tasks, np), dim3(256), 0, stream, sa, d_blks, d_arena, d_dtail, d_winv, xm, ps);
      };
      if (rows) tail_rows(0);
      else
      for (int j = 0; j < p.ntc_max; ++j)
         if (p.fwd[j].cnt > 0)
            hipLaunchKernelGGL(k_mtail_fwd, dim3(p.fwd[j].cnt, np), dim3(256), 0, stream, p.d_tasks + p.fwd[j].off, d_blks, d_arena, d_dtail,
                               d_winv, xm, j, ps);
      if (nsn_total > 0)
         hipLaunchKernelGGL(k_mhead_dscale, dim3(grid_for((long long)nsn_total * MQ, 256), np), dim3(256), 0, stream, d_sns, nsn_total, d_blks,
                            d_arena, xm, ps);
      if (rows) tail_rows(1);
      else
      for (int i = p.ntc_max - 1; i >= 0; --i)
         if (p.bwd[i].cnt > 0)
            hipLaunchKernelGGL(k_mtail_bwd, dim3(p.bwd[i].cnt, np), dim3(256), 0, stream, p.d_tasks + p.bwd[i].off, d_blks, d_arena, d_dtail,
                               d_winv, xm, i, ps);
      for (int l = (int)levels_top.size() - 1; l >= 0; --l) head(levels_top[l], 1);
      for (int l = (int)levels.size() - 1; l >= 0; --l) head(levels[l], 1);
      hipLaunchKernelGGL(k_mpermute, pg, dim3(256), 0, stream, d_blks, d_perm, d_perm_off, X, x_stride, nr, xm, 1, ps);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // multi-RHS solve (DoubleLinearSolver::solve(int nrhss, double* rhss, int*), PardisoSolver.C:276-352): all right-hand
   // sides share every launch (grid.y/z = rhs index); refine_steps unconditional refinement steps.  X_dev: nrhs vectors of
   // length n_total at distance x_stride.
   double *d_mx_xw = nullptr, *d_mx_rhs = nullptr, *d_mx_res = nullptr;
   int mx_cap = 0;
   double *d_mmeasure = nullptr, *h_mmeasure = nullptr;   // solve_multi: the refinement measure per right-hand side (device / pinned)
   int* d_midx = nullptr;                                // ... and the columns that take the correction solve
   double* d_hostx = nullptr;   // device copy of host right-hand sides (pips_hip_ldl_solve), kept between calls
   size_t hostx_cap = 0;
   double* d_hostpack = nullptr;   // packed rows + their indices of pips_hip_ldl_solve_sparse, kept between calls likewise
   size_t hostpack_cap = 0;
   static constexpr int MULTI_CHUNK_MAX = 256;   // right-hand sides per pass (eight panels)
   int ensure_multi_buffers(int want = 32) {
      want = std::min(MULTI_CHUNK_MAX, (std::max(want, 32) + MQ - 1) / MQ * MQ);
      if (mx_cap < want) {
         HIP_TRY(hipSetDevice(device));
         HIP_TRY(hipStreamSynchronize(stream));
         for (double** p : {&d_mx_xw, &d_mx_rhs, &d_mx_res}) { if (*p) (void)hipFree(*p); *p = nullptr; }
         HIP_TRY(hipMalloc((void**)&d_mx_xw, (size_t)want * std::max<long long>(xw_total, 1) * sizeof(double)));
         HIP_TRY(hipMalloc((void**)&d_mx_rhs, (size_t)want * std::max<long long>(n_total, 1) * sizeof(double)));
         HIP_TRY(hipMalloc((void**)&d_mx_res, (size_t)want * std::max<long long>(n_total, 1) * sizeof(double)));
         mx_cap = want;
      }
      if (!d_mmeasure) {   // per right-hand side: the measure of the adaptive refinement (device / pinned host, behind it the list of columns to correct)
         HIP_TRY(hipMalloc((void**)&d_mmeasure, MULTI_CHUNK_MAX * sizeof(double)));
         HIP_TRY(hipMalloc((void**)&d_midx, MULTI_CHUNK_MAX * sizeof(int)));
         HIP_TRY(hipHostMalloc((void**)&h_mmeasure, MULTI_CHUNK_MAX * (sizeof(double) + sizeof(int)), hipHostMallocDefault));
      }
      return PIPS_OK;
   }
   int solve_multi(double* X_dev, int nrhs, long long x_stride) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "solve called before factor");
      HIP_TRY(hipSetDevice(device));
      // deterministic mode: the interleaved sweeps panel by panel with the slot / gather forward substitution (solve_once_multi) where the
      // slots of a panel fit 4 GB (MQ doubles per contribution of the head: a leaf handle, a small batch), else one right-hand side at a time
      const bool det_multi = deterministic && head_slots && use_multi(nrhs) && vslots_total > 0 && (double)vslots_total * MQ * sizeof(double) <= 4e9;
      last_multi_path = det_multi ? 2 : (deterministic || !use_multi(nrhs)) ? 0 : 1;
      if (deterministic && !det_multi) {   // one right-hand side at a time through the atomics-free sweeps
         for (int r = 0; r < nrhs; ++r)
            if (int rc = solve(X_dev + (long long)r * x_stride)) return rc;
         return PIPS_OK;
      }
      if (det_multi && !d_mvslot) HIP_TRY(hipMalloc((void**)&d_mvslot, (size_t)vslots_total * MQ * sizeof(double)));
      last_refine_steps = 0;   // (over the chunks: the steps of the last one that took any)
      const bool multi = use_multi(nrhs);
      const int chunk_max = det_multi ? MQ : multi ? MULTI_CHUNK_MAX : 32;   // (the per-right-hand-side sweeps keep their 32 work vectors)
      int rc0 = ensure_multi_buffers(std::min(nrhs, chunk_max));
      if (rc0) return rc0;
      for (int r0 = 0; r0 < nrhs; r0 += chunk_max) {
         const int nr = std::min(chunk_max, nrhs - r0);
         double* X = X_dev + (long long)r0 * x_stride;
         if (refine_steps > 0)
            HIP_TRY(hipMemcpy2DAsync(d_mx_rhs, (size_t)n_total * sizeof(double), X, (size_t)x_stride * sizeof(double),
                                     (size_t)n_total * sizeof(double), nr, hipMemcpyDeviceToDevice, stream));
         int rc = multi ? solve_once_multi(X, nr, x_stride, d_mx_xw) : solve_once(X, nr, x_stride, d_mx_xw);
         if (rc) return rc;
         for (int it = 0; it < refine_steps; ++it) {
            HIP_TRY(hipMemcpyAsync(d_mx_res, d_mx_rhs, (size_t)nr * n_total * sizeof(double), hipMemcpyDeviceToDevice, stream));
            // r = rhs - K x for every right-hand side: x at stride x_stride, r contiguous -> a contiguous view of x where the caller's is not
            // (d_mx_xw is free between solves; it is at least nr * n_total long because xw_total >= n_total)
            const double* Xc = X;
            if (x_stride != n_total) {
               HIP_TRY(hipMemcpy2DAsync(d_mx_xw, (size_t)n_total * sizeof(double), X, (size_t)x_stride * sizeof(double),
                                        (size_t)n_total * sizeof(double), nr, hipMemcpyDeviceToDevice, stream));
               Xc = d_mx_xw;
            }
            hipLaunchKernelGGL(k_full_spmv_sub, dim3(grid_for(n_total * 8, 256, 65536), nr), dim3(256), 0, stream, d_frowptr, d_fcol, d_fsrc, d_kval,
                               Xc, d_mx_res, n_total, d_rowbase, n_total);
            if (n_flong > 0)
               hipLaunchKernelGGL(k_full_spmv_sub_long, dim3(n_flong, nr), dim3(256), 0, stream, d_flong, d_frowptr, d_fcol, d_fsrc, d_kval,
                                  Xc, d_mx_res, d_rowbase, n_total);
            int n_fix = nr;   // right-hand sides that take the correction solve: all of them, or (adaptive) those whose measure says so
            if (refine_tol > 0.0 && d_mmeasure) {
               // adaptive like the single right-hand side (solve()), per right-hand side like PARDISO: every column's worst block is measured,
               // one small copy back to the host, and the correction solve takes only the columns that were not accurate enough - their
               // residuals moved side by side to the front (ascending: a column never lands on one that is still to move)
               HIP_TRY(hipMemsetAsync(d_mmeasure, 0, (size_t)nr * sizeof(double), stream));
               hipLaunchKernelGGL(k_mrefine_measure, dim3(nblk, nr), dim3(256), 0, stream, d_blks, d_mx_res, n_total, d_mx_rhs, n_total, Xc, n_total,
                                  refine_mode == 1 ? 1.0 / (repl_rel > 0 ? repl_rel : 1.0) : 0.0, d_mmeasure);
               HIP_TRY(hipMemcpyAsync(h_mmeasure, d_mmeasure, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, stream));
               HIP_TRY(hipStreamSynchronize(stream));
               int* fix = (int*)(h_mmeasure + MULTI_CHUNK_MAX);
               n_fix = 0;
               double worst = 0.0;
               for (int q = 0; q < nr; ++q) {
                  if (!(h_mmeasure[q] <= worst)) worst = h_mmeasure[q];
                  if (!(h_mmeasure[q] <= refine_tol)) fix[n_fix++] = q;
               }
               last_refine_measure = worst;
               if (n_fix == 0) break;
               if (n_fix < nr) {
                  for (int i = 0; i < n_fix; ++i)
                     if (fix[i] != i)
                        HIP_TRY(hipMemcpyAsync(d_mx_res + (size_t)i * n_total, d_mx_res + (size_t)fix[i] * n_total, (size_t)n_total * sizeof(double), hipMemcpyDeviceToDevice, stream));
                  HIP_TRY(hipMemcpyAsync(d_midx, fix, (size_t)n_fix * sizeof(int), hipMemcpyHostToDevice, stream));
               }
            }
            rc = (multi && (n_fix > 1 || det_multi)) ? solve_once_multi(d_mx_res, n_fix, n_total, d_mx_xw) : solve_once(d_mx_res, n_fix, n_total, d_mx_xw);
            if (rc) return rc;
            if (n_fix < nr) hipLaunchKernelGGL(k_maxpy_idx, dim3(grid_for(n_total, 256, 1024), n_fix), dim3(256), 0, stream, X, x_stride, d_mx_res, n_total, 1.0, n_total, d_midx);
            else hipLaunchKernelGGL(k_maxpy, dim3(grid_for(n_total, 256, 1024), nr), dim3(256), 0, stream, X, x_stride, d_mx_res, n_total, 1.0, n_total);
            ++last_refine_steps;
         }
      }
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // x := K^-1 x with up to refine_steps steps of iterative refinement against the CSR values on the device.
   // refine_tol > 0 makes it adaptive like PARDISO's iparm[7] (PardisoProjectSolver.C:72): after every solve the residual
   // is formed and the loop stops once max_b ||r_b||inf / ||rhs_b||inf <= refine_tol (one small D2H copy + stream sync).
   int solve(double* x_dev) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "solve called before factor");
      HIP_TRY(hipSetDevice(device));
      last_refine_steps = 0;
      if (refine_steps <= 0) return solve_once(x_dev);
      const size_t bytes = (size_t)n_total * sizeof(double);
      HIP_TRY(hipMemcpyAsync(d_rhs, x_dev, bytes, hipMemcpyDeviceToDevice, stream));
      if (refine_tol > 0.0) {
         HIP_TRY(hipMemsetAsync(d_norms + nblk, 0, (size_t)nblk * sizeof(double), stream));
         hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, d_rhs, d_blks, d_norms + nblk);
      }
      int rc = solve_once(x_dev);
      if (rc) return rc;
      for (int it = 0; it < refine_steps; ++it) {
         timer.begin(stream, 11);
         // r = rhs - K x; without long rows the copy of rhs is folded into the product
         if (n_flong > 0) HIP_TRY(hipMemcpyAsync(d_res, d_rhs, bytes, hipMemcpyDeviceToDevice, stream));
         hipLaunchKernelGGL(k_full_spmv_sub, dim3(grid_for(n_total * 8, 256, 65536)), dim3(256), 0, stream, d_frowptr, d_fcol, d_fsrc, d_kval,
                            x_dev, d_res, n_total, d_rowbase, 0LL, n_flong > 0 ? (const double*)nullptr : d_rhs);
         if (n_flong > 0)
            hipLaunchKernelGGL(k_full_spmv_sub_long, dim3(n_flong), dim3(256), 0, stream, d_flong, d_frowptr, d_fcol, d_fsrc, d_kval, x_dev,
                               d_res, d_rowbase, 0LL);
         if (refine_tol > 0.0) {
            HIP_TRY(hipMemsetAsync(d_norms, 0, (size_t)nblk * sizeof(double), stream));
            hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, d_res, d_blks, d_norms);
            if (refine_mode == 1) {
               HIP_TRY(hipMemsetAsync(d_norms + 2 * nblk, 0, (size_t)nblk * sizeof(double), stream));
               hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, x_dev, d_blks, d_norms + 2 * nblk);
            }
            HIP_TRY(hipMemcpyAsync(h_norms, d_norms, (size_t)3 * nblk * sizeof(double), hipMemcpyDeviceToHost, stream));
            timer.end(stream);
            if (refine_mode == 1 && h_amax.empty()) {
               h_amax.resize(nblk);
               std::vector<BlkDesc> tmp(nblk);
               HIP_TRY(hipMemcpyAsync(tmp.data(), d_blks, (size_t)nblk * sizeof(BlkDesc), hipMemcpyDeviceToHost, stream));
               HIP_TRY(hipStreamSynchronize(stream));
               for (int b = 0; b < nblk; ++b) h_amax[b] = tmp[b].repl_abs / (repl_rel > 0 ? repl_rel : 1.0);   // = max|K_b| (k_block_absmax)
            }
            HIP_TRY(hipStreamSynchronize(stream));
            double worst = 0.0;
            for (int b = 0; b < nblk; ++b) {
               const double den = refine_mode == 1 ? h_amax[b] * h_norms[2 * nblk + b] + h_norms[nblk + b] : h_norms[nblk + b];
               // (a non-finite norm - Inf, or Inf / Inf = NaN - must win the comparison: never "converged" on a poisoned iterate)
               if (den > 0.0) { const double q = h_norms[b] / den; if (!(q <= worst)) worst = q; }
            }
            last_refine_measure = worst;
            if (worst <= refine_tol) break;
         } else
            timer.end(stream);
         rc = solve_once(d_res);
         if (rc) return rc;
         hipLaunchKernelGGL(k_axpy, dim3(grid_for(n_total, 256)), dim3(256), 0, stream, x_dev, d_res, 1.0, n_total);
         ++last_refine_steps;
      }
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // The measure of the adaptive refinement for a solution somebody else produced: worst block of ||rhs - K x||inf over the denominator of
   // refine_mode (normwise backward error, or ||rhs||inf) - one product with the CSR values, the norms, one small copy to the host.
   int residual_measure(const double* rhs_dev, const double* x_dev, double* worst_out) {
      if (refine_steps <= 0 || !d_res || !d_norms) PIPS_FAIL(PIPS_ERR_STATE, "residual_measure: refinement buffers missing");
      const size_t bytes = (size_t)n_total * sizeof(double);
      timer.begin(stream, 11);
      HIP_TRY(hipMemsetAsync(d_norms, 0, (size_t)3 * nblk * sizeof(double), stream));
      hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, rhs_dev, d_blks, d_norms + nblk);
      if (n_flong > 0) HIP_TRY(hipMemcpyAsync(d_res, rhs_dev, bytes, hipMemcpyDeviceToDevice, stream));
      hipLaunchKernelGGL(k_full_spmv_sub, dim3(grid_for(n_total * 8, 256, 65536)), dim3(256), 0, stream, d_frowptr, d_fcol, d_fsrc, d_kval, x_dev, d_res, n_total,
                         d_rowbase, 0LL, n_flong > 0 ? (const double*)nullptr : rhs_dev);
      if (n_flong > 0)
         hipLaunchKernelGGL(k_full_spmv_sub_long, dim3(n_flong), dim3(256), 0, stream, d_flong, d_frowptr, d_fcol, d_fsrc, d_kval, x_dev, d_res, d_rowbase, 0LL);
      hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, d_res, d_blks, d_norms);
      if (refine_mode == 1) hipLaunchKernelGGL(k_vec_block_absmax, dim3(absmax_chunks(), nblk), dim3(256), 0, stream, x_dev, d_blks, d_norms + 2 * nblk);
      HIP_TRY(hipMemcpyAsync(h_norms, d_norms, (size_t)3 * nblk * sizeof(double), hipMemcpyDeviceToHost, stream));
      timer.end(stream);
      if (refine_mode == 1 && h_amax.empty()) {
         h_amax.resize(nblk);
         std::vector<BlkDesc> tmp(nblk);
         HIP_TRY(hipMemcpyAsync(tmp.data(), d_blks, (size_t)nblk * sizeof(BlkDesc), hipMemcpyDeviceToHost, stream));
         HIP_TRY(hipStreamSynchronize(stream));
         for (int b = 0; b < nblk; ++b) h_amax[b] = tmp[b].repl_abs / (repl_rel > 0 ? repl_rel : 1.0);
      }
      HIP_TRY(hipStreamSynchronize(stream));
      double worst = 0.0;
      for (int b = 0; b < nblk; ++b) {
         const double den = refine_mode == 1 ? h_amax[b] * h_norms[2 * nblk + b] + h_norms[nblk + b] : h_norms[nblk + b];
         if (den > 0.0) { const double q = h_norms[b] / den; if (!(q <= worst)) worst = q; }
      }
      *worst_out = worst;
      last_refine_measure = worst;
      return PIPS_OK;
   }

   // the measure of a solveCompressed by sweeps in one launch (k_measure_leaf_rows): x = the result, b = the leaf right-hand side as the
   // caller gave it, x0 = the root solution.  Same quantity and same host formula as residual_measure on (b - Br x0, x); available where
   // every row of K is short and the border is held by leaf row
   bool can_measure_fused() const { return refine_steps > 0 && d_norms && n_flong == 0 && (bt_rows_total == 0 || d_br_rowptr); }   // (deterministic mode too: a row is summed by eight lanes in a fixed tree, the maxima do not depend on an order)
   int residual_measure_fused(const double* b_dev, const double* x0_dev, const double* x_dev, double* worst_out) {
      timer.begin(stream, 11);
      HIP_TRY(hipMemsetAsync(d_norms, 0, (size_t)3 * nblk * sizeof(double), stream));
      hipLaunchKernelGGL(k_measure_leaf_rows, dim3(std::max(absmax_chunks(), (int)std::min<long long>(256, n_total / std::max(nblk, 1) / 2048 + 1)), nblk), dim3(256), 0, stream,
                         d_blks, d_frowptr, d_fcol, d_fsrc, d_kval, x_dev, b_dev, bt_rows_total > 0 ? d_br_rowptr : (const int*)nullptr, d_br_sc, d_br_src, d_bval, x0_dev,
                         d_norms, nblk);
      HIP_TRY(hipMemcpyAsync(h_norms, d_norms, (size_t)3 * nblk * sizeof(double), hipMemcpyDeviceToHost, stream));
      timer.end(stream);
      std::vector<BlkDesc> tmp;
      const bool need_amax = refine_mode == 1 && h_amax.empty();   // (the blocks' largest entries: once per factorisation, with the same wait)
      if (need_amax) {
         tmp.resize(nblk);
         HIP_TRY(hipMemcpyAsync(tmp.data(), d_blks, (size_t)nblk * sizeof(BlkDesc), hipMemcpyDeviceToHost, stream));
      }
      HIP_TRY(hipStreamSynchronize(stream));
      if (need_amax) {
         h_amax.resize(nblk);
         for (int b = 0; b < nblk; ++b) h_amax[b] = tmp[b].repl_abs / (repl_rel > 0 ? repl_rel : 1.0);
      }
      double worst = 0.0;
      for (int b = 0; b < nblk; ++b) {
         const double den = refine_mode == 1 ? h_amax[b] * h_norms[2 * nblk + b] + h_norms[nblk + b] : h_norms[nblk + b];
         if (den > 0.0) { const double q = h_norms[b] / den; if (!(q <= worst)) worst = q; }
      }
      *worst_out = worst;
      last_refine_measure = worst;
      return PIPS_OK;
   }

   // The inertia counters travel to pinned host memory at the end of every factorisation (factor()); a query only waits for that
   // copy - not for whatever was queued behind the factorisation (the leaf solves of an Lsolve, say).
   int* h_inertia_pin = nullptr;
   hipEvent_t ev_inertia = nullptr;
   bool inertia_in_flight = false, inertia_on_host = false;
   int fetch_inertia() {
      if (inertia_in_flight && h_inertia_pin) {   // one wait per factorisation: a query per block (256 of them per IPM iteration) reads the host copy
         HIP_TRY(hipEventSynchronize(ev_inertia));
         std::copy(h_inertia_pin, h_inertia_pin + h_inertia.size(), h_inertia.begin());
         inertia_in_flight = false;
         inertia_on_host = true;
         if (tail_single && h_inertia_pin[3 * nblk]) {
            h_inertia_pin[3 * nblk] = 0;
            int fi[6] = {0, 0, 0, 0, 0, 0};   // the first wait that gave up: kind, tile, K range, block + 1
            (void)hipMemcpy(fi, tsingle.d_ctl + 10, sizeof(fi), hipMemcpyDeviceToHost);
            HIP_TRY(hipMemsetAsync(tsingle.d_ctl + 9, 0, 7 * sizeof(int), stream));
            PIPS_FAIL(PIPS_ERR_HIP, "the single-launch factorisation of the dense tails (PIPS_HIP_TAIL_SINGLE) gave up waiting for a tile after %lld polls - first: task "
                                    "kind %d on tile (%d, %d) of block %d; its factors are unusable.  Another process holding the device for seconds can cause that",
                      tsingle.args.poll_limit, fi[1], fi[2], fi[3], fi[5] - 1);
         }
         return PIPS_OK;
      }
      if (inertia_on_host) return PIPS_OK;
      HIP_TRY(hipMemcpyAsync(h_inertia.data(), d_inertia, h_inertia.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      return PIPS_OK;
   }

   // ---- border-backward sweep ----------------------------------------------------------------------
   // u_i = K_i^-1 Br_i x0 for every block from the augmented factor alone:  Br_i = L D L21^T, so  u = L^-T (L21^T x0)  - the
   // backward sweep of the augmented factor [L 0; L21 I] with a zero right-hand side and the border unknowns fixed to -x0.  No
   // border product, no forward sweep, no diagonal scaling.  Pays where the border rows of the factor are no larger than what a
   // forward sweep reads (decided at analyze time, border_backward_ok); needs Schur mode 1 and the single-launch tail sweeps.
   // There is no refinement in it: the caller uses it only while the factorisation has no perturbed pivot (perturbed_leaf_pivots).
   bool border_backward_ok = false;
   bool aug_sweeps_ok = false;   // solveCompressed by one forward + one backward sweep of the augmented factor (decided at analyze time)
   long long aug_passes = 0;     // passes (forward + backward) of those sweeps so far
   int perturbed_cache = -1;   // perturbed pivots of the current factorisation over all blocks; -1 = not fetched yet
   int perturbed_leaf_pivots(int* out) {
      if (perturbed_cache < 0) {
         int rc = fetch_inertia();
         if (rc) return rc;
         int z = 0;
         for (int b = 0; b < nblk; ++b) z += h_inertia[3 * b + 2];
         perturbed_cache = z;
      }
      *out = perturbed_cache;
      return PIPS_OK;
   }
   // Forward sweep of the augmented factor [L 0; L_b I] on [b; 0]: the work vector keeps y = L^-1 P b (tail rows D^-1-scaled, as the tail
   // sweep leaves them) for backward_augmented, and red += -L_b y = -Br^T K^-1 b, block by block (what Lsolve adds to b0,
   // sLinsysRootAug.C:323-344, without the backward sweep, the residual check and the sparse border product of a full solve).
   int forward_augmented(const double* b_dev, double* red) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "solve called before factor");
      HIP_TRY(hipSetDevice(device));
      timer.begin(stream, 7);
      hipLaunchKernelGGL(k_border_fill, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_bmap, (const double*)nullptr, d_xw, 0.0);
      hipLaunchKernelGGL(k_permute_in, dim3(64, nblk, 1), dim3(256), 0, stream, d_blks, d_perm, d_perm_off, b_dev, 0LL, d_xw, 0LL);
      timer.end(stream);
      timer.begin(stream, 8);
      for (const LevelRange& L : levels) {
         if (L.simple_cnt > 0 && lf_rows > 0)
            hipLaunchKernelGGL(k_leaf_fwd_gather, dim3((unsigned)((lf_rows + 255) / 256), 1), dim3(256), 0, stream, d_lf_rows, d_lf_ptr, d_lf_src, d_lf_val, d_xw,
                               0LL, (int)lf_rows);
         else if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin, L.simple_cnt, d_blks,
                               d_rowidx, d_arena, d_xw, 0LL, 0);
         if (L.simple_cnt > 0 && n_lb > 0)
            hipLaunchKernelGGL(k_leaf_border, dim3((n_lb + 255) / 256), dim3(256), 0, stream, d_lb_list, n_lb, d_sns, d_blks, d_rowidx, d_arena, d_xw, 0);
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0)
            hipLaunchKernelGGL(head_wcap <= 16 ? k_head_fwd_chain<16> : k_head_fwd_chain<HEAD_WMAX>, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin, d_blks,
                               d_rowidx, d_arena, d_xw, 0LL, 1);
      }
      timer.end(stream);
      timer.begin(stream, 9);   // (one record of this phase per pass: the backward half books its tail sweep with the head's)
      TailCtx c = ctx();
      c.timer = nullptr;
      int rc = tail_fwd(c, d_xw);
      if (rc) return rc;
      if (nb_pad_max > 0)
         hipLaunchKernelGGL(k_tail_border_fwd, dim3(nb_pad_max / TILE, nblk), dim3(256), 0, stream, d_blks, d_arena, d_dtail, d_xw);
      hipLaunchKernelGGL(k_border_collect, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_bmap, d_xw, red);
      timer.end(stream);
      ++aug_passes;
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
   // The same sweep in deterministic mode: the head by the slot / gather scheme of solve_once (no atomics), the border slots of every block by
   // k_border_gather_det from the finished head part (target by target, fixed order), the dense tail's border rows by k_tail_border_fwd (one
   // writer per row).  The border slots stay in the work vector: the caller gathers them group-wise (g_bslot_grp) like Br^T z.
   int forward_augmented_det(const double* b_dev) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "solve called before factor");
      if (!det_aug_ready || !head_slots) PIPS_FAIL(PIPS_ERR_STATE, "forward_augmented_det: not prepared");
      HIP_TRY(hipSetDevice(device));
      timer.begin(stream, 7);
      hipLaunchKernelGGL(k_border_fill, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_bmap, (const double*)nullptr, d_xw, 0.0);
      hipLaunchKernelGGL(k_permute_in, dim3(64, nblk, 1), dim3(256), 0, stream, d_blks, d_perm, d_perm_off, b_dev, 0LL, d_xw, 0LL);
      timer.end(stream);
      timer.begin(stream, 8);
      const ScatterCtx sxv{2, nullptr, d_vslot_val, d_xw, nullptr};
      for (size_t li = 0; li < levels.size(); ++li) {
         const LevelRange& L = levels[li];
         gather(gv_levels[li], d_vslot_val, d_xw);
         if (L.simple_cnt > 0 && lf_rows > 0)
            hipLaunchKernelGGL(k_leaf_fwd_gather, dim3((unsigned)((lf_rows + 255) / 256), 1), dim3(256), 0, stream, d_lf_rows, d_lf_ptr, d_lf_src, d_lf_val, d_xw,
                               0LL, (int)lf_rows);
         else if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin, L.simple_cnt, d_blks,
                               d_rowidx, d_arena, d_xw, 0LL, 0, sxv);
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0) hipLaunchKernelGGL(k_head_fwd, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin, d_blks, d_rowidx, d_arena, d_xw, 0LL, sxv);
      }
      gather(gv_tail, d_vslot_val, d_xw);
      hipLaunchKernelGGL(k_border_rowdot_det, dim3(grid_for(n_bg_ent, 256, 1 << 20)), dim3(256), 0, stream, n_bg_ent, d_bg_ent, d_arena, d_xw, d_bg_val);
      hipLaunchKernelGGL(k_border_gather_det, dim3((unsigned)((n_bg_targets + 3) / 4)), dim3(256), 0, stream, n_bg_targets, d_bg_ptr, d_bg_idx, d_bg_slot, d_bg_val, d_xw);
      timer.end(stream);
      timer.begin(stream, 9);
      TailCtx c = ctx();
      c.timer = nullptr;
      int rc = tail_fwd(c, d_xw);
      if (rc) return rc;
      if (nb_pad_max > 0)
         hipLaunchKernelGGL(k_tail_border_fwd, dim3(nb_pad_max / TILE, nblk), dim3(256), 0, stream, d_blks, d_arena, d_dtail, d_xw);
      timer.end(stream);
      ++aug_passes;
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
   // ... and the backward sweep from where forward_augmented stopped: border slots = x0, x = L^-T (D^-1 y - L_b^T x0) = K^-1 (b - Br x0)
   // in original order at out_dev (Ltsolve, sLinsysRootAug.C:346-365 / LniTransMult, with nothing left to combine)
   int backward_augmented(const double* x0_dev, double* out_dev) {
      HIP_TRY(hipSetDevice(device));
      timer.begin(stream, 10);
      hipLaunchKernelGGL(k_border_fill, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_bmap, x0_dev, d_xw, 1.0);
      TailCtx c = ctx();
      c.timer = nullptr;
      int rc = tail_bwd(c, d_xw, 1, 0, 1);
      if (rc) return rc;
      for (int l = (int)levels.size() - 1; l >= 0; --l) {
         const LevelRange& L = levels[l];
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0)
            hipLaunchKernelGGL(head_wcap <= 16 ? k_head_bwd_chain<16> : k_head_bwd_chain<HEAD_WMAX>, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin, d_blks,
                               d_rowidx, d_arena, d_xw, 0LL, 1, 1);
         if (L.simple_cnt > 0 && d_leafdesc)
            hipLaunchKernelGGL(k_leaf_bwd, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_leafdesc, L.simple_cnt, d_rowidx, d_arena, d_xw, 0LL, 1);
         else if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin, L.simple_cnt, d_blks,
                               d_rowidx, d_arena, d_xw, 0LL, 1, sx_atomic(), 0, 0, 1);
         if (L.simple_cnt > 0 && n_lb > 0)
            hipLaunchKernelGGL(k_leaf_border, dim3((n_lb + 255) / 256), dim3(256), 0, stream, d_lb_list, n_lb, d_sns, d_blks, d_rowidx, d_arena, d_xw, 1);
      }
      timer.end(stream);
      timer.begin(stream, 7);
      hipLaunchKernelGGL(k_permute_out, dim3(64, nblk, 1), dim3(256), 0, stream, d_blks, d_perm, d_perm_off, out_dev, 0LL, d_xw, 0LL);
      timer.end(stream);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
   int solve_border_backward(const double* x0_dev, double* out_dev) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "solve called before factor");
      HIP_TRY(hipSetDevice(device));
      HIP_TRY(hipMemsetAsync(d_xw, 0, (size_t)xw_total * sizeof(double), stream));
      hipLaunchKernelGGL(k_border_fill, dim3(8, nblk), dim3(256), 0, stream, d_blks, d_bmap, x0_dev, d_xw);
      TailCtx c = ctx();
      int rc = tail_bwd(c, d_xw, 1, 0, 1);
      if (rc) return rc;
      const ScatterCtx none{0, nullptr, nullptr, nullptr, nullptr};
      if (spine_total > 0)
         hipLaunchKernelGGL(k_head_solve_spine, dim3(nblk, 1), dim3(64), 0, stream, d_spine, d_spine_off, d_sns, d_blks, d_rowidx, d_arena, d_xw, 0LL, 1, 1);
      for (int l = (int)levels.size() - 1; l >= 0; --l) {
         const LevelRange& L = levels[l];
         const int begin = L.small_cnt > 0 ? L.small_begin : L.large_begin;
         const int cnt = L.small_cnt + L.large_cnt;
         if (cnt > 0)
            hipLaunchKernelGGL(head_wcap <= 16 ? k_head_bwd_chain<16> : k_head_bwd_chain<HEAD_WMAX>, dim3(cnt, 1), dim3(64), 0, stream, d_sns, begin,
                               d_blks, d_rowidx, d_arena, d_xw, 0LL, 1, 0);
         if (L.simple_cnt > 0)
            hipLaunchKernelGGL(k_head_solve_simple, dim3((L.simple_cnt + 255) / 256, 1), dim3(256), 0, stream, d_sns, L.simple_begin, L.simple_cnt,
                               d_blks, d_rowidx, d_arena, d_xw, 0LL, 1, none, 1);
      }
      hipLaunchKernelGGL(k_permute_out, dim3(64, nblk, 1), dim3(256), 0, stream, d_blks, d_perm, d_perm_off, out_dev, 0LL, d_xw, 0LL);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }
};

// ---------------------------------------------------------------------------------------------------------------
// dense root solver (DeSymIndefSolver replacement) on the same tile kernels
// ---------------------------------------------------------------------------------------------------------------
extern "C" int pips_hip_allreduce_sum(void* comm, double* buf_dev, size_t n, void* stream);
extern "C" int pips_hip_all_gather(void* comm, double* buf_dev, size_t chunk, int n_parts, void* stream);
extern "C" int pips_hip_broadcast(void* comm, double* buf_dev, size_t n, int root, void* stream);
extern "C" int pips_hip_comm_has_broadcast(void* comm);

__global__ void k_inertia_to_double(const int* __restrict__ in, double* __restrict__ out, int back, int* __restrict__ in_out) {
   if (threadIdx.x < 3) { if (!back) out[threadIdx.x] = (double)in[threadIdx.x]; else in_out[threadIdx.x] = (int)(out[threadIdx.x] + 0.5); }
}

struct DenseLdl {
   int device = 0, n = 0, npad = 0, n_primal = -1;
   // 0: static pivot order with the expected signs of the inertia hint (right for the quasi-definite Schur complement the fused
   //    path builds itself: no search on the critical path of the 128 dependent pivots of a tile);
   // 1: Bunch-Kaufman 1 x 1 / 2 x 2 pivoting bounded to the diagonal tile (k_tile_diag_bk): what a drop-in for DeSymIndefSolver
   //    needs - dsytrf takes any symmetric matrix (DeSymIndefSolver.C:78)
   int pivoting = 0;
   hipStream_t stream = nullptr;
   double thr_rel = 1e-13, repl_rel = 1e-8;
   bool factored = false;
   std::vector<BlkDesc> h_blks;
   TailPlan plan;
   BlkDesc* d_blks = nullptr;
   double *d_R = nullptr, *d_winv = nullptr, *d_dtail = nullptr, *d_xw = nullptr, *d_in = nullptr, *d_pref = nullptr;
   double* d_U = nullptr;   // U = L D (npad x npad), B operand of the updates
   signed char* d_psign = nullptr;
   long long *d_psign_off = nullptr, *d_kptr = nullptr;
   int* d_inertia = nullptr;
   int h_inertia[3] = {0, 0, 0};
   hipStream_t side = nullptr;
   hipEvent_t ev_panel = nullptr, ev_rest = nullptr;
   // staging copy of a host matrix: only the host-pointer entry points need it (the fused KKT path hands over d_SC)
   int ensure_input_buffer() {
      if (!d_in) HIP_TRY(hipMalloc((void**)&d_in, (size_t)std::max(n, 1) * std::max(n, 1) * sizeof(double)));
      return PIPS_OK;
   }

   ~DenseLdl() {
      if (side) (void)hipStreamDestroy(side);
      if (ev_panel) (void)hipEventDestroy(ev_panel);
      if (ev_rest) (void)hipEventDestroy(ev_rest);
      void* ptrs[] = {d_blks, d_R, d_U, d_winv, d_dtail, d_xw, d_in, d_pref, d_psign, d_psign_off, d_kptr, d_inertia, d_dist_tasks, d_panel, d_perm, d_pert_cnt, d_pert_list, d_flagvec,
                      d_C, d_rtasks, d_rflags};
      for (void* p : ptrs)
         if (p) (void)hipFree(p);
      plan.release();
      sweep.release();
   }
   SweepRt sweep;
   // ---- the factorisation as ONE dependency-driven launch (rootkernel.hip.h, rootplan.cpp): static pivot order on one rank.  Bunch-Kaufman
   // (a 454-register diagonal kernel) and the root distributed over ranks keep the launch-per-step driver (tail_factor / factor_distributed);
   // PIPS_HIP_ROOT_LAUNCHES=1 keeps it everywhere (the tests run both).
   bool single_launch = !getenv("PIPS_HIP_ROOT_LAUNCHES");
   double* d_C = nullptr;          // the accumulating tiles (scratch): L goes to d_R, U to d_U, each written once per launch
   TileTask* d_rtasks = nullptr;
   int n_rtasks = 0, n_rbulk = 0;
   int* d_rflags = nullptr;        // ctl[8] | prog[ntc * ntc] | rowdone[ntc] | dready[ntc]
   double plan_makespan_us = 0.0;
   long long root_poll_limit = 400000;   // polls before a wait inside the launch gives up (some 0.1 s: a factorisation takes 2 - 40 ms)
   bool root_error_pending = false;
   int ensure_single_launch() {
      if (d_rtasks) return PIPS_OK;
      const int ntc = npad / TILE;
      RootPlanParams pp;
      if (const char* q = getenv("PIPS_HIP_ROOT_CHAIN_CU")) pp.chain_slots = atoi(q) != 0 ? 2 : 0;   // 0: one list, the chain wherever its workgroups land (A/B)
      if (pp.chain_slots == 0) pp.workers = 512;
      std::vector<int> t, tc;
      int rc = build_root_plan(ntc, pp, t, tc, &plan_makespan_us);
      if (rc) return rc;
      n_rbulk = (int)(t.size() / 4);
      t.insert(t.end(), tc.begin(), tc.end());
      n_rtasks = (int)(t.size() / 4);
      HIP_TRY(hipMalloc((void**)&d_rtasks, std::max<size_t>(t.size(), 4) * sizeof(int)));
      HIP_TRY(hipMemcpy(d_rtasks, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc((void**)&d_rflags, ((size_t)8 + (size_t)ntc * ntc + 2 * (size_t)ntc) * sizeof(int)));
      HIP_TRY(hipMalloc((void**)&d_C, (size_t)npad * npad * sizeof(double)));
      if (const char* pl = getenv("PIPS_HIP_ROOT_POLL_LIMIT")) root_poll_limit = atoll(pl);
      return PIPS_OK;
   }
   int factor_single_launch() {
      const int ntc = npad / TILE;
      HIP_TRY(hipMemsetAsync(d_rflags, 0, ((size_t)8 + (size_t)ntc * ntc + 2 * (size_t)ntc) * sizeof(int), stream));
      RootArgs a{};
      a.tasks = d_rtasks; a.n_tasks = n_rtasks; a.n_bulk = n_rbulk; a.ntc = ntc; a.ld = npad;
      a.C = d_C; a.R = d_R; a.U = d_U; a.winv = d_winv; a.dtail = d_dtail; a.pref = d_pref; a.psign = d_psign; a.inertia = d_inertia;
      a.ctl = d_rflags; a.prog = d_rflags + 8; a.rowdone = a.prog + (size_t)ntc * ntc; a.dready = a.rowdone + ntc;
      a.blk = d_blks; a.poll_limit = root_poll_limit;
      a.diag_blocked = getenv("PIPS_HIP_ROOT_DIAG_BARRIERS") ? 0 : 1;
      const char* trace_file = getenv("PIPS_HIP_ROOT_TRACE");   // diagnostics: per-task clocks of this launch into a file (tools/root_trace.py)
      long long* d_trace = nullptr;
      if (trace_file) {
         HIP_TRY(hipMalloc((void**)&d_trace, ((size_t)3 * n_rtasks + 32 * (size_t)ntc) * sizeof(long long)));
         HIP_TRY(hipMemsetAsync(d_trace, 0, ((size_t)3 * n_rtasks + 32 * (size_t)ntc) * sizeof(long long), stream));
         a.trace = d_trace;
      }
      hipLaunchKernelGGL(k_root_ldl, dim3(n_rtasks), dim3(512), 0, stream, a);
      HIP_TRY(hipGetLastError());
      root_error_pending = true;
      if (trace_file) {
         std::vector<long long> h((size_t)3 * n_rtasks + 32 * (size_t)ntc);
         std::vector<int> ht((size_t)4 * n_rtasks);
         HIP_TRY(hipStreamSynchronize(stream));
         HIP_TRY(hipMemcpy(h.data(), d_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
         HIP_TRY(hipMemcpy(ht.data(), d_rtasks, ht.size() * sizeof(int), hipMemcpyDeviceToHost));
         (void)hipFree(d_trace);
         if (FILE* f = fopen(trace_file, "w")) {
            for (int t = 0; t < n_rtasks; ++t)
               fprintf(f, "%d %d %d %d %d %lld %lld %lld\n", t, ht[4 * t], ht[4 * t + 1], ht[4 * t + 2], ht[4 * t + 3], h[3 * (size_t)t], h[3 * (size_t)t + 1], h[3 * (size_t)t + 2]);
            fclose(f);
         }
         if (FILE* f = fopen((std::string(trace_file) + ".diag").c_str(), "w")) {   // phase clocks of the diagonal tiles (blocked variant)
            for (int jj = 0; jj < ntc; ++jj) {
               for (int q = 0; q < 26; ++q) fprintf(f, "%lld ", h[(size_t)3 * n_rtasks + 32 * (size_t)jj + q]);
               fprintf(f, "\n");
            }
            fclose(f);
         }
      }
      return PIPS_OK;
   }
   // a wait inside the launch that gave up raised the error word; read at the host's next synchronisation point with this handle
   int take_root_error(const char* who) {
      if (!root_error_pending || !d_rflags) return PIPS_OK;
      root_error_pending = false;
      int w = 0;
      HIP_TRY(hipMemcpyAsync(&w, d_rflags + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      if (w) PIPS_FAIL(PIPS_ERR_HIP, "%s: the single-launch root factorisation gave up waiting for a tile after %lld polls (its factors are invalid); "
                                     "PIPS_HIP_ROOT_LAUNCHES=1 selects the launch-per-step factorisation", who, root_poll_limit);
      return PIPS_OK;
   }
   int init() {
      HIP_TRY(hipSetDevice(device));
      int rc = PIPS_OK;
      npad = (n + TILE - 1) / TILE * TILE;
      BlkDesc d{};
      d.n = n; d.n_head = 0; d.m = n; d.m_pad = npad; d.nb = 0; d.nb_pad = 0; d.ldT = npad;
      d.ntc = d.ntr = npad / TILE;
      d.thr_rel = 0; d.repl_rel = 1e-8; d.repl_abs = 1;
      h_blks.assign(1, d);
      if ((rc = dev_upload(&d_blks, h_blks, stream))) return rc;
      // one block only: right-looking (measured on MI355X, tools/root_probe.py: S=2000 4.3 -> 2.3 ms, S=16000 183 -> 45 ms;
      // panels of 1 tile column are best up to S = 8000, 2-3 beyond)
      const int panel = d.ntc <= 64 ? 1 : 2;
      // lookahead (second stream) pays once the trailing update of a panel outlasts a diagonal tile: S >= 6000 measured
      const bool lookahead = d.ntc >= 48;
      if ((rc = plan.build(h_blks, panel, lookahead))) return rc;
      if ((rc = sweep.build(h_blks, nullptr))) return rc;
      if (lookahead) {
         int prio_lo = 0, prio_hi = 0;
         HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
         HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_lo));   // the bulk update: behind the diagonal chain
         HIP_TRY(hipEventCreateWithFlags(&ev_panel, hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&ev_rest, hipEventDisableTiming));
      }
      std::vector<signed char> ps(npad, 1);
      for (int i = 0; i < n; ++i) ps[i] = n_primal < 0 ? 0 : (i < n_primal ? 1 : -1);
      if ((rc = dev_upload(&d_psign, ps, stream))) return rc;
      std::vector<long long> zero(1, 0), kp = {0, (long long)npad};   // fallback magnitude from the diagonal only
      if ((rc = dev_upload(&d_psign_off, zero, stream))) return rc;
      if ((rc = dev_upload(&d_kptr, kp, stream))) return rc;
      HIP_TRY(hipMalloc((void**)&d_R, (size_t)npad * npad * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_U, (size_t)npad * npad * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_winv, (size_t)npad * TILE * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_dtail, (size_t)npad * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_xw, (size_t)npad * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_pref, (size_t)npad * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&d_inertia, 3 * sizeof(int)));
      return PIPS_OK;
   }
   TailCtx ctx() {
      TailCtx c{d_blks, &plan, d_R, d_dtail, d_winv, d_psign, d_psign_off, nullptr, d_inertia, stream, nullptr, d_pref, side, ev_panel, ev_rest, true, nullptr, d_U};
      c.sweep = &sweep;
      c.bunch_kaufman = pivoting == 1;
      if (pivoting == 1) {
         c.d_pert_cnt = d_pert_cnt; c.d_pert_list = d_pert_list;
         c.bk_orig = last_A; c.bk_orig_ld = last_lda; c.bk_orig_rowmajor = last_rowmajor; c.d_bk_perm = perm.empty() ? nullptr : d_perm;
         c.bk_isolate = bk_isolate;
      }
      return c;
   }
   // ---- Bunch-Kaufman beyond the tile.  k_tile_diag_bk searches its 128 x 128 tile; dsytrf searches the whole column
   // (DeSymIndefSolver.C:78).  Where a tile finds no pivot for an index (a zero leading block coupled only to later rows: [[0 A^T]; [A 0]])
   // the kernel records the row of the column's largest entry in the panel below.  check_pivots() - at the next host synchronisation
   // point: the end of the host-pointer factor call, or the join with the root's stream before the first Dsolve - then moves every such
   // row next to its column (a symmetric permutation P kept for the following factorisations: the structure that needed it comes back
   // every iteration), factorises P A P^T again and goes on until no index is left without a pivot (at most BK_RETRIES times).  The
   // pair then sits inside one tile, where the 2 x 2 pivot is found.  Solves permute their right-hand side in and out.
   static constexpr int BK_RETRIES = 8;
   static constexpr int BK_MAX_COLUMNS = 2048;   // columns per round whose original entries travel to the host for the partner choice
   std::vector<int> perm;              // perm[i] = original index at position i (empty: identity)
   int* d_perm = nullptr;
   int *d_pert_cnt = nullptr, *d_pert_list = nullptr;
   const double* last_A = nullptr;     // the matrix of the last factor_dev (the caller keeps it until the next factorisation)
   int last_lda = 0, last_rowmajor = 0;
   bool check_pending = false;
   int bk_isolate = 1;                 // factorisations check_pivots will look at take an index without a pivot OUT of the matrix (k_tile_diag_bk)
   int bk_refactorizations = 0;        // how often check_pivots had to factorise again (diagnostics / tests)
   // indices (positions in the current order) the last factorisation found no pivot for: one rank - what its tile kernels recorded, in
   // their order; a root distributed over several ranks - every rank recorded those of its own tile columns, the union (ascending) reaches
   // every rank through an all-reduce of a 0 / 1 vector, so that all ranks go on to build the SAME pivot order
   double* d_flagvec = nullptr;
   int flagged_positions(std::vector<int>& pos) {
      pos.clear();
      int cnt = 0;
      HIP_TRY(hipMemcpyAsync(&cnt, d_pert_cnt, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      cnt = std::min(cnt, npad);   // (an index flagged by the tile kernel AND the growth check counts twice: the kernels stop writing at npad records)
      std::vector<int> rec((size_t)2 * std::max(cnt, 0));
      if (cnt > 0) HIP_TRY(hipMemcpy(rec.data(), d_pert_list, rec.size() * sizeof(int), hipMemcpyDeviceToHost));
      if (dist_P <= 1) {
         for (int q = 0; q < cnt; ++q) pos.push_back(rec[2 * q]);
         return PIPS_OK;
      }
      std::vector<double> flag((size_t)npad, 0.0);
      for (int q = 0; q < cnt; ++q) if (rec[2 * q] >= 0 && rec[2 * q] < npad) flag[rec[2 * q]] = 1.0;
      if (!d_flagvec) HIP_TRY(hipMalloc((void**)&d_flagvec, (size_t)npad * sizeof(double)));
      HIP_TRY(hipMemcpy(d_flagvec, flag.data(), flag.size() * sizeof(double), hipMemcpyHostToDevice));
      int rc = pips_hip_allreduce_sum(dist_comm, d_flagvec, (size_t)npad, stream);
      if (rc) return rc;
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipMemcpy(flag.data(), d_flagvec, flag.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i) if (flag[i] > 0.0) pos.push_back(i);
      return PIPS_OK;
   }
   int check_pivots() {
      if (!check_pending) return PIPS_OK;
      check_pending = false;
      if (pivoting != 1 || !d_pert_cnt) return PIPS_OK;
      HIP_TRY(hipSetDevice(device));
      std::vector<int> fpos;
      for (int attempt = 0; attempt < BK_RETRIES; ++attempt) {
         int rcf = flagged_positions(fpos);
         if (rcf) return rcf;
         const int cnt = (int)fpos.size();
         if (cnt <= 0) return PIPS_OK;
         std::vector<int> rec((size_t)2 * cnt, -1);
         for (int q = 0; q < cnt; ++q) rec[2 * q] = fpos[q];
         if (dist_P > 1) HIP_TRY(hipMemcpy(d_pert_list, rec.data(), std::min(rec.size(), (size_t)2 * npad) * sizeof(int), hipMemcpyHostToDevice));   // (k_bk_gather_columns reads the positions there)
         // positions (in the current order) -> partner positions.  For every index without a pivot the column of the ORIGINAL matrix comes
         // to the host; in the order the tiles reported them each takes the row with its largest entry that is still free (not itself
         // without a pivot, not taken by an earlier column): the row a 2 x 2 pivot with this column needs ([[0 A^T]; [A 0]]: a row of A).
         const int n_use = std::min(cnt, BK_MAX_COLUMNS);
         std::vector<double> cols((size_t)n_use * n);
         {
            double* d_cols = nullptr;
            HIP_TRY(hipMalloc((void**)&d_cols, cols.size() * sizeof(double)));
            hipLaunchKernelGGL(k_bk_gather_columns, dim3(std::max(1, std::min(64, (n + 255) / 256)), n_use), dim3(256), 0, stream, last_A, last_lda, last_rowmajor,
                               perm.empty() ? (const int*)nullptr : (const int*)d_perm, n, d_pert_list, n_use, d_cols);
            const hipError_t ec = hipMemcpy(cols.data(), d_cols, cols.size() * sizeof(double), hipMemcpyDeviceToHost);
            (void)hipFree(d_cols);
            if (ec != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "dense root: %s", hipGetErrorString(ec));
         }
         std::vector<int> partner(n, -1);
         std::vector<char> taken(n, 0), flagged(n, 0);
         for (int q = 0; q < cnt; ++q) if (rec[2 * q] >= 0 && rec[2 * q] < n) flagged[rec[2 * q]] = 1;
         bool any = false;
         for (int q = 0; q < n_use; ++q) {
            const int c = rec[2 * q];
            if (c < 0 || c >= n || partner[c] >= 0) continue;
            const double* col = cols.data() + (size_t)q * n;
            int best = -1;
            for (int i = 0; i < n; ++i)
               if (!flagged[i] && !taken[i] && col[i] > 0.0 && (best < 0 || col[i] > col[best])) best = i;
            if (best < 0) continue;
            partner[c] = best; taken[best] = 1; any = true;
         }
         if (!any) break;               // nothing below to pair with
         std::vector<int> cur(n);
         for (int i = 0; i < n; ++i) cur[i] = perm.empty() ? i : perm[i];
         std::vector<int> next;
         next.reserve(n);
         std::vector<int> deferred;     // a pair must not straddle a tile boundary: an unpaired index goes in between
         for (int i = 0; i < n; ++i) {
            if (taken[i]) continue;      // emitted right behind its column (wherever it stood)
            if (partner[i] >= 0) {
               if ((int)next.size() % TILE == TILE - 1) {
                  int filler = -1;
                  for (int t = i + 1; t < n && filler < 0; ++t) if (!taken[t] && partner[t] < 0 && t != i && !flagged[t] && std::find(deferred.begin(), deferred.end(), t) == deferred.end()) filler = t;
                  if (filler >= 0) { next.push_back(cur[filler]); deferred.push_back(filler); }
               }
               next.push_back(cur[i]);
               next.push_back(cur[partner[i]]);
            } else if (std::find(deferred.begin(), deferred.end(), i) == deferred.end())
               next.push_back(cur[i]);
         }
         if ((int)next.size() != n) PIPS_FAIL(PIPS_ERR_STATE, "dense root: internal error building the pivot order (%zu of %d)", next.size(), n);
         perm.swap(next);
         if (!d_perm) HIP_TRY(hipMalloc((void**)&d_perm, (size_t)std::max(n, 1) * sizeof(int)));
         HIP_TRY(hipMemcpy(d_perm, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
         ++bk_refactorizations;
         int rc = factor_enqueue();
         if (rc) return rc;
      }
      // indices without a pivot are left: they get the usual small replacement (and are reported as perturbed pivots - what the
      // regularisation loop of the caller reacts to) instead of being taken out of the matrix
      {
         int rcf = flagged_positions(fpos);
         if (rcf) return rcf;
         if (!fpos.empty()) {
            bk_isolate = 0;
            const int rc = factor_enqueue();
            bk_isolate = 1;
            if (rc) return rc;
         }
      }
      return PIPS_OK;
   }

   // ---- distributed factorisation (several ranks, each with the whole reduced Schur complement): tile column j belongs to rank
   // j mod P (1-D column-cyclic).  The owner factorises the diagonal tile and solves the column's panel, the panel (Winv_j, d_j, L(:, j),
   // U(:, j)) goes to every rank, and every rank applies it to the tile columns it owns - 1 / P of the S^3 / 3 flops per rank instead
   // of all of them on every rank (DistributedRootLinearSystem.C:1436-1464 has every rank call dsytrf on the same matrix).  At the
   // end every rank holds the complete factor, so the solves stay local and replicated.  The panel travels by pips_hip_broadcast
   // (ncclBroadcast / the host's broadcast callback; a host communicator without one: as an all-reduce in which the other ranks
   // contribute zeros - exact, twice the bytes).  One column of lookahead (round 5): see factor_distributed.  NOT TIMED: the GPU box has one
   // device; correctness with 2 and 4 processes sharing it (tests/test_dist_root_gpu.py).
   void* dist_comm = nullptr;
   int dist_rank = 0, dist_P = 1;
   std::vector<TaskList> dist_diag, dist_trsm, dist_next, dist_upd;   // dist_next[j]: this rank's tiles of column j + 1 under panel j; dist_upd[j]: of its columns >= j + 2
   TileTask* d_dist_tasks = nullptr;
   double* d_panel = nullptr;
   int set_distributed(void* comm, int rank, int P) {
      if (P <= 1 || !comm) { dist_comm = nullptr; dist_P = 1; return PIPS_OK; }
      if (rank < 0 || rank >= P) PIPS_FAIL(PIPS_ERR_ARG, "distributed root: rank %d of %d", rank, P);
      HIP_TRY(hipSetDevice(device));
      dist_comm = comm; dist_rank = rank; dist_P = P;
      const int ntc = npad / TILE;
      std::vector<TileTask> all;
      dist_diag.assign(ntc, {}); dist_trsm.assign(ntc, {}); dist_next.assign(ntc, {}); dist_upd.assign(ntc, {});
      for (int j = 0; j < ntc; ++j) {
         dist_diag[j].off = (long long)all.size();
         all.push_back({0, j, j, 0});
         dist_diag[j].cnt = 1;
         dist_trsm[j].off = (long long)all.size();
         for (int ti = j + 1; ti < ntc; ++ti) all.push_back({0, ti, j, 0});
         dist_trsm[j].cnt = (int)((long long)all.size() - dist_trsm[j].off);
         dist_next[j].off = (long long)all.size();
         if (j + 1 < ntc && (j + 1) % P == rank)
            for (int ti = j + 1; ti < ntc; ++ti) all.push_back({0, ti, j + 1, j | ((j + 1) << 16)});   // C(ti, tk) -= L(ti, j) U(tk, j)^T
         dist_next[j].cnt = (int)((long long)all.size() - dist_next[j].off);
         dist_upd[j].off = (long long)all.size();
         for (int tk = j + 2; tk < ntc; ++tk)
            if (tk % P == rank)
               for (int ti = tk; ti < ntc; ++ti) all.push_back({0, ti, tk, j | ((j + 1) << 16)});
         dist_upd[j].cnt = (int)((long long)all.size() - dist_upd[j].off);
      }
      if (d_dist_tasks) { (void)hipFree(d_dist_tasks); d_dist_tasks = nullptr; }
      int rc = dev_upload(&d_dist_tasks, all, stream);
      if (rc) return rc;
      if (!d_panel) HIP_TRY(hipMalloc((void**)&d_panel, ((size_t)TILE * TILE + TILE + 8 + 2 * (size_t)npad * TILE) * sizeof(double)));
      if (!side) {   // one column of lookahead: the bulk of a panel's update runs beside the factorisation and the broadcast of the next column
         HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
         HIP_TRY(hipEventCreateWithFlags(&ev_panel, hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&ev_rest, hipEventDisableTiming));
      }
      return PIPS_OK;
   }
   int factor_distributed() {
      const TailCtx c = ctx();
      const int ntc = npad / TILE;
      const size_t ld_bytes = (size_t)npad * sizeof(double);
      bool side_busy = false;
      for (int j = 0; j < ntc; ++j) {
         const int owner = j % dist_P;
         const size_t rows = (size_t)npad - (size_t)(j + 1) * TILE;
         const size_t head = (size_t)TILE * TILE + TILE + 8, count = head + 2 * rows * TILE;
         double* Lp = d_panel + head;
         double* Up = Lp + rows * TILE;
         const size_t col0 = (size_t)(j + 1) * TILE + (size_t)j * TILE * npad;   // first entry below the diagonal tile of column j
         if (owner == dist_rank) {
            if (c.bunch_kaufman)
               hipLaunchKernelGGL(k_tile_diag_bk, dim3(1), dim3(256), 0, stream, d_dist_tasks + dist_diag[j].off, c.d_blks, c.d_arena, c.d_dtail, c.d_winv, c.d_inertia,
                                  c.d_pert_cnt, c.d_pert_list, c.bk_orig, c.bk_orig_ld, c.bk_orig_rowmajor, c.d_bk_perm, c.bk_isolate);
            else
               hipLaunchKernelGGL(k_tile_diag, dim3(1), dim3(256), 0, stream, d_dist_tasks + dist_diag[j].off, c.d_blks, c.d_arena, c.d_dtail, c.d_winv,
                                  c.d_psign, c.d_psign_off, c.d_inertia, c.d_pref);
            if (dist_trsm[j].cnt > 0)
               hipLaunchKernelGGL(k_tile_gemm<1>, dim3((dist_trsm[j].cnt + 7) / 8 * 8), dim3(512), 0, stream, d_dist_tasks + dist_trsm[j].off, dist_trsm[j].cnt,
                                  c.d_blks, c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
            // (the owner looks for multipliers a whole-column search would not allow, as on one rank: DenseLdl::check_pivots)
            if (c.bunch_kaufman && c.d_pert_cnt && dist_trsm[j].cnt > 0)
               hipLaunchKernelGGL(k_bk_growth, dim3(TILE), dim3(256), 0, stream, c.d_blks, c.d_arena, j, 1e8, c.d_pert_cnt, c.d_pert_list);
            HIP_TRY(hipMemcpyAsync(d_panel, d_winv + (size_t)j * TILE * TILE, (size_t)TILE * TILE * sizeof(double), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_panel + (size_t)TILE * TILE, d_dtail + (size_t)j * TILE, TILE * sizeof(double), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemsetAsync(d_panel + (size_t)TILE * TILE + TILE, 0, 8 * sizeof(double), stream));
            if (rows > 0) {
               HIP_TRY(hipMemcpy2DAsync(Lp, rows * sizeof(double), d_R + col0, ld_bytes, rows * sizeof(double), TILE, hipMemcpyDeviceToDevice, stream));
               HIP_TRY(hipMemcpy2DAsync(Up, rows * sizeof(double), d_U + col0, ld_bytes, rows * sizeof(double), TILE, hipMemcpyDeviceToDevice, stream));
            }
         } else if (!pips_hip_comm_has_broadcast(dist_comm))
            HIP_TRY(hipMemsetAsync(d_panel, 0, count * sizeof(double), stream));   // (no broadcast primitive: the panel travels as an all-reduce of zeros)
         int rc = pips_hip_broadcast(dist_comm, d_panel, count, owner, stream);
         if (rc) return rc;
         if (owner != dist_rank) {
            HIP_TRY(hipMemcpyAsync(d_winv + (size_t)j * TILE * TILE, d_panel, (size_t)TILE * TILE * sizeof(double), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_dtail + (size_t)j * TILE, d_panel + (size_t)TILE * TILE, TILE * sizeof(double), hipMemcpyDeviceToDevice, stream));
            if (rows > 0) {
               HIP_TRY(hipMemcpy2DAsync(d_R + col0, ld_bytes, Lp, rows * sizeof(double), rows * sizeof(double), TILE, hipMemcpyDeviceToDevice, stream));
               HIP_TRY(hipMemcpy2DAsync(d_U + col0, ld_bytes, Up, rows * sizeof(double), rows * sizeof(double), TILE, hipMemcpyDeviceToDevice, stream));
            }
         }
         // One column of lookahead.  Panel j is applied in two pieces: this rank's tiles of column j + 1 on the main stream - the owner of that
         // column goes on to factorise and broadcast it - and its columns >= j + 2 (the bulk) on the side stream, beside that.  Column j + 1
         // also takes the bulk of panel j - 1: the main stream joins it first (ev_rest as recorded before this iteration's bulk).
         const bool ahead = side && !getenv("PIPS_HIP_DIST_NO_LOOKAHEAD");
         hipStream_t bulk = ahead ? side : stream;
         if (ahead) {
            if (side_busy) HIP_TRY(hipStreamWaitEvent(stream, ev_rest, 0));
            HIP_TRY(hipEventRecord(ev_panel, stream));          // panel j is in d_R / d_U on this rank
            HIP_TRY(hipStreamWaitEvent(side, ev_panel, 0));
         }
         if (dist_upd[j].cnt > 0)
            hipLaunchKernelGGL(k_tile_gemm<3>, dim3((dist_upd[j].cnt + 7) / 8 * 8), dim3(512), 0, bulk, d_dist_tasks + dist_upd[j].off, dist_upd[j].cnt, c.d_blks,
                               c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
         if (ahead) { HIP_TRY(hipEventRecord(ev_rest, side)); side_busy = true; }
         if (dist_next[j].cnt > 0)
            hipLaunchKernelGGL(k_tile_gemm<3>, dim3((dist_next[j].cnt + 7) / 8 * 8), dim3(512), 0, stream, d_dist_tasks + dist_next[j].off, dist_next[j].cnt, c.d_blks,
                               c.d_arena, c.d_dtail, c.d_winv, c.d_bmap, (double*)nullptr, 0, (const int*)nullptr, c.d_uarena);
      }
      if (side_busy) HIP_TRY(hipStreamWaitEvent(stream, ev_rest, 0));
      // every rank counted the pivots of its own diagonal tiles: the sum is the inertia
      hipLaunchKernelGGL(k_inertia_to_double, dim3(1), dim3(64), 0, stream, d_inertia, d_panel, 0, d_inertia);
      HIP_TRY(hipMemsetAsync(d_panel + 3, 0, 5 * sizeof(double), stream));
      int rc = pips_hip_allreduce_sum(dist_comm, d_panel, 8, stream);
      if (rc) return rc;
      hipLaunchKernelGGL(k_inertia_to_double, dim3(1), dim3(64), 0, stream, d_inertia, d_panel, 1, d_inertia);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   }

   // A_dev: n x n, symmetric, column-major with the lower triangle authoritative (== row-major with the upper one)
   // rowmajor = 1: A_dev is row-major (the reference's DenseStorage), 0: column-major; lower triangle authoritative
   int factor_dev(const double* A_dev, int lda, int rowmajor) {
      last_A = A_dev; last_lda = lda; last_rowmajor = rowmajor;
      int rc = factor_enqueue();
      check_pending = pivoting == 1;
      return rc;
   }
   int factor_enqueue() {
      const double* A_dev = last_A;
      const int lda = last_lda, rowmajor = last_rowmajor;
      HIP_TRY(hipSetDevice(device));
      if (pivoting == 1) {
         if (!d_pert_cnt) {
            HIP_TRY(hipMalloc((void**)&d_pert_cnt, sizeof(int)));
            HIP_TRY(hipMalloc((void**)&d_pert_list, (size_t)2 * std::max(npad, 1) * sizeof(int)));
         }
         HIP_TRY(hipMemsetAsync(d_pert_cnt, 0, sizeof(int), stream));
      }
      const bool single = single_launch && pivoting == 0 && dist_P <= 1;
      if (single) { const int rcs = ensure_single_launch(); if (rcs) return rcs; }
      double* d_work = single ? d_C : d_R;   // where the matrix is accumulated: a scratch copy (single launch) or in place
      hipLaunchKernelGGL(k_copy_lower_to_padded, dim3(grid_for((long long)npad * npad, 256)), dim3(256), 0, stream, A_dev,
                         lda, n, d_work, npad, npad, rowmajor, perm.empty() ? (const int*)nullptr : (const int*)d_perm);
      hipLaunchKernelGGL(k_pref_tail, dim3(8, 1), dim3(256), 0, stream, d_blks, d_work, d_pref, 1);
      hipLaunchKernelGGL(k_block_absmax_init, dim3(1), dim3(256), 0, stream, d_blks, 1);
      hipLaunchKernelGGL(k_block_absmax, dim3(8, 1), dim3(256), 0, stream, d_pref, d_kptr, d_blks);
      hipLaunchKernelGGL(k_block_absmax_finish, dim3(1), dim3(256), 0, stream, d_blks, 1, thr_rel, repl_rel);
      HIP_TRY(hipMemsetAsync(d_inertia, 0, 3 * sizeof(int), stream));
      int rc = single ? factor_single_launch() : (dist_P > 1 ? factor_distributed() : tail_factor(ctx(), nullptr, 0));
      if (rc) return rc;
      factored = true;
      return PIPS_OK;
   }
   int solve_dev(double* x_dev) {
      if (!factored) PIPS_FAIL(PIPS_ERR_STATE, "dense solve called before factor");
      HIP_TRY(hipSetDevice(device));
      int rc = check_pivots();     // (a factorisation whose pivots were not looked at yet: host-pointer callers come here first)
      if (rc) return rc;
      HIP_TRY(hipMemsetAsync(d_xw, 0, (size_t)npad * sizeof(double), stream));
      if (perm.empty()) HIP_TRY(hipMemcpyAsync(d_xw, x_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
      else hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_perm, n, x_dev, d_xw, 0, (double*)nullptr);
      rc = tail_fwd(ctx(), d_xw);
      if (rc) return rc;
      rc = tail_bwd(ctx(), d_xw);
      if (rc) return rc;
      if (perm.empty()) HIP_TRY(hipMemcpyAsync(x_dev, d_xw, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
      else hipLaunchKernelGGL(k_perm_gather, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_perm, n, (const double*)nullptr, d_xw, 1, x_dev);
      return PIPS_OK;
   }
};

// ---------------------------------------------------------------------------------------------------------------
// fused two-level KKT system of one rank: leaves (Engine) + replicated dense root (DenseLdl) + Schur reduction.
// Mirrors DistributedRootLinearSystem::factor2 (:206-243) and DistributedLinearSystem::solveCompressed (:409-420)
// with sLinsysRootAug::{finalizeKKTdense, Lsolve, Dsolve, Ltsolve} (sLinsysRootAug.C:1769-1796, 323-365).
// ---------------------------------------------------------------------------------------------------------------
typedef int (*allreduce_fn)(void* comm, double* buf, size_t n, void* stream);

struct KktSystem {
   Engine* leaves = nullptr;
   std::unique_ptr<DenseLdl> root;
   int n0 = 0, my0 = 0, myl = 0, mzl = 0, S = 0;
   int rank = 0, n_ranks = 1;
   void* comm = nullptr;
   double *d_SC = nullptr, *d_t = nullptr, *d_fin_val = nullptr, *d_c0_val = nullptr, *d_red = nullptr, *d_packed = nullptr;
   long long* d_fin_idx = nullptr;
   long long n_fin = 0;
   double *d_gall = nullptr, *d_gvec_all = nullptr;   // deterministic mode over several ranks: all eight group slots (8 x S x S / 8 x S)
   int mz0 = 0;
   int *d_c0_rp = nullptr, *d_c0_ci = nullptr;
   const double* d_zdiag0 = nullptr;   // caller-owned, set per iteration
   double root_reg_primal = 0.0, root_reg_dual = 0.0;   // pips_hip_kkt_set_root_regularization
   hipStream_t comm_stream = nullptr;   // panel-wise Schur reduction beside the leaf work
   hipEvent_t ev_reduced = nullptr;
   // The dense root is factorised on a stream of its own: it is a latency chain (S = 2000: 16 diagonal tiles, 1.9 ms with the chip
   // nearly idle) and nothing needs its factors before the Dsolve of the next solveCompressed - the leaf solves of that call's
   // Lsolve run beside it.  root_wait() joins the main stream with it (before Dsolve, the next factorisation, an inertia query).
   hipStream_t root_stream = nullptr;
   hipEvent_t ev_sc_final = nullptr, ev_root_done = nullptr;
   bool root_pending = false;
   bool root_own_stream = true;   // pips_hip_kkt_set_root_stream: a caller that asks for the root's inertia after every factorisation (the IPM
                                  // harness) has nothing to run beside the root - the second stream then only costs (measured: section 4.3b)
   int root_wait() {
      if (root_pending) {
         if (root && root->check_pending) {   // Bunch-Kaufman root: were there indices without a pivot inside their tile?  (host wait for the
                                              // root's stream - the work queued on the main stream meanwhile keeps the device busy)
            hipStream_t keep = root->stream;
            root->stream = root_stream;
            const int rc = root->check_pivots();
            root->stream = keep;
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev_root_done, root_stream));
         }
         HIP_TRY(hipStreamWaitEvent(leaves->stream, ev_root_done, 0));
         root_pending = false;
      } else if (root && root->check_pending) {
         // the root was factorised on the main stream (pips_hip_kkt_set_root_stream(0)): the pivot check is still owed, and it reads from the
         // device and waits - it must happen here, before a capture of the solve sequence begins, not inside DenseLdl::solve_dev
         const int rc = root->check_pivots();
         if (rc) return rc;
      }
      return PIPS_OK;
   }
   size_t packed_cap = 0;
   bool use_rsag = false, force_reduce = false;
   bool solve_graph = false;                // pips_hip_kkt_set_solve_graph
   hipGraphExec_t graph_exec = nullptr;
   hipStream_t graph_stream = nullptr;
   // everything a captured launch sequence has baked in: buffer addresses, the Ltsolve path, the number of refinement launches, the
   // elimination of root inequality rows (zdiag0, C0), the root's pivoting mode, the analysis the leaf buffers belong to
   struct GraphKey {
      const void *b0 = nullptr, *bl = nullptr, *zdiag0 = nullptr, *c0_val = nullptr, *c0_rp = nullptr, *c0_ci = nullptr;
      int from_factor = 0, refine_steps = 0, refine_mode = 0, mz0 = 0, pivoting = 0, bk_gen = 0;
      long long analysis_gen = 0;
      bool operator==(const GraphKey& o) const {
         return b0 == o.b0 && bl == o.bl && zdiag0 == o.zdiag0 && c0_val == o.c0_val && c0_rp == o.c0_rp && c0_ci == o.c0_ci &&
                from_factor == o.from_factor && refine_steps == o.refine_steps && refine_mode == o.refine_mode && mz0 == o.mz0 &&
                pivoting == o.pivoting && bk_gen == o.bk_gen && analysis_gen == o.analysis_gen;
      }
   } graph_key;
   long long graph_captures = 0, graph_replays = 0;
   bool last_ltsolve_from_factor = false;   // which Ltsolve the last solveCompressed took (reported per solve, not only at analyze time)
   // Sweeps of the augmented factor for both halves of solveCompressed (Engine::forward_augmented / backward_augmented).  They carry no
   // refinement, so they are taken only on evidence that this factorisation is accurate: no perturbed pivot, and an earlier
   // solveCompressed on the SAME factors went the refined way and every refined leaf solve in it met the backward-error tolerance
   // without a step (aug_validated_gen == factor_gen).  The first solveCompressed after every factorisation is that witness.
   long long factor_gen = 0, aug_validated_gen = -1;
   int last_solve_path = 0;   // 0: two refined leaf solves, 1: refined Lsolve + Ltsolve from the factor, 2: augmented sweeps, 3: augmented sweeps
                              // whose result was checked (below)
   // One rank: the witness is the first sweep pair itself - its result x_i is put into the leaf rows, r_i = b_i - Br_i x0 - K_i x_i, and
   // accepted where the measure of the adaptive refinement is within the tolerance (one product with K instead of a refined solve and
   // its extra backward sweep); a result that fails is thrown away, the saved right-hand side goes the refined way and the sweeps stay
   // off for these factors (aug_failed_gen).  Several ranks keep the refined witness: the decision to repeat a solveCompressed would
   // have to be taken by all ranks together.  PIPS_HIP_AUG_WITNESS=0: the refined witness everywhere.
   long long aug_failed_gen = -1;
   bool checked_witness = env_int("PIPS_HIP_AUG_WITNESS", 1) != 0;
   // Every solveCompressed that goes by sweeps is measured like that (pips_hip_kkt_set_solve_check: every k-th one; 0 = the witness
   // only, rounds 4's behaviour): the reference's PARDISO measures and refines EVERY leaf solve (iparm[7] = 2,
   // PardisoProjectSolver.C:72), and one clean right-hand side does not bound the backward error of the next.  Several ranks decide
   // together: each solveCompressed ends with a one-number all-reduce "did any rank's check fail"; if so every rank restores its
   // right-hand side and all go the refined way (a rank's inaccurate -Br^T K^-1 b taints x0 for everybody).
   int solve_check_every = 1, sweeps_since_check = 0;
   long long solves_since_factor = 0;   // equal on every rank: which solveCompressed calls are scheduled for a measure (every solve_check_every-th)
   double* h_flag = nullptr;            // pinned: the one-number exchange of settle() without a host wait before the collective
   long long checked_solves = 0, failed_checks = 0;
   double* d_flag = nullptr;
   bool joint_aug_any = false;        // several ranks: some rank's analysis chose the sweeps (all-reduced once per analysis)
   long long joint_aug_gen = -1;
   double *d_bsave = nullptr, *d_b0save = nullptr;
   bool root_pivoting_set = false;   // pips_hip_kkt_set_root_pivoting decided; else: Bunch-Kaufman iff root inequality rows are eliminated
   // phase times of one factorize and the solveCompressed calls after it (pips_hip_kkt_get_timing; on with the batch's timing switch):
   // 0 diagonals + zero SC, 1 leaf factorisation, 2 Schur reduction, 3 finalize, 4 root factorisation (its own stream),
   // 5 Lsolve leaf solves, 6 Lsolve border product + b0 reduction, 7 Dsolve, 8 Ltsolve, 9 x_i = z_i - u_i,
   // 10 panel-wise Schur reduction on its own stream (sum over the panels; phase 2 is then only what the main stream waited for it)
   PhaseTimer timer;
   // sparse root: SC lives as the value array of a lower-triangular CSR pattern inside a one-block sparse
   // engine, which factorises and solves it with the leaf machinery (ordering, head / dense tail, refinement)
   bool sparse = false;
   std::unique_ptr<Engine> root_sp;
   std::vector<int> sc_rowptr, sc_colidx, root_perm, root_colcount;
   int root_order_mode = 0;   // sparse root: 0 minimum degree, 1 dense-tile band, 2 dissection around the hubs
   long long *d_xdiag_pos = nullptr, *d_zlink_pos = nullptr;
   int* d_sc_rowptr = nullptr;
   ~KktSystem() {
      if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
      if (graph_stream) (void)hipStreamDestroy(graph_stream);
      if (root_stream) { (void)hipStreamSynchronize(root_stream); (void)hipStreamDestroy(root_stream); }
      if (ev_sc_final) (void)hipEventDestroy(ev_sc_final);
      if (ev_root_done) (void)hipEventDestroy(ev_root_done);
      if (comm_stream) (void)hipStreamDestroy(comm_stream);
      if (ev_reduced) (void)hipEventDestroy(ev_reduced);
      if (h_flag) (void)hipHostFree(h_flag);
      void* ptrs[] = {d_SC, d_t, d_fin_val, d_fin_idx, d_c0_val, d_red, d_c0_rp, d_c0_ci, d_packed, d_xdiag_pos, d_zlink_pos, d_sc_rowptr, d_gall, d_gvec_all, d_bsave, d_b0save, d_flag};
      for (void* p : ptrs)
         if (p) (void)hipFree(p);
   }
};

// vals[pos[i]] += d[i]
__global__ void k_add_at(double* __restrict__ vals, const long long* __restrict__ pos, const double* __restrict__ d, int n) {
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) vals[pos[i]] += d[i];
}

// SC[0:n0,0:n0] -= C0^T diag(zdiag)^-1 C0 (lower triangle; zdiag < 0): schur_complement_add_CTDC_block
// (sLinsysRootAug.C:1276-1338, SparseStorage::matTransDinvMultMat SparseStorage.C:1257).  One thread per row of C0.
// sc_rowptr != nullptr: SC is the value array of the sparse root's CSR pattern, whose x0 block is dense: (i, j), j <= i < n0,
// sits at sc_rowptr[i] + j
__global__ void k_ctdc(int mz0, const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                       const double* __restrict__ zdiag, double* __restrict__ SC, int ld, const int* __restrict__ sc_rowptr) {
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < mz0; k += gridDim.x * blockDim.x) {
      const double dinv = 1.0 / zdiag[k];
      for (int p = rp[k]; p < rp[k + 1]; ++p)
         for (int q = rp[k]; q < rp[k + 1]; ++q) {
            const int i = ci[p], j = ci[q];
            if (i >= j) atomic_add_f64(sc_rowptr ? SC + sc_rowptr[i] + j : SC + i + (long long)j * ld, -v[p] * v[q] * dinv);
         }
   }
}

// solveReducedLinkCons (sLinsysRootAug.C:384-466), z0 elimination: mode 0: t_k = b3_k / zdiag_k ; rhs1 -= C0^T t
//                                                                     mode 1: b3_k = (b3_k - (C0 x1)_k) / zdiag_k
__global__ void k_z0_elim(int mode, int mz0, const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                          const double* __restrict__ zdiag, double* __restrict__ b3, double* __restrict__ x1) {
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < mz0; k += gridDim.x * blockDim.x) {
      if (mode == 0) {
         const double t = b3[k] / zdiag[k];
         for (int p = rp[k]; p < rp[k + 1]; ++p) atomic_add_f64(x1 + ci[p], -v[p] * t);
      } else {
         double s = b3[k];
         for (int p = rp[k]; p < rp[k + 1]; ++p) s -= v[p] * x1[ci[p]];
         b3[k] = s / zdiag[k];
      }
   }
}

// add_regularization_local_kkt (DistributedLeafLinearSystem.C:108-143): K diag += primal on the leading n_primal rows,
// -= dual on the rest
__global__ void k_add_regularization(const BlkDesc* __restrict__ blks, const int* __restrict__ n_primal,
                                     const long long* __restrict__ kdiag, double* __restrict__ kval, double primal, double dual) {
   const BlkDesc bd = blks[blockIdx.y];
   const int np = n_primal[blockIdx.y];
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bd.n; i += gridDim.x * blockDim.x)
      kval[kdiag[bd.x_off + i]] += (np < 0 || i < np) ? primal : -dual;
}

// pack / unpack the lower triangle of the column-major S x S Schur complement (column c holds S - c entries):
// the reference reduces packed triangles too (submatrixAllReduceDiagLower, DistributedRootLinearSystem.C:1661-1707)
__global__ void k_pack_lower(const double* __restrict__ M, int ld, int S, double* __restrict__ packed, int unpack) {
   const int c = blockIdx.y;
   const long long base = (long long)c * S - (long long)c * (c - 1) / 2;
   double* col = const_cast<double*>(M) + (long long)c * ld;
   for (int r = c + blockIdx.x * blockDim.x + threadIdx.x; r < S; r += gridDim.x * blockDim.x) {
      if (unpack) col[r] = packed[base + (r - c)];
      else packed[base + (r - c)] = col[r];
   }
}

// the same for a row panel [R0, R1) of the lower triangle: column c < R1 contributes its rows max(c, R0) .. R1 - 1
__global__ void k_pack_rows(const double* __restrict__ M, int ld, int R0, int R1, double* __restrict__ packed, int unpack) {
   const int c = blockIdx.y;
   const long long h = R1 - R0;
   const long long base = c <= R0 ? (long long)c * h
                                  : (long long)R0 * h + (long long)(c - R0) * R1 - ((long long)c * (c - 1) / 2 - (long long)R0 * (R0 - 1) / 2);
   const int r_first = c > R0 ? c : R0;
   double* col = const_cast<double*>(M) + (long long)c * ld;
   for (int r = r_first + blockIdx.x * blockDim.x + threadIdx.x; r < R1; r += gridDim.x * blockDim.x) {
      if (unpack) col[r] = packed[base + (r - r_first)];
      else packed[base + (r - r_first)] = col[r];
   }
}

// diagonal_add_constant_from: dense column-major SC (rowptr == nullptr) or the CSR lower pattern of the sparse SC, whose rows end
// with their diagonal entry
__global__ void k_add_const_diag(double* __restrict__ M, int ld, const int* __restrict__ rowptr, int first, int n, double value) {
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int r = first + i;
      if (rowptr) M[rowptr[r + 1] - 1] += value;
      else M[(long long)r * ld + r] += value;
   }
}

__global__ void k_add_diag(double* __restrict__ M, int ld, int off, const double* __restrict__ d, int n) {
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      M[(long long)(off + i) * ld + off + i] += d[i];
}

}  // namespace pips

// =================================================================================================================
// C ABI
// =================================================================================================================
using namespace pips;

extern "C" {

const char* pips_hip_last_error(void) { return pips::last_error(); }
long long pips_hip_host_wait_count(void) { return pips::g_host_waits.load(std::memory_order_relaxed); }
int pips_hip_host_wait_sites(char* buf, int cap) {
   int used = 0;
   if (buf && cap > 0) buf[0] = 0;
   for (const pips::WaitSite& w : pips::g_wait_sites) {
      const char* f = w.file.load();
      if (!f || w.n.load() == 0) continue;
      const char* base = strrchr(f, '/');
      const int k = snprintf(buf ? buf + used : nullptr, buf && cap > used ? cap - used : 0, "%s:%d %lld\n", base ? base + 1 : f, w.line.load(), w.n.load());
      if (k < 0 || !buf || used + k >= cap) break;
      used += k;
   }
   return used;
}

int pips_hip_device_count(void) {
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

static int resolve_device(int device, int* out) {
   int cnt = 0;
   if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0)
      PIPS_FAIL(PIPS_ERR_NO_DEVICE, "no HIP device visible: the MI355X backend has no CPU fallback");
   if (device < 0) { HIP_TRY(hipGetDevice(&device)); }
   if (device >= cnt) PIPS_FAIL(PIPS_ERR_ARG, "device %d out of range (%d visible)", device, cnt);
   *out = device;
   return PIPS_OK;
}

// ---- batch -------------------------------------------------------------------------------------------------------
int pips_hip_batch_create(void** handle, int n_blocks, int S, int device, void* stream) {
   if (!handle || n_blocks <= 0 || S < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_create: bad arguments");
   auto e = std::make_unique<Engine>();
   e->nblk = n_blocks;
   e->S = S;
   e->device = device;   // resolved at analyze time so that set_block / symbolic work without a GPU
   e->stream = (hipStream_t)stream;
   e->in.assign(n_blocks, BlockInput());
   if (const char* d = getenv("PIPS_HIP_DETERMINISTIC")) e->deterministic = atoi(d) != 0;   // default of pips_hip_batch_set_deterministic
   *handle = e.release();
   return PIPS_OK;
}

int pips_hip_batch_set_block(void* handle, int b, int n, int n_primal, const int* K_rowptr, const int* K_colidx,
                             const int* Bt_rowptr, const int* Bt_colidx, const double* Bt_val) {
   Engine* e = (Engine*)handle;
   if (!e || b < 0 || b >= e->nblk || n <= 0 || !K_rowptr || !K_colidx) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_set_block: bad arguments");
   BlockInput& in = e->in[b];
   in.n = n;
   in.n_primal = n_primal;
   in.krow.assign(K_rowptr, K_rowptr + n + 1);
   in.kcol.assign(K_colidx, K_colidx + K_rowptr[n]);
   in.btrow.clear(); in.btcol.clear(); in.btval.clear();
   if (Bt_rowptr && e->S > 0) {
      in.btrow.assign(Bt_rowptr, Bt_rowptr + e->S + 1);
      const int nnz = Bt_rowptr[e->S];
      in.btcol.assign(Bt_colidx, Bt_colidx + nnz);
      if (Bt_val) in.btval.assign(Bt_val, Bt_val + nnz); else in.btval.assign(nnz, 0.0);
   }
   e->analyzed = false;
   return PIPS_OK;
}

int pips_hip_batch_set_options(void* handle, int force_n_head, int refine_steps, double thr_rel, double repl_rel) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   e->opt.force_n_head = force_n_head;
   if (refine_steps >= 0) e->refine_steps = refine_steps;
   if (thr_rel >= 0) e->thr_rel = thr_rel;
   if (repl_rel > 0) e->repl_rel = repl_rel;
   return PIPS_OK;
}

int pips_hip_batch_set_schur_mode(void* handle, int mode) {
   Engine* e = (Engine*)handle;
   if (!e || mode < 0 || mode > 2) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_set_schur_mode: mode must be 0 (auto), 1 or 2");
   if (e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_set_schur_mode: call before analyze");
   e->schur_mode = mode;
   return PIPS_OK;
}

int pips_hip_batch_set_deterministic(void* handle, int on) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_set_deterministic: call before pips_hip_batch_analyze");
   e->deterministic = on != 0;
   return PIPS_OK;
}

int pips_hip_batch_get_schur_mode(void* handle, int* mode) {
   Engine* e = (Engine*)handle;
   if (!e || !mode || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_get_schur_mode: analyze first");
   *mode = e->schur_mode_eff;
   return PIPS_OK;
}

int pips_hip_batch_add_regularization(void* handle, double primal, double dual) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_add_regularization: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   hipLaunchKernelGGL(k_add_regularization, dim3(32, e->nblk), dim3(256), 0, e->stream, e->d_blks, e->d_nprimal, e->d_kdiag, e->d_kval,
                      primal, dual);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

int pips_hip_batch_set_refinement(void* handle, int max_steps, double tol) {
   Engine* e = (Engine*)handle;
   if (!e || max_steps < 0 || tol < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_set_refinement: bad arguments");
   e->refine_steps = max_steps;
   e->refine_tol = tol;
   e->refine_mode = 0;
   return PIPS_OK;
}

int pips_hip_batch_set_refinement_backward_error(void* handle, int max_steps, double tol) {
   Engine* e = (Engine*)handle;
   if (!e || max_steps < 0 || tol < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_set_refinement_backward_error: bad arguments");
   e->refine_steps = max_steps;
   e->refine_tol = tol;
   e->refine_mode = 1;
   return PIPS_OK;
}

double pips_hip_batch_last_refinement_measure(void* handle) {
   Engine* e = (Engine*)handle;
   return e ? e->last_refine_measure : -1.0;
}

int pips_hip_batch_last_refinement_steps(void* handle) {
   Engine* e = (Engine*)handle;
   return e ? e->last_refine_steps : -1;
}

int pips_hip_batch_analyze(void* handle, int n_threads) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   int dev;
   int rc = resolve_device(e->device, &dev);
   if (rc) return rc;
   e->device = dev;
   return e->analyze(n_threads);
}

int pips_hip_batch_set_values(void* handle, int b, const double* K_val_host) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed || b < 0 || b >= e->nblk) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_set_values: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   HIP_TRY(hipMemcpyAsync(e->d_kval + e->kptr[b], K_val_host, (size_t)(e->kptr[b + 1] - e->kptr[b]) * sizeof(double),
                          hipMemcpyHostToDevice, e->stream));
   HIP_TRY(hipStreamSynchronize(e->stream));
   return PIPS_OK;
}

int pips_hip_batch_set_diagonals_dev(void* handle, const double* diag_dev) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_set_diagonals: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   hipLaunchKernelGGL(k_put_diag, dim3(grid_for(e->n_total, 256)), dim3(256), 0, e->stream, e->d_kdiag, diag_dev, e->d_kval,
                      e->n_total);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

int pips_hip_batch_set_diagonals(void* handle, const double* diag_host) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_set_diagonals: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   HIP_TRY(hipMemcpyAsync(e->d_stage, diag_host, (size_t)e->n_total * sizeof(double), hipMemcpyHostToDevice, e->stream));
   int rc = pips_hip_batch_set_diagonals_dev(handle, e->d_stage);
   if (rc) return rc;
   HIP_TRY(hipStreamSynchronize(e->stream));
   return PIPS_OK;
}

int pips_hip_batch_factor(void* handle, double* SC_dev, int ldSC) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (SC_dev && ldSC < e->S) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_batch_factor: ldSC %d < S %d", ldSC, e->S);
   return e->factor(SC_dev, ldSC);
}

int pips_hip_batch_solve_dev(void* handle, double* x_dev) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   return e->solve(x_dev);
}

int pips_hip_batch_solve(void* handle, double* x_host) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_solve: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   const size_t bytes = (size_t)e->n_total * sizeof(double);
   HIP_TRY(hipMemcpyAsync(e->d_stage, x_host, bytes, hipMemcpyHostToDevice, e->stream));
   int rc = e->solve(e->d_stage);
   if (rc) return rc;
   HIP_TRY(hipMemcpyAsync(x_host, e->d_stage, bytes, hipMemcpyDeviceToHost, e->stream));
   HIP_TRY(hipStreamSynchronize(e->stream));
   return e->sweep.take_error("pips_hip_batch_solve");
}

int pips_hip_batch_border_tmult_dev(void* handle, const double* z_dev, double* b0_dev, double alpha) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "border_tmult: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   if (e->bt_rows_total > 0 && e->deterministic) {
      hipLaunchKernelGGL(k_border_rowdot, dim3(grid_for(e->bt_rows_total, 256)), dim3(256), 0, e->stream, e->d_bt_rowptr, e->d_bt_colidx, e->d_bval,
                         e->d_bt_xoff, z_dev, e->d_bt_tmp, e->bt_rows_total, alpha);
      e->gather(e->g_btm, e->d_bt_tmp, b0_dev);
   } else if (e->bt_rows_total > 0)
      hipLaunchKernelGGL(k_border_tmult, dim3(grid_for(e->bt_rows_total * BT_LANES, 256, 65536)), dim3(256), 0, e->stream, e->d_bt_rowptr,
                         e->d_bt_colidx, e->d_bval, e->d_bt_rowsc, e->d_bt_xoff, z_dev, b0_dev, e->bt_rows_total, alpha);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

int pips_hip_batch_border_mult_dev(void* handle, const double* x0_dev, double* t_dev, double alpha) {
   Engine* e = (Engine*)handle;
   if (!e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "border_mult: analyze first");
   HIP_TRY(hipSetDevice(e->device));
   if (e->bt_rows_total > 0 && e->deterministic) {
      hipLaunchKernelGGL(k_border_entry_products, dim3(grid_for(e->bt_rows_total, 256)), dim3(256), 0, e->stream, e->d_bt_rowptr, e->d_bval,
                         e->d_bt_rowsc, x0_dev, e->d_bt_tmp, e->bt_rows_total, alpha);
      e->gather(e->g_bm, e->d_bt_tmp, t_dev);
   } else if (e->bt_rows_total > 0 && e->d_br_rowptr)
      hipLaunchKernelGGL(k_border_mult_rows, dim3(grid_for(e->n_total, 256, 1 << 20)), dim3(256), 0, e->stream, e->d_br_rowptr, e->d_br_sc, e->d_br_src,
                         e->d_bval, x0_dev, t_dev, e->n_total, alpha);
   else if (e->bt_rows_total > 0)
      hipLaunchKernelGGL(k_border_mult, dim3(grid_for(e->bt_rows_total * BT_LANES, 256, 65536)), dim3(256), 0, e->stream, e->d_bt_rowptr,
                         e->d_bt_colidx, e->d_bval, e->d_bt_rowsc, e->d_bt_xoff, x0_dev, t_dev, e->bt_rows_total, alpha);
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

int pips_hip_batch_inertia(void* handle, int b, int* pos, int* neg, int* zero) {
   Engine* e = (Engine*)handle;
   if (!e || !e->factored || b < 0 || b >= e->nblk) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_inertia: factor first");
   int rc = e->fetch_inertia();
   if (rc) return rc;
   if (pos) *pos = e->h_inertia[3 * b];
   if (neg) *neg = e->h_inertia[3 * b + 1];
   if (zero) *zero = e->h_inertia[3 * b + 2];
   return PIPS_OK;
}

static void sym_info(const std::vector<BlockSym>& sym, int64_t* what, int n_what) {
   int64_t v[13] = {0};
   for (const BlockSym& s : sym) {
      v[10] += (int64_t)s.upd.size() * 4;
      v[11] += s.nb;
      v[12] += (int64_t)s.a_dst.size();
      v[0] += s.nnzL; v[1] += s.n; v[2] += s.n_head; v[3] += s.m; v[4] += (int64_t)s.sn.size();
      v[5] = std::max<int64_t>(v[5], s.n_levels);
      v[6] += (int64_t)s.flops_factor; v[7] += (int64_t)s.flops_border; v[8] += s.arena * 8;
      v[9] = std::max<int64_t>(v[9], s.m_pad / TILE);
   }
   for (int i = 0; i < n_what && i < 13; ++i) what[i] = v[i];
}

int pips_hip_batch_info(void* handle, int64_t* what, int n_what) {
   Engine* e = (Engine*)handle;
   if (!e || e->sym.empty()) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_batch_info: analyze first");
   sym_info(e->sym, what, n_what);
   if (n_what > 13) what[13] = e->border_backward_ok ? 1 : 0;
   if (n_what > 14) what[14] = e->mf ? 1 : 0;
   if (n_what > 15) { what[15] = 0; for (const BlockSym& s : e->sym) what[15] = std::max<int64_t>(what[15], s.mf_max_front); }
   if (n_what > 16) what[16] = e->mfU_total * 8;
   if (n_what > 17) { what[17] = 0; for (const MfLaunch& m : e->mf_launches) if (m.cls >= 6) what[17] += m.cnt; }
   if (n_what > 19) {   // the sparse head alone: stored entries of L and row indices (algorithmic bytes of its factorisation)
      what[18] = what[19] = 0;
      for (const BlockSym& s : e->sym) {
         what[18] += s.nnzL - (int64_t)s.m * (s.m + 1) / 2;
         what[19] += (int64_t)s.rowidx.size();
      }
   }
   if (n_what > 21) {   // entries of L in border rows (head panels / border-row arena + the tails' border rows): read by the sweeps of the
                        // augmented factor, not by a solve with K_i; and whether those sweeps may serve solveCompressed
      what[20] = 0;
      int64_t tail_border = 0;
      for (const BlockSym& s : e->sym) {
         for (const HeadSupernode& hs : s.sn) what[20] += (int64_t)hs.w * (hs.r - hs.rb);   // (counted in nnzL: stored with the head)
         tail_border += (int64_t)s.nb * s.m;                                                   // (not counted in nnzL)
      }
      what[21] = e->aug_sweeps_ok ? 1 : 0;
      if (n_what > 22) what[22] = e->aug_passes;
      if (n_what > 23) what[23] = tail_border;
   }
   if (n_what > 24) { what[24] = 0; for (const BlockSym& s : e->sym) what[24] += s.mf_split ? 1 : 0; }   // blocks with the border split
   if (n_what > 25) { what[25] = 0; for (const BlockSym& s : e->sym) what[25] += (e->mf && s.mf_konly) ? 1 : 0; }   // ... whose fronts hold the rows of K only
   return PIPS_OK;
}

int pips_hip_batch_sync(void* handle) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   HIP_TRY(hipSetDevice(e->device));
   HIP_TRY(hipStreamSynchronize(e->stream));
   return e->sweep.take_error("pips_hip_batch_sync");
}

int pips_hip_batch_set_timing(void* handle, int on) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   e->timer.on = on != 0;
   return PIPS_OK;
}

int pips_hip_batch_get_timing(void* handle, double* ms, int64_t* cnt, int n) {
   Engine* e = (Engine*)handle;
   if (!e) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   HIP_TRY(hipSetDevice(e->device));
   HIP_TRY(hipStreamSynchronize(e->stream));
   e->timer.collect();
   for (int i = 0; i < n && i < PhaseTimer::NPHASE; ++i) {
      if (ms) ms[i] = e->timer.ms[i];
      if (cnt) cnt[i] = e->timer.cnt[i];
   }
   return PIPS_OK;
}

void pips_hip_batch_destroy(void* handle) { delete (Engine*)handle; }

// ---- single leaf solver handle -------------------------------------------------------------------------------------
struct LdlGroup;
// pips_hip_ldl_factor_schur of single handles: the leaf kernels address the Schur complement as a dense S x S array, but a leaf only touches
// the nb x nb entries of its non-empty border columns - one S x S scratch array per DEVICE serves every handle (a call leaves it with a
// stream synchronisation; 64 handles with an array each were 32 GB at S = 8000), and only the compact nb x nb block travels to the host.
struct SchurScratch {
   std::mutex mu;
   std::map<int, std::pair<double*, size_t>> per_device;   // device -> (array, doubles)
};
static SchurScratch g_schur_scratch;

// C[i + j * nb] = SC[bm[i] + bm[j] * ld] (the block of a leaf's border columns, column-major; the lower triangle is what the caller reads) -
// and SC is left zero on the whole block, whichever triangle a Schur mode wrote
__global__ void k_schur_take_block(double* __restrict__ SC, int ld, const int* __restrict__ bm, int nb, double* __restrict__ C) {
   const int j = blockIdx.y;
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
      double* p = SC + bm[i] + (long long)bm[j] * ld;
      C[i + (long long)j * nb] = *p;
      *p = 0.0;
   }
}

struct LdlHandle {
   Engine eng;
   bool have_perm = false;
   double* d_sc = nullptr;        // compact nb x nb Schur term of this leaf (pips_hip_ldl_factor_schur)
   std::vector<double> h_sc;
   std::shared_ptr<LdlGroup> group;   // the leaves of a rank bound into one batch engine (pips_hip_ldl_factor_schur_batch)
   int group_index = -1;
   // which engine holds this leaf's NEWEST factors: the batch's (set for every member by pips_hip_ldl_factor_schur_batch) or the handle's
   // own (pips_hip_ldl_factor / _factor_schur clear it) - a handle can go through both over its life (matrixChanged() alone, later the
   // host's loop handed over), and solve / inertia must never answer from the older of the two
   bool newest_in_group = false;
   ~LdlHandle();
};
// Array-of-handles entries (INTEGRATION.md level 1.5b): the reference keeps one DoubleLinearSolver per leaf and loops over its children
// (sLinsysRootAug::assembleLocalKKT :210-227, Lsolve / Ltsolve :323-365).  One leaf alone is a latency chain on the device (64 leaves of
// configs[1] one after the other: 1.17 s per factorisation against 0.1 s as one batch), so the handles of a rank can be bound into ONE batch
// engine: the patterns and borders the handles were given, analysed together; factor_schur_batch / solve_batch then run all leaves in every
// launch, one S x S Schur buffer on the device and one transfer of it per factorisation instead of one per leaf.
struct LdlGroup {
   Engine eng;
   std::vector<LdlHandle*> members;
   double *d_sc = nullptr, *d_x = nullptr;
   std::vector<double> h_sc;
   ~LdlGroup() { if (d_sc) (void)hipFree(d_sc); if (d_x) (void)hipFree(d_x); }
};
// a handle that goes away leaves an empty slot in its group: the siblings keep the batch's factors (their own solves go through
// group_solve_rows, which needs no sibling), the next array-of-handles call sees another array and binds anew
LdlHandle::~LdlHandle() {
   if (d_sc) (void)hipFree(d_sc);
   if (group && group_index >= 0 && group_index < (int)group->members.size() && group->members[group_index] == this) group->members[group_index] = nullptr;
}
// array-of-handles solves / queries answer from the batch's factors: refuse if a member has been factorised alone since
static int ldl_group_is_current(const LdlGroup& g, const char* who) {
   if (!g.eng.factored) PIPS_FAIL(PIPS_ERR_STATE, "%s: pips_hip_ldl_factor_schur_batch first", who);
   for (size_t i = 0; i < g.members.size(); ++i)
      if (g.members[i] && !g.members[i]->newest_in_group)
         PIPS_FAIL(PIPS_ERR_STATE, "%s: handle %d was factorised on its own after the batch's factorisation - factorise the array again", who, (int)i);
   return PIPS_OK;
}
static inline bool ldl_uses_group(const LdlHandle* h) { return h->group && h->newest_in_group && h->group->eng.factored; }
// one member's right-hand sides through the batch engine (every block in every launch, the others with zeros - correct, and as long as
// a batch solve: hosts that loop over their leaves should hand the loop over, pips_hip_ldl_solve_batch).  rhs: nrhs rows of length ld,
// on the host or on the device
static int group_solve_rows(LdlGroup& g, int index, int nrhs, double* rhs, long long ld, bool on_device, const char* who) {
   Engine& e = g.eng;
   HIP_TRY(hipSetDevice(e.device));
   if (!g.d_x) HIP_TRY(hipMalloc((void**)&g.d_x, (size_t)std::max<long long>(e.n_total, 1) * sizeof(double)));
   const size_t cnt = (size_t)(e.x_off[index + 1] - e.x_off[index]) * sizeof(double);
   for (int r = 0; r < nrhs; ++r) {
      HIP_TRY(hipMemsetAsync(g.d_x, 0, (size_t)e.n_total * sizeof(double), e.stream));
      HIP_TRY(hipMemcpyAsync(g.d_x + e.x_off[index], rhs + (size_t)r * ld, cnt, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e.stream));
      int rc = e.solve(g.d_x);
      if (rc) return rc;
      HIP_TRY(hipMemcpyAsync(rhs + (size_t)r * ld, g.d_x + e.x_off[index], cnt, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e.stream));
   }
   if (!on_device) HIP_TRY(hipStreamSynchronize(e.stream));
   return e.sweep.take_error(who);
}

int pips_hip_ldl_create(void** handle, int n, const int* krow, const int* jcol, int device, int flags) {
   (void)flags;
   if (!handle || n <= 0 || !krow || !jcol) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_create: bad arguments");
   auto h = std::make_unique<LdlHandle>();
   h->eng.nblk = 1;
   h->eng.S = 0;
   h->eng.device = device;
   h->eng.in.assign(1, BlockInput());
   BlockInput& in = h->eng.in[0];
   in.n = n;
   in.krow.assign(krow, krow + n + 1);
   in.kcol.assign(jcol, jcol + krow[n]);
   *handle = h.release();
   return PIPS_OK;
}

int pips_hip_ldl_set_inertia_hint(void* handle, int n_primal) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   h->eng.in[0].n_primal = n_primal;
   return PIPS_OK;
}

int pips_hip_ldl_set_pivot_rule(void* handle, double thr_rel, double repl_rel) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   h->eng.thr_rel = thr_rel;
   h->eng.repl_rel = repl_rel;
   return PIPS_OK;
}

int pips_hip_ldl_set_refinement(void* handle, int max_steps, double tol) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   h->eng.refine_steps = max_steps;
   h->eng.refine_tol = tol;
   h->eng.refine_mode = 0;
   return PIPS_OK;
}

int pips_hip_ldl_set_deterministic(void* handle, int on) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (h->eng.analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_set_deterministic: before the first factorisation (the symbolic phase builds the slot lists)");
   h->eng.deterministic = on != 0;
   return PIPS_OK;
}

int pips_hip_ldl_set_refinement_backward_error(void* handle, int max_steps, double tol) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || max_steps < 0 || tol < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_set_refinement_backward_error: bad arguments");
   h->eng.refine_steps = max_steps;
   h->eng.refine_tol = tol;
   h->eng.refine_mode = 1;
   return PIPS_OK;
}

int pips_hip_ldl_analyze(void* handle) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   return pips_hip_batch_analyze(&h->eng, 1);
}

int pips_hip_ldl_factor(void* handle, const double* vals_host) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || !vals_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor: bad arguments");
   if (!h->eng.analyzed) {
      int rc = pips_hip_ldl_analyze(handle);
      if (rc) return rc;
   }
   int rc = pips_hip_batch_set_values(&h->eng, 0, vals_host);
   if (rc) return rc;
   rc = h->eng.factor(nullptr, 0);
   if (rc) return rc;
   h->newest_in_group = false;      // (the batch's copy of this leaf is the older one now)
   HIP_TRY(hipStreamSynchronize(h->eng.stream));
   return PIPS_OK;
}

// The leaf's Schur term without dense border columns over PCIe.  The border Br_i^T (S x n CSR, rows = Schur column ids: the
// reference's border_left_transp, DistributedLeafLinearSystem.C:214-252) is declared before the analysis; factor_schur factorises
// K_i and adds  -Br_i^T K_i^-1 Br_i  to the caller's Schur complement: what the K4-K6 chunk loop
// (addBiTLeftKiBiRightToResBlockedParallelSolvers, DistributedLinearSystem.C:766-1047: densify <= 20 T border columns,
// solve(nrhs, ...), sparse product back) computes, but with only the CSR values going up and the S x S term coming down.
int pips_hip_ldl_set_border(void* handle, int S, const int* Bt_rowptr, const int* Bt_colidx) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || S <= 0 || !Bt_rowptr || !Bt_colidx) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_set_border: bad arguments");
   if (h->eng.analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_set_border: declare the border before pips_hip_ldl_analyze / the first factor");
   BlockInput& in = h->eng.in[0];
   for (int p = 0; p < Bt_rowptr[S]; ++p)
      if (Bt_colidx[p] < 0 || Bt_colidx[p] >= in.n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_set_border: column %d outside the leaf (n = %d)", Bt_colidx[p], in.n);
   h->eng.S = S;
   in.btrow.assign(Bt_rowptr, Bt_rowptr + S + 1);
   in.btcol.assign(Bt_colidx, Bt_colidx + Bt_rowptr[S]);
   in.btval.assign((size_t)Bt_rowptr[S], 0.0);
   return PIPS_OK;
}

int pips_hip_ldl_factor_schur(void* handle, const double* K_vals_host, const double* Bt_vals_host, double* SC_host, int ldSC) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || !K_vals_host || !Bt_vals_host || !SC_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur: bad arguments");
   Engine& e = h->eng;
   const int S = e.S;
   if (S <= 0 || e.in[0].btrow.empty()) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_factor_schur: no border declared (pips_hip_ldl_set_border)");
   if (ldSC < S) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur: ldSC %d < S %d", ldSC, S);
   int rc;
   if (!e.analyzed && (rc = pips_hip_ldl_analyze(handle))) return rc;
   HIP_TRY(hipSetDevice(e.device));
   if ((rc = pips_hip_batch_set_values(&e, 0, K_vals_host))) return rc;
   if (e.nnzB_total > 0) HIP_TRY(hipMemcpyAsync(e.d_bval, Bt_vals_host, (size_t)e.nnzB_total * sizeof(double), hipMemcpyHostToDevice, e.stream));
   const BlockSym& bs = e.sym[0];
   const int nb = (int)bs.bmap.size();
   // the device's shared S x S scratch array: zero on entry (allocated zeroed; every call takes its block out again and leaves zeros)
   // (held to the end of the call: host threads that factorise their leaves side by side take turns on the scratch array)
   std::lock_guard<std::mutex> scratch_lock(g_schur_scratch.mu);
   double* scratch = nullptr;
   {
      auto& slot = g_schur_scratch.per_device[e.device];
      if (slot.second < (size_t)S * S) {
         if (slot.first) (void)hipFree(slot.first);
         slot = {nullptr, 0};
         HIP_TRY(hipMalloc((void**)&slot.first, (size_t)S * S * sizeof(double)));
         slot.second = (size_t)S * S;
         HIP_TRY(hipMemset(slot.first, 0, (size_t)S * S * sizeof(double)));
      }
      scratch = slot.first;
   }
   if ((rc = e.factor(scratch, S))) {   // (whatever reached the scratch array must not meet the next handle)
      (void)hipStreamSynchronize(e.stream);
      (void)hipMemset(scratch, 0, (size_t)S * S * sizeof(double));
      return rc;
   }
   h->newest_in_group = false;
   if (nb > 0) {
      if (!h->d_sc) HIP_TRY(hipMalloc((void**)&h->d_sc, (size_t)nb * nb * sizeof(double)));
      hipLaunchKernelGGL(k_schur_take_block, dim3(std::max(1, std::min(16, (nb + 255) / 256)), nb), dim3(256), 0, e.stream, scratch, S, e.d_bmap, nb, h->d_sc);
      h->h_sc.resize((size_t)nb * nb);
      HIP_TRY(hipMemcpyAsync(h->h_sc.data(), h->d_sc, (size_t)nb * nb * sizeof(double), hipMemcpyDeviceToHost, e.stream));
   }
   HIP_TRY(hipStreamSynchronize(e.stream));
   // device: column-major with the lower triangle valid; caller: row-major DenseSymmetricMatrix, lower triangle meaningful
   // (DenseStorage.C:64-83): entry [i][j], i >= j, takes the device's (i, j) - only the rows of this leaf's non-empty border columns differ from zero
   for (int j = 0; j < nb; ++j)
      for (int i = j; i < nb; ++i) SC_host[(size_t)bs.bmap[i] * ldSC + bs.bmap[j]] += h->h_sc[(size_t)i + (size_t)j * nb];
   return e.sweep.take_error("pips_hip_ldl_factor_schur");
}

extern "C" int pips_hip_ldl_solve_batch(void* const* handles, int n, double* const* rhs_inout_host);
// flags[q] = 1 where column q of X (ld apart, n entries) has an entry that is not zero (a NaN counts)
__global__ __launch_bounds__(256) void k_columns_nonzero(const double* __restrict__ X, long long ld, int n, int* __restrict__ flags) {
   const double* v = X + ld * blockIdx.x;
   int any = 0;
   for (int i = threadIdx.x; i < n; i += 256) any |= (v[i] != 0.0) ? 1 : 0;
   any = __syncthreads_or(any);
   if (threadIdx.x == 0) flags[blockIdx.x] = any ? 1 : 0;
}

int pips_hip_ldl_solve(void* handle, int nrhs, double* rhs, int ld) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || nrhs < 0 || !rhs || ld < h->eng.in[0].n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_solve: bad arguments");
   if (ldl_uses_group(h))   // the newest factors of this leaf are the batch's
      return group_solve_rows(*h->group, h->group_index, nrhs, rhs, ld, false, "pips_hip_ldl_solve");
   Engine& e = h->eng;
   if (!e.factored) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_solve: factor first");
   if (nrhs == 0) return PIPS_OK;
   HIP_TRY(hipSetDevice(e.device));
   const size_t row = (size_t)e.n_total * sizeof(double);
   if (nrhs == 1) {
      HIP_TRY(hipMemcpyAsync(e.d_stage, rhs, row, hipMemcpyHostToDevice, e.stream));
      int rc = e.solve(e.d_stage);
      if (rc) return rc;
      HIP_TRY(hipMemcpyAsync(rhs, e.d_stage, row, hipMemcpyDeviceToHost, e.stream));
      HIP_TRY(hipStreamSynchronize(e.stream));
      return PIPS_OK;
   }
   // all right-hand sides in one pass (one RHS per row of length ld).  Like PardisoSolver::solve (PardisoSolver.C:276-352) only the non-zero
   // right-hand sides are solved.  Which ones those are is found on the device AFTER the upload: a scan on the host walks every column up
   // to its first entry - the border columns of a chunk have theirs in the dual rows, behind 10 000 zeros each, 1.3 ms per call of 160 -
   // while the whole chunk crosses the link in 0.35 ms and the flags come back with one small copy.
   // (the device copy of the right-hand sides stays with the handle: the reference's loop calls this hundreds of times per factorisation
   // with the same chunk size, an allocation and a release per call were a fifth of a millisecond each)
   const size_t flag_bytes = ((size_t)nrhs * sizeof(int) + 15) & ~(size_t)15;
   if (e.hostx_cap < (size_t)nrhs * row + flag_bytes) {
      if (e.d_hostx) { HIP_TRY(hipStreamSynchronize(e.stream)); (void)hipFree(e.d_hostx); e.d_hostx = nullptr; e.hostx_cap = 0; }
      HIP_TRY(hipMalloc((void**)&e.d_hostx, (size_t)nrhs * row + flag_bytes));
      e.hostx_cap = (size_t)nrhs * row + flag_bytes;
   }
   double* d_X = e.d_hostx;
   int* d_flags = (int*)((char*)e.d_hostx + (size_t)nrhs * row);
   std::vector<int> flags((size_t)nrhs), nz;
   HIP_TRY(hipMemcpy2DAsync(d_X, row, rhs, (size_t)ld * sizeof(double), row, nrhs, hipMemcpyHostToDevice, e.stream));
   hipLaunchKernelGGL(k_columns_nonzero, dim3(nrhs), dim3(256), 0, e.stream, d_X, (long long)e.n_total, (int)e.n_total, d_flags);
   HIP_TRY(hipMemcpyAsync(flags.data(), d_flags, (size_t)nrhs * sizeof(int), hipMemcpyDeviceToHost, e.stream));
   HIP_TRY(hipStreamSynchronize(e.stream));
   nz.reserve(nrhs);
   for (int r = 0; r < nrhs; ++r) if (flags[r]) nz.push_back(r);
   const int nnz_rhs = (int)nz.size();
   if (nnz_rhs == 0) return PIPS_OK;
   hipError_t err = hipSuccess;
   if (nnz_rhs < nrhs)   // the non-zero columns move to the front (ascending: a column never lands on one that is still to move)
      for (int q = 0; q < nnz_rhs && err == hipSuccess; ++q)
         if (nz[q] != q) err = hipMemcpyAsync(d_X + (size_t)q * e.n_total, d_X + (size_t)nz[q] * e.n_total, row, hipMemcpyDeviceToDevice, e.stream);
   int rc = err == hipSuccess ? (nnz_rhs == 1 ? e.solve(d_X) : e.solve_multi(d_X, nnz_rhs, e.n_total)) : PIPS_ERR_HIP;
   if (!rc) {
      if (nnz_rhs == nrhs)
         err = hipMemcpy2DAsync(rhs, (size_t)ld * sizeof(double), d_X, row, row, nrhs, hipMemcpyDeviceToHost, e.stream);
      else
         for (int q = 0; q < nnz_rhs && err == hipSuccess; ++q)
            err = hipMemcpyAsync(rhs + (size_t)nz[q] * ld, d_X + (size_t)q * e.n_total, row, hipMemcpyDeviceToHost, e.stream);
   }
   if (err == hipSuccess) err = hipStreamSynchronize(e.stream);
   if (rc) return rc;
   if (err != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "pips_hip_ldl_solve: %s", hipGetErrorString(err));
   return PIPS_OK;
}

// ---- device-pointer and sparse-row variants of the per-leaf solve
int pips_hip_ldl_solve_dev(void* handle, int nrhs, double* rhs_inout_dev, long long ld) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || nrhs < 1 || !rhs_inout_dev || ld < h->eng.in[0].n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_solve_dev: bad arguments");
   if (ldl_uses_group(h)) return group_solve_rows(*h->group, h->group_index, nrhs, rhs_inout_dev, ld, true, "pips_hip_ldl_solve_dev");
   Engine& e = h->eng;
   if (!e.factored) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_solve_dev: factor first");
   HIP_TRY(hipSetDevice(e.device));
   return nrhs == 1 ? e.solve(rhs_inout_dev) : e.solve_multi(rhs_inout_dev, nrhs, ld);
}

// rows[i] = rhs row (0 <= row < n) of packed entry i: X[q * ld + rows[i]] = packed[q * n_rows + i]
__global__ void k_expand_rows(const int* __restrict__ rows, int n_rows, const double* __restrict__ packed, double* __restrict__ X, long long ld, int nrhs) {
   const long long total = (long long)n_rows * nrhs;
   for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
      const int q = (int)(idx / n_rows), i = (int)(idx - (long long)q * n_rows);
      X[q * ld + rows[i]] = packed[idx];
   }
}

// = DoubleLinearSolver::solve(int nrhss, double* rhss, int* colSparsity) with the third argument honoured: colSparsity[i] != 0 marks the rows
// that can be non-zero in any of the right-hand sides (the caller's border_left_transp pattern, DistributedLinearSystem.C:903).  Only those
// rows of the non-zero right-hand sides travel to the device (the solutions come back dense: K^-1 fills them).
int pips_hip_ldl_solve_sparse(void* handle, int nrhs, double* rhs, int ld, const int* col_sparsity) {
   if (!col_sparsity) return pips_hip_ldl_solve(handle, nrhs, rhs, ld);
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || nrhs < 0 || !rhs || ld < h->eng.in[0].n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_solve_sparse: bad arguments");
   if (ldl_uses_group(h)) return pips_hip_ldl_solve(handle, nrhs, rhs, ld);   // (through the batch every row travels: colSparsity saves nothing there)
   Engine& e = h->eng;
   if (!e.factored) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_solve_sparse: factor first");
   HIP_TRY(hipSetDevice(e.device));
   const int n = (int)e.n_total;
   std::vector<int> rows, nz;
   for (int i = 0; i < n; ++i) if (col_sparsity[i]) rows.push_back(i);
   // where most rows are marked the whole columns travel faster than a pack on the host (the union over a chunk of 160 border columns of
   // configs[1] marks a fifth of the rows; 19 MB cross the link in 0.35 ms): the pack pays below an eighth
   if ((long long)rows.size() * 8 >= n) return pips_hip_ldl_solve(handle, nrhs, rhs, ld);
   for (int r = 0; r < nrhs; ++r) {
      const double* v = rhs + (size_t)r * ld;
      bool any = false;
      for (int i : rows) if (v[i] != 0.0) { any = true; break; }
      if (any) nz.push_back(r);
   }
   const int nq = (int)nz.size(), nr = (int)rows.size();
   if (nq == 0) return PIPS_OK;
   std::vector<double> packed((size_t)nq * nr);
   for (int q = 0; q < nq; ++q) {
      const double* v = rhs + (size_t)nz[q] * ld;
      for (int i = 0; i < nr; ++i) packed[(size_t)q * nr + i] = v[rows[i]];
   }
   // (device copies kept with the handle, like pips_hip_ldl_solve's: no allocation and release per call)
   const size_t row = (size_t)n * sizeof(double), pack_bytes = packed.size() * sizeof(double) + (size_t)nr * sizeof(int);
   if (e.hostx_cap < (size_t)nq * row) {
      if (e.d_hostx) { HIP_TRY(hipStreamSynchronize(e.stream)); (void)hipFree(e.d_hostx); e.d_hostx = nullptr; e.hostx_cap = 0; }
      HIP_TRY(hipMalloc((void**)&e.d_hostx, (size_t)nq * row));
      e.hostx_cap = (size_t)nq * row;
   }
   if (e.hostpack_cap < pack_bytes) {
      if (e.d_hostpack) { HIP_TRY(hipStreamSynchronize(e.stream)); (void)hipFree(e.d_hostpack); e.d_hostpack = nullptr; e.hostpack_cap = 0; }
      HIP_TRY(hipMalloc((void**)&e.d_hostpack, pack_bytes));
      e.hostpack_cap = pack_bytes;
   }
   double *d_X = e.d_hostx, *d_packed = e.d_hostpack;
   int* d_rows = (int*)(e.d_hostpack + packed.size());
   hipError_t err = hipMemcpyAsync(d_packed, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice, e.stream);
   if (err == hipSuccess) err = hipMemcpyAsync(d_rows, rows.data(), (size_t)nr * sizeof(int), hipMemcpyHostToDevice, e.stream);
   if (err == hipSuccess) err = hipMemsetAsync(d_X, 0, (size_t)nq * row, e.stream);
   int rc = PIPS_OK;
   if (err == hipSuccess) {
      hipLaunchKernelGGL(k_expand_rows, dim3(grid_for((long long)nq * nr, 256)), dim3(256), 0, e.stream, d_rows, nr, d_packed, d_X, (long long)n, nq);
      rc = nq == 1 ? e.solve(d_X) : e.solve_multi(d_X, nq, n);
      if (!rc && nq == nrhs)
         err = hipMemcpy2DAsync(rhs, (size_t)ld * sizeof(double), d_X, row, row, nrhs, hipMemcpyDeviceToHost, e.stream);
      else
         for (int q = 0; q < nq && !rc && err == hipSuccess; ++q)
            err = hipMemcpyAsync(rhs + (size_t)nz[q] * ld, d_X + (size_t)q * n, row, hipMemcpyDeviceToHost, e.stream);
      if (err == hipSuccess) err = hipStreamSynchronize(e.stream);   // (the host vectors packed / rows live until here)
   }
   if (rc) return rc;
   if (err != hipSuccess) PIPS_FAIL(PIPS_ERR_HIP, "pips_hip_ldl_solve_sparse: %s", hipGetErrorString(err));
   return PIPS_OK;
}

// ---- array-of-handles entries
static int ldl_group_of(void* const* handles, int n, std::shared_ptr<LdlGroup>& out) {
   if (!handles || n <= 0) PIPS_FAIL(PIPS_ERR_ARG, "array of leaf handles: bad arguments");
   LdlHandle* h0 = (LdlHandle*)handles[0];
   if (!h0) PIPS_FAIL(PIPS_ERR_ARG, "array of leaf handles: null handle");
   bool same = h0->group && (int)h0->group->members.size() == n;
   for (int i = 0; i < n && same; ++i) same = handles[i] && h0->group->members[i] == (LdlHandle*)handles[i];
   if (same) { out = h0->group; return PIPS_OK; }
   // bind: one batch engine over the patterns (and borders) the handles hold
   auto g = std::make_shared<LdlGroup>();
   Engine& e = g->eng;
   e.nblk = n;
   e.S = h0->eng.S;
   e.device = h0->eng.device;
   e.thr_rel = h0->eng.thr_rel; e.repl_rel = h0->eng.repl_rel;
   e.refine_steps = h0->eng.refine_steps; e.refine_tol = h0->eng.refine_tol; e.refine_mode = h0->eng.refine_mode;
   e.in.resize(n);
   for (int i = 0; i < n; ++i) {
      LdlHandle* h = (LdlHandle*)handles[i];
      if (!h) PIPS_FAIL(PIPS_ERR_ARG, "array of leaf handles: null handle at %d", i);
      if (h->eng.S != e.S) PIPS_FAIL(PIPS_ERR_ARG, "array of leaf handles: handle %d declares Schur dimension %d, handle 0 %d", i, h->eng.S, e.S);
      if (h->eng.device != e.device) PIPS_FAIL(PIPS_ERR_ARG, "array of leaf handles: handles on different devices");
      e.in[i] = h->eng.in[0];
      g->members.push_back(h);
   }
   int rc = pips_hip_batch_analyze(&e, std::min(n, 16));
   if (rc) return rc;
   for (int i = 0; i < n; ++i) { ((LdlHandle*)handles[i])->group = g; ((LdlHandle*)handles[i])->group_index = i; }
   out = g;
   return PIPS_OK;
}

int pips_hip_ldl_factor_schur_batch(void* const* handles, int n, const double* const* K_vals_host, const double* const* Bt_vals_host, double* SC_host,
                                    int ldSC) {
   std::shared_ptr<LdlGroup> g;
   int rc = ldl_group_of(handles, n, g);
   if (rc) return rc;
   if (!K_vals_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur_batch: no values");
   Engine& e = g->eng;
   const int S = e.S;
   const bool schur = SC_host != nullptr;
   if (schur && (S <= 0 || ldSC < S || !Bt_vals_host)) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur_batch: Schur term asked for without border / ldSC %d < S %d", ldSC, S);
   HIP_TRY(hipSetDevice(e.device));
   for (int i = 0; i < n; ++i) {
      if (!K_vals_host[i]) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur_batch: no values for leaf %d", i);
      HIP_TRY(hipMemcpyAsync(e.d_kval + e.kptr[i], K_vals_host[i], (size_t)(e.kptr[i + 1] - e.kptr[i]) * sizeof(double), hipMemcpyHostToDevice, e.stream));
   }
   if (schur && e.nnzB_total > 0) {
      long long off = 0;
      for (int i = 0; i < n; ++i) {
         const size_t cnt = e.in[i].btcol.size();
         if (cnt > 0) {
            if (!Bt_vals_host[i]) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_factor_schur_batch: no border values for leaf %d", i);
            HIP_TRY(hipMemcpyAsync(e.d_bval + off, Bt_vals_host[i], cnt * sizeof(double), hipMemcpyHostToDevice, e.stream));
         }
         off += (long long)cnt;
      }
   }
   if (schur) {
      if (!g->d_sc) HIP_TRY(hipMalloc((void**)&g->d_sc, (size_t)S * S * sizeof(double)));
      HIP_TRY(hipMemsetAsync(g->d_sc, 0, (size_t)S * S * sizeof(double), e.stream));
   }
   if ((rc = e.factor(schur ? g->d_sc : nullptr, S))) return rc;
   for (LdlHandle* m : g->members) if (m) m->newest_in_group = true;
   if (schur) {
      g->h_sc.resize((size_t)S * S);
      HIP_TRY(hipMemcpyAsync(g->h_sc.data(), g->d_sc, (size_t)S * S * sizeof(double), hipMemcpyDeviceToHost, e.stream));
      HIP_TRY(hipStreamSynchronize(e.stream));
      // device: column-major, lower triangle valid; caller: row-major DenseSymmetricMatrix (DenseStorage.C:64-83), lower triangle meaningful
      std::vector<char> used(S, 0);
      for (const BlockSym& bs : e.sym) for (int c : bs.bmap) used[c] = 1;
      std::vector<int> cols;
      for (int c = 0; c < S; ++c) if (used[c]) cols.push_back(c);
      for (int cb : cols)
         for (int ra : cols)
            if (ra >= cb) SC_host[(size_t)ra * ldSC + cb] += g->h_sc[(size_t)ra + (size_t)cb * S];
   } else
      HIP_TRY(hipStreamSynchronize(e.stream));
   return e.sweep.take_error("pips_hip_ldl_factor_schur_batch");
}

int pips_hip_ldl_solve_batch_dev(void* const* handles, int n, double* x_dev) {
   std::shared_ptr<LdlGroup> g;
   int rc = ldl_group_of(handles, n, g);
   if (rc) return rc;
   if (!x_dev) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_solve_batch_dev: null vector");
   if ((rc = ldl_group_is_current(*g, "pips_hip_ldl_solve_batch_dev"))) return rc;
   HIP_TRY(hipSetDevice(g->eng.device));
   return g->eng.solve(x_dev);
}

int pips_hip_ldl_solve_batch(void* const* handles, int n, double* const* rhs_inout_host) {
   std::shared_ptr<LdlGroup> g;
   int rc = ldl_group_of(handles, n, g);
   if (rc) return rc;
   if (!rhs_inout_host) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_ldl_solve_batch: null array");
   Engine& e = g->eng;
   if ((rc = ldl_group_is_current(*g, "pips_hip_ldl_solve_batch"))) return rc;
   HIP_TRY(hipSetDevice(e.device));
   if (!g->d_x) HIP_TRY(hipMalloc((void**)&g->d_x, (size_t)std::max<long long>(e.n_total, 1) * sizeof(double)));
   // a leaf without a right-hand side this time (NULL) is solved with zeros: the batch runs every block in every launch
   HIP_TRY(hipMemsetAsync(g->d_x, 0, (size_t)e.n_total * sizeof(double), e.stream));
   for (int i = 0; i < n; ++i)
      if (rhs_inout_host[i])
         HIP_TRY(hipMemcpyAsync(g->d_x + e.x_off[i], rhs_inout_host[i], (size_t)(e.x_off[i + 1] - e.x_off[i]) * sizeof(double), hipMemcpyHostToDevice, e.stream));
   if ((rc = e.solve(g->d_x))) return rc;
   for (int i = 0; i < n; ++i)
      if (rhs_inout_host[i])
         HIP_TRY(hipMemcpyAsync(rhs_inout_host[i], g->d_x + e.x_off[i], (size_t)(e.x_off[i + 1] - e.x_off[i]) * sizeof(double), hipMemcpyDeviceToHost, e.stream));
   HIP_TRY(hipStreamSynchronize(e.stream));
   return e.sweep.take_error("pips_hip_ldl_solve_batch");
}

int pips_hip_ldl_inertia_batch(void* const* handles, int n, int* pos, int* neg, int* zero) {
   std::shared_ptr<LdlGroup> g;
   int rc = ldl_group_of(handles, n, g);
   if (rc) return rc;
   if ((rc = ldl_group_is_current(*g, "pips_hip_ldl_inertia_batch"))) return rc;
   for (int i = 0; i < n; ++i) {
      int p = 0, q = 0, z = 0;
      if ((rc = pips_hip_batch_inertia(&g->eng, i, &p, &q, &z))) return rc;
      if (pos) pos[i] = p;
      if (neg) neg[i] = q;
      if (zero) zero[i] = z;
   }
   return PIPS_OK;
}

int pips_hip_ldl_inertia(void* handle, int* pos, int* neg, int* zero) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (ldl_uses_group(h))   // the newest factorisation of this leaf was the batch's (pips_hip_ldl_factor_schur_batch)
      return pips_hip_batch_inertia(&h->group->eng, h->group_index, pos, neg, zero);
   return pips_hip_batch_inertia(&h->eng, 0, pos, neg, zero);
}

int pips_hip_ldl_info(void* handle, int64_t* what, int n_what) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || h->eng.sym.empty()) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_info: analyze first");
   const BlockSym& s = h->eng.sym[0];
   int64_t v[8] = {s.nnzL, s.n_head, s.m, (int64_t)s.sn.size(), s.n_levels, (int64_t)s.flops_factor, (int64_t)h->eng.last_refine_steps, (int64_t)h->eng.last_multi_path};
   for (int i = 0; i < n_what && i < 8; ++i) what[i] = v[i];
   return PIPS_OK;
}

int pips_hip_ldl_get_perm(void* handle, int* perm) {
   LdlHandle* h = (LdlHandle*)handle;
   if (!h || h->eng.sym.empty() || !perm) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_ldl_get_perm: analyze first");
   std::copy(h->eng.sym[0].perm.begin(), h->eng.sym[0].perm.end(), perm);
   return PIPS_OK;
}

void pips_hip_ldl_destroy(void* handle) { delete (LdlHandle*)handle; }

// ---- dense root ----------------------------------------------------------------------------------------------------
int pips_hip_dense_ldl_create(void** handle, int n, int n_primal, int device) {
   if (!handle || n <= 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_create: bad arguments");
   int dev;
   int rc = resolve_device(device, &dev);
   if (rc) return rc;
   auto d = std::make_unique<DenseLdl>();
   d->n = n;
   d->n_primal = n_primal;
   d->device = dev;
   // no inertia hint = a plain DeSymIndefSolver replacement: pivot like dsytrf; with a hint the caller vouches for the quasi-definite order
   d->pivoting = n_primal < 0 ? 1 : 0;
   rc = d->init();
   if (rc) return rc;
   *handle = d.release();
   return PIPS_OK;
}

int pips_hip_dense_ldl_set_distributed(void* handle, void* comm, int rank, int n_ranks) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   d->factored = false;
   return d->set_distributed(comm, rank, n_ranks);
}

int pips_hip_dense_ldl_set_pivoting(void* handle, int mode) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || mode < 0 || mode > 1) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_set_pivoting: mode 0 (static order) or 1 (Bunch-Kaufman inside the diagonal tiles)");
   d->pivoting = mode;
   d->factored = false;
   return PIPS_OK;
}

int pips_hip_dense_ldl_factor(void* handle, const double* A_host, int lda) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || !A_host || lda < d->n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_factor: bad arguments");
   HIP_TRY(hipSetDevice(d->device));
   // DenseStorage is row-major with the lower triangle authoritative (DeSymIndefSolver.h:41-42 hands exactly this to
   // dsytrf_('U') as a column-major matrix); upload and let the copy kernel read it transposed.
   if (int rc0 = d->ensure_input_buffer()) return rc0;
   HIP_TRY(hipMemcpy2DAsync(d->d_in, (size_t)d->n * sizeof(double), A_host, (size_t)lda * sizeof(double),
                            (size_t)d->n * sizeof(double), (size_t)d->n, hipMemcpyHostToDevice, d->stream));
   int rc = d->factor_dev(d->d_in, d->n, 1);
   if (rc) return rc;
   if ((rc = d->check_pivots())) return rc;     // (the staged copy d_in stays valid: a new pivot order factorises from it again)
   HIP_TRY(hipStreamSynchronize(d->stream));
   return d->take_root_error("pips_hip_dense_ldl_factor");
}

int pips_hip_dense_ldl_factor_dev(void* handle, const double* A_dev, int lda) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || !A_dev || lda < d->n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_factor_dev: bad arguments");
   return d->factor_dev(A_dev, lda, 0);
}

int pips_hip_dense_ldl_solve_dev(void* handle, double* rhs_inout_dev) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || !rhs_inout_dev) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_solve_dev: bad arguments");
   return d->solve_dev(rhs_inout_dev);
}

int pips_hip_dense_ldl_solve(void* handle, int nrhs, double* rhs, int ld) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || !rhs || nrhs < 0 || ld < d->n) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_dense_ldl_solve: bad arguments");
   HIP_TRY(hipSetDevice(d->device));
   if (int rc0 = d->ensure_input_buffer()) return rc0;
   for (int k = 0; k < nrhs; ++k) {
      double* x = rhs + (size_t)k * ld;
      HIP_TRY(hipMemcpyAsync(d->d_in, x, (size_t)d->n * sizeof(double), hipMemcpyHostToDevice, d->stream));
      int rc = d->solve_dev(d->d_in);
      if (rc) return rc;
      HIP_TRY(hipMemcpyAsync(x, d->d_in, (size_t)d->n * sizeof(double), hipMemcpyDeviceToHost, d->stream));
      HIP_TRY(hipStreamSynchronize(d->stream));
   }
   return d->sweep.take_error("pips_hip_dense_ldl_solve");
}

int pips_hip_dense_ldl_inertia(void* handle, int* pos, int* neg, int* zero) {
   DenseLdl* d = (DenseLdl*)handle;
   if (!d || !d->factored) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_dense_ldl_inertia: factor first");
   HIP_TRY(hipSetDevice(d->device));
   { const int rcp = d->check_pivots(); if (rcp) return rcp; }
   HIP_TRY(hipMemcpyAsync(d->h_inertia, d->d_inertia, 3 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
   HIP_TRY(hipStreamSynchronize(d->stream));
   { const int rce = d->sweep.take_error("pips_hip_dense_ldl_inertia"); if (rce) return rce; }
   { const int rce = d->take_root_error("pips_hip_dense_ldl_inertia"); if (rce) return rce; }
   if (pos) *pos = d->h_inertia[0];
   if (neg) *neg = d->h_inertia[1];
   if (zero) *zero = d->h_inertia[2];
   return PIPS_OK;
}

void pips_hip_dense_ldl_destroy(void* handle) { delete (DenseLdl*)handle; }

// ---- plain device buffers ------------------------------------------------------------------------------------------
int pips_hip_malloc(void** dev_ptr, size_t bytes) {
   if (!dev_ptr) PIPS_FAIL(PIPS_ERR_ARG, "null pointer");
   HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 8));
   return PIPS_OK;
}
int pips_hip_free(void* dev_ptr) {
   if (dev_ptr) HIP_TRY(hipFree(dev_ptr));
   return PIPS_OK;
}
int pips_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
   HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
   return PIPS_OK;
}
int pips_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
   HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
   return PIPS_OK;
}
int pips_hip_memset(void* dst, int value, size_t bytes) {
   HIP_TRY(hipMemset(dst, value, bytes));
   return PIPS_OK;
}

// ---- fused KKT system ---------------------------------------------------------------------------------------------
int pips_hip_allreduce_sum(void* comm, double* buf_dev, size_t n, void* stream);

int pips_hip_kkt_create(void** handle, void* batch, int n0, int my0, int myl, int mzl, const int* A0_rowptr,
                        const int* A0_colidx, const double* A0_val, const int* F0_rowptr, const int* F0_colidx,
                        const double* F0_val, const int* G0_rowptr, const int* G0_colidx, const double* G0_val, void* comm,
                        int rank, int n_ranks) {
   Engine* e = (Engine*)batch;
   if (!handle || !e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_create: the leaf batch must be analyzed first");
   const int S = n0 + my0 + myl + mzl;
   if (S != e->S) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_create: n0+my0+myl+mzl = %d but the batch was created with S = %d", S, e->S);
   auto k = std::make_unique<KktSystem>();
   k->leaves = e;
   k->n0 = n0; k->my0 = my0; k->myl = myl; k->mzl = mzl; k->S = S;
   k->comm = comm; k->rank = rank; k->n_ranks = n_ranks;
   k->force_reduce = comm && getenv("PIPS_HIP_FORCE_REDUCE") != nullptr;
   HIP_TRY(hipSetDevice(e->device));
   k->root = std::make_unique<DenseLdl>();
   k->root->n = S;
   k->root->n_primal = n0;
   k->root->device = e->device;
   k->root->stream = e->stream;
   k->root->thr_rel = e->thr_rel;
   k->root->repl_rel = e->repl_rel;
   int rc = k->root->init();
   if (rc) return rc;
   // several ranks: the dense root factorised column-cyclically over the ranks instead of redundantly on every one of them
   // (PIPS_HIP_ROOT_DISTRIBUTED=1; untimed - see DenseLdl::set_distributed)
   if (comm && n_ranks > 1 && env_int("PIPS_HIP_ROOT_DISTRIBUTED", 0) != 0 && (rc = k->root->set_distributed(comm, rank, n_ranks))) return rc;
   HIP_TRY(hipMalloc((void**)&k->d_SC, (size_t)S * S * sizeof(double)));
   HIP_TRY(hipMalloc((void**)&k->d_t, std::max<size_t>((size_t)e->n_total, 1) * sizeof(double)));
   // constant root blocks added by finalizeKKTdense: A0 at row n0, F0 at row n0+my0, G0 at row n0+my0+myl
   // (sLinsysRootAug.C:270-320, 1782-1796).  SC is column-major with the lower triangle valid: (r,c) -> r + c*S.
   std::vector<long long> idx;
   std::vector<double> val;
   auto add = [&](const int* rp, const int* ci, const double* v, int rows, int r0) {
      if (!rp) return;
      for (int r = 0; r < rows; ++r)
         for (int p = rp[r]; p < rp[r + 1]; ++p) { idx.push_back((long long)(r0 + r) + (long long)ci[p] * S); val.push_back(v[p]); }
   };
   add(A0_rowptr, A0_colidx, A0_val, my0, n0);
   add(F0_rowptr, F0_colidx, F0_val, myl, n0 + my0);
   add(G0_rowptr, G0_colidx, G0_val, mzl, n0 + my0 + myl);
   k->n_fin = (long long)idx.size();
   if ((rc = dev_upload(&k->d_fin_idx, idx, nullptr))) return rc;
   if ((rc = dev_upload(&k->d_fin_val, val, nullptr))) return rc;
   // several ranks: Schur SYRK in row panels, each reduced as soon as it is final (PIPS_HIP_SC_PANELS, default 4 for S >= 1024; 1 =
   // one reduction after all leaf work); PIPS_HIP_SC_REDUCE=rsag: reduce-scatter + all-gather instead of the all-reduce
   if (e->deterministic && (rc = e->set_det_groups(rank, n_ranks))) return rc;
   if (comm && (n_ranks > 1 || k->force_reduce) && !e->deterministic) {
      // default: panels only where the reduction is worth hiding (S >= 4096: >= 64 MB packed; splitting the SYRK costs ~1 ms)
      int panels = S >= 4096 ? 4 : 1;
      if (const char* pp = getenv("PIPS_HIP_SC_PANELS")) panels = atoi(pp);
      if ((rc = e->set_sc_panels(panels))) return rc;
      if (const char* m = getenv("PIPS_HIP_SC_REDUCE")) k->use_rsag = std::string(m) == "rsag";
   }
   *handle = k.release();
   return PIPS_OK;
}

// Largest column count the dissected sparse root admits in its head: a front of colcount + 1 rows must fit the LDS as a packed triangle
// (19 200 doubles: 195 rows).  Measured (tools/sparse_root_probe.py, 64 blocks): 31 linking rows per pair (fronts <= 188 rows) factorize
// 4.3 ms as a band, 2.6 ms dissected; 100 rows per pair (fronts of 308 rows: update matrices in device memory) 10.1 ms as a band, 11.1 ms
// dissected and solveCompressed 2.1 -> 4.3 ms - those stay a band.
constexpr int ROOT_ND_MAX_COLCOUNT = 192;

// Sparse-root variant (createSchurCompSymbSparseUpper, DistributedProblem.cpp:2235+; finalizeKKTsparse, sLinsysRootAug.C:
// 1629-1739).  Pattern of SC (lower): the dense x0 block, for every block the clique on its non-empty border columns, the
// root rows A0 / F0 / G0 and a full diagonal.  With 2-link structure (a linking row touches two blocks) it stays sparse.
// blk_cols_ptr / blk_cols (optional): the border column sets of ALL blocks of the problem (needed with n_ranks > 1, where a
// rank only knows its own blocks but every rank must reduce the same value array); NULL: the blocks of this batch.
int pips_hip_kkt_create_sparse(void** handle, void* batch, int n0, int my0, int myl, int mzl, const int* A0_rowptr,
                               const int* A0_colidx, const double* A0_val, const int* F0_rowptr, const int* F0_colidx,
                               const double* F0_val, const int* G0_rowptr, const int* G0_colidx, const double* G0_val,
                               int n_blocks_global, const int* blk_cols_ptr, const int* blk_cols, void* comm, int rank,
                               int n_ranks) {
   Engine* e = (Engine*)batch;
   if (!handle || !e || !e->analyzed) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_create_sparse: the batch must be analyzed");
   const int S = n0 + my0 + myl + mzl;
   if (S != e->S) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_create_sparse: n0+my0+myl+mzl = %d but the batch was created with S = %d", S, e->S);
   if (n_ranks > 1 && !blk_cols_ptr) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_create_sparse: n_ranks > 1 needs the border column sets of all blocks");
   auto k = std::make_unique<KktSystem>();
   k->leaves = e;
   k->n0 = n0; k->my0 = my0; k->myl = myl; k->mzl = mzl; k->S = S;
   k->comm = comm; k->rank = rank; k->n_ranks = n_ranks;
   k->force_reduce = comm && getenv("PIPS_HIP_FORCE_REDUCE") != nullptr;
   k->sparse = true;
   HIP_TRY(hipSetDevice(e->device));
   // ---- pattern, row by row (lower triangle, sorted, explicit diagonal)
   std::vector<std::vector<int>> rows(S);
   for (int r = 0; r < S; ++r) rows[r].push_back(r);
   for (int r = 0; r < n0; ++r)
      for (int c = 0; c < r; ++c) rows[r].push_back(c);
   auto add_rows = [&](const int* rp, const int* ci, int nrows, int r0) {
      if (!rp) return;
      for (int r = 0; r < nrows; ++r)
         for (int p = rp[r]; p < rp[r + 1]; ++p) rows[r0 + r].push_back(ci[p]);
   };
   add_rows(A0_rowptr, A0_colidx, my0, n0);
   add_rows(F0_rowptr, F0_colidx, myl, n0 + my0);
   add_rows(G0_rowptr, G0_colidx, mzl, n0 + my0 + myl);
   auto add_clique = [&](const int* cols, int nc) {
      for (int a = 0; a < nc; ++a)
         for (int b = 0; b <= a; ++b) rows[cols[a]].push_back(cols[b]);   // cols ascending: cols[a] >= cols[b]
   };
   if (blk_cols_ptr) {
      for (int b = 0; b < n_blocks_global; ++b) add_clique(blk_cols + blk_cols_ptr[b], blk_cols_ptr[b + 1] - blk_cols_ptr[b]);
   }
   for (int b = 0; b < e->nblk; ++b) add_clique(e->sym[b].bmap.data(), (int)e->sym[b].bmap.size());
   k->sc_rowptr.assign(S + 1, 0);
   for (int r = 0; r < S; ++r) {
      std::sort(rows[r].begin(), rows[r].end());
      rows[r].erase(std::unique(rows[r].begin(), rows[r].end()), rows[r].end());
      k->sc_rowptr[r + 1] = k->sc_rowptr[r] + (int)rows[r].size();
   }
   k->sc_colidx.reserve(k->sc_rowptr[S]);
   for (int r = 0; r < S; ++r) k->sc_colidx.insert(k->sc_colidx.end(), rows[r].begin(), rows[r].end());
   auto pos_of = [&](int r, int c) -> long long {
      const int* b0 = k->sc_colidx.data() + k->sc_rowptr[r];
      const int* b1 = k->sc_colidx.data() + k->sc_rowptr[r + 1];
      const int* it = std::lower_bound(b0, b1, c);
      return (it != b1 && *it == c) ? (long long)(it - k->sc_colidx.data()) : -1;
   };
   // ---- per-block position tables for the leaf kernels
   std::vector<int> tab;
   std::vector<long long> off(e->nblk, 0);
   for (int b = 0; b < e->nblk; ++b) {
      const std::vector<int>& bm = e->sym[b].bmap;
      const int nb = (int)bm.size();
      off[b] = (long long)tab.size();
      tab.resize(tab.size() + (size_t)nb * nb, 0);
      for (int la = 0; la < nb; ++la)
         for (int lb = 0; lb <= la; ++lb) tab[off[b] + (long long)la * nb + lb] = (int)pos_of(bm[la], bm[lb]);
   }
   int rc = e->set_sc_tables(tab, off, (long long)k->sc_rowptr[S]);
   if (rc) return rc;
   if (e->deterministic && (rc = e->set_det_groups(rank, n_ranks))) return rc;   // group buffers as long as the value array
   // ---- the root as a one-block sparse engine; its value array is the Schur complement
   k->root_sp = std::make_unique<Engine>();
   Engine* r = k->root_sp.get();
   r->nblk = 1; r->S = 0; r->device = e->device; r->stream = e->stream;
   r->thr_rel = e->thr_rel; r->repl_rel = e->repl_rel;
   r->deterministic = e->deterministic;   // (the root's own factorisation and sweeps: slots and fixed-order gathers instead of atomics)
   r->in.assign(1, BlockInput());
   r->in[0].n = S; r->in[0].n_primal = n0;
   r->in[0].krow = k->sc_rowptr; r->in[0].kcol = k->sc_colidx;
   // Elimination order.  The link-link block of SC is negative definite on its own (-sum F_i (K_i^-1)_xx F_i^T), so the linking
   // rows can go first with their expected signs, then x0, then the root equality rows y0 (zero diagonal block: they need x0
   // before them).  In that order a 2-link Schur complement is banded: if its tile envelope is thin the root is factorised as
   // an all-dense-tile band (TailPlan envelope: band^2 work per column on the MFMA kernels) - otherwise minimum degree with
   // the usual head / tail split decides (linking rows still before x0 unless y0 rows exist).
   k->root_perm.clear();
   for (int i = n0 + my0; i < S; ++i) k->root_perm.push_back(i);
   for (int i = 0; i < n0 + my0; ++i) k->root_perm.push_back(i);
   {
      std::vector<int> ipos(S);
      for (int t = 0; t < S; ++t) ipos[k->root_perm[t]] = t;
      const int nt = (S + TILE - 1) / TILE;
      std::vector<int> first(nt);
      for (int t = 0; t < nt; ++t) first[t] = t;
      for (int rr = 0; rr < S; ++rr)
         for (int p = k->sc_rowptr[rr]; p < k->sc_rowptr[rr + 1]; ++p) {
            const int a = ipos[rr], b = ipos[k->sc_colidx[p]];
            const int tr = std::max(a, b) / TILE, tc = std::min(a, b) / TILE;
            first[tr] = std::min(first[tr], tc);
         }
      double env = 0;
      for (int t = 0; t < nt; ++t) env += t - first[t] + 1;
      bool banded = env <= 0.25 * 0.5 * nt * (nt + 1.0);
      // A thin band is a chain: as dense tiles its diagonal tiles are factorised one after the other (63 tiles of 86 us at S = 8000 -
      // as long as the dense root).  Dissected around the hubs x0 / y0 (ordered last) the linking rows become a tree of small fronts
      // for the multifrontal head, a dozen dependent launches deep; only the hubs and the top separators stay dense.
      int mode = banded ? 2 : 0;   // 0 minimum degree, 1 dense-tile band, 2 dissection (falls back to 1 without separators)
      if (const char* f = getenv("PIPS_HIP_SPARSE_ROOT_BAND")) mode = atoi(f);   // tests: force a path
      if (mode == 2) {
         std::vector<int> ap(S + 1, 0), ai, hubs, nd_perm;
         for (int rr = 0; rr < S; ++rr)
            for (int p = k->sc_rowptr[rr]; p < k->sc_rowptr[rr + 1]; ++p)
               if (k->sc_colidx[p] != rr) { ++ap[rr + 1]; ++ap[k->sc_colidx[p] + 1]; }
         for (int i = 0; i < S; ++i) ap[i + 1] += ap[i];
         ai.resize(ap[S]);
         {
            std::vector<int> fill(ap.begin(), ap.end() - 1);
            for (int rr = 0; rr < S; ++rr)
               for (int p = k->sc_rowptr[rr]; p < k->sc_rowptr[rr + 1]; ++p) {
                  const int c = k->sc_colidx[p];
                  if (c != rr) { ai[fill[rr]++] = c; ai[fill[c]++] = rr; }
               }
         }
         for (int i = 0; i < n0 + my0; ++i) hubs.push_back(i);
         // head = the dissected rows as long as their fronts stay LDS-resident in k_front (ROOT_ND_MAX_COLCOUNT); the cost model is no
         // guide here (it prices a scattering head against MFMA throughput, and the band's cost is the latency of its chain of diagonal tiles)
         int cut = 0;
         const bool dissected = hub_dissected_order(S, ap, ai, hubs, 48, nd_perm, k->root_colcount) && (int)nd_perm.size() == S;
         if (dissected) {
            const int n_rest = S - (int)hubs.size();
            while (cut < n_rest && k->root_colcount[cut] <= ROOT_ND_MAX_COLCOUNT) ++cut;
         }
         // (without separators - one linking row or none, say - nd_perm is empty: never take it, whatever the threshold says)
         if (dissected && cut > 0 && cut >= (S - (int)hubs.size()) / 2) { k->root_perm = nd_perm; r->opt.force_n_head = cut; r->sn_width = HEAD_WMAX; }
         else mode = 1;   // (the band order - linking rows, then x0 - is always built above: root_perm holds it)
      }
      if (mode == 2) {
         r->opt.user_perm = k->root_perm.data();
         r->opt.user_colcount = k->root_colcount.data();
      } else if (mode == 1) {
         r->opt.user_perm = k->root_perm.data();
         r->opt.force_n_head = 0;
      } else {
         r->opt.constrain_order = my0 > 0;
      }
      k->root_order_mode = mode;
   }
   if ((rc = r->analyze(4))) return rc;
   // ---- constant root entries and the diagonals added by finalizeKKT
   std::vector<long long> idx, xpos(n0), zpos(mzl);
   std::vector<double> val;
   auto add = [&](const int* rp, const int* ci, const double* v, int nrows, int r0) {
      if (!rp) return;
      for (int rr = 0; rr < nrows; ++rr)
         for (int p = rp[rr]; p < rp[rr + 1]; ++p) { idx.push_back(pos_of(r0 + rr, ci[p])); val.push_back(v[p]); }
   };
   add(A0_rowptr, A0_colidx, A0_val, my0, n0);
   add(F0_rowptr, F0_colidx, F0_val, myl, n0 + my0);
   add(G0_rowptr, G0_colidx, G0_val, mzl, n0 + my0 + myl);
   for (int i = 0; i < n0; ++i) xpos[i] = pos_of(i, i);
   for (int i = 0; i < mzl; ++i) zpos[i] = pos_of(n0 + my0 + myl + i, n0 + my0 + myl + i);
   k->n_fin = (long long)idx.size();
   if ((rc = dev_upload(&k->d_fin_idx, idx, nullptr))) return rc;
   if ((rc = dev_upload(&k->d_fin_val, val, nullptr))) return rc;
   if ((rc = dev_upload(&k->d_xdiag_pos, xpos, nullptr))) return rc;
   if ((rc = dev_upload(&k->d_zlink_pos, zpos, nullptr))) return rc;
   if ((rc = dev_upload(&k->d_sc_rowptr, k->sc_rowptr, nullptr))) return rc;
   HIP_TRY(hipMalloc((void**)&k->d_t, std::max<size_t>((size_t)e->n_total, 1) * sizeof(double)));
   *handle = k.release();
   return PIPS_OK;
}

static int kkt_factorize_sparse(KktSystem* k, const double* leaf_diag_dev, const double* xdiag0_dev, const double* zdiag_link_dev) {
   Engine* e = k->leaves;
   Engine* r = k->root_sp.get();
   int rc;
   if (leaf_diag_dev && (rc = pips_hip_batch_set_diagonals_dev(e, leaf_diag_dev))) return rc;
   const size_t nnz = (size_t)k->sc_rowptr[k->S];
   PhaseTimer& tm = k->timer;
   tm.on = e->timer.on;
   tm.reset();
   if ((rc = k->root_wait())) return rc;                                         // the previous root factorisation still reads the value array
   HIP_TRY(hipMemsetAsync(r->d_kval, 0, nnz * sizeof(double), e->stream));
   tm.begin(e->stream, 1);
   e->defer_group_reduce = (k->n_ranks > 1 || k->force_reduce) && e->deterministic && e->det_global;
   rc = e->factor(r->d_kval, 0);
   e->defer_group_reduce = false;
   if (rc) return rc;
   tm.end(e->stream);
   const bool reduce = k->n_ranks > 1 || k->force_reduce;
   tm.begin(e->stream, 2);
   if (reduce && e->deterministic && e->det_global && e->d_gbuf) {
      // deterministic mode over several ranks: all eight group slots of the value array on every rank, one fixed tree (as in pips_hip_kkt_factorize)
      if (!k->comm) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: n_ranks > 1 needs a communicator");
      if (!k->d_gall) HIP_TRY(hipMalloc((void**)&k->d_gall, 8 * nnz * sizeof(double)));
      HIP_TRY(hipMemsetAsync(k->d_gall, 0, 8 * nnz * sizeof(double), e->stream));
      HIP_TRY(hipMemcpyAsync(k->d_gall + (size_t)e->det_first_slot * nnz, e->d_gbuf, (size_t)e->det_n_groups * nnz * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      if ((rc = pips_hip_all_gather(k->comm, k->d_gall, (size_t)e->det_slots * nnz, 8 / e->det_slots, e->stream))) return rc;   // every rank's slots to every rank: 1 x the bytes
      hipLaunchKernelGGL(k_reduce_groups, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(1024, (nnz + 255) / 256)), 1), dim3(256), 0, e->stream, r->d_kval, 0, (int)nnz,
                         k->d_gall, (long long)nnz, 8, 0);
   } else if (reduce) {
      if (!k->comm) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: n_ranks > 1 needs a communicator");
      if ((rc = pips_hip_allreduce_sum(k->comm, r->d_kval, nnz, e->stream))) return rc;
   }
   tm.end(e->stream);
   if (xdiag0_dev && k->n0 > 0)
      hipLaunchKernelGGL(k_add_at, dim3(grid_for(k->n0, 256)), dim3(256), 0, e->stream, r->d_kval, k->d_xdiag_pos, xdiag0_dev, k->n0);
   if (k->n_fin > 0)
      hipLaunchKernelGGL(k_add_entries, dim3(grid_for(k->n_fin, 256)), dim3(256), 0, e->stream, r->d_kval, k->d_fin_idx, k->d_fin_val,
                         k->n_fin);
   if (k->mz0 > 0) {
      if (!k->d_zdiag0) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: mz0 > 0 needs pips_hip_kkt_set_root_inequalities + a zdiag0 vector");
      hipLaunchKernelGGL(k_ctdc, e->deterministic ? dim3(1) : dim3(grid_for(k->mz0, 128)), e->deterministic ? dim3(1) : dim3(128), 0, e->stream, k->mz0, k->d_c0_rp,
                         k->d_c0_ci, k->d_c0_val, k->d_zdiag0, r->d_kval, 0, k->d_sc_rowptr);
   }
   if (zdiag_link_dev && k->mzl > 0)
      hipLaunchKernelGGL(k_add_at, dim3(grid_for(k->mzl, 256)), dim3(256), 0, e->stream, r->d_kval, k->d_zlink_pos, zdiag_link_dev, k->mzl);
   if (k->root_reg_primal != 0.0 && k->n0 > 0)
      hipLaunchKernelGGL(k_add_const_diag, dim3(grid_for(k->n0, 256)), dim3(256), 0, e->stream, r->d_kval, 0, k->d_sc_rowptr, 0, k->n0, k->root_reg_primal);
   if (k->root_reg_dual != 0.0 && k->S > k->n0)
      hipLaunchKernelGGL(k_add_const_diag, dim3(grid_for(k->S - k->n0, 256)), dim3(256), 0, e->stream, r->d_kval, 0, k->d_sc_rowptr, k->n0,
                         k->S - k->n0, -k->root_reg_dual);
   HIP_TRY(hipGetLastError());
   // The root engine's factorisation is a chain of small launches (the dissected root: 26 levels of fronts + the hubs' tile): on a stream
   // of its own it runs beside the leaf sweeps of the next solveCompressed's Lsolve, as the dense root does (root_wait() joins before
   // Dsolve, the next factorisation, queries): 39.9 -> 38.9 ms per unit on the configs[3] shape, 43.5 -> 42.7 on the 256-block chain
   // (tools/ab_async_root.sh, alternating on one box).  Default since round 5 (PIPS_HIP_SPARSE_ROOT_ASYNC=0 / PIPS_HIP_ROOT_SYNC keep the
   // main stream): round 4 had one bench run of about two dozen with it not finish inside its time limit and made it opt-in; 148 full-size
   // runs and 60 small ones in round 5 (tools/stress_exit.sh, tools/stress_async.sh, every run under a watchdog) all ended, and the
   // mechanism is the dense root's, which has been the default since round 2.
   static const bool root_async_env = env_int("PIPS_HIP_SPARSE_ROOT_ASYNC", 1) != 0 && !getenv("PIPS_HIP_ROOT_SYNC");
   const bool root_async = root_async_env && k->root_own_stream;
   if (!root_async) {
      const int rec_main = tm.begin_i(e->stream, 13);     // (phase 13 = the root factorisation where it sits on the main stream: critical path)
      tm.begin(e->stream, 4);
      rc = r->factor(nullptr, 0);
      tm.end(e->stream);
      tm.end_i(rec_main, e->stream);
      return rc;
   }
   if (!k->root_stream) {
      int prio_lo = 0, prio_hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      HIP_TRY(hipStreamCreateWithPriority(&k->root_stream, hipStreamNonBlocking, prio_hi));
      HIP_TRY(hipEventCreateWithFlags(&k->ev_sc_final, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&k->ev_root_done, hipEventDisableTiming));
   }
   HIP_TRY(hipEventRecord(k->ev_sc_final, e->stream));
   HIP_TRY(hipStreamWaitEvent(k->root_stream, k->ev_sc_final, 0));
   r->stream = k->root_stream;
   tm.begin(k->root_stream, 4);
   rc = r->factor(nullptr, 0);
   tm.end(k->root_stream);
   r->stream = e->stream;                                                         // solves and queries run on the main stream
   if (rc) return rc;
   HIP_TRY(hipEventRecord(k->ev_root_done, k->root_stream));
   k->root_pending = true;
   return PIPS_OK;
}

int pips_hip_kkt_factorize(void* handle, const double* leaf_diag_dev, const double* xdiag0_dev, const double* zdiag_link_dev) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   Engine* e = k->leaves;
   HIP_TRY(hipSetDevice(e->device));
   ++k->factor_gen;
   k->solves_since_factor = 0;
   if (k->sparse) return kkt_factorize_sparse(k, leaf_diag_dev, xdiag0_dev, zdiag_link_dev);
   int rc;
   PhaseTimer& tm = k->timer;
   tm.on = e->timer.on;
   tm.reset();
   tm.begin(e->stream, 0);
   if (leaf_diag_dev && (rc = pips_hip_batch_set_diagonals_dev(e, leaf_diag_dev))) return rc;
   const size_t n = (size_t)k->S * k->S;
   if ((rc = k->root_wait())) return rc;                                         // the previous root factorisation still reads d_SC
   HIP_TRY(hipMemsetAsync(k->d_SC, 0, n * sizeof(double), e->stream));            // initializeKKT (:840-847)
   tm.end(e->stream);
   tm.begin(e->stream, 1);
   e->defer_group_reduce = (k->n_ranks > 1 || k->force_reduce) && e->deterministic && e->det_global;
   rc = e->factor(k->d_SC, k->S);                                                // children factor2 + assembleLocalKKT
   e->defer_group_reduce = false;
   if (rc) return rc;
   tm.end(e->stream);
   // reduceKKT (:860-881).  PIPS_HIP_FORCE_REDUCE exercises the reduction path with a one-rank communicator (tests).
   const bool reduce = k->n_ranks > 1 || k->force_reduce;
   const int rec_reduce = tm.begin_i(e->stream, 2);   // what the main stream waits for the reduction: its exposed part
   if (reduce && e->deterministic && e->det_global && e->d_gbuf) {
      // deterministic mode over several ranks: every rank's group buffers to every rank, then ALL eight slots in the one fixed tree (the
      // leaf engine left its groups unreduced, Engine::det_global) - equal bits for 1, 2, 4 and 8 ranks
      if (!k->comm) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: n_ranks > 1 needs a communicator");
      const size_t gs = (size_t)k->S * k->S;
      if (!k->d_gall) HIP_TRY(hipMalloc((void**)&k->d_gall, 8 * gs * sizeof(double)));
      HIP_TRY(hipMemsetAsync(k->d_gall, 0, 8 * gs * sizeof(double), e->stream));
      HIP_TRY(hipMemcpyAsync(k->d_gall + (size_t)e->det_first_slot * gs, e->d_gbuf, (size_t)e->det_n_groups * gs * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      if ((rc = pips_hip_all_gather(k->comm, k->d_gall, (size_t)e->det_slots * gs, 8 / e->det_slots, e->stream))) return rc;   // (the all-reduce of zeros it replaces moved 8 x the bytes)
      hipLaunchKernelGGL(k_reduce_groups, dim3(std::max(1, std::min(64, (k->S + 255) / 256)), k->S), dim3(256), 0, e->stream, k->d_SC, k->S, k->S, k->d_gall,
                         (long long)gs, 8, 0);
   } else if (reduce) {
      if (!k->comm) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: n_ranks > 1 needs a communicator");
      // only the lower triangle is authoritative: reduce S(S+1)/2 packed doubles instead of S^2
      const size_t np = (size_t)k->S * (k->S + 1) / 2;
      const size_t n_groups = std::max<size_t>(e->sc_groups.size(), 1);
      const size_t P = (size_t)std::max(1, pips_hip_comm_size(k->comm));
      const size_t cap = np + (n_groups + 1) * P;   // reduce-scatter pads every piece to a multiple of the rank count
      if (!k->d_packed || k->packed_cap < cap) {
         if (k->d_packed) (void)hipFree(k->d_packed);
         HIP_TRY(hipMalloc((void**)&k->d_packed, cap * sizeof(double)));
         k->packed_cap = cap;
      }
      auto reduce_piece = [&](double* buf, size_t cnt, hipStream_t st) -> int {
         return k->use_rsag ? pips_hip_allreduce_sum_rsag(k->comm, buf, cnt, st) : pips_hip_allreduce_sum(k->comm, buf, cnt, st);
      };
      if (!e->sc_groups.empty()) {
         // Panel-wise: the Schur SYRK ran in row-panel groups (Engine::set_sc_panels); the rows of panel q are final on this rank
         // once group q has run, so their reduction goes out on a second stream while the leaves compute the later groups -
         // the overlap of leaf work with MPI_Allreduce that DistributedRootLinearSystem.C:860-881 cannot have (it reduces
         // after all children are done).  Everything was enqueued by e->factor(); here only the reductions are issued, in order.
         if (!k->comm_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&k->comm_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&k->ev_reduced, hipEventDisableTiming));
         }
         size_t off = 0;
         for (size_t q = 0; q < e->sc_groups.size(); ++q) {
            const int R0 = e->sc_row_begin[q], R1 = e->sc_row_begin[q + 1];
            if (R1 <= R0) continue;
            const size_t h = (size_t)(R1 - R0);
            const size_t cnt = (size_t)R0 * h + h * (h + 1) / 2;
            HIP_TRY(hipStreamWaitEvent(k->comm_stream, e->ev_sc[q], 0));
            const dim3 pg(std::max(1, std::min(64, (R1 - R0 + 255) / 256)), R1);
            const int rec_panel = tm.begin_i(k->comm_stream, 10);   // pack + collective + unpack of this panel, beside the leaf work
            hipLaunchKernelGGL(k_pack_rows, pg, dim3(256), 0, k->comm_stream, k->d_SC, k->S, R0, R1, k->d_packed + off, 0);
            if ((rc = reduce_piece(k->d_packed + off, cnt, k->comm_stream))) return rc;
            hipLaunchKernelGGL(k_pack_rows, pg, dim3(256), 0, k->comm_stream, k->d_SC, k->S, R0, R1, k->d_packed + off, 1);
            tm.end_i(rec_panel, k->comm_stream);
            off += (cnt + P - 1) / P * P;
         }
         HIP_TRY(hipEventRecord(k->ev_reduced, k->comm_stream));
         HIP_TRY(hipStreamWaitEvent(e->stream, k->ev_reduced, 0));
      } else {
         const dim3 pg(std::max(1, std::min(64, (k->S + 255) / 256)), k->S);
         hipLaunchKernelGGL(k_pack_lower, pg, dim3(256), 0, e->stream, k->d_SC, k->S, k->S, k->d_packed, 0);
         if ((rc = reduce_piece(k->d_packed, np, e->stream))) return rc;
         hipLaunchKernelGGL(k_pack_lower, pg, dim3(256), 0, e->stream, k->d_SC, k->S, k->S, k->d_packed, 1);
      }
   }
   tm.end_i(rec_reduce, e->stream);
   // finalizeKKTdense
   tm.begin(e->stream, 3);
   if (xdiag0_dev && k->n0 > 0)
      hipLaunchKernelGGL(k_add_diag, dim3(grid_for(k->n0, 256)), dim3(256), 0, e->stream, k->d_SC, k->S, 0, xdiag0_dev, k->n0);
   if (k->n_fin > 0)
      hipLaunchKernelGGL(k_add_entries, dim3(grid_for(k->n_fin, 256)), dim3(256), 0, e->stream, k->d_SC, k->d_fin_idx,
                         k->d_fin_val, k->n_fin);
   if (k->mz0 > 0) {
      if (!k->d_zdiag0) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_factorize: mz0 > 0 needs pips_hip_kkt_set_root_inequalities + a zdiag0 vector");
      // deterministic mode: one thread walks the rows of C0 (the kernel's atomics then arrive in row order)
      hipLaunchKernelGGL(k_ctdc, e->deterministic ? dim3(1) : dim3(grid_for(k->mz0, 128)), e->deterministic ? dim3(1) : dim3(128), 0, e->stream, k->mz0,
                         k->d_c0_rp, k->d_c0_ci, k->d_c0_val, k->d_zdiag0, k->d_SC, k->S, (const int*)nullptr);
   }
   if (zdiag_link_dev && k->mzl > 0)
      hipLaunchKernelGGL(k_add_diag, dim3(grid_for(k->mzl, 256)), dim3(256), 0, e->stream, k->d_SC, k->S,
                         k->n0 + k->my0 + k->myl, zdiag_link_dev, k->mzl);
   if (k->root_reg_primal != 0.0 && k->n0 > 0)
      hipLaunchKernelGGL(k_add_const_diag, dim3(grid_for(k->n0, 256)), dim3(256), 0, e->stream, k->d_SC, k->S, (const int*)nullptr, 0, k->n0,
                         k->root_reg_primal);
   if (k->root_reg_dual != 0.0 && k->S > k->n0)
      hipLaunchKernelGGL(k_add_const_diag, dim3(grid_for(k->S - k->n0, 256)), dim3(256), 0, e->stream, k->d_SC, k->S, (const int*)nullptr, k->n0,
                         k->S - k->n0, -k->root_reg_dual);
   HIP_TRY(hipGetLastError());
   tm.end(e->stream);
   static const bool root_async_env = !getenv("PIPS_HIP_ROOT_SYNC");
   const bool root_async = root_async_env && k->root_own_stream && k->root->dist_P <= 1;   // the distributed root issues collectives: main stream
   if (!root_async) {
      const int rec_main = tm.begin_i(e->stream, 13);
      tm.begin(e->stream, 4);
      rc = k->root->factor_dev(k->d_SC, k->S, 0);                                  // factorizeKKT (:1436-1464)
      tm.end(e->stream);
      tm.end_i(rec_main, e->stream);
      return rc;
   }
   if (!k->root_stream) {
      int prio_lo = 0, prio_hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      HIP_TRY(hipStreamCreateWithPriority(&k->root_stream, hipStreamNonBlocking, prio_hi));
      HIP_TRY(hipEventCreateWithFlags(&k->ev_sc_final, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&k->ev_root_done, hipEventDisableTiming));
   }
   HIP_TRY(hipEventRecord(k->ev_sc_final, e->stream));
   HIP_TRY(hipStreamWaitEvent(k->root_stream, k->ev_sc_final, 0));
   k->root->stream = k->root_stream;
   tm.begin(k->root_stream, 4);
   rc = k->root->factor_dev(k->d_SC, k->S, 0);
   tm.end(k->root_stream);
   k->root->stream = e->stream;                                                   // solves and queries run on the main stream
   if (rc) return rc;
   HIP_TRY(hipEventRecord(k->ev_root_done, k->root_stream));
   k->root_pending = true;
   return PIPS_OK;
}

int pips_hip_kkt_set_root_regularization(void* handle, double primal, double dual) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || primal < 0.0 || dual < 0.0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_set_root_regularization: bad arguments");
   k->root_reg_primal = primal;
   k->root_reg_dual = dual;
   return PIPS_OK;
}

// the launch sequence of one solveCompressed; `capturing`: inside a stream capture (no host-side decisions, no waits on events recorded outside)
static int kkt_solve_compressed_enqueue(KktSystem* k, double* b0_dev, double* b_leaf_dev, bool capturing) {
   Engine* e = k->leaves;
   int rc;
   bool use_aug = false, verify = false;
   int lsolve_steps = 0;
   // with mz0 > 0 the caller's vector is [x0 | y0 | z0 | ylink | zlink]; the Schur system lives on the reduced vector
   // [x0 | y0 | ylink | zlink] (solveReducedLinkCons, sLinsysRootAug.C:397-433)
   double* red = b0_dev;
   const int head = k->n0 + k->my0, tailn = k->myl + k->mzl;
   if (k->mz0 > 0) {
      red = k->d_red;
      HIP_TRY(hipMemcpyAsync(red, b0_dev, (size_t)head * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(red + head, b0_dev + head + k->mz0, (size_t)tailn * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
   }
   // several ranks (or the forced reduction of the tests): the checks are decided together - see KktSystem::solve_check_every
   const bool joint = k->n_ranks > 1 || k->force_reduce;
   const bool can_measure = !capturing && e->refine_tol > 0.0 && e->refine_steps > 0;
   // (whether the ranks exchange the outcome may depend only on what is equal on every rank: the settings the host gives all ranks alike,
   // and "some rank's analysis chose the sweeps" - the cost model decides per rank - settled once per analysis by an all-reduce)
   if (joint && can_measure && k->solve_check_every > 0 && k->joint_aug_gen != e->analysis_gen) {
      if (!k->d_flag) HIP_TRY(hipMalloc((void**)&k->d_flag, sizeof(double)));
      double any = e->aug_sweeps_ok ? 1.0 : 0.0;
      HIP_TRY(hipMemcpyAsync(k->d_flag, &any, sizeof(double), hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
      if ((rc = pips_hip_allreduce_sum(k->comm, k->d_flag, 1, e->stream))) return rc;
      HIP_TRY(hipMemcpyAsync(&any, k->d_flag, sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
      k->joint_aug_any = any > 0.0;
      k->joint_aug_gen = e->analysis_gen;
   }
   const bool joint_check = joint && can_measure && k->solve_check_every > 0 && k->joint_aug_any;
   // Which calls measure is decided by a counter that is equal on every rank (solveCompressed calls since the factorisation; the first one
   // is always scheduled): several ranks then exchange the outcome only on scheduled calls - none could have measured on the others -
   // instead of ending every call with a latency-bound collective and two host waits.
   const bool scheduled = k->solve_check_every > 0 && (k->solves_since_factor++ % k->solve_check_every) == 0;
   if (can_measure && e->aug_sweeps_ok && k->aug_failed_gen != k->factor_gen) {
      const bool validated = k->aug_validated_gen == k->factor_gen;
      const bool may_check = !joint || joint_check;      // (a measure may fail: several ranks must be able to act on it together)
      if (validated || (k->checked_witness && may_check && (!joint || scheduled))) {   // the first solve after a factorisation: a checked sweep pair, or the refined pass
         int pert = 1;
         if ((rc = e->perturbed_leaf_pivots(&pert))) return rc;
         use_aug = pert == 0;
         const bool due = use_aug && validated && may_check && scheduled;
         verify = use_aug && (!validated || due);
      }
   }
   if (verify || joint_check) {   // the right-hand side as the caller gave it: needed for the check, and for the refined pass if a check fails
      if (!k->d_bsave) HIP_TRY(hipMalloc((void**)&k->d_bsave, std::max<size_t>((size_t)e->n_total, 1) * sizeof(double)));
      if (!k->d_b0save) HIP_TRY(hipMalloc((void**)&k->d_b0save, (size_t)(k->S + k->mz0 + 1) * sizeof(double)));
      HIP_TRY(hipMemcpyAsync(k->d_bsave, b_leaf_dev, (size_t)e->n_total * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(k->d_b0save, b0_dev, (size_t)(k->S + k->mz0) * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
   }
   // the joint decision at the end of the call: any rank's failed check sends every rank back to its saved right-hand side
   auto settle = [&](bool my_check_failed) -> int {
      bool redo = my_check_failed;
      if (joint_check) {
         if (!k->d_flag) HIP_TRY(hipMalloc((void**)&k->d_flag, sizeof(double)));
         if (!k->h_flag) HIP_TRY(hipHostMalloc((void**)&k->h_flag, 2 * sizeof(double), hipHostMallocDefault));
         k->h_flag[0] = my_check_failed ? 1.0 : 0.0;   // (pinned: the copy is queued, nothing waits before the collective)
         HIP_TRY(hipMemcpyAsync(k->d_flag, k->h_flag, sizeof(double), hipMemcpyHostToDevice, e->stream));
         int rcf = pips_hip_allreduce_sum(k->comm, k->d_flag, 1, e->stream);
         if (rcf) return rcf;
         HIP_TRY(hipMemcpyAsync(k->h_flag + 1, k->d_flag, sizeof(double), hipMemcpyDeviceToHost, e->stream));
         HIP_TRY(hipStreamSynchronize(e->stream));
         redo = k->h_flag[1] > 0.0;
      }
      if (!redo) return PIPS_OK;
      ++k->failed_checks;
      k->aug_failed_gen = k->factor_gen;   // no sweeps on these factors any more, on any rank
      HIP_TRY(hipMemcpyAsync(b_leaf_dev, k->d_bsave, (size_t)e->n_total * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(b0_dev, k->d_b0save, (size_t)(k->S + k->mz0) * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      return kkt_solve_compressed_enqueue(k, b0_dev, b_leaf_dev, capturing);
   };
   if (e->deterministic && e->d_gvec) {
      // deterministic Lsolve: t = -sum_i Br_i^T K_i^-1 b_i is formed on its own - group-wise in block order, the (at most eight)
      // groups in the fixed tree of k_reduce_groups, the ranks' parts by the all-reduce - and added to b0 on every rank.  Guarantee:
      // run-to-run reproducibility for any rank count, and equal bits for 1 and 2 ranks (a two-operand all-reduce has one order);
      // with 4 or 8 ranks the association of the per-rank partial sums is the all-reduce's (ring / tree, per chunk), not this tree
      k->timer.begin(e->stream, 5);
      if (use_aug) { if ((rc = e->forward_augmented_det(b_leaf_dev))) return rc; }   // (the blocks' border slots hold -L_b y = -Br^T K^-1 b)
      else {
         if ((rc = e->solve(b_leaf_dev))) return rc;
         lsolve_steps = e->last_refine_steps;
      }
      k->timer.end(e->stream);
      k->timer.begin(e->stream, 6);
      HIP_TRY(hipMemsetAsync(e->d_gvec, 0, (size_t)8 * k->S * sizeof(double), e->stream));
      HIP_TRY(hipMemsetAsync(e->d_tvec, 0, (size_t)k->S * sizeof(double), e->stream));
      if (use_aug) e->gather(e->g_bslot_grp, e->d_xw, e->d_gvec);
      else if (e->bt_rows_total > 0) {
         hipLaunchKernelGGL(k_border_rowdot, dim3(grid_for(e->bt_rows_total, 256)), dim3(256), 0, e->stream, e->d_bt_rowptr, e->d_bt_colidx, e->d_bval,
                            e->d_bt_xoff, b_leaf_dev, e->d_bt_tmp, e->bt_rows_total, -1.0);
         e->gather(e->g_btm_grp, e->d_bt_tmp, e->d_gvec);
      }
      if (e->det_global && k->n_ranks > 1) {   // all eight group slots on every rank, one tree (see pips_hip_kkt_factorize)
         if (!k->d_gvec_all) HIP_TRY(hipMalloc((void**)&k->d_gvec_all, (size_t)8 * k->S * sizeof(double)));
         HIP_TRY(hipMemsetAsync(k->d_gvec_all, 0, (size_t)8 * k->S * sizeof(double), e->stream));
         HIP_TRY(hipMemcpyAsync(k->d_gvec_all + (size_t)e->det_first_slot * k->S, e->d_gvec, (size_t)e->det_n_groups * k->S * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
         if ((rc = pips_hip_all_gather(k->comm, k->d_gvec_all, (size_t)e->det_slots * k->S, 8 / e->det_slots, e->stream))) return rc;
         hipLaunchKernelGGL(k_reduce_groups, dim3(std::max(1, std::min(64, (k->S + 255) / 256)), 1), dim3(256), 0, e->stream, e->d_tvec, k->S, k->S, k->d_gvec_all,
                            (long long)k->S, 8, 0);
      } else {
      hipLaunchKernelGGL(k_reduce_groups, dim3(std::max(1, std::min(64, (k->S + 255) / 256)), 1), dim3(256), 0, e->stream, e->d_tvec, k->S, k->S, e->d_gvec,
                         (long long)k->S, e->det_n_groups, e->det_first_slot);
      if ((k->n_ranks > 1 || k->force_reduce) && (rc = pips_hip_allreduce_sum(k->comm, e->d_tvec, (size_t)k->S, e->stream))) return rc;
      }
      hipLaunchKernelGGL(k_axpy, dim3(grid_for(k->S, 256)), dim3(256), 0, e->stream, red, e->d_tvec, 1.0, (long long)k->S);
      k->timer.end(e->stream);
   } else {
   // Lsolve: ranks > 0 zero b0, every child adds -Br^T K^-1 b_i, all-reduce (sLinsysRootAug.C:323-344)
   if (k->n_ranks > 1 && k->rank > 0) HIP_TRY(hipMemsetAsync(red, 0, (size_t)k->S * sizeof(double), e->stream));
   k->timer.begin(e->stream, 5);
   if (use_aug) { if ((rc = e->forward_augmented(b_leaf_dev, red))) return rc; }
   else {
      if ((rc = e->solve(b_leaf_dev))) return rc;
      lsolve_steps = e->last_refine_steps;
   }
   k->timer.end(e->stream);
   k->timer.begin(e->stream, 6);
   if (!use_aug && (rc = pips_hip_batch_border_tmult_dev(e, b_leaf_dev, red, -1.0))) return rc;
   if ((k->n_ranks > 1 || k->force_reduce) && (rc = pips_hip_allreduce_sum(k->comm, red, (size_t)k->S, e->stream)))
      return rc;
   k->timer.end(e->stream);
   }
   // Dsolve: eliminate z0 through C0, solve with the Schur complement, recover z0 (solveReducedLinkCons :384-466)
   // the join with the root's stream is a phase of its own (11): what the main stream waits there is the part of the root factorisation
   // that the first Lsolve did not hide - the exposed root time, measured instead of estimated
   if (!capturing) {   // (a captured sequence: joined before the capture began)
      const bool pending = k->root_pending;
      if (pending) k->timer.begin(e->stream, 11);
      if ((rc = k->root_wait())) return rc;
      if (pending) k->timer.end(e->stream);
   }
   k->timer.begin(e->stream, 7);
   if (k->mz0 > 0)
      hipLaunchKernelGGL(k_z0_elim, e->deterministic ? dim3(1) : dim3(grid_for(k->mz0, 128)), e->deterministic ? dim3(1) : dim3(128), 0, e->stream, 0, k->mz0, k->d_c0_rp, k->d_c0_ci, k->d_c0_val,
                         k->d_zdiag0, b0_dev + head, red);
   if (k->sparse) {
      if ((rc = k->root_sp->solve(red))) return rc;
   } else {
      if ((rc = k->root->solve_dev(red))) return rc;
   }
   if (k->mz0 > 0) {
      hipLaunchKernelGGL(k_z0_elim, dim3(grid_for(k->mz0, 128)), dim3(128), 0, e->stream, 1, k->mz0, k->d_c0_rp, k->d_c0_ci, k->d_c0_val,
                         k->d_zdiag0, b0_dev + head, red);
      HIP_TRY(hipMemcpyAsync(b0_dev, red, (size_t)head * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(b0_dev + head + k->mz0, red + head, (size_t)tailn * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
   }
   // Ltsolve: b_i -= K_i^-1 Br_i x0 (LniTransMult, DistributedLinearSystem.C:430-483).  Where the stored border rows are thin
   // enough and no pivot of the factorisation was perturbed: from the augmented factor with one backward sweep
   // (Engine::solve_border_backward); else border product + full solve with refinement.
   k->timer.end(e->stream);
   k->timer.begin(e->stream, 8);
   // The sweep carries no refinement, so it is taken only on evidence that the factors are accurate: no perturbed pivot (the
   // counters reached pinned memory with the factorisation: no wait for the solves queued behind it) AND, with adaptive refinement,
   // the refined leaf solve of this call's Lsolve - same factors - was satisfied by its first solve (backward error below the
   // tolerance without a step).  A pivot that kept its sign but is rounding noise passes the first test, not the second.
   if (use_aug) {
      if ((rc = e->backward_augmented(red, b_leaf_dev))) return rc;
      k->last_ltsolve_from_factor = true;
      k->last_solve_path = 2;
      bool failed = false;
      if (verify) {
         k->timer.end(e->stream);
         k->timer.begin(e->stream, 12);   // (phase 12: the measure of the sweeps' result)
         // r_i = (b_i - Br_i x0) - K_i x_i over the blocks, measured like a refinement step would measure it
         double worst = 0.0;
         if (e->can_measure_fused()) {
            if ((rc = e->residual_measure_fused(k->d_bsave, red, b_leaf_dev, &worst))) return rc;
         } else {
            HIP_TRY(hipMemcpyAsync(k->d_t, k->d_bsave, (size_t)e->n_total * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            if ((rc = pips_hip_batch_border_mult_dev(e, red, k->d_t, -1.0))) return rc;
            if ((rc = e->residual_measure(k->d_t, b_leaf_dev, &worst))) return rc;
         }
         ++k->checked_solves;
         k->sweeps_since_check = 0;
         if (worst <= e->refine_tol) {
            k->aug_validated_gen = k->factor_gen;
            k->last_solve_path = 3;
         } else
            failed = true;   // not good enough without refinement: the refined path on the saved right-hand side, no sweeps on these factors
      }
      k->timer.end(e->stream);
      if (failed || (joint_check && scheduled)) return settle(failed);
      HIP_TRY(hipGetLastError());
      return PIPS_OK;
   } else {
   if (!capturing) {
      int pert = 1;
      if ((e->border_backward_ok || e->aug_sweeps_ok) && (rc = e->perturbed_leaf_pivots(&pert))) return rc;
      const bool lsolve_clean = e->refine_tol > 0.0 ? e->last_refine_steps == 0 : true;
      k->last_ltsolve_from_factor = e->border_backward_ok && pert == 0 && lsolve_clean;   // (dense or sparse root: x0 comes in Schur numbering either way)
   }
   int ltsolve_steps = 0;
   if (k->last_ltsolve_from_factor) {
      if ((rc = e->solve_border_backward(red, k->d_t))) return rc;
   } else {
      HIP_TRY(hipMemsetAsync(k->d_t, 0, (size_t)e->n_total * sizeof(double), e->stream));
      if ((rc = pips_hip_batch_border_mult_dev(e, red, k->d_t, 1.0))) return rc;
      if ((rc = e->solve(k->d_t))) return rc;
      ltsolve_steps = e->last_refine_steps;
   }
   k->last_solve_path = k->last_ltsolve_from_factor ? 1 : 0;
   // this refined pass is the witness for the factors it ran on (see KktSystem::aug_validated_gen)
   // (a pass that was allowed no step proves nothing: refine_steps > 0)
   if (can_measure && e->aug_sweeps_ok && lsolve_steps == 0 && ltsolve_steps == 0) k->aug_validated_gen = k->factor_gen;
   k->timer.end(e->stream);
   k->timer.begin(e->stream, 9);
   hipLaunchKernelGGL(k_axpy, dim3(grid_for(e->n_total, 256)), dim3(256), 0, e->stream, b_leaf_dev, k->d_t, -1.0, e->n_total);
   k->timer.end(e->stream);
   if (joint_check && scheduled) return settle(false);   // (another rank's check may have failed)
   }
   HIP_TRY(hipGetLastError());
   return PIPS_OK;
}

// solveCompressed as a replayed HIP graph (pips_hip_kkt_set_solve_graph): the launch sequence of one call is
// fixed between factorisations - dozens of launches on a launch-bound problem (configs[0]: ~50 kernels of a few microseconds each) -
// so it is captured once per (right-hand-side pointers, Ltsolve path) and replayed.  What a capture cannot contain keeps the
// direct path: adaptive refinement (it reads norms on the host between steps), reductions over several ranks, the sparse root,
// deterministic mode, phase timing.  The single-launch sweeps take their epoch from device memory for this (k_sweep_bump).
static bool kkt_graph_eligible(const KktSystem* k) {
   const Engine* e = k->leaves;
   return k->solve_graph && !k->sparse && k->n_ranks <= 1 && !k->force_reduce && e->refine_tol == 0.0 && !e->deterministic && !e->timer.on &&
          !k->timer.on;
}

int pips_hip_kkt_solve_compressed(void* handle, double* b0_dev, double* b_leaf_dev) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || !b0_dev || !b_leaf_dev) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_solve_compressed: bad arguments");
   Engine* e = k->leaves;
   HIP_TRY(hipSetDevice(e->device));
   if (!kkt_graph_eligible(k)) return kkt_solve_compressed_enqueue(k, b0_dev, b_leaf_dev, false);
   int rc;
   // host-side decisions and joins first: they are part of the key, not of the graph
   if ((rc = k->root_wait())) return rc;
   int pert = 1;
   if (e->border_backward_ok && (rc = e->perturbed_leaf_pivots(&pert))) return rc;
   k->last_ltsolve_from_factor = e->border_backward_ok && pert == 0;
   KktSystem::GraphKey key;
   key.b0 = b0_dev; key.bl = b_leaf_dev; key.zdiag0 = k->d_zdiag0; key.c0_val = k->d_c0_val; key.c0_rp = k->d_c0_rp; key.c0_ci = k->d_c0_ci;
   key.from_factor = k->last_ltsolve_from_factor ? 1 : 0; key.refine_steps = e->refine_steps; key.refine_mode = e->refine_mode; key.mz0 = k->mz0;
   key.pivoting = k->root ? k->root->pivoting : 0; key.analysis_gen = e->analysis_gen;
   key.bk_gen = k->root ? k->root->bk_refactorizations : 0;   // (a new pivot order: the solve permutes its right-hand side)
   if (k->graph_exec && !(k->graph_key == key)) {
      (void)hipGraphExecDestroy(k->graph_exec);
      k->graph_exec = nullptr;
   }
   if (!k->graph_exec) {
      // The capture runs on a stream of its own (the handle's stream may be the legacy default stream, which cannot be captured):
      // the engine's and the root's stream members point there for the duration of the enqueue; the graph is then launched into the
      // handle's own stream like any other work.
      if (!k->graph_stream) HIP_TRY(hipStreamCreateWithFlags(&k->graph_stream, hipStreamNonBlocking));
      hipGraph_t g = nullptr;
      hipStream_t keep_e = e->stream, keep_r = k->root->stream;
      e->stream = k->graph_stream; k->root->stream = k->graph_stream;
      hipError_t eb = hipStreamBeginCapture(k->graph_stream, hipStreamCaptureModeRelaxed);
      rc = eb == hipSuccess ? kkt_solve_compressed_enqueue(k, b0_dev, b_leaf_dev, true) : PIPS_OK;
      const hipError_t ec = eb == hipSuccess ? hipStreamEndCapture(k->graph_stream, &g) : eb;
      e->stream = keep_e; k->root->stream = keep_r;
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (ec != hipSuccess || !g) PIPS_FAIL(PIPS_ERR_HIP, "pips_hip_kkt_solve_compressed: stream capture failed: %s", hipGetErrorString(ec));
      const hipError_t ei = hipGraphInstantiate(&k->graph_exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ei != hipSuccess) { k->graph_exec = nullptr; PIPS_FAIL(PIPS_ERR_HIP, "pips_hip_kkt_solve_compressed: hipGraphInstantiate: %s", hipGetErrorString(ei)); }
      k->graph_key = key;
      ++k->graph_captures;
   }
   HIP_TRY(hipGraphLaunch(k->graph_exec, e->stream));
   ++k->graph_replays;
   return PIPS_OK;
}

int pips_hip_kkt_set_root_stream(void* handle, int own_stream) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   int rc = k->root_wait();
   if (rc) return rc;
   k->root_own_stream = own_stream != 0;
   return PIPS_OK;
}

int pips_hip_kkt_set_solve_graph(void* handle, int on) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   k->solve_graph = on != 0;
   if (!on && k->graph_exec) { (void)hipGraphExecDestroy(k->graph_exec); k->graph_exec = nullptr; }
   return PIPS_OK;
}

int pips_hip_kkt_solve_graph_stats(void* handle, int64_t* captures, int64_t* replays) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (captures) *captures = k->graph_captures;
   if (replays) *replays = k->graph_replays;
   return PIPS_OK;
}

int pips_hip_kkt_set_solve_check(void* handle, int every) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || every < 0) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_set_solve_check: bad arguments");
   k->solve_check_every = every;
   k->sweeps_since_check = 0;
   return PIPS_OK;
}

int pips_hip_kkt_solve_check_counts(void* handle, long long* checked, long long* failed) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (checked) *checked = k->checked_solves;
   if (failed) *failed = k->failed_checks;
   return PIPS_OK;
}

int pips_hip_kkt_last_solve_path(void* handle, int* path) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || !path) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_last_solve_path: bad arguments");
   *path = k->last_solve_path;
   return PIPS_OK;
}

int pips_hip_kkt_last_ltsolve_from_factor(void* handle, int* flag) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || !flag) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_last_ltsolve_from_factor: bad arguments");
   *flag = k->last_ltsolve_from_factor ? 1 : 0;
   return PIPS_OK;
}

int pips_hip_kkt_get_timing(void* handle, double* ms, int64_t* cnt, int n) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   HIP_TRY(hipSetDevice(k->leaves->device));
   HIP_TRY(hipStreamSynchronize(k->leaves->stream));
   if (k->root_stream) HIP_TRY(hipStreamSynchronize(k->root_stream));
   k->timer.collect();
   for (int i = 0; i < n && i < PhaseTimer::NPHASE; ++i) {
      if (ms) ms[i] = k->timer.ms[i];
      if (cnt) cnt[i] = k->timer.cnt[i];
   }
   return PIPS_OK;
}

int pips_hip_kkt_set_root_inequalities(void* handle, int mz0, const int* C0_rowptr, const int* C0_colidx, const double* C0_val) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || mz0 < 0 || (mz0 > 0 && (!C0_rowptr || !C0_colidx || !C0_val))) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_set_root_inequalities: bad arguments");
   HIP_TRY(hipSetDevice(k->leaves->device));
   k->mz0 = mz0;
   // -C0^T Omega^-1 C0 in the x0 block (sLinsysRootAug.C:1276-1294): an active row makes it a huge low-rank matrix plus an O(1) rest,
   // the static pivot rule then takes the cancelled pivots for zeros - the reference leaves that to dsytrf, so does the root here
   if (k->root && !k->root_pivoting_set) k->root->pivoting = mz0 > 0 ? 1 : 0;
   if (mz0 == 0) return PIPS_OK;
   std::vector<int> rp(C0_rowptr, C0_rowptr + mz0 + 1), ci(C0_colidx, C0_colidx + C0_rowptr[mz0]);
   std::vector<double> v(C0_val, C0_val + C0_rowptr[mz0]);
   int rc;
   if ((rc = dev_upload(&k->d_c0_rp, rp, nullptr)) || (rc = dev_upload(&k->d_c0_ci, ci, nullptr)) || (rc = dev_upload(&k->d_c0_val, v, nullptr))) return rc;
   HIP_TRY(hipMalloc((void**)&k->d_red, (size_t)std::max(k->S, 1) * sizeof(double)));
   return PIPS_OK;
}

int pips_hip_kkt_set_root_pivoting(void* handle, int mode) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || mode < 0 || mode > 1) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_set_root_pivoting: mode 0 (static order) or 1 (Bunch-Kaufman inside the diagonal tiles)");
   if (k->sparse) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_set_root_pivoting: the sparse root is factorised by the leaf engine (static pivot order)");
   k->root->pivoting = mode;
   k->root_pivoting_set = true;
   return PIPS_OK;
}

int pips_hip_kkt_set_zdiag0_dev(void* handle, const double* zdiag0_dev) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   k->d_zdiag0 = zdiag0_dev;
   return PIPS_OK;
}

int pips_hip_kkt_get_schur(void* handle, double** SC_dev, int* ld) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   if (k->sparse) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_get_schur: sparse-root system, use pips_hip_kkt_get_schur_sparse");
   if (SC_dev) *SC_dev = k->d_SC;
   if (ld) *ld = k->S;
   return PIPS_OK;
}

int pips_hip_kkt_root_inertia(void* handle, int* pos, int* neg, int* zero) {
   KktSystem* k = (KktSystem*)handle;
   if (!k) PIPS_FAIL(PIPS_ERR_ARG, "null handle");
   int rcw = k->root_wait();
   if (rcw) return rcw;
   if (k->sparse) return pips_hip_batch_inertia(k->root_sp.get(), 0, pos, neg, zero);
   return pips_hip_dense_ldl_inertia(k->root.get(), pos, neg, zero);
}

int pips_hip_kkt_get_schur_sparse(void* handle, int* nnz, int* rowptr, int* colidx, double* val_host) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || !k->sparse) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_get_schur_sparse: not a sparse-root system");
   const int n = k->sc_rowptr[k->S];
   if (nnz) *nnz = n;
   if (rowptr) std::copy(k->sc_rowptr.begin(), k->sc_rowptr.end(), rowptr);
   if (colidx) std::copy(k->sc_colidx.begin(), k->sc_colidx.end(), colidx);
   if (val_host) {
      HIP_TRY(hipSetDevice(k->leaves->device));
      HIP_TRY(hipStreamSynchronize(k->leaves->stream));
      HIP_TRY(hipMemcpy(val_host, k->root_sp->d_kval, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
   }
   return PIPS_OK;
}

// sparse root: what[0] = elimination order taken (0 minimum degree, 1 dense-tile band, 2 dissection around x0 / y0), then the entries
// of pips_hip_batch_info for the root's one-block engine (what[1 + i] = info[i])
int pips_hip_kkt_sparse_root_info(void* handle, int64_t* what, int n_what) {
   KktSystem* k = (KktSystem*)handle;
   if (!k || !what || n_what < 1) PIPS_FAIL(PIPS_ERR_ARG, "pips_hip_kkt_sparse_root_info: bad arguments");
   if (!k->sparse || !k->root_sp) PIPS_FAIL(PIPS_ERR_STATE, "pips_hip_kkt_sparse_root_info: not a sparse-root system");
   what[0] = k->root_order_mode;
   return n_what > 1 ? pips_hip_batch_info(k->root_sp.get(), what + 1, n_what - 1) : PIPS_OK;
}

void pips_hip_kkt_destroy(void* handle) { delete (KktSystem*)handle; }

// ---- symbolic probe (CPU only) -------------------------------------------------------------------------------------
int pips_symbolic_probe(int n, int n_primal, const int* krow, const int* jcol, int S, const int* Bt_rowptr,
                        const int* Bt_colidx, int force_n_head, int64_t* what, int n_what, int* perm, int* colcount) {
   if (n <= 0 || !krow || !jcol) PIPS_FAIL(PIPS_ERR_ARG, "pips_symbolic_probe: bad arguments");
   AnalyzeOptions opt;
   apply_tuning(opt);
   opt.force_n_head = force_n_head;
   opt.mf_konly = opt.mf_split_nb_max > 0 && opt.max_sn_width <= 16 && env_int("PIPS_HIP_MF_KONLY", 0) != 0;   // (as Engine::analyze outside deterministic mode)
   CsrPattern K{n, n, krow, jcol};
   CsrPattern B{0, n, nullptr, nullptr};
   if (Bt_rowptr && S > 0) B = CsrPattern{S, n, Bt_rowptr, Bt_colidx};
   std::vector<BlockSym> sym(1);
   int rc = analyze_block(K, B, n_primal, opt, sym[0]);
   if (rc) return rc;
   if (what) sym_info(sym, what, n_what);
   if (what && n_what > 16) {   // multifrontal head: usable, border split taken, doubles of update matrices, doubles of border rows kept beside the panels
      const BlockSym& bs = sym[0];
      what[13] = bs.mf_ok ? 1 : 0;
      what[14] = bs.mf_split ? (bs.mf_konly ? 2 : 1) : 0;
      what[15] = bs.mf_U_total;
      what[16] = 0;
      for (const HeadSupernode& hs : bs.sn) if (hs.ld < hs.w + hs.r) what[16] += (int64_t)hs.w * (hs.r - hs.rb);
   }
   if (const char* dump = getenv("PIPS_HIP_DUMP_SN")) {   // development aid: one line per head supernode
      if (FILE* f = fopen(dump, "w")) {
         const BlockSym& bs = sym[0];
         for (size_t si = 0; si < bs.sn.size(); ++si) {
            const HeadSupernode& s = bs.sn[si];
            const int* rows = bs.rowidx.data() + s.rows;
            int nh = 0, nt = 0;
            for (int a = 0; a < s.r; ++a) { if (rows[a] < bs.n_head) ++nh; else if (rows[a] < bs.n) ++nt; }
            // c0 w r level rows-in-head rows-in-tail border-rows update-segments parent-supernode (index, -1: none in the head)
            fprintf(f, "%d %d %d %d %d %d %d %d %d\n", s.c0, s.w, s.r, s.level, nh, nt, s.r - nh - nt, s.n_useg,
                    si < bs.sn_parent.size() ? bs.sn_parent[si] : -1);
         }
         fclose(f);
         fprintf(stderr, "[pips_hip] multifrontal: ok %d, largest front %d, update matrices %lld doubles, records %zu ints\n", (int)bs.mf_ok, bs.mf_max_front,
                 (long long)bs.mf_U_total, bs.mf_int.size());
         if (bs.mf_konly) {   // gather-form records: pairs per front and their sizes, level by level
            std::vector<long long> lp, lf, lmax, lwork;
            long long pairs = 0, fronts = 0, work = 0, rowsum = 0;
            for (size_t si = 0; si < bs.sn.size(); ++si) {
               if (bs.kb_off[si] < 0) continue;
               const HeadSupernode& s = bs.sn[si];
               const int* R = bs.kb_rec.data() + bs.kb_off[si];
               if ((int)lp.size() <= s.level) { lp.resize(s.level + 1, 0); lf.resize(s.level + 1, 0); lmax.resize(s.level + 1, 0); lwork.resize(s.level + 1, 0); }
               long long wk = 0;
               for (int q = 0; q < R[0]; ++q) {
                  const HeadSupernode& c = bs.sn[R[2 + 2 * q]];
                  const int ns = (R[3 + 2 * q] >> 16) - (R[3 + 2 * q] & 0xffff);
                  wk += (long long)(c.r - c.rb) * ns * c.w;
                  rowsum += c.r - c.rb;
               }
               lp[s.level] += R[0]; ++lf[s.level]; lmax[s.level] = std::max<long long>(lmax[s.level], R[0]); lwork[s.level] += wk;
               pairs += R[0]; ++fronts; work += wk;
            }
            fprintf(stderr, "[pips_hip] gather-form border rows: %lld fronts, %lld pairs (%.1f border rows each), %lld multiply-adds, %zu tail runs\n", fronts, pairs,
                    pairs ? (double)rowsum / pairs : 0.0, work, bs.kb_tail.size() / 2);
            for (size_t l = 0; l < lp.size(); ++l)
               fprintf(stderr, "   level %2zu: %6lld fronts %7lld pairs, most %4lld, %9lld multiply-adds\n", l, lf[l], lp[l], lmax[l], lwork[l]);
         }
      }
   }
   if (perm) std::copy(sym[0].perm.begin(), sym[0].perm.end(), perm);
   if (colcount) std::copy(sym[0].colcount.begin(), sym[0].colcount.end(), colcount);
   return PIPS_OK;
}

// the order the sparse root takes for a chain-like Schur complement (hub_dissected_order: hubs last, the rest dissected) and the
// symbolic analysis under it; PIPS_ERR_STATE when the graph without the hubs has no separators
int pips_symbolic_probe_hubs(int n, int n_primal, const int* krow, const int* jcol, int n_hubs, const int* hubs, int min_size, int64_t* what,
                             int n_what, int* perm, int* colcount) {
   if (n <= 0 || !krow || !jcol || n_hubs < 0 || (n_hubs > 0 && !hubs)) PIPS_FAIL(PIPS_ERR_ARG, "pips_symbolic_probe_hubs: bad arguments");
   std::vector<int> ap(n + 1, 0), ai;
   for (int r = 0; r < n; ++r)
      for (int p = krow[r]; p < krow[r + 1]; ++p) {
         if (jcol[p] < 0 || jcol[p] > r) PIPS_FAIL(PIPS_ERR_ARG, "pips_symbolic_probe_hubs: K must be lower-triangular CSR");
         if (jcol[p] != r) { ++ap[r + 1]; ++ap[jcol[p] + 1]; }
      }
   for (int i = 0; i < n; ++i) ap[i + 1] += ap[i];
   ai.resize(ap[n]);
   {
      std::vector<int> fill(ap.begin(), ap.end() - 1);
      for (int r = 0; r < n; ++r)
         for (int p = krow[r]; p < krow[r + 1]; ++p)
            if (jcol[p] != r) { ai[fill[r]++] = jcol[p]; ai[fill[jcol[p]]++] = r; }
   }
   std::vector<int> hv(hubs, hubs + n_hubs), pv, cc;
   if (!hub_dissected_order(n, ap, ai, hv, min_size, pv, cc)) PIPS_FAIL(PIPS_ERR_STATE, "pips_symbolic_probe_hubs: no separators");
   AnalyzeOptions opt;
   apply_tuning(opt);
   opt.user_perm = pv.data();
   opt.user_colcount = cc.data();
   {  // (the cut pips_hip_kkt_create_sparse takes)
      int cut = 0;
      while (cut < n - n_hubs && cc[cut] <= ROOT_ND_MAX_COLCOUNT) ++cut;
      opt.force_n_head = cut;
   }
   CsrPattern K{n, n, krow, jcol};
   CsrPattern B{0, n, nullptr, nullptr};
   std::vector<BlockSym> sym(1);
   int rc = analyze_block(K, B, n_primal, opt, sym[0]);
   if (rc) return rc;
   if (what) sym_info(sym, what, n_what);
   if (perm) std::copy(sym[0].perm.begin(), sym[0].perm.end(), perm);
   if (colcount) std::copy(sym[0].colcount.begin(), sym[0].colcount.end(), colcount);
   return PIPS_OK;
}

}  // extern "C"
