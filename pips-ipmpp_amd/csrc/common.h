// Shared host-side declarations for the MI355X KKT backend (libpipship.so).
// Everything here is internal; the public surface is include/pips_hip.h.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace pips {

// ---- error handling --------------------------------------------------------------------------
// C-ABI entry points return 0 on success; the adapter maps non-zero onto the reference's abort
// convention (PardisoSolver.C:201-204).  The message of the last failure is kept per thread.
void set_last_error(const std::string& msg);
const char* last_error();

enum : int {
   PIPS_OK = 0,
   PIPS_ERR_ARG = 1,
   PIPS_ERR_STATE = 2,
   PIPS_ERR_HIP = 3,
   PIPS_ERR_NUMERIC = 4,
   PIPS_ERR_NO_DEVICE = 5,
   PIPS_ERR_RCCL = 6,
};

#define PIPS_FAIL(code, ...)                                   \
   do {                                                        \
      char _buf[512];                                          \
      snprintf(_buf, sizeof(_buf), __VA_ARGS__);               \
      ::pips::set_last_error(_buf);                            \
      return (code);                                           \
   } while (0)

// ---- ordering (order.cpp) --------------------------------------------------------------------
// Constrained approximate-minimum-degree ordering of a symmetric pattern.
//  n            dimension
//  ap/ai        full symmetric adjacency (both triangles, no diagonal), CSR
//  n_primal     rows [0,n_primal) are primal (expected positive pivots); rows >= n_primal are dual rows
//               (expected negative pivots, possibly structurally zero diagonal).  A dual row becomes
//               eligible only after all its primal neighbours are eliminated, which keeps every leading
//               principal block of [D W^T; W 0] nonsingular (see DESIGN.md "static pivoting").
//               n_primal < 0: unconstrained.
//  perm[k]      = original index eliminated k-th ; colcount[k] = #off-diagonal entries in column k of L
void constrained_amd(int n, const std::vector<int>& ap, const std::vector<int>& ai, int n_primal,
                     std::vector<int>& perm, std::vector<int>& colcount);

// Partial nested dissection (order.cpp): up to max_depth levels of dual-row separators, constrained minimum degree inside
// the leaves, exact column counts.  Returns false (perm untouched or unusable) when the block has no small separators -
// random sparsity - or is too small; the caller then uses constrained_amd on the whole block.
bool dissected_order(int n, const std::vector<int>& ap, const std::vector<int>& ai, int n_primal, int max_depth,
                     std::vector<int>& perm, std::vector<int>& colcount, int min_size = 512);

// Nested dissection around hub vertices (order.cpp): hubs_last are ordered last in the given order, the rest is dissected (segments
// below min_size rows stay whole, minimum degree inside); unconstrained.  false: no separators found.
bool hub_dissected_order(int n, const std::vector<int>& ap, const std::vector<int>& ai, const std::vector<int>& hubs_last, int min_size,
                         std::vector<int>& perm, std::vector<int>& colcount);

// ---- symbolic analysis of one leaf block (symbolic.cpp) --------------------------------------
struct HeadSupernode {
   int c0;          // first column (permuted index)
   int w;           // width
   int r;           // number of rows below the diagonal block
   int level;       // elimination-tree level among head supernodes (0 = leaves)
   int64_t panel;   // offset (doubles) of the ld x w column-major panel inside the block arena
   int64_t rows;    // offset (ints) of the r row indices inside BlockSym::rowidx
   int64_t upd;     // offset (ints) of this supernode's head-to-head update segments inside BlockSym::upd
   int n_useg;      // number of such segments (distinct head supernodes among the below-rows)
   int rb;          // index of the first border row among the below-rows (== r when there is none)
   int ld;          // leading dimension of the stored panel: w + r, or - a front under the border split - w + rb: its border rows live
                    // only in the border-row arena (the solve sweeps then read a compact panel of rows of K)
};

struct BlockSym {
   int n = 0;          // dimension of K_i
   int n_primal = -1;  // leading primal rows (inertia hint)
   int nb = 0;         // number of border columns of this block that are non-empty (compressed)
   int n_head = 0;     // columns [0,n_head) (permuted) are factorised by the sparse head kernels
   int m = 0;          // dense tail dimension = n - n_head
   int m_pad = 0;      // m rounded up to the tile size (identity padding)
   int nb_pad = 0;     // nb rounded up to the tile size (zero padding)
   int ldT = 0;        // leading dimension of the tail panel = m_pad + nb_pad
   int64_t T_off = 0;  // offset of the tail panel in the block arena
   int64_t arena = 0;  // arena size in doubles
   int n_levels = 0;
   std::vector<int> perm, iperm;        // perm[new] = old ; iperm[old] = new
   std::vector<int> bmap;               // compressed border index -> Schur column id
   std::vector<HeadSupernode> sn;       // head supernodes, ascending c0
   std::vector<int> sn_of_col;          // [n_head]
   std::vector<int> rowidx;             // concatenated below-rows of head supernodes
                                        //   value < n : permuted row of K ; value >= n : n + compressed border index
   std::vector<int> upd;                // head-to-head update segments, see symbolic.cpp "update segments"
   std::vector<int64_t> a_dst;          // [nnz(K lower)] arena offset of every CSR entry
   std::vector<char> a_front, b_front;  // 1: the entry lands in the panel of a front (k_front takes it from the value array itself: no scatter)
   std::vector<int64_t> b_dst;          // [nnz(border)]  arena offset of every border entry, -1 if it lands in SC (never)
   std::vector<signed char> psign;      // [n] expected pivot sign in permuted order (+1/-1/0)
   int64_t nnzL = 0;                    // stored entries of L (head panels + dense tail lower triangle)
   double flops_factor = 0;             // head + tail factor flops
   double flops_border = 0;             // border TRSM + Schur SYRK flops
   std::vector<int> colcount;           // AMD column counts (diagnostics)
   std::vector<int> tile_first;         // per tile row of the tail panel (m_pad + nb_pad rows): first tile column inside the
                                        // row envelope of the tail's Schur complement (0 = dense row); fill stays inside it
   // ---- multifrontal head (symbolic.cpp "multifrontal metadata"): every head supernode that is not a simple leaf is a front
   //      (w + r) x (w + r) whose update matrix (r x r, packed lower) goes to its parent front instead of being scattered
   bool mf_ok = false;                  // every front fits the LDS budget (AnalyzeOptions::mf_lds_doubles)
   // mf_konly: per front with border rows the record k_border_rows reads { n_pairs, n_ent, (local id of C, b0 | b1 << 16) ..., (a | k << 16, index of the border value) ... },
   // kb_off[s] its offset (-1: none); kb_tail: (local id of C, q0 | q1 << 16) runs of at most 16 tail rows of supernodes with border rows (k_border_tail)
   std::vector<int> kb_rec, kb_tail;
   std::vector<int64_t> kb_off;
   bool mf_konly = false;               // (with mf_split) the fronts hold the rows of K only: their border rows are formed afterwards in gather form (k_border_rows)
   bool mf_split = false;               // border split (symbolic.cpp): update matrices keep only the columns of K rows, the border x border
                                        // part of the Schur contribution comes from the finished panels (k_border_schur)
   int mf_max_front = 0;                // largest w + r among the fronts
   int64_t mf_U_total = 0;              // doubles of update-matrix storage of this block
   std::vector<int> sn_parent;          // per head supernode: the head supernode that holds its first below-row, else -1
   std::vector<int64_t> mf_U;           // per head supernode: offset of its packed update matrix; simple leaf below a front: offset of
                                        // its 1 + r values inside the block's leaf-value region; -1: neither
   int64_t mf_LV_total = 0;             // doubles of leaf values of this block
   std::vector<int64_t> mf_meta;        // per head supernode: offset of its front record inside mf_int, -1 for simple leaves
   std::vector<int> mf_int;             // front records
   std::vector<int64_t> mf_fix;         // positions inside mf_int that hold LOCAL supernode ids (the engine renumbers them)
};

// Front record of supernode J inside BlockSym::mf_int (offsets relative to the record's start):
//   [0]                    number of child fronts
//   [1]                    number of simple leaves hanging below this front
//   [2]                    bit 0: the front has a parent front (its update matrix / vector is picked up there); bits 1..: number of
//                          entries of K and of the border in the front's panel (listed behind the leaf part)
//   [3]                    length of the leaf part
//   [4]                    number of leaf items
//   [5]                    doubles of leaf values, sum of (1 + r_c)
//   [6]                    sum of the children's r_c
//   [7]                    offset of the leaves' values (d_c, l_c: 1 + r_c doubles per leaf, leaves ascending) inside the block's
//                          leaf-value region: the leaf kernel writes them there, the front reads them as one piece
//   [8 ..)                 child table, 3 ints per child front, children ascending:
//                             { offset of its update matrix minus this front's, r_c | uc_c << 16 (uc_c: update columns it hands over),
//                               offset of its update VECTOR (solves) minus this front's }
//   then                   per child, in the same order, the position of each of its r_c below-rows inside THIS front (0 .. w + r)
//   then                   the leaf part
//   leaf part              colptr[w + r + 1] | items (2 ints each) | leaf table (4 ints per leaf) | position lists
//   leaf table entry       { the leaf's column (permuted index), offset of its values, r_c, offset of its position list inside the leaf part }
//   leaf item              { offset of the leaf's 1 + r_c values << 9 | r_c << 4 | b,  offset of its position list inside the leaf part }
//                          - the leaf's b-th row is this front column; items are sorted by front column, inside a column by leaf
//   position list          position of each of the leaf's r_c rows inside the front
//   behind the leaf part   the front's own entries of K and of the border: count, then (position in the packed panel, index into the
//                          block's K values; border entries as -1 - index into the block's border values) pairs
constexpr int MF_HDR = 8;
constexpr int MF_MAX_FRONT = 512;    // a thread per front row

struct CsrPattern {
   int nrows = 0, ncols = 0;
   const int* rowptr = nullptr;
   const int* colidx = nullptr;
};

struct AnalyzeOptions {
   int tile = 128;            // dense tile size
   int max_sn_width = 32;     // head supernode width cap
   // dual-row nested dissection tried before minimum degree (0 levels = off).  Time-coupled blocks are chains: every level halves
   // them.  Measured on the energy-like family (tools/config3_probe.py, 32 blocks x 50 000): depth 4 / segments >= 512 rows
   // (round 1's setting) 119 levels, factorize 50 ms, solveCompressed 20 ms, nnz(L) 112 M; depth 12 / >= 128 rows 14 levels, 20.7 ms,
   // 9.0 ms, 82 M; deeper brings nothing more.  Blocks without thin separators (config 2) fall through to minimum degree unchanged.
   int nd_depth = 12;
   int nd_min_size = 128;     // a segment with fewer dual rows is not dissected further
   const int* user_perm = nullptr;   // given elimination order (perm[new] = old) instead of minimum degree / dissection
   const int* user_colcount = nullptr;   // ... with its column counts (the head / tail cut is priced with them); nullptr: zeros
   bool constrain_order = true;   // dual rows only after their primal neighbours (leaf KKT blocks); false: plain minimum degree,
                                  // the inertia hint still supplies the expected pivot signs (sparse Schur complement)
   int simple_rmax = 16;      // width-1 tree leaves with at most this many rows are "simple leaves" (one thread each on the device)
   int64_t mf_lds_doubles = 19200;   // LDS budget of one front (150 KB of the 160 KB): the packed front if it fits, else its w panel
                                     // columns (the update matrix then stays in device memory); neither, or more than
                                     // MF_MAX_FRONT rows: the block is not multifrontal
   bool mf_konly = false;       // fronts on the rows of K only where the border split applies (BlockSym::mf_konly)
   int mf_split_nb_max = 176;   // border split of the update matrices where the block has at most this many non-empty border columns
                                // (k_border_schur keeps the packed nb x nb triangle in LDS: 176 -> 122 KB); 0 = never
   double relax_zeros = 0.4;  // supernode amalgamation: admissible share of explicit zeros in a panel (0 = fundamental)
   int min_tail = 256;        // do not open a dense tail smaller than this
   int force_n_head = -1;     // >=0: override the cost model (tests)
   double head_cost = 8.0e-11;  // seconds per scattered update entry (cost model; measured optimum on MI355X, config 2)
   double mfma_rate = 5.0e13;   // sustained dense FP64 flop/s of the tile kernels (cost model)
};

// K: lower-triangular CSR pattern of K_i (n x n).  border: CSR with S rows (Schur column ids) over the n rows of K_i,
// i.e. the pattern of Br_i^T (border_left_transp in DistributedLeafLinearSystem.C:214-252).  May be empty (nrows = 0).
int analyze_block(const CsrPattern& K, const CsrPattern& border, int n_primal, const AnalyzeOptions& opt,
                  BlockSym& out);

// ---- task list of the single-launch dense root (rootplan.cpp; kernel: rootkernel.hip.h) -------------------------------
// Cost model of the list schedule in microseconds (measured on MI355X): a K = 128 step of the update kernel with two
// workgroups per compute unit / with the unit to itself, the fixed cost of an update task (operand latency + C tile round trip), trsm
// and diagonal tile.
struct RootPlanParams {
   int workers = 510;       // workgroup slots for the bulk: 256 compute units x 2, less the chain's unit
   int chain_slots = 2;     // > 0: the chain of the diagonal tiles has a compute unit of its own (a task list of its own); 0: one list
   int boost = 10, boost_width = 3;   // update priority: tiles within boost_width of the diagonal as if their column were boost columns nearer
   int max_depth = 24;              // tile columns per update task at most
   int chain_width = 1;     // tiles this close to the diagonal have their trsm and completing update on the chain's list
   int qmin = 4;            // an update that does not finish its tile waits until it is this many tile columns deep
   int urgent = 1;          // tiles within this distance of the chain's diagonal tile are updated whatever the depth
   double t_step = 31.0, t_step_alone = 22.0, t0 = 8.0, t_trsm = 40.0, t_trsm_alone = 26.0, t_diag = 85.0;
};
// tasks / chain_tasks: (kind 0 UPD / 1 TRSM / 2 DIAG, i, j, k0 | k1 << 16) quadruples in ticket order
int build_root_plan(int ntc, const RootPlanParams& p, std::vector<int>& tasks, std::vector<int>& chain_tasks, double* makespan_us);

}  // namespace pips

// ---- host waits, counted ------------------------------------------------------------------------
// Every place where the host waits for the device inside this library - stream / event / device synchronisation and the blocking copies -
// goes through these wrappers, so that "how often does an IPM iteration stop the host" is a number (pips_hip_host_wait_count; bench.py
// reports it per iteration and per work unit) instead of an estimate.  Only in translation units that include the HIP runtime.
#ifdef HIP_INCLUDE_HIP_HIP_RUNTIME_H
#include <atomic>
namespace pips {
extern std::atomic<long long> g_host_waits;
void note_host_wait(const char* file, int line);   // engine.hip: the total, and a table per call site (pips_hip_host_wait_sites)
inline hipError_t counted_stream_sync(hipStream_t s, const char* f, int l) { note_host_wait(f, l); return (hipStreamSynchronize)(s); }
inline hipError_t counted_device_sync(const char* f, int l) { note_host_wait(f, l); return (hipDeviceSynchronize)(); }
inline hipError_t counted_event_sync(hipEvent_t e, const char* f, int l) { note_host_wait(f, l); return (hipEventSynchronize)(e); }
inline hipError_t counted_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k, const char* f, int l) { note_host_wait(f, l); return (hipMemcpy)(d, s, n, k); }
}
#define hipStreamSynchronize(s) ::pips::counted_stream_sync(s, __FILE__, __LINE__)
#define hipDeviceSynchronize() ::pips::counted_device_sync(__FILE__, __LINE__)
#define hipEventSynchronize(e) ::pips::counted_event_sync(e, __FILE__, __LINE__)
#define hipMemcpy(d, s, n, k) ::pips::counted_memcpy(d, s, n, k, __FILE__, __LINE__)
#endif
