// HIP kernels (gfx950 / CDNA4) of the KKT backend.  Included once by engine.hip.
//
//  head  : supernodal sparse LDL^T for the low-fill leading part of every leaf block.  One workgroup per supernode and
//          elimination-tree level; the w x w pivot block is factorised in LDS, the sub-diagonal panel is solved with
//          coalesced column reads, and the Schur update is scattered into ancestors / the dense tail / the root Schur
//          complement with hardware FP64 atomics.
//  tail  : left-looking tiled dense LDL^T (128 x 128 tiles) on the FP64 matrix cores (v_mfma_f64_16x16x4_f64),
//          operands staged through LDS with register prefetch.  The border rows of the augmented system ride along as
//          extra tile rows, so the same GEMM kernel produces  L21 = Br^T K^-T  and finally  SC -= L21 D L21^T.
//  solve : level-scheduled head substitution + tiled dense TRSV using the pre-inverted diagonal tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pips {

constexpr int TILE = 128;
constexpr int KB = 16;          // k-depth of one LDS stage of the tile GEMM (16; 32 needs 132 KB of LDS: one workgroup per CU)
constexpr int LDS_PAD = 16;     // LDS row padding (doubles): 144*8 B = 1152 B -> half-wave k-groups hit disjoint banks
constexpr int LDSW = TILE + LDS_PAD;

struct SnDesc {
   long long panel;  // global arena offset of the ld x w panel
   long long rows;   // global offset into rowidx
   long long upd;    // global offset into upd (head-to-head update segments)
   int w, r, c0, blk;
   int n_useg, rb;   // number of update segments; index of the first border row among the r below-rows
   int ld, pad_;     // leading dimension of the stored panel: w + r, or w + rb for a front under the border split (its border rows live
                     // only in the border-row arena at bb: Lt[k * rpb + a - rb], rpb = r - rb rounded up to 4)
   long long slot;   // deterministic mode: first contribution slot of the factorisation scatter (r (r + 1) / 2 slots: pair (a, b),
                     // a >= b, has slot + b r - b (b - 1) / 2 + a - b)
   long long vslot;  // ... and of the forward-substitution scatter (r slots)
   long long U;      // multifrontal head: offset of the packed r x r update matrix inside the update arena, -1 if none
   long long mf;     // multifrontal head: offset of the front record inside mfint (common.h "Front record"), -1 for simple leaves
   long long bb;     // border split: offset of the supernode's border rows inside the border-row arena (k_border_schur), -1 if none
};


struct BlkDesc {
   long long arena_off;  // block arena base (doubles)
   long long T;          // global arena offset of the tail panel
   long long sncol_off;  // offset into sn_of_col (values are global supernode ids)
   long long xw_off;     // offset of the permuted work vector (length n_head + m_pad)
   long long x_off;      // offset into flat original-order vectors (sum of n over preceding blocks)
   long long bmap_off;   // offset into bmap
   long long winv_off;   // offset into winv (ntc tiles of TILE*TILE)
   long long dt_off;     // offset into dtail (m_pad)
   long long sctab_off;  // offset of this block's nb x nb position table inside sctab (sparse Schur complement), else 0
   int n, n_head, m, m_pad, nb, nb_pad, ldT, ntc, ntr;
   int mf_split;         // multifrontal head with the border split (BlockSym::mf_split)
   long long U;          // offset of the block's scaled tail copy U = L D (m_pad x m_pad, ld = m_pad) inside the U arena
   double thr_rel, repl_rel;  // pivot threshold / replacement relative to the pivot's reference magnitude pref[k]
   double repl_abs;           // replacement when no reference magnitude exists (structurally zero diagonal)
   long long lv_off;          // multifrontal head: offset of the block's leaf values inside the leaf-value arena
   long long k_off, b_off;    // offsets of the block's K values / border values (Engine::d_kval, d_bval): k_front reads its panel entries there
   long long T_in;            // where the tail panel is ASSEMBLED (scatter, root fronts, border rows of the head) and accumulated: = T when the tail is
                              // factorised in place (launch per step), a scratch region behind the panels when it is one launch (tailkernel.hip.h)
};

struct TileTask { int blk, ti, tj, pad; };

typedef double double4_t __attribute__((ext_vector_type(4)));

// Column pointer and stride of below-row a of a head supernode: rows of K sit in the panel; the border rows of a front under the border
// split only in the border-row arena (which lives behind the panels in the same allocation: sn.bb is an arena offset like sn.panel)
struct BelowRow { const double* p; long long stride; };
__device__ __forceinline__ BelowRow below_row(const double* __restrict__ arena, const SnDesc& sn, int a) {
   if (a < sn.ld - sn.w) return BelowRow{arena + sn.panel + sn.w + a, (long long)sn.ld};
   return BelowRow{arena + sn.bb + (a - sn.rb), (long long)((sn.r - sn.rb + 3) & ~3)};
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
   __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// How a scattered contribution reaches its target.  mode 0: hardware FP64 atomic (order of arrival decides the last bits).
// Deterministic mode: every contribution owns a slot; mode 1 (once, at analyze time) records where slot k goes, mode 2 stores
// the value into its slot, and k_gather_slots adds the slots of every target in a fixed order.
struct ScatterCtx {
   int mode;
   long long* rec;
   double* val;
   const double* base;      // arena (factorisation) / work vector (solve)
   const double* sc_base;   // Schur complement
};
constexpr long long SCATTER_SC_FLAG = 1LL << 62;
__device__ __forceinline__ void scatter_add(const ScatterCtx& sx, double* ptr, bool is_sc, long long slot, double v) {
   if (sx.mode == 0) atomic_add_f64(ptr, v);
   else if (sx.mode == 1) sx.rec[slot] = is_sc ? (SCATTER_SC_FLAG | (long long)(ptr - sx.sc_base)) : (long long)(ptr - sx.base);
   else sx.val[slot] = v;
}
__device__ __forceinline__ long long pair_slot(long long base, int a, int b, int r) {
   return base + (long long)b * r - (long long)b * (b - 1) / 2 + (a - b);
}

// target[tgt[t]] += sum of the slots off[t] .. off[t+1] in list order
__global__ void k_gather_slots(long long n_targets, const long long* __restrict__ tgt, const long long* __restrict__ off,
                               const long long* __restrict__ slots, const double* __restrict__ val, double* __restrict__ target) {
   for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n_targets; t += (long long)gridDim.x * blockDim.x) {
      double s = 0.0;
      for (long long p = off[t]; p < off[t + 1]; ++p) s += val[slots[p]];
      target[tgt[t]] += s;
   }
}


// Where the (la, lb) entry of a block's Schur contribution lives (la >= lb: compressed border ids, ascending like the Schur
// column ids they map to).  Dense root: column-major S x S array, bm = the block's bmap.  Sparse root (sctab != nullptr): the
// value array of the lower-triangular CSR pattern of SC, looked up in the block's nb x nb position table.
__device__ __forceinline__ double* sc_entry(double* SC, int ldSC, const int* __restrict__ bm, const int* __restrict__ sctab,
                                            long long tab_off, int nb, int la, int lb) {
   return sctab ? SC + sctab[tab_off + (long long)la * nb + lb] : SC + bm[la] + (long long)bm[lb] * ldSC;
}

// Static pivoting rule.  Every pivot has a reference magnitude pref (|original diagonal entry|, refreshed for the dense
// tail with the diagonal after the sparse head has been eliminated).  With the expected sign known (quasi-definite KKT
// blocks: all contributions to a pivot carry its own sign until the dual-dual eliminations start) a pivot is accepted
// iff sign*d > thr_rel*pref; otherwise it is replaced by sign*repl_rel*pref (repl_abs when pref == 0) and counted as
// perturbed.  This is scale-invariant, unlike a threshold relative to max|K|: IPM diagonals span 1e-8..1e8.
__device__ __forceinline__ double fix_pivot(double d, int sign, double pref, double thr_rel, double repl_rel,
                                            double repl_abs, bool& perturbed) {
   // branch-free (selects only): the rule sits on the critical path of every pivot
   const double thr = thr_rel * pref;
   const double repl = pref > 0.0 ? repl_rel * pref : repl_abs;
   const double s = sign > 0 ? 1.0 : (sign < 0 ? -1.0 : (d < 0.0 ? -1.0 : 1.0));   // no hint: keep the pivot's own sign
   const bool ok = s * d > thr;                                                    // false for NaN
   perturbed = !ok;
   return ok ? d : s * repl;
}

// ------------------------------------------------------------------------------------------------
// value scatter: arena[dst[p]] = val[p]
// ------------------------------------------------------------------------------------------------
__global__ void k_scatter(const long long* __restrict__ dst, const double* __restrict__ val, double* __restrict__ arena,
                          long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
   {
      const long long d = dst[i];
      if (d >= 0) arena[d] = val[i];   // (< 0: an entry a front takes from the value array itself)
   }
}

// arena := 0 where a factorisation reads or accumulates: the head panels and, in the dense tail, every column from the
// top of its diagonal tile down (the tiles above the diagonal are never touched: a third of the arena at config 2).
// grid (x, block); 16-byte stores
__global__ __launch_bounds__(256) void k_arena_clear(const BlkDesc* __restrict__ blks, double* __restrict__ arena, int tail_only = 0) {
   const BlkDesc bd = blks[blockIdx.y];
   typedef double double2_t __attribute__((ext_vector_type(2)));
   const double2_t z = {0.0, 0.0};
   const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gstride = (long long)gridDim.x * blockDim.x;
   double2_t* head = (double2_t*)(arena + bd.arena_off);
   const long long nh = (bd.T - bd.arena_off) / 2;   // panels are padded to 16 doubles
   if (!tail_only)
      for (long long i = gtid; i < nh; i += gstride) head[i] = z;
   // tail: a workgroup takes whole columns, its threads run down the rows
   for (int c = blockIdx.x; c < bd.m_pad; c += gridDim.x) {
      const int r0 = c / TILE * TILE;
      double2_t* col = (double2_t*)(arena + bd.T_in + (long long)c * bd.ldT + r0);
      const int n2 = (bd.ldT - r0) / 2;
      for (int i = threadIdx.x; i < n2; i += blockDim.x) col[i] = z;
   }
}

// K values <- diagonal vector (a2: put_primal_diagonal / put_dual_inequalites_diagonal / regularisation)
__global__ void k_put_diag(const long long* __restrict__ kdiag, const double* __restrict__ diag, double* __restrict__ kval,
                           long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      kval[kdiag[i]] = diag[i];
}

__global__ void k_tail_pad_diag(const BlkDesc* __restrict__ blks, double* __restrict__ arena, int nblk) {
   const int b = blockIdx.x;
   if (b >= nblk) return;
   const BlkDesc bd = blks[b];
   for (int t = bd.m + threadIdx.x; t < bd.m_pad; t += blockDim.x) arena[bd.T_in + t + (long long)t * bd.ldT] = 1.0;
}

// pivot reference magnitudes:
//    primal row: |a_kk|.  dual row: |a_kk| + sum_j K_kj^2 / |K_jj| over its primal neighbours j, i.e. the magnitude of the
//    normal-equation diagonal (W D^-1 W^T)_kk the pivot is built from - so that a pivot which cancels to rounding noise
//    (rank-deficient W) is recognised whatever sign the noise has, also when the dual diagonal itself is zero.
// k_pref_rows: per row in the caller's order, a tile of 256 rows per workgroup - the tile's entries of the lower CSR (consecutive) stream
// through LDS, all threads side by side (K_kj^2 / |K_jj| does not depend on the row), then a thread per row adds its segment.  (A thread per
// row of the permuted order walked its entries alone, two dependent gathers each: 1.07 ms per factorisation of the configs[3] share, the waves
// waiting 95 % of their cycles; with the tiles 0.83 ms - a vector of |a_jj| beside it, one gather instead of two: no better.)  k_pref_init: into the permuted order, same layout as the work vectors (xw_off, length n_head + m_pad; the
// tail's padding gets 1).
__global__ __launch_bounds__(256) void k_pref_rows(const BlkDesc* __restrict__ blks, const double* __restrict__ kval, const long long* __restrict__ kdiag,
                                                   double* __restrict__ rowref, const int* __restrict__ n_primal, const int* __restrict__ krowptr,
                                                   const int* __restrict__ kcolidx) {
   constexpr int PCH = 4096;
   __shared__ double prod[PCH];
   const BlkDesc bd = blks[blockIdx.y];
   const int np = n_primal[blockIdx.y], tid = threadIdx.x;
   for (int k0 = blockIdx.x * 256; k0 < bd.n; k0 += gridDim.x * 256) {
      const int i = k0 + tid;
      const bool row = i < bd.n;
      const long long gi = bd.x_off + (row ? i : bd.n - 1);
      double v = fabs(kval[kdiag[gi]]);
      if (np >= 0 && k0 + 256 > np) {    // (a tile of primal rows only: nothing to add)
         const int p0 = krowptr[gi], p1 = row ? krowptr[gi + 1] : p0;
         const int p_lo = krowptr[bd.x_off + k0], p_hi = krowptr[bd.x_off + min(k0 + 256, bd.n)];
         for (int c0 = p_lo; c0 < p_hi; c0 += PCH) {
            const int c1 = min(c0 + PCH, p_hi);
            for (int q = c0 + tid; q < c1; q += 256) {
               const int j = kcolidx[q];
               double pr = 0.0;
               if (j < np) {
                  const double dj = fabs(kval[kdiag[bd.x_off + j]]), a = kval[q];
                  if (dj > 0.0) pr = a * a / dj;
               }
               prod[q - c0] = pr;
            }
            __syncthreads();
            if (i >= np)
               for (int q = max(p0, c0); q < min(p1, c1); ++q) v += prod[q - c0];
            __syncthreads();
         }
      }
      if (row) rowref[gi] = v;
   }
}
__global__ void k_pref_init(const BlkDesc* __restrict__ blks, const int* __restrict__ perm, const long long* __restrict__ perm_off,
                            const double* __restrict__ rowref, double* __restrict__ pref) {
   const BlkDesc bd = blks[blockIdx.y];
   const int* p = perm + perm_off[blockIdx.y];
   const int len = bd.n_head + bd.m_pad;
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < len; k += gridDim.x * blockDim.x)
      pref[bd.xw_off + k] = k < bd.n ? rowref[bd.x_off + p[k]] : 1.0;
}

// tail columns: reference := max(reference, |diagonal after the head has been eliminated|)
__global__ void k_pref_tail(const BlkDesc* __restrict__ blks, const double* __restrict__ arena, double* __restrict__ pref,
                            int overwrite) {
   const BlkDesc bd = blks[blockIdx.y];
   for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < bd.m_pad; t += gridDim.x * blockDim.x) {
      const double v = fabs(arena[bd.T_in + t + (long long)t * bd.ldT]);
      double* q = pref + bd.xw_off + bd.n_head + t;
      *q = overwrite ? v : fmax(*q, v);
   }
}

// out[b] = max(out[b], max |v| over the rows of block b) (flat block-after-block vector); grid (chunks, block): every workgroup
// reduces a slice and folds it in with an integer atomic max (bit patterns of non-negative doubles are ordered like the numbers) - the
// caller zeroes out[] first.  (One workgroup per block streamed 600 KB alone on the time-coupled share: 115 us per call, three calls
// per refinement check.)
__global__ void k_vec_block_absmax(const double* __restrict__ v, const BlkDesc* __restrict__ blks, double* __restrict__ out) {
   const BlkDesc bd = blks[blockIdx.y];
   double mx = 0.0;
   // a NaN entry counts as +Inf (fmax would drop it and the refinement check would see a clean residual on a poisoned iterate)
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bd.n; i += gridDim.x * blockDim.x) {
      const double a = fabs(v[bd.x_off + i]);
      mx = fmax(mx, a <= 1.7976931348623157e308 ? a : __longlong_as_double(0x7ff0000000000000LL));
   }
   __shared__ double red[256];
   red[threadIdx.x] = mx;
   __syncthreads();
   for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0 && red[0] > 0.0) atomicMax((unsigned long long*)(out + blockIdx.y), (unsigned long long)__double_as_longlong(red[0]));
}

// max |K| per block -> fallback replacement magnitude.  grid (chunks, block): every workgroup reduces a slice of the block's
// values and folds it into blks[b].repl_abs with an integer atomic max (bit patterns of non-negative doubles are ordered like
// the numbers); k_block_absmax_finish turns the maximum into the replacement magnitude.  A single workgroup per block took
// 2.2 ms on the one-block sparse root.
__global__ void k_block_absmax(const double* __restrict__ kval, const long long* __restrict__ kptr, BlkDesc* blks) {
   const int b = blockIdx.y;
   double mx = 0.0;
   for (long long i = kptr[b] + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < kptr[b + 1]; i += (long long)gridDim.x * blockDim.x)
      mx = fmax(mx, fabs(kval[i]));
   __shared__ double red[256];
   red[threadIdx.x] = mx;
   __syncthreads();
   for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
      __syncthreads();
   }
   if (threadIdx.x == 0 && red[0] > 0.0)
      atomicMax((unsigned long long*)&blks[b].repl_abs, (unsigned long long)__double_as_longlong(red[0]));
}

__global__ void k_block_absmax_init(BlkDesc* blks, int nblk) {
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b < nblk) blks[b].repl_abs = 0.0;
}

__global__ void k_block_absmax_finish(BlkDesc* blks, int nblk, double thr_rel, double repl_rel) {
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= nblk) return;
   const double a = blks[b].repl_abs > 0.0 ? blks[b].repl_abs : 1.0;
   blks[b].thr_rel = thr_rel;
   blks[b].repl_rel = repl_rel;
   blks[b].repl_abs = repl_rel * a;
}

// ------------------------------------------------------------------------------------------------
// head supernode factorisation (one workgroup per supernode)
// ------------------------------------------------------------------------------------------------
constexpr int HEAD_WMAX = 32;   // widest head supernode (solve kernels)

// Head-to-head update segments (symbolic.cpp "update segments"): 8-int header + positions, read-only on the device.
constexpr int USEG_HDR = 8;
constexpr int MF_HDR_DEV = 4;

// BLOCK threads; WMAX widest supernode handled; LCAP doubles of L21 cached in LDS.
// Latency matters more than throughput here (a chain-like elimination tree runs one supernode per block per launch), so
// everything on the dependent path is kept short: pivot references / signs are fetched once, the pivot block needs one
// barrier per column (all threads derive the pivot, columns stay unscaled in LDS, 1/d is applied on the fly), and the
// scatter positions come precomputed from the analysis instead of being searched.
template <int BLOCK, int WMAX, int LCAP>
__device__ __forceinline__ void head_factor_body(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                                 const int* __restrict__ upd, const signed char* __restrict__ psign,
                                                 const long long* __restrict__ psign_off, const int* __restrict__ bmap,
                                                 double* __restrict__ arena, double* __restrict__ SC, int ldSC,
                                                 int* __restrict__ inertia, const double* __restrict__ pref,
                                                 const int* __restrict__ sctab, const ScatterCtx& sx, double* Ls, int lcap) {
   // Ls / lcap: LDS cache for L21 and its capacity in doubles (at most LCAP).  The level kernels size it per launch to what the
   // supernodes of that launch need (dynamic LDS): a fixed 48 KB held a launch of tens of thousands of small supernodes
   // (time-coupled blocks: w ~ 20, r ~ 60) to two workgroups per compute unit.
   __shared__ double Ld[WMAX * WMAX];  // pivot block, column-major ld = w; column k keeps l_ik * d_k (unscaled)
   __shared__ double dk[WMAX];
   __shared__ double prf[WMAX];
   __shared__ int sgn[WMAX];

   const int w = sn.w, r = sn.r, ld = sn.ld, tid = threadIdx.x;
   double* P = arena + sn.panel;
   const int* rows = rowidx + sn.rows;

   // Every thread owns NQ entries of the pivot block in registers: row ti, columns tj + CT*q.  Step k: the owners of
   // column k publish it (unscaled) in LDS, one barrier, everybody derives the pivot and updates its own entries.
   constexpr int CT = BLOCK / WMAX, NQ = WMAX / CT;
   static_assert(CT * WMAX == BLOCK && NQ * CT == WMAX, "thread grid must tile the pivot block");
   const int ti = tid % WMAX, tj = tid / WMAX;
   double areg[NQ];
#pragma unroll
   for (int q = 0; q < NQ; ++q) {
      const int j = tj + CT * q;
      areg[q] = (ti < w && j <= ti) ? P[ti + (long long)j * ld] : 0.0;
   }
   if (tid < w) {
      prf[tid] = pref[bd.xw_off + sn.c0 + tid];
      sgn[tid] = psign[psign_off[sn.blk] + sn.c0 + tid];
   }

   // ---- LDL^T of the w x w pivot block (right-looking); L11 goes straight to the panel
   int c_pos = 0, c_neg = 0, c_pert = 0;
   for (int k = 0; k < w; ++k) {
      const bool own = (k % CT) == tj && ti >= k && ti < w;
      double mine = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
         if (tj + CT * q == k) mine = areg[q];
      if (own) Ld[ti + k * w] = mine;
      __syncthreads();
      bool pert;
      const double d = fix_pivot(Ld[k + k * w], sgn[k], prf[k], bd.thr_rel, bd.repl_rel, bd.repl_abs, pert);
      if (pert) ++c_pert; else if (d > 0) ++c_pos; else ++c_neg;
      const double rd = 1.0 / d;
      if (ti > k && ti < w) {
         const double ci = Ld[ti + k * w];
#pragma unroll
         for (int q = 0; q < NQ; ++q) {
            const int j = tj + CT * q;
            if (j > k && j <= ti) areg[q] -= ci * (Ld[j + k * w] * rd);
         }
      }
      if (own) P[ti + (long long)k * ld] = ti == k ? d : mine * rd;
      if (tid == 0) dk[k] = d;
   }
   __syncthreads();
   if (tid == 0) {
      if (c_pos) atomicAdd(&inertia[3 * sn.blk + 0], c_pos);
      if (c_neg) atomicAdd(&inertia[3 * sn.blk + 1], c_neg);
      if (c_pert) atomicAdd(&inertia[3 * sn.blk + 2], c_pert);
   }
   if (r == 0) return;

   // ---- L21 := A21 L11^-T D^-1, one row per thread (coalesced along the rows of each column).  With the unscaled
   //      pivot block the recurrence reads  l_k = (a_k - sum_{l<k} l_l * Ld[k,l]) / d_k.
   const int lw = w | 1;   // odd LDS row stride: conflict-free when lanes walk over rows
   const bool cacheL = (long long)r * lw <= lcap;
   for (int a = tid; a < r; a += BLOCK) {
      double y[WMAX];
      double* row = P + w + a;
#pragma unroll
      for (int k = 0; k < WMAX; ++k) {
         if (k < w) {
            double v = row[(long long)k * ld];
#pragma unroll
            for (int l = 0; l < k; ++l) v -= y[l] * Ld[k + l * w];
            y[k] = v / dk[k];
         }
      }
#pragma unroll
      for (int k = 0; k < WMAX; ++k) {
         if (k < w) {
            row[(long long)k * ld] = y[k];
            if (cacheL) Ls[a * lw + k] = y[k];
         }
      }
   }
   __syncthreads();

   // ---- Schur update, scattered with FP64 atomics.  Rows below are sorted: head columns < tail columns < border.
   auto entry = [&](int a, int b) -> double {   // (L21 D L21^T)[a, b]
      double u = 0.0;
      for (int k = 0; k < w; ++k) {
         const double la = cacheL ? Ls[a * lw + k] : P[w + a + (long long)k * ld];
         const double lb = cacheL ? Ls[b * lw + k] : P[w + b + (long long)k * ld];
         u += la * lb * dk[k];
      }
      return u;
   };
   // all pairs b in [bs, be), a in [b, r): flattened over the bounding rectangle when a column alone cannot fill the
   // workgroup (short supernodes on a chain), column by column otherwise
   auto for_pairs = [&](int bs, int be, auto&& f) {
      const int np = r - bs;
      if (np >= 2 * BLOCK) {
         for (int b = bs; b < be; ++b)
            for (int a = b + tid; a < r; a += BLOCK) f(a, b);
      } else {
         const int tot = (be - bs) * np;
         for (int idx = tid; idx < tot; idx += BLOCK) {
            const int bb = idx / np, aa = idx - bb * np;
            if (aa >= bb) f(bs + aa, bs + bb);
         }
      }
   };

   int b0 = 0;
   const int* U = upd + sn.upd;
   for (int sg = 0; sg < sn.n_useg; ++sg) {   // targets inside the head: positions precomputed
      const int sb0 = U[0], sb1 = U[1], tld = U[3];
      const long long tpanel = (long long)(((unsigned long long)(unsigned)U[5] << 32) | (unsigned long long)(unsigned)U[4]);
      const int* pp = U + USEG_HDR;
      double* TP = arena + bd.arena_off + tpanel;
      for_pairs(sb0, sb1, [&](int a, int b) {
         scatter_add(sx, TP + pp[a - sb0] + (long long)pp[b - sb0] * tld, false, pair_slot(sn.slot, a, b, r), -entry(a, b));
      });
      U += USEG_HDR + (r - sb0);
      b0 = sb1;
   }
   const int n = bd.n, n_head = bd.n_head;
   if (b0 < sn.rb) {   // target columns in the dense tail
      double* T = arena + bd.T_in;
      for_pairs(b0, sn.rb, [&](int a, int b) {
         const int ra = rows[a], cb = rows[b];
         const int tr = ra < n ? ra - n_head : bd.m_pad + (ra - n);
         scatter_add(sx, T + tr + (long long)(cb - n_head) * bd.ldT, false, pair_slot(sn.slot, a, b, r), -entry(a, b));
      });
      b0 = sn.rb;
   }
   if (b0 < r && SC) {   // border x border: the Schur complement itself (SC == nullptr: factor-only call)
      const int* bm = bmap + bd.bmap_off;
      for_pairs(b0, r, [&](int a, int b) {
         scatter_add(sx, sc_entry(SC, ldSC, bm, sctab, bd.sctab_off, bd.nb, rows[a] - n, rows[b] - n), true, pair_slot(sn.slot, a, b, r), -entry(a, b));
      });
   }
}

template <int BLOCK, int WMAX, int LCAP>
__global__ __launch_bounds__(BLOCK) void k_head_factor(const SnDesc* __restrict__ sns, int sn_begin,
                                                      const BlkDesc* __restrict__ blks,
                                                      const int* __restrict__ rowidx, const int* __restrict__ upd,
                                                      const signed char* __restrict__ psign,
                                                      const long long* __restrict__ psign_off,
                                                      const int* __restrict__ bmap, double* __restrict__ arena,
                                                      double* __restrict__ SC, int ldSC, int* __restrict__ inertia,
                                                      const double* __restrict__ pref, const int* __restrict__ sctab,
                                                      ScatterCtx sx = ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}, int lcap = LCAP) {
   extern __shared__ double head_Ls[];   // lcap doubles (launch parameter)
   const SnDesc sn = sns[sn_begin + blockIdx.x];
   const BlkDesc bd = blks[sn.blk];
   head_factor_body<BLOCK, WMAX, LCAP>(sn, bd, rowidx, upd, psign, psign_off, bmap, arena, SC, ldSC, inertia, pref, sctab, sx, head_Ls, lcap);
}

// "Spine" of a chain-like elimination tree: the top levels that hold at most two supernodes per block.  Level scheduling
// would spend one launch per level on them; here one workgroup per block walks its spine supernodes in postorder inside a
// single launch.  A supernode's panel receives atomic updates (performed at L2) from earlier ones of the same workgroup:
// the agent-scope fence drains this thread's atomics / stores and invalidates the L1 lines a neighbouring panel may have
// left behind, the barrier orders the workgroup.
template <int BLOCK, int WMAX, int LCAP>
__global__ __launch_bounds__(BLOCK) void k_head_factor_spine(const int* __restrict__ spine, const int* __restrict__ spine_off,
                                                            const SnDesc* __restrict__ sns, const BlkDesc* __restrict__ blks,
                                                            const int* __restrict__ rowidx, const int* __restrict__ upd,
                                                            const signed char* __restrict__ psign,
                                                            const long long* __restrict__ psign_off,
                                                            const int* __restrict__ bmap, double* __restrict__ arena,
                                                            double* __restrict__ SC, int ldSC, int* __restrict__ inertia,
                                                            const double* __restrict__ pref, const int* __restrict__ sctab) {
   __shared__ double Ls[LCAP];
   const int p0 = spine_off[blockIdx.x], p1 = spine_off[blockIdx.x + 1];
   if (p0 == p1) return;
   const BlkDesc bd = blks[blockIdx.x];
   for (int p = p0; p < p1; ++p) {
      const SnDesc sn = sns[spine[p]];
      head_factor_body<BLOCK, WMAX, LCAP>(sn, bd, rowidx, upd, psign, psign_off, bmap, arena, SC, ldSC, inertia, pref, sctab,
                                          ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}, Ls, LCAP);
      __threadfence();
      __syncthreads();
   }
}

// ------------------------------------------------------------------------------------------------
// "simple leaf" head supernodes: width 1, at most 16 rows below, elimination-tree leaves (every primal column of an LP block
// looks like this: ~10^4 per block; their rows lie in the dense tail / border, or - time-coupled blocks - in the head, where
// the positions come from the segment tables).  One THREAD per supernode:
// the launch is throughput-bound instead of paying a workgroup's dependent-load latency chain per 21-flop supernode.
// ------------------------------------------------------------------------------------------------
constexpr int SIMPLE_RMAX = 16;

__global__ __launch_bounds__(256) void k_head_factor_simple(const SnDesc* __restrict__ sns, int sn_begin, int cnt,
                                                           const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                           const int* __restrict__ upd, const signed char* __restrict__ psign,
                                                           const long long* __restrict__ psign_off, const int* __restrict__ bmap,
                                                           double* __restrict__ arena, double* __restrict__ SC, int ldSC,
                                                           int* __restrict__ inertia, const double* __restrict__ pref,
                                                           const int* __restrict__ sctab,
                                                           ScatterCtx sx = ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}, int mf = 0,
                                                           double* __restrict__ lvals = nullptr, const int* __restrict__ lfpos = nullptr,
                                                           double* __restrict__ lfval = nullptr, double* __restrict__ bbarena = nullptr) {
   __shared__ int cnt_s[3];
   __shared__ int blk_s;
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (threadIdx.x == 0) blk_s = sns[sn_begin + min(blockIdx.x * blockDim.x, (unsigned)cnt - 1)].blk;
   if (threadIdx.x < 3) cnt_s[threadIdx.x] = 0;
   __syncthreads();
   if (t < cnt) {
      const SnDesc sn = sns[sn_begin + t];
      const BlkDesc bd = blks[sn.blk];
      const int r = sn.r;
      double* P = arena + sn.panel;
      const int* rows = rowidx + sn.rows;
      bool pert;
      const double d = fix_pivot(P[0], psign[psign_off[sn.blk] + sn.c0], pref[bd.xw_off + sn.c0], bd.thr_rel, bd.repl_rel, bd.repl_abs, pert);
      P[0] = d;
      const int which = pert ? 2 : (d > 0 ? 0 : 1);
      if (sn.blk == blk_s) atomicAdd(&cnt_s[which], 1); else atomicAdd(&inertia[3 * sn.blk + which], 1);
      double l[SIMPLE_RMAX];
      int ro[SIMPLE_RMAX];
#pragma unroll
      for (int a = 0; a < SIMPLE_RMAX; ++a)
         if (a < r) { ro[a] = rows[a]; l[a] = P[1 + a] / d; P[1 + a] = l[a]; }
      if (lfpos) {   // ... and once more in the order the forward gather reads them (k_leaf_fwd_gather)
         const int* lp = lfpos + sn.rows;
#pragma unroll
         for (int a = 0; a < SIMPLE_RMAX; ++a)
            if (a < r && ro[a] < bd.n) lfval[lp[a]] = l[a];
      }
      double* T = arena + bd.T_in;
      const int* bm = bmap + bd.bmap_off;
      const int n = bd.n, n_head = bd.n_head;
      // multifrontal head: a leaf below a front leaves its rank-one update to that front (k_front reads d and l from the panel)
      const bool to_parent = mf && sn.n_useg > 0;
      if (to_parent) {   // ... as one contiguous piece per front: d, l_0 .. l_{r-1}
         double* lv = lvals + sn.U;
         lv[0] = d;
#pragma unroll
         for (int a = 0; a < SIMPLE_RMAX; ++a)
            if (a < r) lv[1 + a] = l[a];
         if (sn.bb >= 0) {   // border split: the leaf's border rows for k_border_schur (w = 1: Lt[a], then the pivot)
            double* Q = bbarena + sn.bb;
#pragma unroll
            for (int a = 0; a < SIMPLE_RMAX; ++a)
               if (a < r && a >= sn.rb) Q[a - sn.rb] = l[a];
            Q[(r - sn.rb + 3) & ~3] = d;
         }
      }
      // target columns inside the head (time-coupled blocks: the rows of a primal column are dual rows the dissection keeps in the
      // head): positions from the precomputed segment tables, as in head_factor_body; l is re-read from the panel (L1 hits)
      // because the segment bounds are run-time values and a dynamically indexed register array would go to scratch
      int b_head = 0;
      if (to_parent) b_head = r;
      else {
         const int* U = upd + sn.upd;
         const int nseg = sn.n_useg < r ? sn.n_useg : r;   // every segment holds at least one of the r rows
         for (int sg = 0; sg < nseg; ++sg) {
            const int sb0 = U[0], sb1 = U[1], tld = U[3];
            if (sb0 != b_head || sb1 <= sb0 || sb1 > r) break;   // segments tile the head rows in order; anything else is not a table of this supernode
            const long long tpanel = (long long)(((unsigned long long)(unsigned)U[5] << 32) | (unsigned long long)(unsigned)U[4]);
            const int* pp = U + USEG_HDR;
            double* TP = arena + bd.arena_off + tpanel;
            for (int b = sb0; b < sb1; ++b) {
               const double lbd = P[1 + b] * d;
               const long long colo = (long long)pp[b - sb0] * tld;
               for (int a = b; a < r; ++a)
                  scatter_add(sx, TP + pp[a - sb0] + colo, false, pair_slot(sn.slot, a, b, r), -P[1 + a] * lbd);
            }
            U += USEG_HDR + (r - sb0);
            b_head = sb1;
         }
      }
#pragma unroll
      for (int b = 0; b < SIMPLE_RMAX; ++b) {
         if (b < r && b >= b_head) {
            const int cb = ro[b];
            const double lbd = l[b] * d;
#pragma unroll
            for (int a = b; a < SIMPLE_RMAX; ++a) {
               if (a < r) {
                  const int ra = ro[a];
                  const double u = l[a] * lbd;
                  if (cb < n) {
                     const int tr = ra < n ? ra - n_head : bd.m_pad + (ra - n);
                     scatter_add(sx, T + tr + (long long)(cb - n_head) * bd.ldT, false, pair_slot(sn.slot, a, b, r), -u);
                  } else if (SC) {
                     scatter_add(sx, sc_entry(SC, ldSC, bm, sctab, bd.sctab_off, bd.nb, ra - n, cb - n), true, pair_slot(sn.slot, a, b, r), -u);
                  }
               }
            }
         }
      }
   }
   __syncthreads();
   if (threadIdx.x < 3 && cnt_s[threadIdx.x]) atomicAdd(&inertia[3 * blk_s + threadIdx.x], cnt_s[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// multifrontal head: one workgroup per front, no FP64 atomics between fronts.
//   front of supernode J = lower triangle on (its w columns, its r below-rows), packed by columns:
//       F(i, j) = F[co(j) + i - j],  co(j) = j nf - j (j - 1) / 2,  nf = w + r
//   the columns j >= w are the update matrix U_J - in the same packed layout a matrix of dimension r has, so it leaves
//   (to the update arena) and arrives (from the children) as one contiguous piece.  The whole front lives in LDS (UG = false);
//   fronts too large for that keep only their w panel columns there and work on U_J where it ends up anyway, in the update
//   arena (UG = true: plain loads and stores of one workgroup, ordered by its barriers).
//   A front is a latency chain, not a stream; the kernel is organised around that:
//   0. thread i requests row i of the panel (K entries, scattered into the arena by k_scatter) straight into registers, and the
//      first w lanes of EVERY wave the pivot rows - up to 64 loads in flight per thread while the LDS front is zeroed;
//   1. assemble what the children left: child fronts' update matrices are added at the recorded positions, one child after
//      the other (inside a child the positions are distinct: fire-and-forget LDS adds); the rank-one updates of the simple
//      leaves below, a thread per front column walking its item list.  Fixed order everywhere: results do not depend on timing;
//   2. factorise the w panel columns.  Every wave holds the pivot rows itself (lane j = row j) and eliminates them redundantly,
//      applying each column to its own rows as it goes: pivot, reciprocal and the column's entries travel by v_readlane, there
//      is no barrier and no LDS traffic in the w dependent steps;
//   3. U -= L21 D L21^T in 4 x 4 register tiles from a 16-byte aligned copy of L21;
//   4. U to the update arena - or, for a front whose parent column lies in the dense tail, into the tail / Schur complement (the
//      only atomics left: these targets are shared between fronts).  The panel went to the arena from the registers in step 2.
// BLOCK >= w + r threads; WMAX >= w.
// ------------------------------------------------------------------------------------------------
// packed lower-triangular index p (columns of a matrix of dimension n one after the other) -> column b; row a = b + p - cu(b)
__device__ __forceinline__ int packed_col(int p, int n) {
   const float t = 2.0f * n + 1.0f;
   int b = (int)((t - sqrtf(t * t - 8.0f * (float)p)) * 0.5f);
   b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
   while (b > 0 && b * n - b * (b - 1) / 2 > p) --b;
   while (b + 1 < n && (b + 1) * n - (b + 1) * b / 2 <= p) ++b;
   return b;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {   // lane: wave-uniform
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
   return __hiloint2double(hi, lo);
}

// fire-and-forget LDS add (ds_add_f64): no read latency on the issuing wave; where several adds hit one address their order
// is the program order of the one thread that issues them, or separated by a barrier
__device__ __forceinline__ void lds_add(double* p, double v) {
   (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int BLOCK, int WMAX, bool UG>
__global__ __launch_bounds__(BLOCK) void k_front(const SnDesc* __restrict__ sns, int sn_begin, const BlkDesc* __restrict__ blks,
                                                const int* __restrict__ rowidx, const int* __restrict__ mfint,
                                                const signed char* __restrict__ psign, const long long* __restrict__ psign_off,
                                                const int* __restrict__ bmap, double* __restrict__ arena,
                                                double* __restrict__ uarena, double* __restrict__ SC, int ldSC,
                                                int* __restrict__ inertia, const double* __restrict__ pref,
                                                const int* __restrict__ sctab, long long* __restrict__ dbg,
                                                const double* __restrict__ lvals, const double* __restrict__ kval,
                                                const double* __restrict__ bval, int ordered, double* __restrict__ bbarena) {
   extern __shared__ __attribute__((aligned(16))) double mf_F[];
   __shared__ double dk[WMAX];
   // development aid (PIPS_HIP_MF_CLOCKS): thread 0 stamps the phase boundaries, 8 stamps per front
#define MF_STAMP(q) do { if (dbg && threadIdx.x == 0) dbg[(long long)(sn_begin + blockIdx.x) * 8 + (q)] = (long long)wall_clock64(); } while (0)
   MF_STAMP(0);
   double* F = mf_F;
   const SnDesc sn = sns[sn_begin + blockIdx.x];
   const BlkDesc bd = blks[sn.blk];
   // fronts on the rows of K only (BlkDesc::mf_split == 2): the front ends with its rb rows of K - its border rows are formed afterwards from
   // the finished panels (k_border_rows), nothing of the border travels from front to front
   const int w = sn.w, r = bd.mf_split == 2 ? sn.rb : sn.r, nf = w + r, tid = threadIdx.x, lane = tid & 63, i = tid;
   const int* H = mfint + sn.mf;
   const int n_child = H[0], n_leaf = H[1], n_ent = H[2] >> 1, n_leafpart = H[3], n_items = H[4], n_vals = H[5], sum_rc = H[6];
   // update columns the front keeps and hands on: all r, or (border split, BlkDesc::mf_split) only those of its rb rows of K - the
   // border x border part is formed by k_border_schur from the finished panels
   const int uc = bd.mf_split ? sn.rb : r;
   const int np = uc * r - uc * (uc - 1) / 2;
   auto co = [nf](int j) { return j * nf - j * (j - 1) / 2; };
   const int cw = co(w);                                       // packed size of the panel columns
   const int rp = (r + 3) & ~3;                                // row stride of the aligned L21 copy (step 3)
   const int pw = cw > w * rp ? cw : w * rp;   // the panel region holds the packed panel, later that copy
   double* FU = UG ? uarena + sn.U : F + pw;                   // U(a, b) = FU[cu(b) + a - b]
   auto cu = [r](int b) { return b * r - b * (b - 1) / 2; };
   // F(i, j) += v
   auto front_add = [&](int fi, int fj, double v) {
      if (fj < w) lds_add(F + co(fj) + (fi - fj), v);
      else if (UG) FU[cu(fj - w) + (fi - fj)] += v;
      else lds_add(FU + cu(fj - w) + (fi - fj), v);
   };
   double* P = arena + sn.panel;
   // LDS behind the front: the leaves' values, then ints: the children's position lists, the leaf part of the record
   double* vals = F + pw + (UG ? 0 : np) + 8;
   int* relbuf = (int*)(vals + n_vals);
   int* leafpart = relbuf + sum_rc;

   const double prf = lane < w ? pref[bd.xw_off + sn.c0 + lane] : 1.0;
   const int sgn = lane < w ? (int)psign[psign_off[sn.blk] + sn.c0 + lane] : 0;

   // ---- 1. assemble
   {
      typedef double double2_t __attribute__((ext_vector_type(2)));
      const int nz = (pw + (UG ? 0 : np) + 8 + 1) >> 1;          // (the region behind is overwritten below: an odd tail is harmless)
      double2_t* F2 = (double2_t*)F;
      const double2_t z = {0.0, 0.0};
      for (int idx = tid; idx < nz; idx += BLOCK) F2[idx] = z;
      if (UG) for (int idx = tid; idx < np; idx += BLOCK) FU[idx] = 0.0;
   }
   __syncthreads();   // zeros before the staged data (the zeroing may reach one double into the values)
   MF_STAMP(1);
   {   // the children's position lists and the leaf part: one contiguous piece of the record
      const int* src = H + MF_HDR + 3 * n_child;
      for (int idx = tid; idx < sum_rc + n_leafpart; idx += BLOCK) relbuf[idx] = src[idx];
   }
   __syncthreads();
   MF_STAMP(2);
   if (n_leaf) {   // the leaves' d and l: written as one piece by the leaf kernel
      const double* lv = lvals + bd.lv_off + H[7];
      for (int idx = tid; idx < n_vals; idx += BLOCK) vals[idx] = lv[idx];
   }
   {  // the panel's own entries of K and of the border, straight from the value arrays (the record lists them behind the leaf part; the
      // panel in the arena is neither cleared nor scattered into).  (Requested here and not ahead of the zeroing: holding them across it
      // costs the registers that keep four waves on a SIMD - measured 27.5 against 25.7 ms.)
      const int* E = H + MF_HDR + 3 * n_child + sum_rc + n_leafpart + 1;
      const double* kv = kval + bd.k_off;
      const double* bv = bval + bd.b_off;
      for (int e = tid; e < n_ent; e += BLOCK) {
         const int pos = E[2 * e], src = E[2 * e + 1];
         lds_add(F + pos, src >= 0 ? kv[src] : bv[-1 - src]);
      }
   }
   {
      // A wave takes a PAIR of columns (b, rc - 1 - b) of a child's packed update matrix - together rc + 1 entries whatever b is -
      // and walks down it 64 rows at a time: coalesced reads, the target column is wave-uniform, one LDS read (the row's position)
      // and one fire-and-forget add per entry.  (pair, 64-row piece) items go round-robin over the waves, NB loads in flight each.
      // A child that hands over only its first ucc columns (border split): a wave per (column, 64-row piece), the columns are long.
      constexpr int NW = BLOCK / 64, NB = 12;
      const int wave = tid >> 6;
      int off = 0;
      for (int c = 0; c < n_child; ++c) {
         const int rcw = H[MF_HDR + 3 * c + 1], rc = rcw & 0xffff, ucc = rcw >> 16;
         const double* Uc = uarena + sn.U + H[MF_HDR + 3 * c];
         const int* rel = relbuf + off;
         const bool whole = ucc == rc;
         const int npairs = (rc + 1) >> 1, pieces = ((whole ? rc + 1 : rc) + 63) >> 6, nitems = (whole ? npairs : ucc) * pieces;
         for (int s0 = wave; s0 < nitems; s0 += NW * NB) {
            double v[NB];
            int ea[NB], eb[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
               const int item = s0 + u * NW;
               ea[u] = -1; eb[u] = 0; v[u] = 0.0;
               if (item < nitems) {
                  const int pr = item / pieces, e = (item - pr * pieces) * 64 + lane;   // pr, pieces: wave-uniform
                  int a = -1, b = pr;
                  if (whole) {
                     const int b1 = pr, b2 = rc - 1 - pr, n1 = rc - b1;
                     if (e < n1) a = b1 + e;
                     else if (b2 != b1 && e - n1 <= pr) { b = b2; a = b2 + (e - n1); }
                  } else if (pr + e < rc) a = pr + e;
                  if (a >= 0) { ea[u] = a; eb[u] = b; v[u] = Uc[b * rc - b * (b - 1) / 2 + (a - b)]; }
               }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
               if (ea[u] >= 0) front_add(rel[ea[u]], rel[eb[u]], v[u]);
         }
         off += rc;
         __syncthreads();
      }
   }
   MF_STAMP(3);
   if (n_leaf) {
      if (n_child == 0) __syncthreads();   // the leaf values are in place
      const int* colptr = leafpart;
      const int* it2 = leafpart + (nf + 1);
      if (ordered || UG) {   // a thread per front column walks that column's items: every sum in a fixed order (deterministic mode), and
                             // one writer per target (the update matrix in device memory takes plain adds)
         for (int q = tid; q < nf; q += BLOCK) {
            for (int it = colptr[q]; it < colptr[q + 1]; ++it) {
               const int i0 = it2[2 * it], i1 = it2[2 * it + 1];
               const int b = i0 & 15, rc = (i0 >> 4) & 31;
               const double* lv = vals + (i0 >> 9);
               const int* rel = leafpart + i1;
               const double lbd = -lv[1 + b] * lv[0];
               for (int a = b; a < rc; ++a) front_add(rel[a], q, lv[1 + a] * lbd);
            }
         }
      } else {         // a thread per item (leaf, b): all threads busy instead of one per front column; the adds are atomic
         for (int it = tid; it < n_items; it += BLOCK) {
            const int i0 = it2[2 * it], i1 = it2[2 * it + 1];
            const int b = i0 & 15, rc = (i0 >> 4) & 31;
            const double* lv = vals + (i0 >> 9);
            const int* rel = leafpart + i1;
            const int q = rel[b];
            const double lbd = -lv[1 + b] * lv[0];
            for (int a = b; a < rc; ++a) front_add(rel[a], q, lv[1 + a] * lbd);
         }
      }
   }
   __syncthreads();

   // ---- 2. panel factorisation
   MF_STAMP(4);
   // thread i takes row i of the panel (K entries, scattered into the arena by k_scatter, + what was assembled in LDS); the first
   // w lanes of EVERY wave also take the pivot rows
   double y[WMAX], yp[WMAX];
   const bool wave_has_rows = (tid & ~63) < nf;               // wave-uniform: the waves beyond the last row only help in the other phases
   if (wave_has_rows) {
#pragma unroll
   for (int k = 0; k < WMAX; ++k) y[k] = (i < nf && k < w && k <= i) ? F[co(k) + i - k] : 0.0;
#pragma unroll
   for (int k = 0; k < WMAX; ++k) yp[k] = (lane < w && k <= lane) ? F[co(k) + lane - k] : 0.0;
   int c_pos = 0, c_neg = 0, c_pert = 0;
   // The pivot of column k + 1 is known as soon as column k has been applied to row k + 1: it is fixed and inverted right there, so
   // that the reciprocal's dependent chain runs beside the other updates of column k instead of after them.
   bool pert;
   double d = fix_pivot(readlane_f64(yp[0], 0), __builtin_amdgcn_readlane(sgn, 0), readlane_f64(prf, 0), bd.thr_rel, bd.repl_rel, bd.repl_abs, pert);
   double rd = 1.0 / d;
#pragma unroll
   for (int k = 0; k < WMAX; ++k) {
      if (k < w) {
         if (pert) ++c_pert; else if (d > 0) ++c_pos; else ++c_neg;
         const double uik = y[k], upk = yp[k];
         const double tcol = upk * rd;                     // lane j: l_jk
         // no branch per column: beyond column w - 1 tcol is 0 (yp is), and entries right of the diagonal of the pivot block
         // collect garbage that nobody reads
         double dn = 1.0, rdn = 1.0;
         bool pertn = false;
         if (k + 1 < WMAX) {
            const double t = readlane_f64(tcol, k + 1);
            y[k + 1] -= uik * t;
            yp[k + 1] -= upk * t;
            dn = fix_pivot(readlane_f64(yp[k + 1], k + 1), __builtin_amdgcn_readlane(sgn, k + 1), readlane_f64(prf, k + 1), bd.thr_rel,
                           bd.repl_rel, bd.repl_abs, pertn);
            rdn = 1.0 / dn;
         }
#pragma unroll
         for (int j = k + 2; j < WMAX; ++j) {
            const double t = readlane_f64(tcol, j);
            y[j] -= uik * t;
            yp[j] -= upk * t;
         }
         y[k] = i == k ? d : uik * rd;
         if (tid == 0) dk[k] = d;
         d = dn; rd = rdn; pert = pertn;
      }
   }
   if (tid == 0) {
      if (c_pos) atomicAdd(&inertia[3 * sn.blk + 0], c_pos);
      if (c_neg) atomicAdd(&inertia[3 * sn.blk + 1], c_neg);
      if (c_pert) atomicAdd(&inertia[3 * sn.blk + 2], c_pert);
   }
   if (i < nf) {
#pragma unroll
      for (int k = 0; k < WMAX; ++k)
         if (k < w && k <= i && i < sn.ld) P[i + (long long)k * sn.ld] = y[k];    // l_ik, d_k on the diagonal: what the solves read
      if (sn.bb >= 0 && i >= w + sn.rb) {   // border split: the border rows go to the border-row arena, in the layout k_border_schur stages
         const int nbj = r - sn.rb, rpb = (nbj + 3) & ~3;
         double* Q = bbarena + sn.bb + (i - w - sn.rb);
#pragma unroll
         for (int k = 0; k < WMAX; ++k)
            if (k < w) Q[k * rpb] = y[k];
      }
   }
   }
   __syncthreads();   // every wave has taken its rows out of the packed panel: the region becomes the L21 copy Lt[k * rp + a]
   if (sn.bb >= 0 && tid < w) { const int rpb = (sn.r - sn.rb + 3) & ~3; bbarena[sn.bb + w * rpb + tid] = dk[tid]; }
   if (i >= w && i < nf) {
#pragma unroll
      for (int k = 0; k < WMAX; ++k)
         if (k < w) F[k * rp + (i - w)] = y[k];
   }
   for (int idx = tid; idx < w * (rp - r); idx += BLOCK) { const int k = idx / (rp - r); F[k * rp + r + (idx - k * (rp - r))] = 0.0; }
   __syncthreads();

   // ---- 3. U -= L21 D L21^T, 4 x 4 tiles over the lower triangle (tile pairs ta >= tb, column-major)
   MF_STAMP(5);
   if (r > 0) {
      typedef double double2_t __attribute__((ext_vector_type(2)));
      const int nt = rp >> 2, ntb = (uc + 3) >> 2, ntiles = ntb * nt - ntb * (ntb - 1) / 2;   // column tiles tb < ntb only
      for (int t = tid; t < ntiles; t += BLOCK) {
         const int tb = packed_col(t, nt), ta = tb + (t - (tb * nt - tb * (tb - 1) / 2));
         const int a0 = 4 * ta, b0 = 4 * tb;
         double acc[4][4];
#pragma unroll
         for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int z = 0; z < 4; ++z) acc[x][z] = 0.0;
#pragma unroll 4
         for (int k = 0; k < w; ++k) {
            const double2_t* ca = (const double2_t*)(F + k * rp + a0);
            const double2_t* cb = (const double2_t*)(F + k * rp + b0);
            const double2_t a01 = ca[0], a23 = ca[1], b01 = cb[0], b23 = cb[1];
            const double d = dk[k];
            const double la[4] = {a01.x, a01.y, a23.x, a23.y};
            const double lb[4] = {b01.x * d, b01.y * d, b23.x * d, b23.y * d};
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
               for (int z = 0; z < 4; ++z) acc[x][z] += la[x] * lb[z];
         }
#pragma unroll
         for (int z = 0; z < 4; ++z) {
            const int b = b0 + z;
            if (b < uc) {
               double* ub = FU + cu(b) - b;
#pragma unroll
               for (int x = 0; x < 4; ++x) {
                  const int a = a0 + x;
                  if (a < r && a >= b) { if (UG) ub[a] -= acc[x][z]; else lds_add(ub + a, -acc[x][z]); }
               }
            }
         }
      }
   }
   __syncthreads();

   // ---- 4. the update matrix leaves
   MF_STAMP(6);
   if (np == 0) { MF_STAMP(7); return; }
   // the update matrix goes to the update arena: the parent front picks it up there; for a front without a head parent (parent
   // column in the dense tail) k_root_assemble adds it to the tail / Schur complement after the last level, front by front in a
   // fixed order - no atomics on targets that several fronts of a block share
   if (!UG) {
      double* Ug = uarena + sn.U;
      for (int idx = tid; idx < np; idx += BLOCK) Ug[idx] = FU[idx];
   }
   MF_STAMP(7);
}

// Update matrices of the fronts without a head parent -> tail panel and Schur complement.  One workgroup per block walks the
// block's root fronts in ascending order (inside a front the targets are distinct; a barrier separates the fronts): plain adds on
// the block's own tail.  Schur targets are shared between blocks: FP64 atomics by default; in deterministic mode (gbuf != nullptr)
// a launch holds at most one block of every group and adds into the group's buffer, so the blocks of a group arrive in order.
// One workgroup per (block, chunk of target columns): blockIdx.y < nt takes a slice of the tail's columns, the others a slice of
// the block's border columns (Schur complement).  A target entry belongs to the chunk of its column - the smaller of its two block-local
// indices, the same for every front that reaches it - so each entry has ONE writer, which walks the fronts in ascending order: the sums
// have a fixed order and the tail needs no atomics.  (One workgroup per block walked all fronts alone: 7.5 ms on the configs[1] blocks,
// where nearly every front is a root with ~130 tail rows.)  The fronts are scanned 256 at a time, one thread each, for the piece of their
// (ascending) row list that falls into the chunk.
// The launch is latency (a barrier and a read-modify-write round trip per front and workgroup), so the engine asks for about as many
// workgroups as are resident at once (8 per compute unit): nt + ns chunks per block, between 2 and 64.

__global__ __launch_bounds__(256) void k_root_assemble(const int* __restrict__ blk_list, const int* __restrict__ root_off,
                                                      const int* __restrict__ roots, const SnDesc* __restrict__ sns,
                                                      const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                      const int* __restrict__ bmap, double* __restrict__ arena,
                                                      const double* __restrict__ uarena, double* __restrict__ SC, int ldSC,
                                                      const int* __restrict__ sctab, double* __restrict__ gbuf, long long gstride,
                                                      const int* __restrict__ blk_group, int nt, int ns) {
   __shared__ int s_b0[256], s_b1[256], s_r[256], s_rb[256];
   __shared__ long long s_rows[256], s_U[256];
   const int blk = blk_list ? blk_list[blockIdx.x] : blockIdx.x;
   const BlkDesc bd = blks[blk];
   double* T = arena + bd.T_in;
   const int* bm = bmap + bd.bmap_off;
   const int n = bd.n, n_head = bd.n_head, c = blockIdx.y;
   double* S_ = gbuf ? gbuf + gstride * blk_group[blk] : SC;
   const bool tail_chunk = c < nt;
   if (!tail_chunk && (!S_ || bd.mf_split)) return;   // (border split: the update matrices hold no border x border part, k_border_schur forms it)
   // block-local row ids [lo, hi) whose columns this workgroup owns
   int lo, hi;
   if (tail_chunk) {
      const int cw = (bd.m_pad + nt - 1) / nt;
      lo = n_head + c * cw; hi = min(lo + cw, n);
   } else {
      const int cw = (bd.nb + ns - 1) / ns;
      lo = n + (c - nt) * cw; hi = min(lo + cw, n + bd.nb);
   }
   if (lo >= hi) return;
   const int q_begin = root_off[blk], q_end = root_off[blk + 1];
   for (int q0 = q_begin; q0 < q_end; q0 += 256) {
      const int nq = min(256, q_end - q0);
      __syncthreads();
      if ((int)threadIdx.x < nq) {
         const SnDesc sn = sns[roots[q0 + threadIdx.x]];
         const int* rows = rowidx + sn.rows;
         // first row >= lo and first row >= hi in the ascending list
         int a = 0, b = sn.r;
         while (a < b) { const int m = (a + b) >> 1; if (rows[m] < lo) a = m + 1; else b = m; }
         const int b0 = a;
         b = sn.r;
         while (a < b) { const int m = (a + b) >> 1; if (rows[m] < hi) a = m + 1; else b = m; }
         s_b0[threadIdx.x] = b0; s_b1[threadIdx.x] = a; s_r[threadIdx.x] = sn.r; s_rb[threadIdx.x] = sn.rb;
         s_rows[threadIdx.x] = sn.rows; s_U[threadIdx.x] = sn.U;
      }
      __syncthreads();
      for (int k = 0; k < nq; ++k) {
         const int b0 = s_b0[k], b1 = s_b1[k];
         if (b0 >= b1) continue;   // (uniform: every thread reads the same LDS words)
         const int rb = s_rb[k], r = bd.mf_split == 2 ? rb : s_r[k];   // (fronts on the rows of K only hand over a K x K update matrix)
         const int* rows = rowidx + s_rows[k];
         const double* U = uarena + s_U[k];
         // entries (a, b), b in [b0, b1), a in [b, r): column b holds r - b of them
         const int per0 = r - b0, total = (b1 - b0) * per0;   // the rectangle [b0, r) x [b0, b1); its corner above the diagonal is skipped
         for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int bi = idx / per0, a = b0 + (idx - bi * per0), b = b0 + bi;
            if (a < b) continue;
            const double u = U[b * r - b * (b - 1) / 2 + a - b];
            const int ra = rows[a], cb = rows[b];
            if (b < rb) {
               const int tr = ra < n ? ra - n_head : bd.m_pad + (ra - n);
               T[tr + (long long)(cb - n_head) * bd.ldT] += u;
            } else {
               double* tgt = sc_entry(S_, ldSC, bm, sctab, bd.sctab_off, bd.nb, ra - n, cb - n);
               if (gbuf) *tgt += u; else atomic_add_f64(tgt, u);
            }
         }
         __syncthreads();   // the next front may reach the same entries from other threads
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Fronts on the rows of K only (BlockSym::mf_konly): the border rows of a supernode in GATHER form.
//    L_b(J) = ( B_J - sum_C L_b(C) D_C L_{J,C}^T ) L_JJ^-T D_J^-1
// over the supernodes C below J that hold border rows and have below-rows q in [b0, b1) inside J's columns (L_{J,C}: those rows of C's
// panel; the update segments of the symbolic phase list exactly these pairs).  A border row never influences a pivot, so none of this
// needs to ride through the fronts: k_front factorises the rows of K (fronts of ~33 rows instead of ~130 on the time-coupled blocks),
// and this kernel forms the border rows afterwards, level by level, from finished panels.
// One workgroup per supernode J, its border rows X (nbj x w) in LDS; the waves take J's pairs round-robin, the lanes along the border rows of C
// (L_b(C) from the border-row arena, coalesced; matched to J's rows through an LDS map border id -> row of J), d_C(c) L_C(q, c) wave-uniform,
// the products are added to X with LDS atomics (no return value).  A first version held X in registers (one wave per J, the pairs one after the
// other behind three barriers each, the border's entries in a serial loop): 19.6 ms per factorisation of the configs[3] share, latency from end to end.
// rec: per J  { n_pairs, n_ent, (C, b0 | b1 << 16) x n_pairs, (a | k << 16, src) x n_ent } with the entries of the border that fall into J's
// columns (src: index into the block's border values).  fin: the arena again, read-only (finished panels and border rows of lower levels).
// ------------------------------------------------------------------------------------------------
constexpr int BR_IDMAX = 192;      // border ids of a block under the border split: < 176
constexpr int BR_PCHUNK = 64;      // pairs whose descriptors are staged at once
struct BrPair { long long panel, bb, rows; int wC, nbc, ldC, ns, b0, pad; };   // what a pair needs of its supernode C (panel: at row w_C + b0)
template <int BLOCK, int WMAX>
__global__ __launch_bounds__(BLOCK) void k_border_rows(const int* __restrict__ list, const SnDesc* __restrict__ sns, const BlkDesc* __restrict__ blks,
                                                       const int* __restrict__ rowidx, const int* __restrict__ rec, const long long* __restrict__ rec_off,
                                                       double* __restrict__ arena, const double* __restrict__ fin, const double* __restrict__ bval) {
   extern __shared__ __attribute__((aligned(16))) double br_X[];       // X[k * rp + a]: border row a of J, column k
   constexpr int NW = BLOCK / 64;
   __shared__ short map[BR_IDMAX], kq[NW * WMAX];
   __shared__ double ljj[WMAX * WMAX], dj[WMAX], dl[NW * WMAX * WMAX];
   __shared__ BrPair pd[BR_PCHUNK];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int jid = list[blockIdx.x];
   const SnDesc J = sns[jid];
   const int* H = rec + rec_off[jid];
   const int bn = blks[J.blk].n;
   const long long b_off = blks[J.blk].b_off;
   const int w = J.w, nbj = J.r - J.rb, rp = (nbj + 3) & ~3;
   const int n_pairs = H[0], n_ent = H[1];
   double* X = br_X;
   // ---- everything the front needs that does not depend on another load of this kernel is requested here, in one go
   for (int p = tid; p < n_pairs && p < BR_PCHUNK; p += BLOCK) {
      const int seg = H[3 + 2 * p], b0 = seg & 0xffff;
      const SnDesc C = sns[H[2 + 2 * p]];
      pd[p] = BrPair{C.panel + C.w + b0, C.bb, C.rows, C.w, C.r - C.rb, C.ld, (seg >> 16) - b0, b0, C.rb};
   }
   for (int idx = tid; idx < w * rp; idx += BLOCK) X[idx] = 0.0;
   for (int g = tid; g < BR_IDMAX; g += BLOCK) map[g] = -1;
   {  // the pivot block of J, for the last step
      const double* PJ = fin + J.panel;
      for (int idx = tid; idx < w * w; idx += BLOCK) {
         const int k = idx / w, j = idx - k * w;
         ljj[k * WMAX + j] = k > j ? PJ[k + (long long)j * J.ld] : 0.0;
      }
      if (tid < w) dj[tid] = PJ[tid + (long long)tid * J.ld];
   }
   __syncthreads();
   for (int a = tid; a < nbj; a += BLOCK) map[rowidx[J.rows + J.rb + a] - bn] = (short)a;
   {  // the border's own entries in J's columns: distinct (row, column) positions
      const int* E = H + 2 + 2 * n_pairs;
      const double* bv = bval + b_off;
      for (int e = tid; e < n_ent; e += BLOCK) {
         const int ak = E[2 * e];
         X[(ak >> 16) * rp + (ak & 0xffff)] = bv[E[2 * e + 1]];
      }
   }
   __syncthreads();
   // a wave per pair, round-robin; the lanes along the border rows of C.  The segment's rows of C's panel are staged in the wave's own piece of
   // LDS (coalesced reads, then wave-uniform LDS reads: straight-line code, every load of a pair in flight at once - a first version read them
   // with uniform global loads behind a branch per column and waited for each)
   double* dlw = dl + wave * (WMAX * WMAX);
   short* kqw = kq + wave * WMAX;
   for (int p0 = 0; p0 < n_pairs; p0 += BR_PCHUNK) {
      if (p0 > 0) {   // (fronts with more pairs than one chunk of descriptors: the next chunk)
         __syncthreads();
         for (int p = p0 + tid; p < n_pairs && p < p0 + BR_PCHUNK; p += BLOCK) {
            const int seg = H[3 + 2 * p], b0 = seg & 0xffff;
            const SnDesc C = sns[H[2 + 2 * p]];
            pd[p - p0] = BrPair{C.panel + C.w + b0, C.bb, C.rows, C.w, C.r - C.rb, C.ld, (seg >> 16) - b0, b0, C.rb};
         }
         __syncthreads();
      }
      const int pn = n_pairs - p0 < BR_PCHUNK ? n_pairs - p0 : BR_PCHUNK;
      for (int p = wave; p < pn; p += NW) {
         const BrPair P = pd[p];
         const int wC = P.wC, nbc = P.nbc, rpC = (nbc + 3) & ~3, ns = P.ns, w4 = (wC + 3) & ~3;
         const double* PC = fin + P.panel;
         const double* LbC = fin + P.bb;
         for (int idx = lane; idx < ns * w4; idx += 64) {
            const int c = idx / ns, qi = idx - c * ns;
            dlw[qi * WMAX + c] = c < wC ? PC[qi + (long long)c * P.ldC] : 0.0;
         }
         if (lane < ns) kqw[lane] = (short)(rowidx[P.rows + P.b0 + lane] - J.c0);
         __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
         __builtin_amdgcn_wave_barrier();
         for (int t = lane; t < nbc; t += 64) {
            const int aj = map[rowidx[P.rows + P.pad + t] - bn];      // (P.pad: rb of C.  Every border row of C is one of J's: C has a row in J's columns)
            double lbd[WMAX];
#pragma unroll
            for (int c = 0; c < WMAX; ++c) {                          // (clamped index: unconditional loads, all in flight; the pivots follow the rows in the border-row arena)
               const int cc = c < wC ? c : wC - 1;
               const double v = LbC[cc * rpC + t] * LbC[wC * rpC + cc];
               lbd[c] = c < wC ? v : 0.0;
            }
            if (aj < 0) continue;
            for (int qi = 0; qi < ns; ++qi) {
               const double* dq = dlw + qi * WMAX;
               double sum = 0.0;
#pragma unroll
               for (int c = 0; c < 4; ++c) sum += lbd[c] * dq[c];
               if (wC > 4) {
#pragma unroll
                  for (int c = 4; c < 8; ++c) sum += lbd[c] * dq[c];
               }
               if (wC > 8) {
#pragma unroll
                  for (int c = 8; c < 12; ++c) sum += lbd[c] * dq[c];
               }
               if (wC > 12) {
#pragma unroll
                  for (int c = 12; c < 16; ++c) sum += lbd[c] * dq[c];
               }
               lds_add(X + kqw[qi] * rp + aj, -sum);
            }
         }
         __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
         __builtin_amdgcn_wave_barrier();
      }
   }
   __syncthreads();
   // Y = X L_JJ^-T (unit lower L_JJ): Y(:, k) = X(:, k) - sum_{j < k} Y(:, j) L_JJ(k, j);  L_b(:, k) = Y(:, k) / d_k
   // (in place in LDS, every thread on its own rows: no array of w values in registers, the compiler would hoist all of L_JJ beside it)
   double* Q = arena + J.bb;
   for (int a = tid; a < nbj; a += BLOCK) {
      Q[a] = X[a] / dj[0];
      for (int k = 1; k < w; ++k) {
         double v = X[k * rp + a];
         for (int j = 0; j < k; ++j) v -= X[j * rp + a] * ljj[k * WMAX + j];
         X[k * rp + a] = v;
         Q[k * rp + a] = v / dj[k];
      }
   }
}

// ... and the same contribution for the border rows of the DENSE TAIL (the tail panel's rows m_pad .. of the columns the supernode C reaches):
//    T(border row a, tail column of row q) -= sum_c L_b(C)(a, c) d_C(c) L_C(q, c)       q in [q0, q1): the tail rows of C
// for every supernode C with border rows and tail rows - with whole fronts this part of the update matrices reached the tail through
// k_root_assemble.  One wave per (C, chunk of its tail rows); targets are shared between supernodes: FP64 atomics (the mode is not taken in
// deterministic mode).  list: (C, q0 | q1 << 16) pairs.
template <int WMAX>
__global__ __launch_bounds__(64) void k_border_tail(const int* __restrict__ list, const SnDesc* __restrict__ sns, const BlkDesc* __restrict__ blks,
                                                    const int* __restrict__ rowidx, double* __restrict__ arena) {
   __shared__ double dl[WMAX * WMAX];
   __shared__ int tcol[WMAX];
   const int lane = threadIdx.x;
   const SnDesc C = sns[list[2 * blockIdx.x]];
   const BlkDesc bd = blks[C.blk];
   const int seg = list[2 * blockIdx.x + 1], q0 = seg & 0xffff, q1 = seg >> 16, ns = q1 - q0;   // ns <= WMAX (the host cuts longer runs)
   const int wC = C.w, nbc = C.r - C.rb, rpC = (nbc + 3) & ~3, ldC = C.ld;
   const double* PC = arena + C.panel;
   for (int idx = lane; idx < ns * wC; idx += 64) {
      const int qi = idx / wC, c = idx - qi * wC;
      dl[qi * WMAX + c] = PC[c + (long long)c * ldC] * PC[(wC + q0 + qi) + (long long)c * ldC];
   }
   if (lane < ns) tcol[lane] = rowidx[C.rows + q0 + lane] - bd.n_head;
   __syncthreads();
   const double* LbC = arena + C.bb;
   double* T = arena + bd.T_in;
   for (int a = lane; a < nbc; a += 64) {
      double lb[WMAX];
#pragma unroll
      for (int c = 0; c < WMAX; ++c) lb[c] = c < wC ? LbC[c * rpC + a] : 0.0;
      const int tr = bd.m_pad + (rowidx[C.rows + C.rb + a] - bd.n);
      for (int qi = 0; qi < ns; ++qi) {
         double sum = 0.0;
#pragma unroll
         for (int c = 0; c < WMAX; ++c) sum += c < wC ? lb[c] * dl[qi * WMAX + c] : 0.0;
         atomic_add_f64(T + tr + (long long)tcol[qi] * bd.ldT, -sum);
      }
   }
}

// Border split (BlockSym::mf_split): the border x border part of a block's Schur contribution, formed from the finished head panels,
//    C(i, j) = - sum_J sum_k L_J(i, k) d_k L_J(j, k),   i >= j compressed border ids,
// over the head supernodes J of the block that hold border rows and take part in the multifrontal scheme (fronts, simple leaves
// below a front) - what the update matrices would otherwise carry from front to front up to the root fronts.
// The border rows of those supernodes are kept a second time, in the layout this kernel stages (k_front / k_head_factor_simple write
// them as they write the panel): per supernode Lt[k * rp + a] (rp = nbj rounded up to 4, the padding stays zero from the analysis) and
// its w pivots, supernode after supernode - a BATCH of up to BB_GMAX supernodes is one contiguous piece of that arena.
// One workgroup per (block, share of the block's batches): C (packed lower triangle, nb (nb + 1) / 2 doubles) stays in LDS for the
// whole walk; per batch the piece is copied to LDS (requested into registers one batch ahead), every thread forms 4 x 4 tiles of
// L_b D L_b^T of the batch's supernodes in registers and adds them at the rows' border positions with LDS atomics (ds_add_f64: no
// return value, no wait).  ordered != 0 (deterministic mode): the supernodes of a batch one after the other with a barrier between
// them - inside a supernode the targets are distinct, so every sum has a fixed order.  At the end C is added to the Schur complement:
// FP64 atomics (targets shared between blocks and shares), or - gbuf != nullptr - plain adds into the group's buffer by launches that
// hold at most one block of every group.
constexpr int BB_GMAX = 8;
// register tile of k_border_schur: BB_TR rows x 4 columns of L_b D L_b^T per thread and step (BB_TR = 4: square tiles over the lower triangle;
// 8: two row groups per tile - six LDS reads per 32 multiply-adds instead of four per 16)
constexpr int BB_TR = 4;
// tiles of a supernode with rp (a multiple of 4) padded border rows; tile t -> (column group tb of 4, row group ta of BB_TR)
__host__ __device__ inline int bb_tile_count(int rp) {
   const int nt4 = rp >> 2;
   if (BB_TR == 4) return nt4 * (nt4 + 1) / 2;
   const int nt8 = (rp + 7) >> 3;
   int cnt = 0;
   for (int tb = 0; tb < nt4; ++tb) cnt += nt8 - (tb >> 1);
   return cnt;
}
struct BbMeta { int lt_off, pos_off, w, nbj, tile0, pad0, pad1, pad2; };   // staging offset of Lt (doubles; the pivots follow at + w * rp),
                                                                          // offset of the rows' positions inside the batch's list, tiles before it
struct BbBatch {
   long long src;     // offset of the batch inside the border-row arena
   long long pos;     // offset of its rows' positions (compressed border ids) inside bbpos
   int first, cnt;    // its supernodes inside the BbMeta array
   int ndoubles, ntiles, npos, pad;
};

template <int BLOCK, int NPF>
__global__ __launch_bounds__(BLOCK) void k_border_schur(const int* __restrict__ blk_list, const int* __restrict__ batch_off,
                                                       const BbBatch* __restrict__ batches, const BbMeta* __restrict__ metas,
                                                       const int* __restrict__ bbpos, const BlkDesc* __restrict__ blks,
                                                       const int* __restrict__ bmap, const double* __restrict__ bbarena,
                                                       double* __restrict__ SC, int ldSC, const int* __restrict__ sctab,
                                                       double* __restrict__ gbuf, long long gstride, const int* __restrict__ blk_group,
                                                       int stg_doubles, int pos_cap, int ordered, double* __restrict__ blk_out = nullptr,
                                                       long long blk_stride = 0, int c_cap = 0) {
   extern __shared__ __attribute__((aligned(16))) double bs_C[];
   typedef double double2_t __attribute__((ext_vector_type(2)));
   const int blk = blk_list ? blk_list[blockIdx.x] : blockIdx.x;
   const BlkDesc bd = blks[blk];
   const int nb = bd.nb, tid = threadIdx.x;
   const int q_begin = batch_off[blk], q_end = batch_off[blk + 1];
   double* S_ = gbuf ? gbuf + gstride * blk_group[blk] : SC;
   if (q_begin >= q_end || (!S_ && !blk_out)) return;
   // gridDim.z workgroups share the triangle by column ranges of (about) equal area: part z keeps the packed columns [jlo, jhi) in its LDS
   // and forms only the tiles that reach them - half the LDS lets two workgroups share a compute unit, which is what hides the barriers
   // and the LDS latency of the walk (one workgroup per compute unit: waves 65 % waiting)
   const int tri = nb * (nb + 1) / 2;
   auto coff = [nb](int j) { return j * nb - j * (j - 1) / 2; };
   int jlo = 0, jhi = nb;
   if (gridDim.z > 1) {
      const int want_lo = (int)((long long)tri * blockIdx.z / gridDim.z), want_hi = (int)((long long)tri * (blockIdx.z + 1) / gridDim.z);
      jlo = blockIdx.z == 0 ? 0 : packed_col(want_lo, nb);
      jhi = blockIdx.z + 1 == gridDim.z ? nb : packed_col(want_hi, nb);
   }
   const int cbase = coff(jlo), ncp = c_cap > 0 ? c_cap : ((tri + 1) & ~1);
   double* C = bs_C;
   double* stage = C + ncp;                              // stg_doubles
   int* spos = (int*)(stage + stg_doubles);              // pos_cap
   BbMeta* smeta = (BbMeta*)(spos + ((pos_cap + 3) & ~3));   // BB_GMAX
   for (int idx = tid; idx < ncp; idx += BLOCK) C[idx] = 0.0;
   if (coff(jhi) - cbase > ncp) return;   // (the host sized the slice for the largest block: cannot happen)
   constexpr int NPP = 4;   // positions per thread (pos_cap <= NPP * BLOCK, checked by the host)
   double2_t pv[NPF];
   int pp[NPP];
   auto request = [&](const BbBatch& bt) {
      const double2_t* src = (const double2_t*)(bbarena + bt.src);
      const int n2 = bt.ndoubles >> 1;
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
         const int e = tid + u * BLOCK;
         if (e < n2) pv[u] = src[e];
      }
      const int* ps = bbpos + bt.pos;
#pragma unroll
      for (int u = 0; u < NPP; ++u) {
         const int e = tid + u * BLOCK;
         if (e < bt.npos) pp[u] = ps[e];
      }
   };
   int q = q_begin + blockIdx.y;
   if (q < q_end) request(batches[q]);
   for (; q < q_end; q += gridDim.y) {
      const BbBatch bt = batches[q];
      __syncthreads();   // the tiles of the previous batch are done with the staging area (first pass: C is zeroed)
      {
         double2_t* st2 = (double2_t*)stage;
         const int n2 = bt.ndoubles >> 1;
#pragma unroll
         for (int u = 0; u < NPF; ++u) {
            const int e = tid + u * BLOCK;
            if (e < n2) st2[e] = pv[u];
         }
#pragma unroll
         for (int u = 0; u < NPP; ++u) {
            const int e = tid + u * BLOCK;
            if (e < bt.npos) spos[e] = pp[u];
         }
         if (tid < bt.cnt * 8) ((int*)smeta)[tid] = ((const int*)(metas + bt.first))[tid];
      }
      __syncthreads();
      if (q + (int)gridDim.y < q_end) request(batches[q + gridDim.y]);
      auto do_tile = [&](const BbMeta& m, int t) {
         const int w = m.w, nbj = m.nbj, rp = (nbj + 3) & ~3, nt = rp >> 2;
         const double* Lt = stage + m.lt_off;
         const double* dk = Lt + w * rp;
         const int* pos = spos + m.pos_off;
         int tb, a0;
         if (BB_TR == 4) {
            tb = packed_col(t, nt);
            a0 = 4 * (tb + (t - (tb * nt - tb * (tb - 1) / 2)));
         } else {   // column groups 2 j and 2 j + 1 both start at row group j: nt8 - j tiles each
            const int nt8 = (rp + 7) >> 3;
            int j = packed_col(t >> 1, nt8);
            while (j > 0 && 2 * (j * nt8 - j * (j - 1) / 2) > t) --j;
            while (j + 1 < nt8 && 2 * ((j + 1) * nt8 - (j + 1) * j / 2) <= t) ++j;
            const int rem = t - 2 * (j * nt8 - j * (j - 1) / 2), per = nt8 - j;
            tb = 2 * j + (rem >= per ? 1 : 0);
            a0 = 8 * (j + (rem >= per ? rem - per : rem));
         }
         const int b0 = 4 * tb;
         if (pos[b0] >= jhi || pos[min(b0 + 3, nbj - 1)] < jlo) return;   // none of the tile's columns belongs to this part (positions ascend)
         double acc[BB_TR][4];
#pragma unroll
         for (int x = 0; x < BB_TR; ++x)
#pragma unroll
            for (int z = 0; z < 4; ++z) acc[x][z] = 0.0;
#pragma unroll 4
         for (int k = 0; k < w; ++k) {
            const double2_t* ca = (const double2_t*)(Lt + k * rp + a0);
            const double2_t* cb = (const double2_t*)(Lt + k * rp + b0);
            double la[BB_TR];
#pragma unroll
            for (int x = 0; x < BB_TR; x += 2) { const double2_t v = ca[x >> 1]; la[x] = v.x; la[x + 1] = v.y; }   // (rows beyond rp: the next column's first entries - their sums are never stored)
            const double2_t b01 = cb[0], b23 = cb[1];
            const double d = dk[k];
            const double lb[4] = {b01.x * d, b01.y * d, b23.x * d, b23.y * d};
#pragma unroll
            for (int x = 0; x < BB_TR; ++x)
#pragma unroll
               for (int z = 0; z < 4; ++z) acc[x][z] += la[x] * lb[z];
         }
#pragma unroll
         for (int z = 0; z < 4; ++z) {
            const int b = b0 + z;
            if (b < nbj && pos[b] >= jlo && pos[b] < jhi) {
               const int j = pos[b];
               double* cj = C + (j * nb - j * (j - 1) / 2) - j - cbase;
#pragma unroll
               for (int x = 0; x < BB_TR; ++x) {
                  const int a = a0 + x;
                  if (a < nbj && a >= b) lds_add(cj + pos[a], -acc[x][z]);
               }
            }
         }
      };
      if (!ordered) {
         for (int t = tid; t < bt.ntiles; t += BLOCK) {
            int g = 0;
#pragma unroll
            for (int h = 1; h < BB_GMAX; ++h) g += (h < bt.cnt && smeta[h].tile0 <= t) ? 1 : 0;
            do_tile(smeta[g], t - smeta[g].tile0);
         }
      } else {
         for (int g = 0; g < bt.cnt; ++g) {
            const int t_end = (g + 1 < bt.cnt ? smeta[g + 1].tile0 : bt.ntiles) - smeta[g].tile0;
            for (int t = tid; t < t_end; t += BLOCK) do_tile(smeta[g], t);
            __syncthreads();
         }
      }
   }
   __syncthreads();
   const int n_mine = coff(jhi) - cbase;
   if (blk_out) {   // deterministic mode: the block's triangle as it stands; k_border_schur_add puts the blocks of a group in order
      double* out = blk_out + blk_stride * blk + cbase;
      for (int idx = tid; idx < n_mine; idx += BLOCK) out[idx] = C[idx];
      return;
   }
   const int* bm = bmap + bd.bmap_off;
   for (int li = tid; li < n_mine; li += BLOCK) {
      const double v = C[li];
      if (v == 0.0) continue;   // (never touched)
      const int idx = li + cbase;
      const int j = packed_col(idx, nb), i = j + (idx - (j * nb - j * (j - 1) / 2));
      double* tgt = sc_entry(S_, ldSC, bm, sctab, bd.sctab_off, nb, i, j);
      if (gbuf) *tgt += v; else atomic_add_f64(tgt, v);
   }
}

// deterministic mode: the triangles k_border_schur left per block go into their group's buffer - a launch holds at most one block of
// every group (plain adds), the launches follow the blocks' order inside the groups
__global__ __launch_bounds__(256) void k_border_schur_add(const int* __restrict__ blk_list, const BlkDesc* __restrict__ blks, const int* __restrict__ bmap,
                                                         const double* __restrict__ blk_out, long long blk_stride, int ldSC,
                                                         const int* __restrict__ sctab, double* __restrict__ gbuf, long long gstride,
                                                         const int* __restrict__ blk_group) {
   const int blk = blk_list[blockIdx.x];
   const BlkDesc bd = blks[blk];
   const int nb = bd.nb;
   const double* in = blk_out + blk_stride * blk;
   double* S_ = gbuf + gstride * blk_group[blk];
   const int* bm = bmap + bd.bmap_off;
   for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < nb * (nb + 1) / 2; idx += gridDim.y * blockDim.x) {
      const double v = in[idx];
      if (v == 0.0) continue;
      const int j = packed_col(idx, nb), i = j + (idx - (j * nb - j * (j - 1) / 2));
      *sc_entry(S_, ldSC, bm, sctab, bd.sctab_off, nb, i, j) += v;
   }
}

// Position of row `ra` of a supernode's row list in the block's work vector: rows of K_i at ra, border rows (ra >= n, only
// touched by the border-backward sweep, Engine::solve_border_backward) behind the padded tail.
__device__ __forceinline__ int xw_row(const BlkDesc& bd, int ra) { return ra < bd.n ? ra : bd.n_head + bd.m_pad + (ra - bd.n); }



// forward / backward substitution for the simple leaves: y = b_c (unit pivot block); b[rows] -= l y   /   x_c = z_c - l^T x[rows]
// Forward substitution of all simple leaves at once, from the side of the rows they update: row rows[t] of the work vector loses
// sum_p val[p] * x[src[p]] over the leaves p in [ptr[t], ptr[t + 1]) that have an entry in it (ascending leaves: a fixed order).
// The leaves' own entries are final on entry (no column below them), and no leaf column is a target - reads and writes are disjoint.
__global__ __launch_bounds__(256) void k_leaf_fwd_gather(const int* __restrict__ rows, const int* __restrict__ ptr, const int* __restrict__ src,
                                                         const double* __restrict__ val, double* __restrict__ xw, long long xw_stride, int n) {
   // a tile of 256 target rows per workgroup: the tile's entries (consecutive in val / src) stream through LDS, all threads side by side, then a
   // thread per row adds its segment - one chain of dependent loads (pointers -> entries -> x) per 256 rows.  (Eight lanes per row, 32 rows per
   // workgroup: 0.39 ms per sweep on the configs[3] shape with the waves waiting 93 % of their cycles.)
   constexpr int LCH = 4096;
   __shared__ double prod[LCH];
   const int tid = threadIdx.x, t0 = blockIdx.x * 256, t = t0 + tid;
   double* x = xw + xw_stride * blockIdx.y;
   const bool row = t < n;
   const int p0 = row ? ptr[t] : 0, p1 = row ? ptr[t + 1] : 0;
   const int target = row ? rows[t] : 0;
   const int p_lo = ptr[t0], p_hi = ptr[min(t0 + 256, n)];
   double s = 0.0;
   for (int c0 = p_lo; c0 < p_hi; c0 += LCH) {
      const int c1 = min(c0 + LCH, p_hi);
      for (int q = c0 + tid; q < c1; q += 256) prod[q - c0] = val[q] * x[src[q]];
      __syncthreads();
      for (int q = max(p0, c0); q < min(p1, c1); ++q) s += prod[q - c0];
      __syncthreads();
   }
   if (row) x[target] -= s;
}

// Backward substitution of the simple leaves from a compact record (24 bytes instead of the 88-byte SnDesc + BlkDesc the general
// kernel reads - on the time-coupled blocks the descriptors were most of this kernel's traffic): x_c = x_c / d - sum_a l_a x[rows_a]
struct LeafDesc {
   long long panel;   // d, l_0 .. l_{r-1} in the arena
   int rows;          // offset into rowidx
   int xoff;          // the block's offset in the work vector
   int c0;            // the leaf's column (block-local, permuted)
   int r_in;          // rows inside the block (the border rows behind them take no part in solves with K_i)
};

__global__ __launch_bounds__(256) void k_leaf_bwd(const LeafDesc* __restrict__ leaves, int cnt, const int* __restrict__ rowidx,
                                                  const double* __restrict__ arena, double* __restrict__ xw, long long xw_stride, int dscale) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= cnt) return;
   const LeafDesc lf = leaves[t];
   const double* P = arena + lf.panel;
   const int* rows = rowidx + lf.rows;
   double* xb = xw + xw_stride * blockIdx.y + lf.xoff;
   // four entries at a time, their loads issued together (a leaf has up to SIMPLE_RMAX rows; one entry per trip made every trip wait for
   // rows[a] and then for x[rows[a]]: the waves waited 89 % of their cycles)
   const double xc = xb[lf.c0], d = dscale ? P[0] : 1.0;
   double s = 0.0;
   for (int a0 = 0; a0 < lf.r_in; a0 += 4) {
      int idx[4];
      double pv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const bool in = a0 + u < lf.r_in;
         idx[u] = in ? rows[a0 + u] : lf.c0;
         pv[u] = in ? P[1 + a0 + u] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = xb[idx[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += pv[u] * xv[u];
   }
   xb[lf.c0] = (dscale ? xc / d : xc) - s;
}

__global__ __launch_bounds__(256) void k_head_solve_simple(const SnDesc* __restrict__ sns, int sn_begin, int cnt,
                                                          const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                          const double* __restrict__ arena, double* __restrict__ xw,
                                                          long long xw_stride, int backward,
                                                          ScatterCtx sx = ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}, int border = 0, int mf = 0,
                                                          int dscale = 0) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= cnt) return;
   const SnDesc sn = sns[sn_begin + t];
   if (mf && sn.n_useg > 0) return;   // multifrontal solves: the front above this leaf does its part (k_front_fwd / k_front_bwd)
   const BlkDesc bd = blks[sn.blk];
   const double* P = arena + sn.panel;
   const int* rows = rowidx + sn.rows;
   double* xb = xw + xw_stride * blockIdx.y + bd.xw_off;
   if (!backward) {
      const double y = xb[sn.c0];
      for (int a = 0; a < sn.r; ++a) {
         const int ra = rows[a];
         if (ra >= bd.n) break;
         scatter_add(sx, xb + ra, false, sn.vslot + a, -P[1 + a] * y);
      }
   } else {
      double s = 0.0;
      for (int a = 0; a < sn.r; ++a) {
         const int ra = rows[a];
         if (ra >= bd.n && !border) break;
         s += P[1 + a] * xb[xw_row(bd, ra)];
      }
      xb[sn.c0] = (dscale ? xb[sn.c0] / P[0] : xb[sn.c0]) - s;
   }
}

// ------------------------------------------------------------------------------------------------
// tile GEMM on the FP64 matrix cores.
//   MODE 0 (update): C(ti,tj) -= L(ti,0:K) U(tj,0:K)^T             K = tj*TILE, L/C tiles of the tail panel, U = L D its
//                    scaled copy (tail rows only): the diagonal scaling costs no v_mul_f64 in the inner loop - FP64 VALU
//                    instructions share the matrix pipe's issue slots, four per 32 MFMAs were 3 % of the kernel
//   MODE 1 (trsm)  : C(ti,tj)  = C(ti,tj) Winv(tj)^T               K = TILE; tail rows also store U(ti,tj) = C(ti,tj) D(tj)
//   MODE 2 (schur) : SC[bmap(ti), bmap(tj)] -= A(ti,0:K) diag(d) B(tj,0:K)^T   ti,tj border tile rows, K = m_pad
//   MODE 3 / 4     : MODE 0 under names of their own - 3 the dense root, 4 the diagonal tiles of a leaf column that are
//                    updated ahead of the rest (so that profiles keep the leaf update kernel apart)
// 512 threads = 8 waves in a 2 x 4 grid, each wave owns a 64 x 32 sub-tile (32 accumulators): ~110 VGPRs, so four
// waves share a SIMD and hide each other's LDS / barrier / DMA-issue stalls (two 256-thread workgroups per CU).
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction).  Measured on MI355X
// (tools/mb2.hip, profiles/r1_fp64_issue_rates.txt) it sustains 72.6 TFLOP/s (17 cycles/instruction/SIMD) whereas
// v_mfma_f64_16x16x4_f64 tops out at 47 TFLOP/s (106-138 cycles), so the 4x4x4 form is the FP64 roofline path.
// Lane maps (probed with tools/probe44.hip):  A lane = 16k + 4b + i,  B lane = 16k + 4b + j,  D lane = 16i + 4b + j.
// We put C rows on (b,j) — 16 consecutive rows, a different row per block — and 4 C columns on i, so one instruction
// produces a 16 x 4 sub-tile: the row-panel fragment is the plain [k][row] LDS image (one ds_read_b64, 16 rows x 4 k),
// the column-panel fragment is 4 columns x 4 k broadcast to the four blocks (LDS broadcast read, no conflict).
// ------------------------------------------------------------------------------------------------
// 16-byte-per-lane LDS-DMA: the wave writes 1 KiB contiguously at lds_base (wave-uniform) + 16 * lane
// Written as inline assembly, not with __builtin_amdgcn_global_load_lds: the compiler cannot prove that the LDS image the
// DMA fills is disjoint from the one the MFMA fragments are read from and puts an s_waitcnt vmcnt(0) in front of the next
// ds_read, i.e. every wave sat out the full global-memory latency in the middle of every stage.  The asm form is invisible
// to that bookkeeping; the wait is placed by hand (dma_wait) right before the barrier that hands the buffer over.
__device__ __forceinline__ void glds16(const double* gptr_lane, double* lds_base) {
   const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_base;
   asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr_lane), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Staging: both panels are column-major with the tile's 128 rows contiguous, so one LDS-DMA wave-instruction moves one
// k-column (1 KiB) straight into the [k][row] LDS image (no VGPR round trip, no ds_write).  Two LDS buffers: the DMA of
// stage s+1 is in flight while stage s is multiplied; one barrier per stage.  The diagonal scaling d_k of the update is
// applied to the column-panel fragment after the LDS read (one v_mul_f64 per fragment, hidden beside the MFMAs).
struct GemmShared {
   __attribute__((aligned(16))) double As[2][KB * LDSW];
   __attribute__((aligned(16))) double Bs[2][KB * LDSW];
   __attribute__((aligned(16))) double Ds[2][KB];   // diagonal scaling d_k of the stage
};

// one tile task (index tix of the list); every thread of the workgroup calls it with the same tix
template <int MODE>
__device__ __forceinline__ void tile_gemm_body(int tix, GemmShared& sh, const TileTask* __restrict__ tasks, int n_tasks,
                                               const BlkDesc* __restrict__ blks, double* __restrict__ arena,
                                               const double* __restrict__ dtail, const double* __restrict__ winv,
                                               const int* __restrict__ bmap, double* __restrict__ SC, int ldSC,
                                               const int* __restrict__ sctab, double* __restrict__ uarena,
                                               double* __restrict__ gbuf = nullptr, long long gstride = 0,
                                               const int* __restrict__ blk_group = nullptr) {
   constexpr bool SCALE = (MODE == 2);   // in-loop diagonal scaling: only the Schur SYRK (border rows have no U copy)
   auto& As = sh.As;
   auto& Bs = sh.Bs;
   auto& Ds = sh.Ds;
   if (tix >= n_tasks) return;
   const TileTask task = tasks[tix];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int wr = wave & 1, wc = wave >> 1;   // 8 waves: 2 (rows) x 4 (cols), each 64 rows x 32 columns
   const int ld = bd.ldT;
   double* T = arena + bd.T;

   int K;
   const double* Ap;  // A panel: rows of C
   const double* Bp;  // B panel: columns of C
   long long ldb;
   const double* dv = nullptr;
   if (MODE == 0 || MODE >= 3) {
      // K range in tile columns [k0, k1): pad = k0 | k1 << 16, k1 == 0 meaning "up to the tile's own column"
      const int k0 = task.pad & 0xffff, k1 = (task.pad >> 16) ? (task.pad >> 16) : task.tj;
      K = (k1 - k0) * TILE;
      Ap = T + (long long)task.ti * TILE + (long long)k0 * TILE * ld;
      Bp = uarena + bd.U + (long long)task.tj * TILE + (long long)k0 * TILE * bd.m_pad;
      ldb = bd.m_pad;
   } else if (MODE == 1) {
      K = TILE;
      Ap = T + (long long)task.ti * TILE + (long long)task.tj * TILE * ld;
      Bp = winv + bd.winv_off + (long long)task.tj * TILE * TILE;
      ldb = TILE;
   } else {
      K = bd.m_pad;
      Ap = T + bd.m_pad + (long long)task.ti * TILE;
      Bp = T + bd.m_pad + (long long)task.tj * TILE;
      ldb = ld;
      dv = dtail + bd.dt_off;
   }

   double acc[4][8];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[i][c] = 0.0;

   const int nst = K / KB;
   const double* Al = Ap + 2 * lane;
   const double* Bl = Bp + 2 * lane;
   auto issue = [&](int st, int buf) {
#pragma unroll
      for (int q = 0; q < KB / 8; ++q) {
         const int k = wave + 8 * q;   // this wave's k-columns of the stage (two at KB = 16)
         glds16(Al + (long long)(st * KB + k) * ld, &As[buf][k * LDSW]);
         glds16(Bl + (long long)(st * KB + k) * ldb, &Bs[buf][k * LDSW]);
      }
   };
   // the KB scaling factors of a stage: one more LDS-DMA, KB/2 lanes of the last wave (16 bytes each)
   auto load_d = [&](int st, int buf) {
      if (SCALE && wave == 7 && lane < KB / 2) glds16(dv + st * KB + 2 * lane, &Ds[buf][0]);
   };
   if (nst > 0) {
      issue(0, 0);
      load_d(0, 0);
   }
   const int rlane = wr * 64 + (lane & 15);   // row-panel fragment offset
   const int clane = wc * 32 + (lane & 3);    // column-panel fragment offset (broadcast over the 4 blocks)
   for (int st = 0; st < nst; ++st) {
      const int buf = st & 1;
      dma_wait();        // own DMA of this stage retired
      __syncthreads();   // everybody's DMA of this stage visible + buffer buf^1 free again
      const double* Ab = As[buf] + (lane >> 4) * LDSW + rlane;
      const double* Bb = Bs[buf] + (lane >> 4) * LDSW + clane;
#pragma unroll
      for (int q = 0; q < KB / 4; ++q) {
         // the DMA of the next stage is issued after the first quarter of this stage's MFMAs, not right behind the barrier: the
         // matrix pipe is already busy when the address arithmetic and the four LDS-DMA instructions go out (+1.5 %; issuing
         // later still, or one instruction per quarter, loses 6-8 %: profiles/r1_fp64_issue_rates.txt)
         if (q == KB / 16 && st + 1 < nst) { issue(st + 1, buf ^ 1); load_d(st + 1, buf ^ 1); }
         double fr[4], fc[8];
#pragma unroll
         for (int i = 0; i < 4; ++i) fr[i] = Ab[(4 * q) * LDSW + i * 16];
#pragma unroll
         for (int c = 0; c < 8; ++c) fc[c] = Bb[(4 * q) * LDSW + c * 4];
         if (SCALE) {
            const double dq = Ds[buf][4 * q + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 4; ++i) fr[i] *= dq;   // A diag(d) B^T: scale the 4 row fragments, not the column ones
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c)
               acc[i][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], fr[i], acc[i][c], 0, 0, 0);
      }
   }

   // epilogue: lane holds C(row = wr*64 + 16 i + (lane&15), col = wc*32 + 4 c + (lane>>4))
   if (MODE == 0 || MODE >= 3) {
      // read-modify-write of the C tile: all 32 loads go out before the first store (written as `*cp -= v` per element the
      // compiler serialises load -> wait -> store 32 times, a global round trip each)
      double* c0 = T + (long long)task.ti * TILE + wr * 64 + (lane & 15) + ((long long)task.tj * TILE + wc * 32 + (lane >> 4)) * ld;
#pragma unroll
      for (int h = 0; h < 4; ++h) {   // 8 elements at a time: the accumulators leave 40-odd registers
         double cv[4][2];
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) cv[i][c] = c0[i * 16 + (long long)((2 * h + c) * 4) * ld];
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) c0[i * 16 + (long long)((2 * h + c) * 4) * ld] = cv[i][c] - acc[i][2 * h + c];
         __builtin_amdgcn_sched_barrier(0);
      }
      return;
   }
   if (MODE == 1) {
      // L tile, and for tail rows its scaled copy U = L D (B operand of the updates); the eight d_j a lane needs are
      // fetched in one go before the stores
      const int col0 = task.tj * TILE + wc * 32 + (lane >> 4), row0 = task.ti * TILE + wr * 64 + (lane & 15);
      const bool tail_row = task.ti < bd.ntc;
      double dsc[8];
      if (tail_row) {
#pragma unroll
         for (int c = 0; c < 8; ++c) dsc[c] = dtail[bd.dt_off + col0 + 4 * c];
      }
      double* c0 = T + row0 + (long long)col0 * ld;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int c = 0; c < 8; ++c) c0[i * 16 + (long long)(c * 4) * ld] = acc[i][c];
      if (tail_row) {
         double* u0 = uarena + bd.U + row0 + (long long)col0 * bd.m_pad;
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) u0[i * 16 + (long long)(c * 4) * bd.m_pad] = acc[i][c] * dsc[c];
      }
      return;
   }
#pragma unroll
   for (int i = 0; i < 4; ++i) {
      const int row = wr * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
         const int col = wc * 32 + c * 4 + (lane >> 4);
         const double v = acc[i][c];
         {
            const int gi = task.ti * TILE + row, gj = task.tj * TILE + col;
            if (gi < bd.nb && gj < bd.nb && gi >= gj) {
               const int* bm = bmap + bd.bmap_off;
               if (gbuf)   // deterministic mode: the block's group buffer, which no other workgroup of this launch touches
                  *sc_entry(gbuf + blk_group[task.blk] * gstride, ldSC, bm, sctab, bd.sctab_off, bd.nb, gi, gj) -= v;
               else
                  atomic_add_f64(sc_entry(SC, ldSC, bm, sctab, bd.sctab_off, bd.nb, gi, gj), -v);
            }
         }
      }
   }
}

// one workgroup per task.  XCD-aware task order: workgroups w and w+8 share an XCD (round-robin dispatch), so each XCD gets a
// contiguous slice of the task list: tasks of one block (which share the B panel) then meet in one L2.
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_tile_gemm(const TileTask* __restrict__ tasks, int n_tasks,
                                                     const BlkDesc* __restrict__ blks, double* __restrict__ arena,
                                                     const double* __restrict__ dtail, const double* __restrict__ winv,
                                                     const int* __restrict__ bmap, double* __restrict__ SC, int ldSC,
                                                     const int* __restrict__ sctab = nullptr, double* __restrict__ uarena = nullptr,
                                                     double* __restrict__ gbuf = nullptr, long long gstride = 0,
                                                     const int* __restrict__ blk_group = nullptr) {
   __shared__ GemmShared sh;
   const int per = (n_tasks + 7) >> 3;
   const int tix = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
   tile_gemm_body<MODE>(tix, sh, tasks, n_tasks, blks, arena, dtail, winv, bmap, SC, ldSC, sctab, uarena, gbuf, gstride, blk_group);
}

// One task per workgroup as above, but the task is drawn from the XCD's counter when the workgroup starts, and a workgroup whose
// XCD has run dry takes from the others.  The launch has an eighth more workgroups than tasks: the XCDs run at different
// clocks under this load (1827 .. 1920 MHz on one box: profiles/r2_gemm_experiments.txt) and with equal static shares the
// slow ones finish 5 % after the fast ones; here the fast XCD's surplus workgroups finish the slow XCDs' slices and the slow
// XCDs' surplus workgroups find nothing and leave.  Unlike the persistent variant no workgroup outlives its tile, so the
// diagonal-tile chain on the side stream still gets its slots as tiles retire.
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_tile_gemm_bal(const TileTask* __restrict__ tasks, int n_tasks,
                                                         const BlkDesc* __restrict__ blks, double* __restrict__ arena,
                                                         const double* __restrict__ dtail, const double* __restrict__ winv,
                                                         const int* __restrict__ bmap, double* __restrict__ SC, int ldSC,
                                                         const int* __restrict__ sctab, double* __restrict__ uarena,
                                                         int* __restrict__ ctr) {
   __shared__ GemmShared sh;
   __shared__ int s_tix;
   if (threadIdx.x == 0) {
      const int per = (n_tasks + 7) >> 3;
      const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20 /* XCC_ID[3:0] */) & 7);
      int got = -1;
      for (int d = 0; d < 8 && got < 0; ++d) {
         const int x = (xcc + d) & 7;
         if (__hip_atomic_load(ctr + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per) continue;   // slice known to be empty
         const int t = __hip_atomic_fetch_add(ctr + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         const int tix = x * per + t;
         if (t < per && tix < n_tasks) got = tix;
      }
      s_tix = got;
   }
   __syncthreads();
   const int tix = s_tix;
   if (tix < 0) return;
   tile_gemm_body<MODE>(tix, sh, tasks, n_tasks, blks, arena, dtail, winv, bmap, SC, ldSC, sctab, uarena);
}

// deterministic mode: SC += ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7)) over the (at most eight) group buffers, lower triangle.
// The order is a function of the global block partition only, so one rank with eight groups and two ranks with four groups each
// followed by the two-operand sum of the all-reduce give the same bits.  (Not so for 4 or 8 ranks: there the all-reduce associates
// the ranks' partial sums in its own order - reproducible run to run, but not this tree.)
__global__ void k_reduce_groups(double* __restrict__ SC, int ld, int S, const double* __restrict__ gbuf, long long gstride, int n_groups,
                                int first_slot) {
   const int c = blockIdx.y;
   for (int r = c + blockIdx.x * blockDim.x + threadIdx.x; r < S; r += gridDim.x * blockDim.x) {
      const long long off = r + (long long)c * ld;
      double p[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
         const int lg = g - first_slot;   // this rank's groups occupy the slots first_slot .. first_slot + n_groups - 1 of the global eight
         p[g] = (lg >= 0 && lg < n_groups) ? gbuf[lg * gstride + off] : 0.0;
      }
      SC[off] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
   }
}


// ------------------------------------------------------------------------------------------------
// diagonal tile: LDL^T of the 128 x 128 tile held in registers, together with Winv = D^-1 L^-1
// ------------------------------------------------------------------------------------------------

struct DiagShared {
   double colk[2][TILE];
   double xrow[2][TILE];
   double dk[TILE];
   double prs[TILE];
   int sgn[TILE];
};

// 1/d on the pivot's critical path: hardware reciprocal + two Newton steps (full double precision for the normal-range
// pivots the static rule lets through) instead of the IEEE division sequence
__device__ __forceinline__ double pivot_rcp(double d) {
   double r = __builtin_amdgcn_rcp(d);
   r = fma(fma(-d, r, 1.0), r, r);
   r = fma(fma(-d, r, 1.0), r, r);
   return r;
}

// the 16 pivots k = 16 KB .. 16 KB + 15; KB is a template parameter so that every register-array index is a constant.
// Each step is one barrier plus a dependent chain LDS -> pivot rule -> reciprocal -> FMAs, so the step is written
// branch-free: every LDS operand is fetched unconditionally and up front, masks are applied with selects (the first
// version let the compiler turn `i > k ? lds[i] : 0` into serialized conditional loads: 92 us per tile instead of 77;
// what remains is the instruction stream of one wave per SIMD, ~150 VALU/LDS instructions per pivot).
// block (a, b), b <= a, of the lower block triangle
__device__ __forceinline__ constexpr int LI(int a, int b) { return a * (a + 1) / 2 + b; }

template <int KB>
__device__ __forceinline__ void diag_block(double (&A)[36], double (&X)[36], DiagShared& sh, const BlkDesc& bd,
                                           int tx, int ty, int tid, int gk0, int3& cnt) {
#pragma unroll 1
   for (int kt = 0; kt < 16; ++kt) {
      const int k = KB * 16 + kt, buf = kt & 1;
      if (ty == kt) {
#pragma unroll
         for (int a = KB; a < 8; ++a) sh.colk[buf][tx + 16 * a] = A[LI(a, KB)];   // A(i, k), i >= 16 KB
      }
      if (tx == kt) {
#pragma unroll
         for (int b = 0; b <= KB; ++b) sh.xrow[buf][ty + 16 * b] = X[LI(KB, b)];  // X(k, c), c < 16 (KB+1)
      }
      __syncthreads();
      const double piv = sh.colk[buf][k], pr = sh.prs[k];
      const int sg = sh.sgn[k];
      double ci[8], cj[8], xr[8];
#pragma unroll
      for (int a = KB; a < 8; ++a) ci[a] = sh.colk[buf][tx + 16 * a];
#pragma unroll
      for (int b = KB; b < 8; ++b) cj[b] = sh.colk[buf][ty + 16 * b];
#pragma unroll
      for (int b = 0; b <= KB; ++b) xr[b] = sh.xrow[buf][ty + 16 * b];
      bool pert;
      const double d = fix_pivot(piv, sg, pr, bd.thr_rel, bd.repl_rel, bd.repl_abs, pert);
      if (gk0 + k < bd.m) { cnt.z += pert; cnt.x += (!pert && d > 0); cnt.y += (!pert && !(d > 0)); }   // alike in every thread
      if (tid == 0) sh.dk[k] = d;
      const double dinv = pivot_rcp(d);
      // Row / column masks: i = tx + 16 a is below the pivot k = 16 KB + kt for every a > KB and, inside the pivot's own
      // block (a == KB), iff tx > kt - so only the a == KB (b == KB) terms need a run-time select.
      double li[8];
      li[KB] = tx > kt ? ci[KB] * dinv : 0.0;
#pragma unroll
      for (int a = KB + 1; a < 8; ++a) li[a] = ci[a] * dinv;
      // A(i,j) -= l_ik a_jk  for i >= j > k   (blocks above the diagonal are never touched)
      {
         const double ajk = ty > kt ? cj[KB] : 0.0;
#pragma unroll
         for (int a = KB; a < 8; ++a) A[LI(a, KB)] -= li[a] * ajk;
      }
#pragma unroll
      for (int b = KB + 1; b < 8; ++b) {
#pragma unroll
         for (int a = b; a < 8; ++a) A[LI(a, b)] -= li[a] * cj[b];
      }
      // X(i,c) -= l_ik X(k,c) for c < k ;  X(i,k) = -l_ik     (c = ty + 16 b < k for every b < KB)
#pragma unroll
      for (int b = 0; b < KB; ++b) {
#pragma unroll
         for (int a = KB; a < 8; ++a) X[LI(a, b)] -= li[a] * xr[b];
      }
      {
         const double xkc = ty < kt ? xr[KB] : (ty == kt ? 1.0 : 0.0);
#pragma unroll
         for (int a = KB; a < 8; ++a) X[LI(a, KB)] -= li[a] * xkc;
      }
   }
}

// One workgroup per diagonal tile, the tile held in REGISTERS: thread (tx,ty) of the 16 x 16 thread grid owns the 8 x 8
// elements A(tx+16a, ty+16b) of the 128 x 128 tile and the same elements of X = L^-1 (both lower triangular).  Per
// pivot k the owners publish column k of A and row k of X through a double-buffered LDS line, everybody applies the
// rank-1 update to its registers: one barrier per column, no LDS read-modify-write chains.  X is accumulated by
// applying each elimination to an identity:  X <- (I - l_k e_k^T) X.
__global__ __launch_bounds__(256, 2) void k_tile_diag(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                  double* __restrict__ arena, double* __restrict__ dtail,
                                                  double* __restrict__ winv, const signed char* __restrict__ psign,
                                                  const long long* __restrict__ psign_off, int* __restrict__ inertia,
                                                  const double* __restrict__ pref) {
   __shared__ DiagShared sh;
   // The 128 pivots are one dependent chain, and this workgroup shares its SIMDs with waves of the update kernel that always have
   // an MFMA ready: highest issue priority for the chain (the update loses nothing measurable, the chain no longer queues).
   __builtin_amdgcn_s_setprio(3);
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, tj = task.tj, ld = bd.ldT;
   const int tx = tid & 15, ty = tid >> 4;
   double* C = arena + bd.T + (long long)tj * TILE + (long long)tj * TILE * ld;
   if (tid < TILE) {
      sh.prs[tid] = pref[bd.xw_off + bd.n_head + tj * TILE + tid];
      sh.sgn[tid] = tj * TILE + tid < bd.m ? (int)psign[psign_off[task.blk] + bd.n_head + tj * TILE + tid] : 1;
   }
   // only the 36 blocks on and below the block diagonal exist (LI): 72 doubles per thread; the full 8 x 8 arrays made the
   // kernel a 496-register one that needs a completely empty CU, i.e. it could only start once the update launch beside it drained
   double A[36], X[36];
#pragma unroll
   for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) {
         const int i = tx + 16 * a, j = ty + 16 * b;
         A[LI(a, b)] = i >= j ? C[i + (long long)j * ld] : 0.0;
         X[LI(a, b)] = 0.0;
      }
   __syncthreads();
   const int gk0 = tj * TILE;
   int3 cnt = make_int3(0, 0, 0);   // accepted positive / negative / perturbed pivots (identical in every thread)
   diag_block<0>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<1>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<2>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<3>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<4>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<5>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<6>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   diag_block<7>(A, X, sh, bd, tx, ty, tid, gk0, cnt);
   double* dk = sh.dk;
   __syncthreads();
   // store L (unit lower, scaled), D, and Winv[n][k] = X[LI(n, k)] / d_n
   double* W = winv + bd.winv_off + (long long)tj * TILE * TILE;
#pragma unroll
   for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
         const int i = tx + 16 * a, j = ty + 16 * b;
         if (b > a) { W[i + (long long)j * TILE] = 0.0; continue; }
         const double av = A[LI(a < b ? b : a, b)], xv = X[LI(a < b ? b : a, b)];   // (the index is only evaluated for b <= a)
         if (i > j) C[i + (long long)j * ld] = av / dk[j];
         else if (i == j) C[i + (long long)j * ld] = dk[j];
         const double x = i == j ? 1.0 : (i > j ? xv : 0.0);
         W[i + (long long)j * TILE] = x / dk[i];
      }
   if (tid < TILE) dtail[bd.dt_off + tj * TILE + tid] = dk[tid];
   if (tid == 0) {
      if (cnt.x) atomicAdd(&inertia[3 * task.blk + 0], cnt.x);
      if (cnt.y) atomicAdd(&inertia[3 * task.blk + 1], cnt.y);
      if (cnt.z) atomicAdd(&inertia[3 * task.blk + 2], cnt.z);
   }
}

// ------------------------------------------------------------------------------------------------
// Diagonal tile with Bunch-Kaufman pivoting (1 x 1 and 2 x 2 pivots, search bounded to the tile): the dense root as a drop-in
// for DeSymIndefSolver (LAPACK dsytrf, DeSymIndefSolver.C:56-97), which takes whatever the host hands it - e.g. the x0 block with
// -C0^T Omega^-1 C0 folded in (sLinsysRootAug.C:1276-1294), where a static pivot order loses pivots to cancellation.
//   No interchange is ever performed.  The pivot of a step is an INDEX p (or a pair p, q) among the tile's not yet eliminated
//   ones; eliminating it from the remaining rows and columns is  A <- E A E^T  with  E = I - l e_p^T.  With G the product of all
//   E in pivot order,  G A G^T = Lambda  is diagonal in the tile's ORIGINAL index space, i.e.  A = M Lambda M^T,  M = G^-1: the
//   permutation of the textbook P L D L^T P^T is absorbed by M, which nobody needs to be triangular - the trsm multiplies with
//   Winv = Lambda^-1 G, the sweeps of the solves apply Winv and Winv^T as dense 128 x 128 tiles, exactly as with static pivots.
//   A 2 x 2 pivot block [a b; b c] is diagonalised by a plane rotation that goes into rows p, q of G, so Lambda stays diagonal
//   everywhere outside this kernel; its two eigenvalues carry the inertia the way DeSymIndefSolver::get_inertia counts a 2 x 2
//   block (:135-160: one positive, one negative).
//   Pivot choice per step (Bunch-Kaufman, alpha = (1 + sqrt 17) / 8; k = first remaining index, lambda = largest |A(i, k)| among
//   the remaining i != k, attained at r; sigma = the same for column r):  |a_kk| >= alpha lambda -> k;  |a_kk| sigma >= alpha lambda^2
//   -> k;  |a_rr| >= alpha sigma -> r;  else the pair (k, r).  A column that is zero on and below... everywhere counts as perturbed.
//   The search sees the tile only: entries of the panel below it are not consulted (growth there is not bounded by alpha).
// Thread (tx, ty) owns A(tx + 16 a, ty + 16 b) for the 36 blocks on and below the block diagonal (diagonal blocks are kept
// symmetric in full) and G(tx + 16 a, ty + 16 b) for all 64.  The column maxima are DPP reductions over the 16 lanes of a row
// (the 16 values of tx cover every row of the tile), so the common case - the first test passes - costs no extra barrier.
// ------------------------------------------------------------------------------------------------
struct DiagBkShared {
   double col[4][TILE];    // published columns: [step & 1] the candidate k, [2 + (step & 1)] the partner r
   double xrow[4][TILE];   // rows of G of the same indices
   double dk[TILE];
};

template <int N>
__device__ __forceinline__ double row16_ror(double v) {   // value of the lane N places to the right inside the 16-lane row (cyclic)
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
   hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_max(double v) {
   v = fmax(v, row16_ror<8>(v)); v = fmax(v, row16_ror<4>(v)); v = fmax(v, row16_ror<2>(v)); v = fmax(v, row16_ror<1>(v));
   return v;
}
__device__ __forceinline__ int row16_min(int v) {
   v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));
   v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));
   return v;
}

// column p = 16 PA + pm of the symmetric matrix and row p of G into LDS lines; eliminated indices are published as zeros
template <int PA>
__device__ __forceinline__ void bk_publish(const double (&A)[36], const double (&X)[64], double* line, double* xline, int pm, int tx, int ty,
                                           unsigned long long e0, unsigned long long e1) {
   auto gone = [&](int i) { return ((i < 64 ? e0 >> i : e1 >> (i - 64)) & 1ull) != 0; };
   if (ty == pm) {
#pragma unroll
      for (int a = PA; a < 8; ++a) { const int i = tx + 16 * a; line[i] = gone(i) ? 0.0 : A[LI(a, PA)]; }
   }
   if (tx == pm) {
#pragma unroll
      for (int b = 0; b < PA; ++b) { const int j = ty + 16 * b; line[j] = gone(j) ? 0.0 : A[LI(PA, b)]; }
#pragma unroll
      for (int b = 0; b < 8; ++b) xline[ty + 16 * b] = X[PA * 8 + b];
   }
}
__device__ __forceinline__ void bk_publish_any(const double (&A)[36], const double (&X)[64], double* line, double* xline, int p, int tx, int ty,
                                               unsigned long long e0, unsigned long long e1) {
   const int pm = p & 15;
   switch (p >> 4) {   // wave-uniform
      case 0: bk_publish<0>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 1: bk_publish<1>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 2: bk_publish<2>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 3: bk_publish<3>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 4: bk_publish<4>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 5: bk_publish<5>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      case 6: bk_publish<6>(A, X, line, xline, pm, tx, ty, e0, e1); break;
      default: bk_publish<7>(A, X, line, xline, pm, tx, ty, e0, e1); break;
   }
}

__global__ __launch_bounds__(256) void k_tile_diag_bk(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                     double* __restrict__ arena, double* __restrict__ dtail,
                                                     double* __restrict__ winv, int* __restrict__ inertia,
                                                     int* __restrict__ pert_cnt = nullptr, int* __restrict__ pert_list = nullptr,
                                                     const double* __restrict__ orig = nullptr, int orig_ld = 0, int orig_rowmajor = 0,
                                                     const int* __restrict__ perm = nullptr, int isolate = 0) {
   __shared__ DiagBkShared sh;
   __shared__ double s_best[256];
   __shared__ int s_idx[256];
   __builtin_amdgcn_s_setprio(3);
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, tj = task.tj, ld = bd.ldT;
   const int tx = tid & 15, ty = tid >> 4;
   double* C = arena + bd.T + (long long)tj * TILE + (long long)tj * TILE * ld;
   double A[36], X[64];
#pragma unroll
   for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
         const int i = tx + 16 * a, j = ty + 16 * b;
         if (b <= a) A[LI(a, b)] = i >= j ? C[i + (long long)j * ld] : C[j + (long long)i * ld];   // diagonal blocks: mirrored
         X[a * 8 + b] = i == j ? 1.0 : 0.0;
      }
   const int gk0 = tj * TILE;
   const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
   unsigned long long e0 = 0ull, e1 = 0ull;     // eliminated indices (identical in every thread)
   int c_pos = 0, c_neg = 0, c_pert = 0;
   auto count = [&](int p, double d, bool pert) {
      if (gk0 + p < bd.m) { c_pert += pert; c_pos += (!pert && d > 0); c_neg += (!pert && !(d > 0)); }
   };
   int step = 0;
   for (int n_done = 0; n_done < TILE; ++step) {
      const int buf = step & 1;
      const int k = e0 != ~0ull ? __ffsll((long long)~e0) - 1 : 64 + __ffsll((long long)~e1) - 1;
      bk_publish_any(A, X, sh.col[buf], sh.xrow[buf], k, tx, ty, e0, e1);
      __syncthreads();
      double ci[8], cj[8], xr[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) { ci[a] = sh.col[buf][tx + 16 * a]; cj[a] = sh.col[buf][ty + 16 * a]; xr[a] = sh.xrow[buf][ty + 16 * a]; }
      const double akk = sh.col[buf][k];
      double part = 0.0;
#pragma unroll
      for (int a = 0; a < 8; ++a) part = fmax(part, tx + 16 * a == k ? 0.0 : fabs(ci[a]));
      const double lam = row16_max(part);
      int p = k, q = -1;          // pivot index / partner of a 2 x 2 pivot
      double cri[8], crj[8], xq[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) { cri[a] = 0.0; crj[a] = 0.0; xq[a] = 0.0; }
      double arr = 0.0, akr = 0.0;
      if (!(fabs(akk) >= alpha * lam) && lam > 0.0) {     // (wave-uniform: every thread holds the same akk, lam)
         int cand = 1 << 20;
#pragma unroll
         for (int a = 0; a < 8; ++a) if (tx + 16 * a != k && fabs(ci[a]) == lam) cand = min(cand, tx + 16 * a);
         const int r = row16_min(cand);
         bk_publish_any(A, X, sh.col[2 + buf], sh.xrow[2 + buf], r, tx, ty, e0, e1);
         __syncthreads();
#pragma unroll
         for (int a = 0; a < 8; ++a) { cri[a] = sh.col[2 + buf][tx + 16 * a]; crj[a] = sh.col[2 + buf][ty + 16 * a]; xq[a] = sh.xrow[2 + buf][ty + 16 * a]; }
         arr = sh.col[2 + buf][r];
         akr = sh.col[buf][r];
         double ps = 0.0;
#pragma unroll
         for (int a = 0; a < 8; ++a) ps = fmax(ps, tx + 16 * a == r ? 0.0 : fabs(cri[a]));
         const double sig = row16_max(ps);
         if (fabs(akk) * sig >= alpha * lam * lam) {
            // k after all
         } else if (fabs(arr) >= alpha * sig) {
            p = r;
#pragma unroll
            for (int a = 0; a < 8; ++a) { ci[a] = cri[a]; cj[a] = crj[a]; xr[a] = xq[a]; }
         } else
            q = r;
      }
      if (q < 0) {
         // ---- 1 x 1 pivot at index p
         double d = p == k ? akk : arr;
         const bool pert = !(fmax(fabs(d), p == k ? lam : 0.0) > 1e-290) || !(fabs(d) > 0.0);
         if (pert && pert_cnt) {
            // No acceptable pivot for this index inside the tile.  dsytrf would look down the whole column (DeSymIndefSolver.C:78): record
            // the row of the largest entry of column p BELOW the tile - DenseLdl::check_pivots moves that row next to p and factorises
            // again.  The entries are taken from the ORIGINAL matrix (orig, under the current order perm): what the factorisation holds in
            // the panel below is polluted by the replaced pivots of earlier tiles, and the row that couples to p in A is the partner a
            // 2 x 2 pivot needs ([[0 A^T]; [A 0]]); without orig: from the panel as it stands.
            double best = 0.0;
            int bi = -1;
            const double* colp = arena + bd.T + (long long)(gk0 + p) * ld;
            const int pc = perm ? perm[gk0 + p] : gk0 + p;
            for (int i = gk0 + TILE + tid; i < bd.m; i += 256) {
               double v;
               if (orig) {
                  const int pi = perm ? perm[i] : i, hi = pi > pc ? pi : pc, lo = pi > pc ? pc : pi;
                  v = fabs(orig_rowmajor ? orig[(long long)hi * orig_ld + lo] : orig[hi + (long long)lo * orig_ld]);
               } else v = fabs(colp[i]);
               if (v > best) { best = v; bi = i; }
            }
            s_best[tid] = best; s_idx[tid] = bi;
            __syncthreads();
            for (int h = 128; h > 0; h >>= 1) {
               if (tid < h && (s_best[tid + h] > s_best[tid] || (s_best[tid + h] == s_best[tid] && s_idx[tid + h] >= 0 && (s_idx[tid] < 0 || s_idx[tid + h] < s_idx[tid])))) {
                  s_best[tid] = s_best[tid + h]; s_idx[tid] = s_idx[tid + h];
               }
               __syncthreads();
            }
            if (tid == 0 && gk0 + p < bd.m) { const int slot = atomicAdd(pert_cnt, 1); if (slot < bd.m_pad) { pert_list[2 * slot] = gk0 + p; pert_list[2 * slot + 1] = s_best[0] > 0.0 ? s_idx[0] : -1; } }   // (the list holds m_pad records)
            __syncthreads();
         }
         // isolate != 0 (a factorisation DenseLdl::check_pivots will look at): an index without a pivot is taken out of the matrix - a pivot
         // so large that its column vanishes from every update - instead of replaced by a small one, whose multipliers would flood the
         // trailing matrix and make every later tile fail with it: the later tiles then report their own indices only
         if (pert) d = isolate ? 1e300 : (bd.repl_abs > 0.0 ? bd.repl_abs : 1.0);
         count(p, d, pert);
         if (tid == 0) sh.dk[p] = d;
         const double dinv = 1.0 / d;
         double li[8];
#pragma unroll
         for (int a = 0; a < 8; ++a) li[a] = tx + 16 * a == p ? 0.0 : ci[a] * dinv;
#pragma unroll
         for (int b = 0; b < 8; ++b) if (ty + 16 * b == p) cj[b] = 0.0;
#pragma unroll
         for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) A[LI(a, b)] -= li[a] * cj[b];
#pragma unroll
         for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) X[a * 8 + b] -= li[a] * xr[b];
         if (p < 64) e0 |= 1ull << p; else e1 |= 1ull << (p - 64);
         n_done += 1;
      } else {
         // ---- 2 x 2 pivot on (k, q): [a_ b_; b_ c_]
         const double a_ = akk, b_ = akr, c_ = arr;
         const double det = a_ * c_ - b_ * b_, idet = 1.0 / det;
         double wk[8], wq[8];
#pragma unroll
         for (int a = 0; a < 8; ++a) {
            const int i = tx + 16 * a;
            const bool out = i == k || i == q;
            wk[a] = out ? 0.0 : (ci[a] * c_ - cri[a] * b_) * idet;
            wq[a] = out ? 0.0 : (cri[a] * a_ - ci[a] * b_) * idet;
         }
#pragma unroll
         for (int b = 0; b < 8; ++b) { const int j = ty + 16 * b; if (j == k || j == q) { cj[b] = 0.0; crj[b] = 0.0; } }
#pragma unroll
         for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) A[LI(a, b)] -= wk[a] * cj[b] + wq[a] * crj[b];
#pragma unroll
         for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) X[a * 8 + b] -= wk[a] * xr[b] + wq[a] * xq[b];
         // [a_ b_; b_ c_] = R diag(l1, l2) R^T with R = [cs -sn; sn cs]; rows k, q of G become R^T (G_k; G_q)
         const double hd = 0.5 * (a_ - c_), rad = sqrt(hd * hd + b_ * b_), mid = 0.5 * (a_ + c_);
         const double l1 = mid + rad, l2 = mid - rad;
         // eigenvector of l1: (b_, l1 - a_) or (l1 - c_, b_), the better conditioned of the two
         double vx = fabs(l1 - c_) >= fabs(l1 - a_) ? l1 - c_ : b_, vy = fabs(l1 - c_) >= fabs(l1 - a_) ? b_ : l1 - a_;
         const double nv = 1.0 / sqrt(vx * vx + vy * vy);
         const double cs = vx * nv, sn = vy * nv;
#pragma unroll
         for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
               const int i = tx + 16 * a;
               if (i == k) X[a * 8 + b] = cs * xr[b] + sn * xq[b];
               else if (i == q) X[a * 8 + b] = -sn * xr[b] + cs * xq[b];
            }
         if (tid == 0) { sh.dk[k] = l1; sh.dk[q] = l2; }
         count(k, l1, false);
         count(q, l2, false);
         if (k < 64) e0 |= 1ull << k; else e1 |= 1ull << (k - 64);
         if (q < 64) e0 |= 1ull << q; else e1 |= 1ull << (q - 64);
         n_done += 2;
      }
   }
   __syncthreads();
   // Winv[n][c] = G(n, c) / lambda_n ; the tile itself keeps Lambda on its diagonal (nothing reads the rest: every consumer goes
   // through Winv and dtail)
   double* W = winv + bd.winv_off + (long long)tj * TILE * TILE;
#pragma unroll
   for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
         const int i = tx + 16 * a, j = ty + 16 * b;
         W[i + (long long)j * TILE] = X[a * 8 + b] / sh.dk[i];
         if (i >= j) C[i + (long long)j * ld] = i == j ? sh.dk[i] : 0.0;
      }
   if (tid < TILE) dtail[bd.dt_off + tj * TILE + tid] = sh.dk[tid];
   if (tid == 0) {
      if (c_pos) atomicAdd(&inertia[3 * task.blk + 0], c_pos);
      if (c_neg) atomicAdd(&inertia[3 * task.blk + 1], c_neg);
      if (c_pert) atomicAdd(&inertia[3 * task.blk + 2], c_pert);
   }
}

// ------------------------------------------------------------------------------------------------
// solves
// ------------------------------------------------------------------------------------------------
// gather/scatter between original-order flat vectors and the permuted work vectors
__global__ void k_permute_in(const BlkDesc* __restrict__ blks, const int* __restrict__ perm,
                             const long long* __restrict__ perm_off, const double* __restrict__ x, long long x_stride,
                             double* __restrict__ xw, long long xw_stride) {
   // blockIdx.y = block, blockIdx.z = right-hand side (rhs r lives at x + r*x_stride, its work vector at xw + r*xw_stride)
   const int b = blockIdx.y;
   const BlkDesc bd = blks[b];
   const int* p = perm + perm_off[b];
   const int len = bd.n_head + bd.m_pad;
   x += x_stride * blockIdx.z;
   xw += xw_stride * blockIdx.z;
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < len; k += gridDim.x * blockDim.x)
      xw[bd.xw_off + k] = k < bd.n ? x[bd.x_off + p[k]] : 0.0;
}

__global__ void k_permute_out(const BlkDesc* __restrict__ blks, const int* __restrict__ perm,
                              const long long* __restrict__ perm_off, double* __restrict__ x, long long x_stride,
                              const double* __restrict__ xw, long long xw_stride) {
   const int b = blockIdx.y;
   const BlkDesc bd = blks[b];
   const int* p = perm + perm_off[b];
   x += x_stride * blockIdx.z;
   xw += xw_stride * blockIdx.z;
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < bd.n; k += gridDim.x * blockDim.x)
      x[bd.x_off + p[k]] = xw[bd.xw_off + k];
}

// head forward: y_J = L11^-1 b_J ; b[rows] -= L21 y_J (atomics) ; one wave per supernode.  Throughput variant (few
// registers, high occupancy) for launches with many supernodes
__global__ __launch_bounds__(64) void k_head_fwd(const SnDesc* __restrict__ sns, int sn_begin,
                                                const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                const double* __restrict__ arena, double* __restrict__ xw, long long xw_stride,
                                                ScatterCtx sx = ScatterCtx{0, nullptr, nullptr, nullptr, nullptr}) {
   __shared__ double y[HEAD_WMAX];
   const SnDesc sn = sns[sn_begin + blockIdx.x];
   const BlkDesc bd = blks[sn.blk];
   const int w = sn.w, r = sn.r, ld = sn.ld, tid = threadIdx.x;
   const double* P = arena + sn.panel;
   double* xb = xw + xw_stride * blockIdx.y + bd.xw_off;   // blockIdx.y = right-hand side
   if (tid < w) y[tid] = xb[sn.c0 + tid];
   __syncthreads();
   for (int k = 0; k < w; ++k) {
      const double yk = y[k];
      if (tid > k && tid < w) y[tid] -= P[tid + (long long)k * ld] * yk;
      __syncthreads();
   }
   if (tid < w) xb[sn.c0 + tid] = y[tid];
   const int* rows = rowidx + sn.rows;
   for (int a = tid; a < r; a += 64) {
      const int ra = rows[a];
      if (ra >= bd.n) break;  // border rows do not take part in solves with K_i
      double s = 0.0;
      for (int k = 0; k < w; ++k) s += P[w + a + (long long)k * ld] * y[k];
      scatter_add(sx, xb + ra, false, sn.vslot + a, -s);
   }
}


// head forward, latency-lean variant for launches with few supernodes (engine picks it per level).
// On chain-like elimination trees a launch holds one supernode per block, so the kernel is a latency chain: all loads
// are issued up front (row tid of L11 sits in registers), the substitution runs on wave shuffles without barriers.
// LDS hand-over inside ONE wave (the bodies below are executed by single waves, each with LDS of its own): DS operations
// of a wave execute in order, so a workgroup-scope fence (-> s_waitcnt lgkmcnt(0)) plus a scheduling barrier is enough.
__device__ __forceinline__ void wave_lds_sync() {
   __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
   __builtin_amdgcn_wave_barrier();
}

template <int WB>   // WB: compile-time bound of the supernode width (register arrays, unrolled substitution)
__device__ __forceinline__ void head_fwd_body(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                              const double* __restrict__ arena, double* __restrict__ xb, double* ys, int border = 0) {
   // one wave; xb = the permuted work vector of this block and right-hand side; ys = HEAD_WMAX doubles of LDS of this wave
   const int w = sn.w, r = sn.r, ld = sn.ld, tid = threadIdx.x & 63;
   const double* P = arena + sn.panel;
   const int* rows = rowidx + sn.rows;
   const int a0 = tid;
   const int ra0 = a0 < r ? rows[a0] : bd.n;               // first GEMV row of this lane, fetched early
   double l[WB];
#pragma unroll
   for (int k = 0; k < WB; ++k) l[k] = (k < tid && tid < w) ? P[tid + (long long)k * ld] : 0.0;
   double y = tid < w ? xb[sn.c0 + tid] : 0.0;
#pragma unroll
   for (int k = 0; k < WB; ++k)
      if (k < w) y -= l[k] * __shfl(y, k);                 // l[k] == 0 for k >= tid
   if (tid < w) { xb[sn.c0 + tid] = y; ys[tid] = y; }
   wave_lds_sync();
   for (int a = a0; a < r; a += 64) {
      const int ra = a == a0 ? ra0 : rows[a];
      if (ra >= bd.n && !border) break;  // border rows do not take part in solves with K_i (border != 0: forward sweep of the augmented
                                         // factor - the border slots of the work vector collect -L_b y)
      const BelowRow br = below_row(arena, sn, a);
      // the w factors of the row: unconditional loads (clamped index), all in flight at once - a loop of run-time length waits for every one
      double pv[WB];
#pragma unroll
      for (int k = 0; k < WB; ++k) pv[k] = br.p[(k < w ? k : w - 1) * br.stride];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < WB; ++k) s += k < w ? pv[k] * ys[k] : 0.0;
      atomic_add_f64(xb + xw_row(bd, ra), -s);
   }
   wave_lds_sync();   // ys is reused by the caller's next supernode / right-hand side
}

// WCAP: the widest supernode the caller can meet (the engine knows it from the analysis): the register budget of the kernel is that
// of its widest variant, and it decides how many waves share a SIMD
template <int WCAP = HEAD_WMAX>
__device__ __forceinline__ void head_fwd_any(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                             const double* __restrict__ arena, double* __restrict__ xb, double* ys, int border = 0) {
   if (sn.w == 1) head_fwd_body<1>(sn, bd, rowidx, arena, xb, ys, border);
   else if (sn.w <= 8) head_fwd_body<8>(sn, bd, rowidx, arena, xb, ys, border);
   else if (sn.w <= 16 || WCAP <= 16) head_fwd_body<16>(sn, bd, rowidx, arena, xb, ys, border);
   else head_fwd_body<HEAD_WMAX>(sn, bd, rowidx, arena, xb, ys, border);
}

template <int WCAP>
__global__ __launch_bounds__(64) void k_head_fwd_chain(const SnDesc* __restrict__ sns, int sn_begin,
                                                const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                const double* __restrict__ arena, double* __restrict__ xw, long long xw_stride, int border = 0) {
   __shared__ double ys[HEAD_WMAX];
   const SnDesc sn = sns[sn_begin + blockIdx.x];
   const BlkDesc bd = blks[sn.blk];
   head_fwd_any<WCAP>(sn, bd, rowidx, arena, xw + xw_stride * blockIdx.y + bd.xw_off, ys, border);   // blockIdx.y = right-hand side
}

// head diagonal scaling: z = D^-1 y for the head columns
__global__ void k_head_dscale(const SnDesc* __restrict__ sns, int nsn, const BlkDesc* __restrict__ blks,
                              const double* __restrict__ arena, double* __restrict__ xw, long long xw_stride, int mf = 0) {
   xw += xw_stride * blockIdx.y;
   for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nsn; s += gridDim.x * blockDim.x) {
      const SnDesc sn = sns[s];
      if (mf && (sn.mf >= 0 || sn.n_useg > 0)) continue;   // fused into k_front_bwd
      const BlkDesc bd = blks[sn.blk];
      const int ld = sn.ld;
      for (int k = 0; k < sn.w; ++k) xw[bd.xw_off + sn.c0 + k] /= arena[sn.panel + k + (long long)k * ld];
   }
}

// head backward, latency-lean variant (few supernodes per launch), same structure as k_head_fwd_chain:
// lane a gathers x[rows[a]] once and forms its share of all w dot products, the w sums are finished through LDS, the
// transposed substitution runs on shuffles with column tid of L11 in registers.
constexpr int RED_ROWS = 8;    // dot products finished per pass through the wave's LDS scratch (RED_ROWS x 65 doubles)

template <int WB>
__device__ __forceinline__ void head_bwd_body(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                              const double* __restrict__ arena, double* __restrict__ xb, double (*red)[65], int border = 0,
                                              int dscale = 0) {
   const int w = sn.w, r = sn.r, ld = sn.ld, tid = threadIdx.x & 63;
   const double* P = arena + sn.panel;
   const int* rows = rowidx + sn.rows;
   double y = tid < w ? xb[sn.c0 + tid] : 0.0;
   if (dscale && tid < w) y /= P[tid + (long long)tid * ld];   // D^-1 fused: no separate pass over every supernode descriptor
   double part[WB];
#pragma unroll
   for (int k = 0; k < WB; ++k) part[k] = 0.0;
   for (int a = tid; a < r; a += 64) {
      const int ra = rows[a];
      if (ra >= bd.n && !border) break;
      const double xa = xb[xw_row(bd, ra)];
      const BelowRow br = below_row(arena, sn, a);   // (border rows - border-backward sweep only - may live in the border-row arena)
      double pv[WB];                                  // (clamped index: unconditional loads, all in flight at once)
#pragma unroll
      for (int k = 0; k < WB; ++k) pv[k] = br.p[(k < w ? k : w - 1) * br.stride];
#pragma unroll
      for (int k = 0; k < WB; ++k) part[k] += k < w ? pv[k] * xa : 0.0;
   }
   if (WB == 1) {   // a single sum: plain wave reduction
      double s = part[0];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
      if (tid == 0) y -= s;
   } else {         // w sums, RED_ROWS at a time: transpose through LDS, lane (k, g) adds 16 of the 64 partial values of sum k
      constexpr int RH = WB < RED_ROWS ? WB : RED_ROWS;
#pragma unroll
      for (int k0 = 0; k0 < WB; k0 += RH) {
         if (k0 < w) {
#pragma unroll
            for (int k = 0; k < RH; ++k) red[k][tid] = part[k0 + k];
            wave_lds_sync();
            const int k = tid % RH, g = tid / RH;          // 64 / RH groups of lanes per sum
            constexpr int PER = 64 / (64 / RH);            // partial values per group = RH
            double s = 0.0;
            for (int j = 0; j < PER; ++j) s += red[k][g * PER + j];
            for (int off = 32; off >= RH; off >>= 1) s += __shfl_down(s, off);   // lanes < RH hold the RH totals
            const double mine = __shfl(s, (tid - k0) & 63);
            if (tid >= k0 && tid < k0 + RH && tid < w) y -= mine;
            wave_lds_sync();
         }
      }
   }
   // column tid of L11 below the diagonal - read only now: its cache lines came in with the first rows of L21 above, and the registers
   // were free for the partial sums until here (96 -> fewer VGPRs: one more wave per SIMD)
   double c[WB];
#pragma unroll
   for (int k = 0; k < WB; ++k) c[k] = (k > tid && k < w) ? P[k + (long long)tid * ld] : 0.0;
#pragma unroll
   for (int k = WB - 1; k >= 0; --k)
      if (k < w) y -= c[k] * __shfl(y, k);                 // c[k] == 0 for k <= tid
   if (tid < w) xb[sn.c0 + tid] = y;
}

template <int WCAP = HEAD_WMAX>
__device__ __forceinline__ void head_bwd_any(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                             const double* __restrict__ arena, double* __restrict__ xb, double (*red)[65], int border = 0,
                                             int dscale = 0) {
   if (sn.w == 1) head_bwd_body<1>(sn, bd, rowidx, arena, xb, red, border, dscale);
   else if (sn.w <= 8) head_bwd_body<8>(sn, bd, rowidx, arena, xb, red, border, dscale);
   else if (sn.w <= 16 || WCAP <= 16) head_bwd_body<16>(sn, bd, rowidx, arena, xb, red, border, dscale);
   else head_bwd_body<HEAD_WMAX>(sn, bd, rowidx, arena, xb, red, border, dscale);
}

template <int WCAP>
__global__ __launch_bounds__(64) void k_head_bwd_chain(const SnDesc* __restrict__ sns, int sn_begin,
                                                const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                                const double* __restrict__ arena, double* __restrict__ xw, long long xw_stride,
                                                int border = 0, int dscale = 0) {
   __shared__ double red[RED_ROWS][65];
   const SnDesc sn = sns[sn_begin + blockIdx.x];
   const BlkDesc bd = blks[sn.blk];
   head_bwd_any<WCAP>(sn, bd, rowidx, arena, xw + xw_stride * blockIdx.y + bd.xw_off, red, border, dscale);
}

// spine sweeps of the solve: one wave per (block, right-hand side) walks the block's spine supernodes inside one launch
// (ascending for the forward sweep, descending for the backward one); see k_head_factor_spine for the fence
__global__ __launch_bounds__(64) void k_head_solve_spine(const int* __restrict__ spine, const int* __restrict__ spine_off,
                                                        const SnDesc* __restrict__ sns, const BlkDesc* __restrict__ blks,
                                                        const int* __restrict__ rowidx, const double* __restrict__ arena,
                                                        double* __restrict__ xw, long long xw_stride, int backward, int border = 0) {
   __shared__ double ys[HEAD_WMAX];
   __shared__ double red[RED_ROWS][65];
   const int p0 = spine_off[blockIdx.x], p1 = spine_off[blockIdx.x + 1];
   if (p0 == p1) return;
   const BlkDesc bd = blks[blockIdx.x];
   double* xb = xw + xw_stride * blockIdx.y + bd.xw_off;
   for (int q = 0; q < p1 - p0; ++q) {
      const SnDesc sn = sns[spine[backward ? p1 - 1 - q : p0 + q]];
      if (!backward) head_fwd_any(sn, bd, rowidx, arena, xb, ys);
      else head_bwd_any(sn, bd, rowidx, arena, xb, red, border);
      __threadfence();
      __syncthreads();
   }
}

// tail forward step j: tiles i >= j of block b:  b_i -= L(i,j-1) (d z)_{j-1}  (j >= 1) ; tile i == j: z_j = Winv_j b_j
// 256 threads: thread (row, half) accumulates 64 of the 128 columns, halves combined through LDS.  A launch is a chain of
// dependent global-memory round trips (the late tile columns have 64 workgroups: pure latency), so everything that does not
// depend on the step before is requested first: the L tile - and in the workgroup of the diagonal tile the Winv tile as well -
// is requested before the vector of the previous step is looked at (-1 % fwd, -4 % bwd per sweep).  Tried and dropped
// (profiles/r2_gemm_experiments.txt, sweeps): Winv tile of the diagonal workgroup prefetched into registers (256 VGPRs, one wave
// per SIMD: +10 %) or into the L2 by touching its lines (+7 %: the extra requests queue in front of the tiles everybody waits for).
__device__ __forceinline__ void tile_load_half(double (&m)[64], const double* __restrict__ M, long long ldm, int row, int half) {
   const double* pc = M + row + (long long)(half * 64) * ldm;   // one running address, not 64 precomputed ones (128 VGPRs)
#pragma unroll
   for (int c = 0; c < 64; ++c) {
      m[c] = *pc;
      pc += ldm;
   }
}
__device__ __forceinline__ double tile_dot_half(const double (&m)[64], const double* v, int half) {
   double s = 0.0;
#pragma unroll
   for (int c = 0; c < 64; ++c) s += m[c] * v[half * 64 + c];
   return s;
}

__global__ __launch_bounds__(256) void k_tail_fwd(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                 const double* __restrict__ arena, const double* __restrict__ dtail,
                                                 const double* __restrict__ winv, double* __restrict__ xw, int j,
                                                 long long xw_stride) {
   __shared__ double v[TILE];
   __shared__ double part[TILE];
   xw += xw_stride * blockIdx.y;   // blockIdx.y = right-hand side
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, row = tid & 127, half = tid >> 7, ti = task.ti, ld = bd.ldT;
   double* xt = xw + bd.xw_off + bd.n_head;
   // requests in the order they are needed (loads return in order): vector of the previous step, own entry, L tile, Winv tile
   double vprev = 0.0;
   if (j >= 1 && tid < TILE) vprev = xt[(j - 1) * TILE + tid] * dtail[bd.dt_off + (j - 1) * TILE + tid];
   double acc = xt[ti * TILE + row];
   double m[64];
   if (j >= 1) tile_load_half(m, arena + bd.T + (long long)ti * TILE + (long long)(j - 1) * TILE * ld, ld, row, half);
   const double* Wj = winv + bd.winv_off + (long long)j * TILE * TILE;
   if (j >= 1) {
      if (tid < TILE) v[tid] = vprev;
      __syncthreads();
      const double s = tile_dot_half(m, v, half);
      if (half == 1) part[row] = s;
      __syncthreads();
      if (half == 0) acc -= s + part[row];
      __syncthreads();
   }
   if (ti == j) {
      if (half == 0) v[row] = acc;
      __syncthreads();
      tile_load_half(m, Wj, TILE, row, half);
      const double s = tile_dot_half(m, v, half);
      if (half == 1) part[row] = s;
      __syncthreads();
      if (half == 0) acc = s + part[row];
   }
   if (half == 0) xt[ti * TILE + row] = acc;
}

// out[c] = sum_r M[r + c*ldm] v[r] for a 128 x 128 column-major tile.  A column is 1 KiB contiguous: one 16-byte-per-lane load
// instruction of a wave fetches it whole (lane l holds rows 2l, 2l+1), each of the four waves takes 32 columns and has all its
// 32 loads in flight at once (one HBM round trip per tile).  The 32 per-lane partial sums are then reduced over the 64 lanes
// with a transposing butterfly: in the step with distance 32, 16, 8, 4, 2 every lane hands half of its values to its partner
// and adds the partner's contribution to the half it keeps (31 exchanges instead of 32 x 6), after which lane l holds column
// l >> 1 summed over the lanes of equal parity; one more exchange with distance 1 finishes it.  No LDS staging, no barrier
// inside (round 1 staged 32-column chunks through LDS with two barriers each and ran 20 % behind the forward sweep).
// 256 threads; v and out are LDS arrays of 128 doubles; the caller synchronises before v is read and after out is written.
// Load and arithmetic are separate calls so that a kernel can have several tiles in flight before it needs the first.
typedef double tg_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tile_tload(tg_d2 (&m)[32], const double* __restrict__ M, long long ldm, int tid) {
   const int lane = tid & 63, w = tid >> 6;
   const double* pc = M + 2 * lane + (long long)(w * 32) * ldm;
#pragma unroll
   for (int e = 0; e < 32; ++e) {
      m[e] = *(const tg_d2*)pc;
      pc += ldm;
   }
}
__device__ __forceinline__ void tile_tdot(const tg_d2 (&m)[32], const double* v, double* out, int tid) {
   const int lane = tid & 63, w = tid >> 6;
   const double x0 = v[2 * lane], x1 = v[2 * lane + 1];
   double s[32];
#pragma unroll
   for (int e = 0; e < 32; ++e) s[e] = m[e].x * x0 + m[e].y * x1;
#pragma unroll
   for (int half = 16; half >= 1; half >>= 1) {
      const int mask = 2 * half;   // lane distance 32, 16, 8, 4, 2
      const bool up = (lane & mask) != 0;
#pragma unroll
      for (int e = 0; e < half; ++e) {
         const double send = up ? s[e] : s[e + half];
         const double keep = up ? s[e + half] : s[e];
         s[e] = keep + __shfl_xor(send, mask);
      }
   }
   const double r = s[0] + __shfl_xor(s[0], 1);
   if ((lane & 1) == 0) out[w * 32 + (lane >> 1)] = r;
}

// tail backward step i (descending): tiles j <= i:  z_j -= L(i+1,j)^T x_{i+1} (if i+1 < ntc) ; tile j == i: x_i = Winv_i^T (d_i z_i)
// Same ordering of the requests as in k_tail_fwd: the L tile is in flight before the vector of the step before is looked at.
__global__ __launch_bounds__(256) void k_tail_bwd(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                 const double* __restrict__ arena, const double* __restrict__ dtail,
                                                 const double* __restrict__ winv, double* __restrict__ xw, int i,
                                                 long long xw_stride) {
   xw += xw_stride * blockIdx.y;
   __shared__ double v[TILE];
   __shared__ double outp[TILE];
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, tj = task.ti, ld = bd.ldT;
   double* xt = xw + bd.xw_off + bd.n_head;
   const bool upd = i + 1 < bd.ntc;
   double vnext = 0.0, dsc = 0.0;
   if (upd && tid < TILE) vnext = xt[(i + 1) * TILE + tid];
   double acc = tid < TILE ? xt[tj * TILE + tid] : 0.0;
   if (tj == i && tid < TILE) dsc = dtail[bd.dt_off + i * TILE + tid];
   tg_d2 m[32];
   if (upd) tile_tload(m, arena + bd.T + (long long)(i + 1) * TILE + (long long)tj * TILE * ld, ld, tid);
   const double* Wi = winv + bd.winv_off + (long long)i * TILE * TILE;
   if (upd) {
      if (tid < TILE) v[tid] = vnext;
      __syncthreads();
      tile_tdot(m, v, outp, tid);
      __syncthreads();
      if (tid < TILE) acc -= outp[tid];
   }
   if (tj == i) {
      __syncthreads();
      if (tid < TILE) v[tid] = acc * dsc;
      __syncthreads();
      tile_tload(m, Wi, TILE, tid);
      tile_tdot(m, v, outp, tid);   // x[c] = sum_n Winv[n][c] v[n]
      __syncthreads();
      if (tid < TILE) acc = outp[tid];
   }
   if (tid < TILE) xt[tj * TILE + tid] = acc;
}

// ------------------------------------------------------------------------------------------------
// Tail sweeps as ONE launch (k_tail_rows_fwd / k_tail_rows_bwd).  The column-at-a-time kernels above cost a launch per tile
// column: 7.5 us of fixed cost each (35 per sweep at config 2, a quarter of the sweep; 125 for a root of 16 000) on top of the
// bytes.  Here a workgroup owns one tile ROW (forward) or tile COLUMN (backward) of one block for the whole sweep: it walks
// along its tiles, waits for the piece of the solution each one needs (a flag per tile column of the block, set by the workgroup
// that owns that piece), and finally computes and publishes its own piece.  The next tile is requested before the wait, so the
// stream of L overlaps the dependency chain instead of alternating with it.
// Progress without co-residency: workgroups take a ticket when they start and the ticket, not blockIdx, picks the task; the
// task list is sorted so that a task only ever waits for tasks with smaller tickets - whoever holds the smallest unfinished
// ticket is running and waits for nobody, so the sweep completes whatever share of the chip the launch gets (other
// processes on the device included).  A wait that exceeds ~10^7 polls gives up, raises *err and poisons its output with NaN.
// Arithmetic and summation order are those of k_tail_fwd / k_tail_bwd: the results are bit-identical to the launch-per-column path.
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
   const TileTask* tasks;        // (blk, ti = row): sorted by row, then block
   int n_tasks;
   int* ticket;                  // [0] next ticket, [1] finished workgroups (the last one resets both)
   int* flags;                   // one per (block, tile column): == epoch when that piece of the solution is final
   const long long* flag_off;    // per block
   const int* tfirst;            // per (block, tile row): first tile column inside the envelope; nullptr = 0
   const long long* tfirst_off;
   const int* epoch_ptr;         // the sweep's epoch lives in device memory (bumped by k_sweep_bump before every sweep): a captured
                                 // launch sequence can be replayed, which a value baked into the kernel arguments would forbid
   int* err;
   // several right-hand sides: blockIdx.y = right-hand side, each with its own tickets (2 ints), flags and work vector
   long long flag_stride, xw_stride;
   long long poll_limit;         // polls after which a wait gives up (SWEEP_POLL_LIMIT; PIPS_HIP_SWEEP_POLL_LIMIT for tests)
};
constexpr long long SWEEP_POLL_LIMIT = 4000000;   // ~5 s of polling (4e6 polls): three orders of magnitude above the longest legitimate wait
constexpr int SWEEP_NRHS_MAX = 32;

__global__ void k_sweep_bump(int* epoch) { *epoch += 1; }

__device__ __forceinline__ bool sweep_wait(const int* f, int epoch, long long poll_limit = SWEEP_POLL_LIMIT) {
   // No agent-scope fence anywhere in the sweeps: an acquire invalidates and a release writes back the whole L2 of the XCD,
   // thousands of times per sweep (measured: 1.45 ms per sweep with the fences against 1.1 ms for the launch-per-column
   // kernels).  Flags and solution pieces are written and read with agent-scope atomics, which go past the caches; everything
   // else a workgroup reads during the sweep was written before the launch.
   long long spins = 0;
   while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > poll_limit) return false;
   }
   return true;
}
// ticket -> task; every thread of the workgroup gets the same answer
__device__ __forceinline__ int sweep_ticket(const SweepArgs& a, int* sh) {
   if (threadIdx.x == 0) *sh = __hip_atomic_fetch_add(a.ticket + 2 * blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   __syncthreads();
   return *sh;
}
__device__ __forceinline__ void sweep_done(const SweepArgs& a) {
   if (threadIdx.x == 0) {
      int* t = a.ticket + 2 * blockIdx.y;
      const int d = __hip_atomic_fetch_add(t + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == a.n_tasks - 1) {   // everybody has taken a ticket and finished: ready for the next launch
         __hip_atomic_store(t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(t + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
}
__device__ __forceinline__ double sweep_load(const double* p) {   // a value another workgroup of this launch has written
   return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sweep_store(double* p, double x) {   // ... and one this workgroup publishes
   __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave waits until its stores have been acknowledged (the agent-scope ones are written through), then the flag goes up.
// The wait is explicit: a workgroup-scope release does not emit one, and without it the flag overtakes the data.
__device__ __forceinline__ void sweep_publish(int* flag, int epoch) {
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
   if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double sweep_nan() { return __builtin_nan(""); }

__global__ __launch_bounds__(256) void k_tail_rows_fwd(SweepArgs a, const BlkDesc* __restrict__ blks, const double* __restrict__ arena,
                                                      const double* __restrict__ dtail, const double* __restrict__ winv,
                                                      double* __restrict__ xw) {
   __shared__ double v[TILE];
   __shared__ double part[TILE];
   __shared__ int sh_t, sh_ok;
   const int epoch = *a.epoch_ptr;
   const int t = sweep_ticket(a, &sh_t);
   const TileTask task = a.tasks[t];
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, row = tid & 127, half = tid >> 7, i = task.ti, ld = bd.ldT;
   double* xt = xw + a.xw_stride * blockIdx.y + bd.xw_off + bd.n_head;
   int* fl = a.flags + a.flag_stride * blockIdx.y + a.flag_off[task.blk];
   const int j0 = a.tfirst ? min(i, a.tfirst[a.tfirst_off[task.blk] + i]) : 0;
   const double* Lrow = arena + bd.T + (long long)i * TILE;
   double acc = half == 0 ? xt[i * TILE + row] : 0.0;
   double m[64];
   bool ok = true;
   if (j0 < i) {
      tile_load_half(m, Lrow + (long long)j0 * TILE * ld, ld, row, half);
      for (int j = j0;; ++j) {
         if (tid == 0) sh_ok = sweep_wait(fl + j, epoch, a.poll_limit) ? 1 : 0;
         __syncthreads();
         if (!sh_ok) { ok = false; break; }
         if (tid < TILE) v[tid] = sweep_load(xt + j * TILE + tid) * dtail[bd.dt_off + j * TILE + tid];
         __syncthreads();
         const double s = tile_dot_half(m, v, half);
         if (half == 1) part[row] = s;
         __syncthreads();
         if (half == 0) acc -= s + part[row];
         if (j + 1 >= i) break;
         // the same registers take the next tile, which is then on its way while this workgroup waits for the next piece
         __builtin_amdgcn_sched_barrier(0);
         tile_load_half(m, Lrow + (long long)(j + 1) * TILE * ld, ld, row, half);
      }
   }
   if (ok) {
      __syncthreads();
      if (half == 0) v[row] = acc;
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      tile_load_half(m, winv + bd.winv_off + (long long)i * TILE * TILE, TILE, row, half);
      const double s = tile_dot_half(m, v, half);
      if (half == 1) part[row] = s;
      __syncthreads();
      if (half == 0) acc = s + part[row];
   } else {
      acc = sweep_nan();
      if (tid == 0) *a.err = 1;
   }
   if (half == 0) sweep_store(xt + i * TILE + row, acc);
   sweep_publish(fl + i, epoch);
   sweep_done(a);
}

// backward: the workgroup owns tile column i:  x_i = Winv_i^T ( d_i ( z_i - sum_{k > i} L(k,i)^T x_k ) ), k descending
// border != 0 (Engine::solve_border_backward): the border tile rows of the augmented factor take part - L21(k, i) for the tile rows
// k >= ntc below the tail, whose "solution" is given (minus the root's x0, at xt[k * TILE ..]) and needs no flag.
__global__ __launch_bounds__(256) void k_tail_rows_bwd(SweepArgs a, const BlkDesc* __restrict__ blks, const double* __restrict__ arena,
                                                      const double* __restrict__ dtail, const double* __restrict__ winv,
                                                      double* __restrict__ xw, int border = 0) {
   __shared__ double v[TILE];
   __shared__ double outp[TILE];
   __shared__ int sh_t, sh_ok;
   const int epoch = *a.epoch_ptr;
   const int t = sweep_ticket(a, &sh_t);
   const TileTask task = a.tasks[a.n_tasks - 1 - t];   // last tile column first
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, i = task.ti, ld = bd.ldT;
   double* xt = xw + a.xw_stride * blockIdx.y + bd.xw_off + bd.n_head;
   int* fl = a.flags + a.flag_stride * blockIdx.y + a.flag_off[task.blk];
   const int* tf = a.tfirst ? a.tfirst + a.tfirst_off[task.blk] : nullptr;
   const double* Lcol = arena + bd.T + (long long)i * TILE * ld;
   // tile rows k > i whose envelope reaches column i, from the last one down
   auto next_k = [&](int k) {
      while (k > i && tf && tf[k] > i) --k;
      return k;
   };
   double acc = tid < TILE ? xt[i * TILE + tid] : 0.0;
   const double dsc = tid < TILE ? dtail[bd.dt_off + i * TILE + tid] : 0.0;
   tg_d2 m[32];
   int k = next_k((border ? bd.ntr : bd.ntc) - 1);
   bool ok = true;
   if (k > i) {
      tile_tload(m, Lcol + (long long)k * TILE, ld, tid);
      for (;;) {
         if (k < bd.ntc) {
            if (tid == 0) sh_ok = sweep_wait(fl + k, epoch, a.poll_limit) ? 1 : 0;
            __syncthreads();
            if (!sh_ok) { ok = false; break; }
            if (tid < TILE) v[tid] = sweep_load(xt + k * TILE + tid);
         } else {
            __syncthreads();
            if (tid < TILE) v[tid] = xt[k * TILE + tid];   // border rows: written before the launch
         }
         __syncthreads();
         tile_tdot(m, v, outp, tid);
         __syncthreads();
         if (tid < TILE) acc -= outp[tid];
         k = next_k(k - 1);
         if (k <= i) break;
         __builtin_amdgcn_sched_barrier(0);
         tile_tload(m, Lcol + (long long)k * TILE, ld, tid);
      }
   }
   if (ok) {
      __syncthreads();
      if (tid < TILE) v[tid] = acc * dsc;
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      tile_tload(m, winv + bd.winv_off + (long long)i * TILE * TILE, TILE, tid);
      tile_tdot(m, v, outp, tid);   // x[c] = sum_n Winv[n][c] v[n]
      __syncthreads();
      if (tid < TILE) acc = outp[tid];
   } else {
      acc = sweep_nan();
      if (tid == 0) *a.err = 1;
   }
   if (tid < TILE) sweep_store(xt + i * TILE + tid, acc);
   sweep_publish(fl + i, epoch);
   sweep_done(a);
}

// border part of the work vectors for the border-backward sweep: minus the root solution at the block's border rows, zero padding
__global__ void k_border_fill(const BlkDesc* __restrict__ blks, const int* __restrict__ bmap, const double* __restrict__ x0,
                              double* __restrict__ xw, double sign = -1.0) {
   const BlkDesc bd = blks[blockIdx.y];
   double* xbd = xw + bd.xw_off + bd.n_head + bd.m_pad;
   const int* bm = bmap + bd.bmap_off;
   for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < bd.nb_pad; r += gridDim.x * blockDim.x) xbd[r] = (r < bd.nb && x0) ? sign * x0[bm[r]] : 0.0;
}

// Sweeps of the AUGMENTED factor [L 0; L_b I] (Engine::forward_augmented / backward_augmented): with the border slots of the work vector
// as targets the forward sweep leaves -L_b y = -Br^T K^-1 b there (L_b = Br^T L^-T D^-1), and the backward sweep started from
// D^-1 y with the border slots holding x0 gives K^-1 (b - Br x0): one forward and one backward sweep per solveCompressed instead of two
// full solves.  The pieces the level kernels do not cover:
// ... the simple leaves that own border rows (thread per leaf): forward x_border -= l_b y_c, backward x_c -= l_b^T x_border
__global__ __launch_bounds__(256) void k_leaf_border(const int* __restrict__ list, int cnt, const SnDesc* __restrict__ sns, const BlkDesc* __restrict__ blks,
                                                     const int* __restrict__ rowidx, const double* __restrict__ arena, double* __restrict__ xw, int backward) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= cnt) return;
   const SnDesc sn = sns[list[t]];
   const BlkDesc bd = blks[sn.blk];
   const double* P = arena + sn.panel;
   const int* rows = rowidx + sn.rows;
   double* xb = xw + bd.xw_off;
   if (!backward) {
      const double y = xb[sn.c0];
      for (int a = sn.rb; a < sn.r; ++a) atomic_add_f64(xb + xw_row(bd, rows[a]), -P[1 + a] * y);
   } else {
      double s = 0.0;
      for (int a = sn.rb; a < sn.r; ++a) s += P[1 + a] * xb[xw_row(bd, rows[a])];
      xb[sn.c0] -= s;
   }
}

// ... deterministic mode, forward: the border slot of (block, border id) from the head supernodes that hold that border row,
//    x_border = - sum_J sum_k L_b(J)(a, k) y_J(k),
// in two steps without atomics: every (supernode, border row) product by a thread of its own - the entries are in the order of the supernode
// array, a supernode's rows side by side, so the loads along k are coalesced - then one wave per target adds the products of its list
// (entry p of the list on lane p % 64: a fixed assignment; the lanes' sums in the fixed tree of the shuffles): equal bits from run to run.
// Entry: where the w factors of the row lie (stride between them) and the first of J's columns in the work vector.
// (A first version gathered the factors per target: 1560 entries per wave, every factor a sector of its own - 10 ms per sweep on the configs[3] share.)
struct BgEntry { long long off; unsigned y; unsigned short stride, w; };
__global__ void k_border_rowdot_det(long long n_ent, const BgEntry* __restrict__ ent, const double* __restrict__ arena, const double* __restrict__ xw,
                                    double* __restrict__ val) {
   for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n_ent; p += (long long)gridDim.x * blockDim.x) {
      const BgEntry e = ent[p];
      const double* L = arena + e.off;
      const double* y = xw + e.y;
      double s = 0.0;
      for (int k = 0; k < (int)e.w; ++k) s += L[(long long)k * e.stride] * y[k];
      val[p] = s;
   }
}
__global__ __launch_bounds__(256) void k_border_gather_det(long long n_targets, const long long* __restrict__ ptr, const int* __restrict__ idx,
                                                           const long long* __restrict__ slot, const double* __restrict__ val, double* __restrict__ xw) {
   const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
   if (t >= n_targets) return;
   const int lane = threadIdx.x & 63;
   double acc = 0.0;
   for (long long p = ptr[t] + lane; p < ptr[t + 1]; p += 64) acc += val[idx[p]];
   for (int h = 32; h > 0; h >>= 1) acc += __shfl_xor(acc, h);
   if (lane == 0) xw[slot[t]] = -acc;
}

// ... the border rows of the dense tail, forward: x_border(ib) -= sum_j T(border tile row ib, tile column j) (d_j z_j), z = what
// k_tail_rows_fwd left (its rows are D^-1-scaled).  grid (border tile rows, blocks), 256 threads: thread = (row, half of the columns)
__global__ __launch_bounds__(256) void k_tail_border_fwd(const BlkDesc* __restrict__ blks, const double* __restrict__ arena,
                                                         const double* __restrict__ dtail, double* __restrict__ xw) {
   __shared__ double v[TILE];
   __shared__ double part[TILE];
   const BlkDesc bd = blks[blockIdx.y];
   if (bd.m <= 0 || (int)blockIdx.x * TILE >= bd.nb_pad) return;
   const int tid = threadIdx.x, row = tid & 127, half = tid >> 7, ld = bd.ldT;
   const double* xt = xw + bd.xw_off + bd.n_head;
   const double* Trow = arena + bd.T + bd.m_pad + (long long)blockIdx.x * TILE + row;
   double acc = 0.0;
   for (int j = 0; j < bd.m_pad; j += TILE) {
      __syncthreads();
      if (tid < TILE) v[tid] = xt[j + tid] * dtail[bd.dt_off + j + tid];
      __syncthreads();
      const double* Tc = Trow + (long long)(j + half * 64) * ld;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) acc += Tc[(long long)c * ld] * v[half * 64 + c];
   }
   if (half == 1) part[row] = acc;
   __syncthreads();
   if (half == 0) {
      double* xbd = xw + bd.xw_off + bd.n_head + bd.m_pad;
      xbd[blockIdx.x * TILE + row] -= acc + part[row];
   }
}

// ... and the border slots of every block added to the root right-hand side: b0[bmap[r]] += x_border[r]
__global__ void k_border_collect(const BlkDesc* __restrict__ blks, const int* __restrict__ bmap, const double* __restrict__ xw,
                                 double* __restrict__ b0) {
   const BlkDesc bd = blks[blockIdx.y];
   const double* xbd = xw + bd.xw_off + bd.n_head + bd.m_pad;
   const int* bm = bmap + bd.bmap_off;
   for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < bd.nb; r += gridDim.x * blockDim.x) atomic_add_f64(b0 + bm[r], xbd[r]);
}

// ------------------------------------------------------------------------------------------------
// Multi-vector solves: DoubleLinearSolver::solve(nrhs, ...) (PardisoSolver.C:276-352) and the blocked Schur path.
// Up to MQ = 32 right-hand sides are interleaved, entry k of right-hand side q at xm[(xw_off + k) * MQ + q], and a lane
// owns one right-hand side (lane & 31) for the whole sweep.  Factor entries are then wave-uniform operands and every entry
// of L is fetched once for the 32 right-hand sides instead of once per right-hand side.
// ------------------------------------------------------------------------------------------------
constexpr int MQ = 32;

// grid (x, block, panel); transposing gather / scatter between nr vectors at distance x_stride and the interleaved work array.  More than
// MQ right-hand sides go in PANELS of MQ: panel p is an interleaved work array of its own at xm + p * panel_stride (right-hand sides
// 32 p .. 32 p + 31), and every multi-vector kernel takes its panel from the grid
__global__ void k_mpermute(const BlkDesc* __restrict__ blks, const int* __restrict__ perm, const long long* __restrict__ perm_off,
                           double* __restrict__ x, long long x_stride, int nr, double* __restrict__ xm, int out, long long panel_stride = 0) {
   const BlkDesc bd = blks[blockIdx.y];
   const int* p = perm + perm_off[blockIdx.y];
   xm += panel_stride * blockIdx.z;
   x += (long long)MQ * blockIdx.z * x_stride;
   nr = min(MQ, nr - MQ * (int)blockIdx.z);
   const long long len = (long long)(out ? bd.n : bd.n_head + bd.m_pad) * MQ;
   for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < len; idx += (long long)gridDim.x * blockDim.x) {
      const int k = (int)(idx / MQ), q = (int)(idx % MQ);
      double* w = xm + (bd.xw_off + k) * MQ + q;
      if (out) { if (q < nr) x[q * x_stride + bd.x_off + p[k]] = *w; }
      else *w = (k < bd.n && q < nr) ? x[q * x_stride + bd.x_off + p[k]] : 0.0;
   }
}

// head forward / backward for one supernode and 32 right-hand sides: one wave, lane = (right-hand side, half); the two halves
// split the below-rows.  sns are taken four per workgroup.
template <int WB>
__device__ __forceinline__ void mhead_body(const SnDesc& sn, const BlkDesc& bd, const int* __restrict__ rowidx,
                                           const double* __restrict__ arena, double* __restrict__ xm, int backward, double* __restrict__ slots) {
   const int w = sn.w, r = sn.r, ld = sn.ld, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
   const double* P = arena + sn.panel;
   double* xb = xm + bd.xw_off * MQ;
   const int* rows = rowidx + sn.rows;
   double y[WB];
#pragma unroll
   for (int k = 0; k < WB; ++k) y[k] = k < w ? xb[(long long)(sn.c0 + k) * MQ + q] : 0.0;
   if (!backward) {
#pragma unroll
      for (int k = 1; k < WB; ++k)
         if (k < w) {
            double v = y[k];
#pragma unroll
            for (int j = 0; j < k; ++j) v -= P[k + (long long)j * ld] * y[j];
            y[k] = v;
         }
      if (h == 0) {
#pragma unroll
         for (int k = 0; k < WB; ++k)
            if (k < w) xb[(long long)(sn.c0 + k) * MQ + q] = y[k];
      }
      for (int a = h; a < r; a += 2) {
         const int ra = rows[a];
         if (ra >= bd.n) break;   // border rows (sorted last) take no part in solves with K_i
         double s = 0.0;
#pragma unroll
         for (int k = 0; k < WB; ++k)
            if (k < w) s += P[w + a + (long long)k * ld] * y[k];
         // deterministic mode: every contribution into its own slot (SnDesc::vslot + a, MQ right-hand sides wide); k_mgather_slots adds them per target
         if (slots) slots[(sn.vslot + a) * MQ + q] = -s;
         else atomic_add_f64(xb + (long long)ra * MQ + q, -s);
      }
   } else {
      double part[WB];
#pragma unroll
      for (int k = 0; k < WB; ++k) part[k] = 0.0;
      for (int a = h; a < r; a += 2) {
         const int ra = rows[a];
         if (ra >= bd.n) break;
         const double xa = xb[(long long)ra * MQ + q];
#pragma unroll
         for (int k = 0; k < WB; ++k)
            if (k < w) part[k] += P[w + a + (long long)k * ld] * xa;
      }
#pragma unroll
      for (int k = 0; k < WB; ++k)
         if (k < w) y[k] -= part[k] + __shfl_xor(part[k], 32);
#pragma unroll
      for (int k = WB - 2; k >= 0; --k)
         if (k < w - 1) {
            double v = y[k];
#pragma unroll
            for (int j = k + 1; j < WB; ++j)
               if (j < w) v -= P[j + (long long)k * ld] * y[j];
            y[k] = v;
         }
      if (h == 0) {
#pragma unroll
         for (int k = 0; k < WB; ++k)
            if (k < w) xb[(long long)(sn.c0 + k) * MQ + q] = y[k];
      }
   }
}

__global__ __launch_bounds__(256) void k_mhead(const SnDesc* __restrict__ sns, int sn_begin, int cnt,
                                              const BlkDesc* __restrict__ blks, const int* __restrict__ rowidx,
                                              const double* __restrict__ arena, double* __restrict__ xm, int backward, long long panel_stride = 0,
                                              double* __restrict__ slots = nullptr) {
   const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
   if (i >= cnt) return;
   xm += panel_stride * blockIdx.y;
   const SnDesc sn = sns[sn_begin + i];
   const BlkDesc bd = blks[sn.blk];
   if (sn.w == 1) mhead_body<1>(sn, bd, rowidx, arena, xm, backward, slots);
   else if (sn.w <= 8) mhead_body<8>(sn, bd, rowidx, arena, xm, backward, slots);
   else mhead_body<HEAD_WMAX>(sn, bd, rowidx, arena, xm, backward, slots);
}

// deterministic mode, one panel of MQ right-hand sides: target row t of the list takes the sum of its slots in the list's order
// (k_gather_slots for MQ interleaved vectors); a wave takes two targets, its halves the right-hand sides of one each
__global__ __launch_bounds__(256) void k_mgather_slots(long long n_targets, const long long* __restrict__ tgt, const long long* __restrict__ off,
                                                      const long long* __restrict__ slots, const double* __restrict__ val, double* __restrict__ xm) {
   const long long t = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 5;
   const int q = threadIdx.x & 31;
   if (t >= n_targets) return;
   double s = 0.0;
   for (long long p = off[t]; p < off[t + 1]; ++p) s += val[slots[p] * MQ + q];
   xm[tgt[t] * MQ + q] += s;
}

// the simple leaves' forward substitution as a gather by target row (k_leaf_fwd_gather) for a panel of MQ interleaved right-hand sides:
// no atomics, the order of a row's sum is the order of its list
__global__ __launch_bounds__(256) void k_mleaf_fwd_gather(const int* __restrict__ rows, const int* __restrict__ ptr, const int* __restrict__ src,
                                                         const double* __restrict__ val, double* __restrict__ xm, int n) {
   const int t = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 5), q = threadIdx.x & 31;
   if (t >= n) return;
   double s = 0.0;
   for (int p = ptr[t]; p < ptr[t + 1]; ++p) s += val[p] * xm[(long long)src[p] * MQ + q];
   xm[(long long)rows[t] * MQ + q] -= s;
}

__global__ void k_mhead_dscale(const SnDesc* __restrict__ sns, int nsn, const BlkDesc* __restrict__ blks,
                               const double* __restrict__ arena, double* __restrict__ xm, long long panel_stride = 0) {
   xm += panel_stride * blockIdx.y;
   for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < (long long)nsn * MQ; idx += (long long)gridDim.x * blockDim.x) {
      const SnDesc sn = sns[idx / MQ];
      const BlkDesc bd = blks[sn.blk];
      const int q = (int)(idx % MQ), ld = sn.ld;
      for (int k = 0; k < sn.w; ++k) xm[(bd.xw_off + sn.c0 + k) * MQ + q] /= arena[sn.panel + k + (long long)k * ld];
   }
}

// acc += sign * M V (TRANSPOSED = 0) or sign * M^T V (TRANSPOSED = 1) for a 128 x 128 column-major tile M and 32 right-hand sides V[k][q]
// (the interleaved layout IS the [k][column] image the matrix-pipe fragments are read from), on the FP64 matrix cores: 256 threads = 4
// waves, wave w owns rows 32 w .. 32 w + 31 of the result, a lane the elements (row 32 w + 16 h + (lane & 15), right-hand side 4 c +
// (lane >> 4)), h < 2, c < 8.  The tile is staged through LDS sixteen k at a time, as the [k][row] image in both cases (the transposed
// case transposes while staging).  (The round-2 version was a scalar multiply-add loop over the LDS tile: DoubleLinearSolver::solve(nrhs)
// behind the adapters spent 11 of its 13 ms per 160 right-hand sides there and in re-reading L per right-hand side.)
constexpr int MVQ = MQ + 8;   // LDS row length of V: the four k of a fragment read land in different banks
// KC k of a tile into registers (KC / 2 doubles per thread), as mtile_apply stages them
template <int TRANSPOSED, int KC>
__device__ __forceinline__ void mtile_fetch(double (&pre)[KC / 2], const double* __restrict__ M, long long ldm, int c0, int tid) {
#pragma unroll
   for (int e = 0; e < KC / 2; ++e) {
      const int idx = e * 256 + tid;
      pre[e] = TRANSPOSED ? M[c0 + (idx % KC) + (long long)(idx / KC) * ldm]       // KC rows x 128 columns, the KC rows of a column contiguous
                          : M[(idx & 127) + (long long)(c0 + (idx >> 7)) * ldm];   // KC columns x 128 rows, rows contiguous
   }
}
// NC column groups of four right-hand sides per workgroup, KC k per LDS stage; prefetched: pre already holds the first KC k of M (the
// caller asked for them before it waited for the flag of V)
template <int TRANSPOSED, int NC = 8, int KC = 16>
__device__ __forceinline__ void mtile_apply(double (&acc)[2][NC], const double* __restrict__ M, long long ldm,
                                            const double (*V)[MVQ], double (*Lt)[TILE + 1], int tid, double sign, double (&pre)[KC / 2], bool prefetched = false) {
   const int lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, ej = lane & 3;
   // (the next KC k are on their way from memory while these are multiplied: a chunk was a round trip of its own before)
   if (!prefetched) mtile_fetch<TRANSPOSED, KC>(pre, M, ldm, 0, tid);
   for (int c0 = 0; c0 < TILE; c0 += KC) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < KC / 2; ++e) {
         const int idx = e * 256 + tid;
         if (TRANSPOSED) Lt[idx % KC][idx / KC] = pre[e];
         else Lt[idx >> 7][idx & 127] = pre[e];
      }
      if (c0 + KC < TILE) mtile_fetch<TRANSPOSED, KC>(pre, M, ldm, c0 + KC, tid);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < KC / 4; ++st) {
         const double f0 = Lt[4 * st + ek][32 * w + er], f1 = Lt[4 * st + ek][32 * w + 16 + er];
         double fc[NC];
#pragma unroll
         for (int c = 0; c < NC; ++c) fc[c] = sign * V[c0 + 4 * st + ek][4 * c + ej];
#pragma unroll
         for (int c = 0; c < NC; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], f0, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], f1, acc[1][c], 0, 0, 0);
         }
      }
   }
}

// The whole tile in registers (64 doubles per thread), for the single-launch sweeps: a tile row's chain is flag -> V -> product -> next
// flag, and with a stage of the tile requested one stage ahead every stage of every product waited for its round trip (four per product,
// 16 us per step of the chain).  Here a product's tile is complete in registers before its flag is waited for, and while its stages move to
// LDS the stages of the NEXT tile of the row (the next column's, at last the diagonal block's inverse) are requested into their places.
template <int TRANSPOSED, int KC>
__device__ __forceinline__ void mtile_fetch_all(double (&T)[TILE / 2], const double* __restrict__ M, long long ldm, int tid) {
#pragma unroll
   for (int s = 0; s < TILE / KC; ++s) mtile_fetch<TRANSPOSED, KC>(*(double (*)[KC / 2]) & T[s * (KC / 2)], M, ldm, s * KC, tid);
}
// (MV: row length of V - the right-hand sides of the slice, padded so that the four k of a fragment read land in different banks)
template <int NC> constexpr int mtile_mv() { return NC == 2 ? 8 : 4 * NC + 8; }
template <int TRANSPOSED, int NC, int KC>
__device__ __forceinline__ void mtile_apply_whole(double (&acc)[2][NC], double (&T)[TILE / 2], const double (*V)[mtile_mv<NC>()], double (*Lt)[TILE + 1], int tid,
                                                  double sign, const double* __restrict__ Mnext, long long ldnext) {
   const int lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, ej = lane & 3;
#pragma unroll
   for (int s = 0; s < TILE / KC; ++s) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < KC / 2; ++e) {
         const int idx = e * 256 + tid;
         if (TRANSPOSED) Lt[idx % KC][idx / KC] = T[s * (KC / 2) + e];
         else Lt[idx >> 7][idx & 127] = T[s * (KC / 2) + e];
      }
      if (Mnext) mtile_fetch<TRANSPOSED, KC>(*(double (*)[KC / 2]) & T[s * (KC / 2)], Mnext, ldnext, s * KC, tid);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < KC / 4; ++st) {
         const double f0 = Lt[4 * st + ek][32 * w + er], f1 = Lt[4 * st + ek][32 * w + 16 + er];
         double fc[NC];
#pragma unroll
         for (int c = 0; c < NC; ++c) fc[c] = sign * V[s * KC + 4 * st + ek][4 * c + ej];
#pragma unroll
         for (int c = 0; c < NC; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], f0, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[c], f1, acc[1][c], 0, 0, 0);
         }
      }
   }
}

// tail forward step j for 32 right-hand sides: tiles i >= j:  b_i -= L(i,j-1) (d z)_{j-1} ; tile i == j: z_j = Winv_j b_j.  grid (tasks, panels)
__global__ __launch_bounds__(256) void k_mtail_fwd(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                  const double* __restrict__ arena, const double* __restrict__ dtail,
                                                  const double* __restrict__ winv, double* __restrict__ xm, int j, long long panel_stride = 0) {
   __shared__ double V[TILE][MVQ];
   __shared__ double Lt[16][TILE + 1];
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, ti = task.ti, ld = bd.ldT;
   double* xt = xm + panel_stride * blockIdx.y + (bd.xw_off + bd.n_head) * MQ;
   double acc[2][8], pre[8];
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[h][c] = xt[(long long)(ti * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek];
   if (j >= 1) {
      for (int idx = tid; idx < TILE * MQ; idx += 256) {
         const int c = idx >> 5, qq = idx & 31;
         V[c][qq] = xt[(long long)((j - 1) * TILE + c) * MQ + qq] * dtail[bd.dt_off + (j - 1) * TILE + c];
      }
      mtile_apply<0>(acc, arena + bd.T + (long long)ti * TILE + (long long)(j - 1) * TILE * ld, ld, V, Lt, tid, -1.0, pre);
   }
   if (ti == j) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < 8; ++c) { V[32 * w + 16 * h + er][4 * c + ek] = acc[h][c]; acc[h][c] = 0.0; }
      mtile_apply<0>(acc, winv + bd.winv_off + (long long)j * TILE * TILE, TILE, V, Lt, tid, 1.0, pre);
   }
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 8; ++c) xt[(long long)(ti * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek] = acc[h][c];
}

// tail backward step i (descending): tiles j <= i:  z_j -= L(i+1,j)^T x_{i+1} ; tile j == i: x_i = Winv_i^T (d_i z_i).  grid (tasks, panels)
__global__ __launch_bounds__(256) void k_mtail_bwd(const TileTask* __restrict__ tasks, const BlkDesc* __restrict__ blks,
                                                  const double* __restrict__ arena, const double* __restrict__ dtail,
                                                  const double* __restrict__ winv, double* __restrict__ xm, int i, long long panel_stride = 0) {
   __shared__ double V[TILE][MVQ];
   __shared__ double Lt[16][TILE + 1];
   const TileTask task = tasks[blockIdx.x];
   if (task.blk < 0) return;
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, tj = task.ti, ld = bd.ldT;
   double* xt = xm + panel_stride * blockIdx.y + (bd.xw_off + bd.n_head) * MQ;
   double acc[2][8], pre[8];
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[h][c] = xt[(long long)(tj * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek];
   if (i + 1 < bd.ntc) {
      for (int idx = tid; idx < TILE * MQ; idx += 256) V[idx >> 5][idx & 31] = xt[(long long)((i + 1) * TILE + (idx >> 5)) * MQ + (idx & 31)];
      mtile_apply<1>(acc, arena + bd.T + (long long)(i + 1) * TILE + (long long)tj * TILE * ld, ld, V, Lt, tid, -1.0, pre);
   }
   if (tj == i) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < 8; ++c) {
            V[32 * w + 16 * h + er][4 * c + ek] = acc[h][c] * dtail[bd.dt_off + i * TILE + 32 * w + 16 * h + er];
            acc[h][c] = 0.0;
         }
      mtile_apply<1>(acc, winv + bd.winv_off + (long long)i * TILE * TILE, TILE, V, Lt, tid, 1.0, pre);
   }
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 8; ++c) xt[(long long)(tj * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek] = acc[h][c];
}

// The tail sweeps of 32 right-hand sides as ONE launch per direction (grid: (block, tile row) tasks x panels), like k_tail_rows_fwd / _bwd
// for one: a workgroup owns tile row i (forward) / tile column i (backward), takes the pieces of the solution it needs as their flags go up
// and multiplies on the matrix pipe (mtile_apply).  35 + 35 launches of 25 us each per pass were 3.4 of the 4.4 ms of a solve(160) on one
// configs[1] block; the chain of a pass is 35 x (tile product + flag).  Same arithmetic as k_mtail_fwd / _bwd, same order per tile row.
template <int NC>
__global__ __launch_bounds__(256, 2) void k_mtail_rows_fwd(SweepArgs a, const BlkDesc* __restrict__ blks, const double* __restrict__ arena,
                                                       const double* __restrict__ dtail, const double* __restrict__ winv, double* __restrict__ xm,
                                                       long long panel_stride) {
   __shared__ double V[TILE][mtile_mv<NC>()];
   constexpr int KC = NC <= 4 ? 32 : 16;   // (a slice of a panel leaves LDS for stages of 32 k: half the barriers)
   __shared__ double Lt[KC][TILE + 1];
   __shared__ int sh_t, sh_ok;
   const int epoch = *a.epoch_ptr;
   const int t = sweep_ticket(a, &sh_t);
   const TileTask task = a.tasks[t];
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, i = task.ti, ld = bd.ldT;
   constexpr int SL = 8 / NC, QS = 4 * NC;   // slices of a panel, right-hand sides per slice
   double* xt = xm + panel_stride * (blockIdx.y / SL) + (bd.xw_off + bd.n_head) * MQ + QS * (blockIdx.y % SL);
   int* fl = a.flags + a.flag_stride * blockIdx.y + a.flag_off[task.blk];
   const int j0 = a.tfirst ? min(i, a.tfirst[a.tfirst_off[task.blk] + i]) : 0;
   double acc[2][NC], T[TILE / 2];
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[h][c] = xt[(long long)(i * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek];
   bool ok = true;
   const double* Wi = winv + bd.winv_off + (long long)i * TILE * TILE;
   const double* Li = arena + bd.T + (long long)i * TILE;
   if (j0 < i) mtile_fetch_all<0, KC>(T, Li + (long long)j0 * TILE * ld, ld, tid);   // (no tile waits for a flag)
   else mtile_fetch_all<0, KC>(T, Wi, TILE, tid);
   for (int j = j0; j < i; ++j) {
      if (tid == 0) sh_ok = sweep_wait(fl + j, epoch, a.poll_limit) ? 1 : 0;
      __syncthreads();
      if (!sh_ok) { ok = false; break; }
      for (int idx = tid; idx < TILE * QS; idx += 256) {
         const int c = idx / QS, qq = idx % QS;
         V[c][qq] = sweep_load(xt + (long long)(j * TILE + c) * MQ + qq) * dtail[bd.dt_off + j * TILE + c];
      }
      const bool last = j + 1 == i;
      mtile_apply_whole<0, NC, KC>(acc, T, V, Lt, tid, -1.0, last ? Wi : Li + (long long)(j + 1) * TILE * ld, last ? (long long)TILE : (long long)ld);
   }
   if (ok) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < NC; ++c) { V[32 * w + 16 * h + er][4 * c + ek] = acc[h][c]; acc[h][c] = 0.0; }
      mtile_apply_whole<0, NC, KC>(acc, T, V, Lt, tid, 1.0, nullptr, 0);
   } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < NC; ++c) acc[h][c] = sweep_nan();
      if (tid == 0) *a.err = 1;
   }
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < NC; ++c) sweep_store(xt + (long long)(i * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek, acc[h][c]);
   sweep_publish(fl + i, epoch);
   sweep_done(a);
}

template <int NC>
__global__ __launch_bounds__(256, 2) void k_mtail_rows_bwd(SweepArgs a, const BlkDesc* __restrict__ blks, const double* __restrict__ arena,
                                                       const double* __restrict__ dtail, const double* __restrict__ winv, double* __restrict__ xm,
                                                       long long panel_stride) {
   __shared__ double V[TILE][mtile_mv<NC>()];
   constexpr int KC = NC <= 4 ? 32 : 16;   // (a slice of a panel leaves LDS for stages of 32 k: half the barriers)
   __shared__ double Lt[KC][TILE + 1];
   __shared__ int sh_t, sh_ok;
   const int epoch = *a.epoch_ptr;
   const int t = sweep_ticket(a, &sh_t);
   const TileTask task = a.tasks[a.n_tasks - 1 - t];   // last tile column first
   const BlkDesc bd = blks[task.blk];
   const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, er = lane & 15, ek = lane >> 4, i = task.ti, ld = bd.ldT;
   constexpr int SL = 8 / NC, QS = 4 * NC;   // slices of a panel, right-hand sides per slice
   double* xt = xm + panel_stride * (blockIdx.y / SL) + (bd.xw_off + bd.n_head) * MQ + QS * (blockIdx.y % SL);
   int* fl = a.flags + a.flag_stride * blockIdx.y + a.flag_off[task.blk];
   const int* tf = a.tfirst ? a.tfirst + a.tfirst_off[task.blk] : nullptr;
   double acc[2][NC], T[TILE / 2];
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[h][c] = xt[(long long)(i * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek];
   bool ok = true;
   const double* Wi = winv + bd.winv_off + (long long)i * TILE * TILE;
   const double* Ci = arena + bd.T + (long long)i * TILE * ld;   // tile column i
   // the tiles of column i inside the envelope, last tile row first; below(k) = the next one under k, i if none is left
   auto below = [&](int k) { for (--k; k > i; --k) if (!(tf && tf[k] > i)) return k; return i; };
   int k = below(bd.ntc);
   if (k > i) mtile_fetch_all<1, KC>(T, Ci + (long long)k * TILE, ld, tid);   // (no tile waits for a flag)
   else mtile_fetch_all<1, KC>(T, Wi, TILE, tid);
   while (k > i) {
      if (tid == 0) sh_ok = sweep_wait(fl + k, epoch, a.poll_limit) ? 1 : 0;
      __syncthreads();
      if (!sh_ok) { ok = false; break; }
      for (int idx = tid; idx < TILE * QS; idx += 256) V[idx / QS][idx % QS] = sweep_load(xt + (long long)(k * TILE + idx / QS) * MQ + idx % QS);
      const int kn = below(k);
      mtile_apply_whole<1, NC, KC>(acc, T, V, Lt, tid, -1.0, kn > i ? Ci + (long long)kn * TILE : Wi, kn > i ? (long long)ld : (long long)TILE);
      k = kn;
   }
   if (ok) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < NC; ++c) {
            V[32 * w + 16 * h + er][4 * c + ek] = acc[h][c] * dtail[bd.dt_off + i * TILE + 32 * w + 16 * h + er];
            acc[h][c] = 0.0;
         }
      mtile_apply_whole<1, NC, KC>(acc, T, V, Lt, tid, 1.0, nullptr, 0);
   } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
         for (int c = 0; c < NC; ++c) acc[h][c] = sweep_nan();
      if (tid == 0) *a.err = 1;
   }
#pragma unroll
   for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < NC; ++c) sweep_store(xt + (long long)(i * TILE + 32 * w + 16 * h + er) * MQ + 4 * c + ek, acc[h][c]);
   sweep_publish(fl + i, epoch);
   sweep_done(a);
}

// X(:, idx[r]) += a * Y(:, r): the refinement update of the right-hand sides that took a correction solve (their corrections side by side in Y)
__global__ void k_maxpy_idx(double* __restrict__ X, long long x_stride, const double* __restrict__ Y, long long y_stride, double a, long long n,
                            const int* __restrict__ idx) {
   X += x_stride * idx[blockIdx.y];
   Y += y_stride * blockIdx.y;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) X[i] += a * Y[i];
}

// X(:, r) += a * Y(:, r) for nr vectors (grid.y): the refinement update of all right-hand sides in one launch
__global__ void k_maxpy(double* __restrict__ X, long long x_stride, const double* __restrict__ Y, long long y_stride, double a, long long n) {
   X += x_stride * blockIdx.y;
   Y += y_stride * blockIdx.y;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) X[i] += a * Y[i];
}

// The measure of the adaptive refinement for several right-hand sides at once: grid (block, right-hand side); for its pair the workgroup forms
// ||r||inf, ||rhs||inf and ||x||inf over the block's rows and folds  ||r|| / (amax_scale * max|K_b| ||x|| + ||rhs||)  into the number of its
// right-hand side, worst[right-hand side], with an integer atomic max (non-negative doubles order like their bit patterns; a NaN or Inf counts as +Inf); amax_scale = 0:
// the denominator is ||rhs||inf alone (refine_mode 0).  blks[b].repl_abs = repl_rel * max|K_b| (k_block_absmax_finish).
__global__ __launch_bounds__(256) void k_mrefine_measure(const BlkDesc* __restrict__ blks, const double* __restrict__ R, long long r_stride,
                                                        const double* __restrict__ B, long long b_stride, const double* __restrict__ X,
                                                        long long x_stride, double amax_scale, double* __restrict__ worst) {
   const BlkDesc bd = blks[blockIdx.x];
   const double inf = __longlong_as_double(0x7ff0000000000000LL);
   const double* r = R + r_stride * blockIdx.y + bd.x_off;
   const double* b = B + b_stride * blockIdx.y + bd.x_off;
   const double* x = X + x_stride * blockIdx.y + bd.x_off;
   double m[3] = {0.0, 0.0, 0.0};
   for (int i = threadIdx.x; i < bd.n; i += 256) {
      const double v[3] = {fabs(r[i]), fabs(b[i]), fabs(x[i])};
#pragma unroll
      for (int q = 0; q < 3; ++q) m[q] = fmax(m[q], v[q] <= 1.7976931348623157e308 ? v[q] : inf);
   }
   __shared__ double red[3][256];
#pragma unroll
   for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = m[q];
   __syncthreads();
   for (int h = 128; h > 0; h >>= 1) {
      if ((int)threadIdx.x < h)
         for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = fmax(red[q][threadIdx.x], red[q][threadIdx.x + h]);
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      const double den = amax_scale * bd.repl_abs * red[2][0] + red[1][0];
      if (den > 0.0) {
         double q = red[0][0] / den;
         if (!(q <= 1.7976931348623157e308)) q = inf;   // (Inf / Inf = NaN: never "converged" on a poisoned iterate)
         if (q > 0.0) atomicMax((unsigned long long*)(worst + blockIdx.y), (unsigned long long)__double_as_longlong(q));
      }
   }
}

// refinement residual r = b - K x (r holds b on entry): the full (both triangles) row structure is built at analyze time -
// frowptr / fcol (block-local column) / fsrc (index of the value inside kval) - so the product is gather-only, no atomics:
//   y_i -= sum_j K_ij x_j
constexpr int FULL_LONG_ROW = 512;

// y0 != nullptr: y = y0 - K x (every row must then be short: the caller checks that no row goes to k_full_spmv_sub_long)
__global__ void k_full_spmv_sub(const int* __restrict__ frowptr, const int* __restrict__ fcol, const int* __restrict__ fsrc,
                                const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
                                long long nrows_total, const long long* __restrict__ row_blk_base, long long vec_stride,
                                const double* __restrict__ y0 = nullptr) {
   x += vec_stride * blockIdx.y;
   y += vec_stride * blockIdx.y;
   // eight lanes per row (KKT rows hold a handful of entries): a wave reads the entries of eight consecutive rows as one contiguous piece.
   // Four rows per lane group and trip: a row is four dependent memory round trips (row pointers, entries, x, y) around a handful of
   // multiply-adds - the kernel is bound by how many of those are in flight, not by bytes (reading the values through fsrc or from
   // a copy in row order makes no difference: measured)
   constexpr int RU = 4;
   const int l = threadIdx.x & 7;
   const long long step = ((long long)gridDim.x * blockDim.x) >> 3, chunk = step * RU, i_end = (nrows_total + chunk - 1) / chunk * chunk;
   for (long long i0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 3; i0 < i_end; i0 += chunk) {
      int p0[RU], p1[RU];
      long long base[RU];
      double s[RU];
      bool mine[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
         const long long i = i0 + u * step;
         mine[u] = i < nrows_total;
         p0[u] = mine[u] ? frowptr[i] : 0;
         p1[u] = mine[u] ? frowptr[i + 1] : 0;
         base[u] = mine[u] ? row_blk_base[i] : 0;
         mine[u] = mine[u] && p1[u] - p0[u] <= FULL_LONG_ROW;   // the others: k_full_spmv_sub_long
         s[u] = 0.0;
      }
      // the first eight entries of the four rows without control flow between their loads, the rest (rows beyond eight entries) after
      double v[RU];
      int c[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
         const int p = p0[u] + l;
         const bool in = mine[u] && p < p1[u];
         const int q = in ? fsrc[p] : 0;
         c[u] = in ? fcol[p] : -1;
         v[u] = in ? val[q] : 0.0;
      }
      // (inactive lanes contribute an exact zero, not 0 * x[...]: a non-finite entry at the head of a block's x must not leak into the
      // residual of every short row of the block)
#pragma unroll
      for (int u = 0; u < RU; ++u) s[u] = c[u] >= 0 ? v[u] * x[base[u] + c[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < RU; ++u)
         if (mine[u])
            for (int p = p0[u] + l + 8; p < p1[u]; p += 8) s[u] += val[fsrc[p]] * x[base[u] + fcol[p]];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
         s[u] += __shfl_xor(s[u], 1);
         s[u] += __shfl_xor(s[u], 2);
         s[u] += __shfl_xor(s[u], 4);
         if (mine[u] && l == 0) y[i0 + u * step] = (y0 ? y0[i0 + u * step] : y[i0 + u * step]) - s[u];
      }
   }
}

// The measure of a solveCompressed by sweeps in ONE pass (KktSystem: every such solve is measured): for the rows i of block b
//    rhs_i = b_i - (Br x0)_i,   r_i = rhs_i - (K x)_i,   out[b] = max |r_i|,  out[nblk + b] = max |rhs_i|,  out[2 nblk + b] = max |x_i|
// - what a copy of the right-hand side, the border product (k_border_mult_rows), k_full_spmv_sub and three k_vec_block_absmax passes
// computed in six launches over the same vectors (1.2 ms per solve on the 256 x 50 000 time-coupled blocks, 0.3 of them those passes).
// grid (chunks, block), tiles of 256 rows; every row must be short (the caller checks n_flong == 0); a NaN counts as +Inf.  out[] zeroed
// by the caller.
__global__ __launch_bounds__(256) void k_measure_leaf_rows(const BlkDesc* __restrict__ blks, const int* __restrict__ frowptr, const int* __restrict__ fcol,
                                                           const int* __restrict__ fsrc, const double* __restrict__ val, const double* __restrict__ x,
                                                           const double* __restrict__ b, const int* __restrict__ br_rowptr, const int* __restrict__ br_sc,
                                                           const int* __restrict__ br_src, const double* __restrict__ bval, const double* __restrict__ x0,
                                                           double* __restrict__ out, int nblk) {
   // Rows are short (6 entries on the time-coupled blocks): with a few lanes per row every round of 32 rows is a chain of four dependent loads
   // (row pointers -> indices -> values and x -> the row's b and x) and the waves wait 91 % of their cycles (SQ_WAIT_ANY).  Here a tile of 256
   // rows streams ALL its entries through LDS - the threads take the tile's entries side by side (coalesced index reads, one round of gathers
   // for the whole tile) - and then every thread adds up the segment of its own row: one chain per 256 rows instead of one per 32.
   constexpr int MCH = 2048;                       // entries of a tile handled at once
   __shared__ double prod[MCH];
   const BlkDesc bd = blks[blockIdx.y];
   const int tid = threadIdx.x;
   const double inf = __longlong_as_double(0x7ff0000000000000LL);
   double mr = 0.0, mb = 0.0, mx = 0.0;
   for (int k0 = blockIdx.x * 256; k0 < bd.n; k0 += gridDim.x * 256) {
      const int k = k0 + tid;
      const bool row = k < bd.n;
      const long long i = bd.x_off + (row ? k : bd.n - 1);
      const int p0 = frowptr[i], p1 = row ? frowptr[i + 1] : p0;
      const int q0 = br_rowptr ? br_rowptr[i] : 0, q1 = (br_rowptr && row) ? br_rowptr[i + 1] : q0;
      const double bi = b[i], xi0 = x[i];
      const int p_lo = frowptr[bd.x_off + k0], p_hi = frowptr[bd.x_off + min(k0 + 256, bd.n)];
      double s = 0.0, t = 0.0;   // (K x)_i and (Br x0)_i
      for (int q = q0; q < q1; ++q) t += bval[br_src[q]] * x0[br_sc[q]];
      for (int c0 = p_lo; c0 < p_hi; c0 += MCH) {
         const int c1 = min(c0 + MCH, p_hi);
         for (int q = c0 + tid; q < c1; q += 256) prod[q - c0] = val[fsrc[q]] * x[bd.x_off + fcol[q]];
         __syncthreads();
         for (int q = max(p0, c0); q < min(p1, c1); ++q) s += prod[q - c0];
         __syncthreads();
      }
      if (row) {
         const double rhs = bi - t, r = fabs(rhs - s), ar = fabs(rhs), xi = fabs(xi0);
         mr = fmax(mr, r <= 1.7976931348623157e308 ? r : inf);
         mb = fmax(mb, ar <= 1.7976931348623157e308 ? ar : inf);
         mx = fmax(mx, xi <= 1.7976931348623157e308 ? xi : inf);
      }
   }
   __shared__ double red[3][256];
   red[0][threadIdx.x] = mr; red[1][threadIdx.x] = mb; red[2][threadIdx.x] = mx;
   __syncthreads();
   for (int h = 128; h > 0; h >>= 1) {
      if ((int)threadIdx.x < h)
         for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = fmax(red[q][threadIdx.x], red[q][threadIdx.x + h]);
      __syncthreads();
   }
   if (threadIdx.x < 3 && red[threadIdx.x][0] > 0.0)
      atomicMax((unsigned long long*)(out + threadIdx.x * nblk + blockIdx.y), (unsigned long long)__double_as_longlong(red[threadIdx.x][0]));
}

// the long rows (the dense x0 rows of a sparse Schur complement factorised as a one-block system): one workgroup per row
__global__ __launch_bounds__(256) void k_full_spmv_sub_long(const long long* __restrict__ long_rows, const int* __restrict__ frowptr,
                                                           const int* __restrict__ fcol, const int* __restrict__ fsrc,
                                                           const double* __restrict__ val, const double* __restrict__ x,
                                                           double* __restrict__ y, const long long* __restrict__ row_blk_base,
                                                           long long vec_stride) {
   __shared__ double red[256];
   x += vec_stride * blockIdx.y;
   y += vec_stride * blockIdx.y;
   const long long i = long_rows[blockIdx.x];
   const long long base = row_blk_base[i];
   double s = 0.0;
   for (int p = frowptr[i] + threadIdx.x; p < frowptr[i + 1]; p += 256) s += val[fsrc[p]] * x[base + fcol[p]];
   red[threadIdx.x] = s;
   __syncthreads();
   for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
   }
   if (threadIdx.x == 0) y[i] -= red[0];
}

// ------------------------------------------------------------------------------------------------
// border products (K12 / K15):  b0 -= sum_i Br_i^T z_i   and   t_i = Br_i x0
// Bt is the CSR of Br_i^T (rows = Schur column ids), concatenated over blocks.
// ------------------------------------------------------------------------------------------------
// Sixteen lanes per row of Bt: its rows are very uneven (time-coupled blocks: 95 first-stage columns with ~500 entries each beside 62
// linking rows with three, and most (block, Schur column) pairs empty) - a thread per row walked the long rows alone, entry by entry.
constexpr int BT_LANES = 16;

__global__ void k_border_tmult(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                               const double* __restrict__ val, const int* __restrict__ row_sc,
                               const long long* __restrict__ row_xoff, const double* __restrict__ z,
                               double* __restrict__ b0, long long nrows, double alpha) {
   const int l = threadIdx.x & (BT_LANES - 1);
   const long long step = ((long long)gridDim.x * blockDim.x) / BT_LANES, i_end = (nrows + step - 1) / step * step;   // whole waves (shuffles)
   for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) / BT_LANES; i < i_end; i += step) {
      double s = 0.0;
      if (i < nrows) {
         const int p1 = rowptr[i + 1];
         const long long xo = row_xoff[i];
         for (int p = rowptr[i] + l; p < p1; p += BT_LANES) s += val[p] * z[xo + colidx[p]];
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      if (i < nrows && l == 0 && s != 0.0) atomic_add_f64(b0 + row_sc[i], alpha * s);
   }
}

// deterministic mode: the same two products without atomics - per-row dot products / per-entry products into a scratch vector,
// then k_gather_slots adds them up per Schur column / per leaf entry in a fixed order
__global__ void k_border_rowdot(const int* __restrict__ rowptr, const int* __restrict__ colidx, const double* __restrict__ val,
                                const long long* __restrict__ row_xoff, const double* __restrict__ z, double* __restrict__ out,
                                long long nrows, double alpha) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nrows; i += (long long)gridDim.x * blockDim.x) {
      double s = 0.0;
      const long long xo = row_xoff[i];
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * z[xo + colidx[p]];
      out[i] = alpha * s;
   }
}
__global__ void k_border_entry_products(const int* __restrict__ rowptr, const double* __restrict__ val, const int* __restrict__ row_sc,
                                        const double* __restrict__ x0, double* __restrict__ out, long long nrows, double alpha) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nrows; i += (long long)gridDim.x * blockDim.x) {
      const double xs = alpha * x0[row_sc[i]];
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) out[p] = val[p] * xs;
   }
}

// t += alpha Br x0 from the side of the leaf rows (Engine::d_br_rowptr: the border transposed once at analyze time): a thread per row
// sums its few entries in a fixed order - no atomics
__global__ void k_border_mult_rows(const int* __restrict__ rowptr, const int* __restrict__ sc, const int* __restrict__ src,
                                   const double* __restrict__ val, const double* __restrict__ x0, double* __restrict__ t, long long nrows,
                                   double alpha) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nrows; i += (long long)gridDim.x * blockDim.x) {
      const int p0 = rowptr[i], p1 = rowptr[i + 1];
      if (p0 == p1) continue;
      double s = 0.0;
      for (int p = p0; p < p1; ++p) s += val[src[p]] * x0[sc[p]];
      t[i] += alpha * s;
   }
}

__global__ void k_border_mult(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                              const double* __restrict__ val, const int* __restrict__ row_sc,
                              const long long* __restrict__ row_xoff, const double* __restrict__ x0,
                              double* __restrict__ t, long long nrows, double alpha) {
   const int l = threadIdx.x & (BT_LANES - 1);
   const long long step = ((long long)gridDim.x * blockDim.x) / BT_LANES;
   for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) / BT_LANES; i < nrows; i += step) {
      const int p0 = rowptr[i], p1 = rowptr[i + 1];
      if (p0 == p1) continue;
      const double xs = alpha * x0[row_sc[i]];
      if (xs == 0.0) continue;
      const long long xo = row_xoff[i];
      for (int p = p0 + l; p < p1; p += BT_LANES) atomic_add_f64(t + xo + colidx[p], val[p] * xs);
   }
}

// Blocked Schur path (the reference's K4-K6: addTermToSchurComplBlocked, DistributedLeafLinearSystem.C:214-252 +
// DistributedLinearSystem.C:766-1047): a chunk of border columns is densified, solved with all blocks at once (one
// right-hand side = the column's entries in every block), and multiplied back with the sparse border.
//   slot[sc] = position of Schur column sc among the non-empty columns; the chunk holds positions [c0, c0 + nr)
__global__ void k_border_rows_to_dense(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                       const double* __restrict__ val, const int* __restrict__ row_sc,
                                       const long long* __restrict__ row_xoff, const int* __restrict__ slot, int c0, int nr,
                                       double* __restrict__ R, long long r_stride, long long nrows) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nrows; i += (long long)gridDim.x * blockDim.x) {
      const int r = slot[row_sc[i]] - c0;
      if (r < 0 || r >= nr) continue;
      double* dst = R + r * r_stride + row_xoff[i];
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) dst[colidx[p]] = val[p];
   }
}

// SC(s', col_r) -= Br_b^T(s', :) X_r  for every border row s' of every block and every right-hand side r of the chunk
// (addLeftBorderTimesDenseColsToResTranspDense, DistributedLinearSystem.C:1115-1175); lower triangle only
__global__ void k_border_tmult_chunk(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                     const double* __restrict__ val, const int* __restrict__ row_sc,
                                     const long long* __restrict__ row_xoff, const int* __restrict__ chunk_cols, int nr,
                                     const double* __restrict__ X, long long x_stride, double* __restrict__ SC, int ldSC,
                                     long long nrows) {
   const int r = blockIdx.y;
   if (r >= nr) return;
   const int col = chunk_cols[r];
   const double* x = X + r * x_stride;
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nrows; i += (long long)gridDim.x * blockDim.x) {
      const int srow = row_sc[i];
      if (srow < col || rowptr[i] == rowptr[i + 1]) continue;
      const long long xo = row_xoff[i];
      double s = 0.0;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * x[xo + colidx[p]];
      if (s != 0.0) atomic_add_f64(SC + srow + (long long)col * ldSC, -s);
   }
}

// dense helpers for the root system
__global__ void k_copy_lower_to_padded(const double* __restrict__ src, int lds, int n, double* __restrict__ dst, int ldd,
                                       int npad, int rowmajor, const int* __restrict__ perm = nullptr) {
   // dst (col-major, npad x npad) lower := src lower (col-major, symmetric storage with the lower triangle authoritative);
   // identity on the padding.  perm != nullptr: dst = P src P^T, dst(r, c) = src(perm[r], perm[c])
   for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < (long long)npad * npad;
        idx += (long long)gridDim.x * blockDim.x) {
      const int r = (int)(idx % npad), c = (int)(idx / npad);
      double v = 0.0;
      if (r < n && c < n) {
         if (r >= c) {
            int sr = r, sc = c;
            if (perm) { const int a = perm[r], b = perm[c]; sr = a > b ? a : b; sc = a > b ? b : a; }
            v = rowmajor ? src[(long long)sr * lds + sc] : src[sr + (long long)sc * lds];
         }
      }
      else if (r == c) v = 1.0;
      dst[r + (long long)c * ldd] = v;
   }
}

// Growth check of a tile column under tile-bounded Bunch-Kaufman pivoting (DenseLdl::check_pivots): with the search of dsytrf (the
// whole column, DeSymIndefSolver.C:78) no multiplier exceeds 1 / alpha = 1.56 (2.57 around a 2 x 2 pivot); a pivot chosen inside the
// tile while the column's weight lies below it - the rounding noise of a numerically rank-deficient leading block taken for a pivot -
// shows as multipliers of 1e10 in the rows below.  One workgroup per column of tile column tj: max |L(i, p)| over the rows below the
// tile; beyond `limit` the index p is recorded like one without a pivot, its partner the row of that largest multiplier.
__global__ __launch_bounds__(256) void k_bk_growth(const BlkDesc* __restrict__ blks, const double* __restrict__ arena, int tj, double limit,
                                                  int* __restrict__ pert_cnt, int* __restrict__ pert_list) {
   __shared__ double s_best[256];
   __shared__ int s_idx[256];
   const BlkDesc bd = blks[0];
   const int tid = threadIdx.x, p = tj * TILE + blockIdx.x;
   if (p >= bd.m) return;
   const double* col = arena + bd.T + (long long)p * bd.ldT;
   double best = 0.0;
   int bi = -1;
   for (int i = (tj + 1) * TILE + tid; i < bd.m; i += 256) { const double v = fabs(col[i]); if (v > best || !(v == v)) { best = v == v ? v : 1e308; bi = i; } }
   s_best[tid] = best; s_idx[tid] = bi;
   __syncthreads();
   for (int h = 128; h > 0; h >>= 1) {
      if (tid < h && s_best[tid + h] > s_best[tid]) { s_best[tid] = s_best[tid + h]; s_idx[tid] = s_idx[tid + h]; }
      __syncthreads();
   }
   if (tid == 0 && s_best[0] > limit) { const int slot = atomicAdd(pert_cnt, 1); if (slot < bd.m_pad) { pert_list[2 * slot] = p; pert_list[2 * slot + 1] = s_idx[0]; } }
}

// |A(perm[i], perm[c_q])| for the columns c_q = list[2 q] DenseLdl::check_pivots looks for partners for (A: the caller's symmetric matrix,
// lower triangle authoritative): out[q * n + i]
__global__ void k_bk_gather_columns(const double* __restrict__ A, int lda, int rowmajor, const int* __restrict__ perm, int n, const int* __restrict__ list,
                                    int n_cols, double* __restrict__ out) {
   const int q = blockIdx.y;
   if (q >= n_cols) return;
   const int c = list[2 * q], pc = perm ? perm[c] : c;
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int pi = perm ? perm[i] : i, hi = pi > pc ? pi : pc, lo = pi > pc ? pc : pi;
      out[(long long)q * n + i] = fabs(rowmajor ? A[(long long)hi * lda + lo] : A[hi + (long long)lo * lda]);
   }
}

// work vector of a dense solve under the symmetric permutation of DenseLdl: xw[i] = x[perm[i]] / x[perm[i]] = xw[i]
__global__ void k_perm_gather(const int* __restrict__ perm, int n, const double* __restrict__ x, double* __restrict__ xw, int back, double* __restrict__ xo) {
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      if (!back) xw[i] = x[perm[i]]; else xo[perm[i]] = xw[i];
   }
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, double a, long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      y[i] += a * x[i];
}

__global__ void k_add_entries(double* __restrict__ M, const long long* __restrict__ idx, const double* __restrict__ v,
                              long long n) {
   for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      atomic_add_f64(M + idx[i], v[i]);
}

}  // namespace pips
