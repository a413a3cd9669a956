// Symbolic analysis of one leaf KKT block K_i together with its border Br_i.
//
// Runs once per block on the host (the sparsity pattern is invariant across IPM iterations: the leaf solver is
// constructed once around a non-owning pointer to K_i, DistributedLeafLinearSystem.C:10-42, PardisoSolver.h:49-50).
// It produces everything the device kernels need so that a numeric factorisation only moves values:
//   * a constrained AMD ordering (order.cpp),
//   * the split of the permuted matrix into a sparse "head" (supernodal, LDS-staged panels, scattered updates) and a
//     dense "tail" (tiled MFMA LDL^T) chosen by a cost model,
//   * the border columns of Br_i appended as extra rows of every panel (the augmented partial factorisation that
//     PardisoSchurSolver.C:83-389 obtains from PARDISO's iparm[37]; here it yields  -Br_i^T K_i^-1 Br_i  directly
//     in the trailing S x S block),
//   * scatter maps from CSR entries of K_i / Br_i^T into the factor arena.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.h"

namespace pips {

static inline int round_up(int x, int t) { return (x + t - 1) / t * t; }

int analyze_block(const CsrPattern& K, const CsrPattern& border, int n_primal, const AnalyzeOptions& opt,
                  BlockSym& out) {
   const int n = K.nrows;
   if (n <= 0 || K.ncols != n) PIPS_FAIL(PIPS_ERR_ARG, "analyze_block: K must be square, got %d x %d", K.nrows, K.ncols);
   if (border.nrows > 0 && border.ncols != n)
      PIPS_FAIL(PIPS_ERR_ARG, "analyze_block: border has %d columns, K has %d rows", border.ncols, n);
   out = BlockSym();
   out.n = n;
   out.n_primal = n_primal;

   // ---- full symmetric adjacency without the diagonal
   std::vector<int> ap(n + 1, 0), ai;
   for (int i = 0; i < n; ++i)
      for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
         const int j = K.colidx[p];
         if (j < 0 || j > i) PIPS_FAIL(PIPS_ERR_ARG, "analyze_block: K must be lower-triangular CSR (row %d col %d)", i, j);
         if (j != i) { ++ap[i + 1]; ++ap[j + 1]; }
      }
   for (int i = 0; i < n; ++i) ap[i + 1] += ap[i];
   ai.resize(ap[n]);
   {
      std::vector<int> fill(ap.begin(), ap.end() - 1);
      for (int i = 0; i < n; ++i)
         for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
            const int j = K.colidx[p];
            if (j != i) { ai[fill[i]++] = j; ai[fill[j]++] = i; }
         }
   }
   if (opt.user_perm) {
      out.perm.assign(opt.user_perm, opt.user_perm + n);
      if (opt.user_colcount) out.colcount.assign(opt.user_colcount, opt.user_colcount + n);
      else out.colcount.assign(n, 0);
   } else if (!opt.constrain_order)
      constrained_amd(n, ap, ai, -1, out.perm, out.colcount);
   else if (!dissected_order(n, ap, ai, n_primal, opt.nd_depth, out.perm, out.colcount, opt.nd_min_size))
      constrained_amd(n, ap, ai, n_primal, out.perm, out.colcount);
   out.iperm.assign(n, 0);
   for (int k = 0; k < n; ++k) out.iperm[out.perm[k]] = k;
   out.psign.assign(n, 0);
   if (n_primal >= 0)
      for (int k = 0; k < n; ++k) out.psign[k] = out.perm[k] < n_primal ? 1 : -1;

   // ---- compressed border columns (the reference skips empty border columns too: DistributedLinearSystem.C:870-874)
   std::vector<int> bidx(border.nrows, -1);
   for (int s = 0; s < border.nrows; ++s)
      if (border.rowptr[s + 1] > border.rowptr[s]) { bidx[s] = (int)out.bmap.size(); out.bmap.push_back(s); }
   const int nb = out.nb = (int)out.bmap.size();

   // ---- head / tail cut from the cost model
   int n_head = n;
   if (opt.force_n_head >= 0) {
      n_head = std::min(opt.force_n_head, n);
   } else {
      std::vector<double> head(n + 1, 0.0);
      for (int k = 0; k < n; ++k) {
         const double c = out.colcount[k] + 1.0;
         head[k + 1] = head[k] + opt.head_cost * 0.5 * c * c;
      }
      double best = head[n];
      for (int k = 0; k <= n; ++k) {
         if (n - k < opt.min_tail) break;
         // the tail is processed in full tiles: price the padded dimension, so the cut lands just below a tile boundary
         const double m = round_up(n - k, opt.tile);
         const double tail = (m * m * m / 3.0 + (double)nb * m * m + (double)nb * nb * m) / opt.mfma_rate;
         if (head[k] + tail < best) { best = head[k] + tail; n_head = k; }
      }
   }
   const int m = n - n_head;
   out.n_head = n_head;
   out.m = m;
   out.m_pad = m > 0 ? round_up(m, opt.tile) : 0;
   out.nb_pad = round_up(nb, opt.tile);
   out.ldT = out.m_pad + out.nb_pad;

   // ---- column structures of the head (merge children) + elimination-tree parents, for the current permutation
   std::vector<std::vector<int>> S;
   std::vector<int> parent;
   auto build_structures = [&]() {
      // permuted strict-lower pattern by column (only columns < n_head are needed) + border rows per column
      std::vector<int> cp(n_head + 1, 0);
      auto for_each_head_entry = [&](auto&& f) {
         for (int i = 0; i < n; ++i)
            for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
               const int j = K.colidx[p];
               if (j == i) continue;
               const int a = out.iperm[i], b = out.iperm[j];
               const int c = std::min(a, b), r = std::max(a, b);
               if (c < n_head) f(c, r);
            }
         for (int s = 0; s < border.nrows; ++s)
            for (int p = border.rowptr[s]; p < border.rowptr[s + 1]; ++p) {
               const int c = out.iperm[border.colidx[p]];
               if (c < n_head) f(c, n + bidx[s]);
            }
      };
      for_each_head_entry([&](int c, int) { ++cp[c + 1]; });
      for (int j = 0; j < n_head; ++j) cp[j + 1] += cp[j];
      std::vector<int> ci(cp[n_head]);
      {
         std::vector<int> fill(cp.begin(), cp.end() - 1);
         for_each_head_entry([&](int c, int r) { ci[fill[c]++] = r; });
      }
      S.assign(n_head, {});
      parent.assign(n_head, -1);
      std::vector<int> first_child(n_head, -1), next_sib(n_head, -1), mark(n + nb, -1);
      for (int j = 0; j < n_head; ++j) {
         auto& Sj = S[j];
         for (int p = cp[j]; p < cp[j + 1]; ++p) {
            const int r = ci[p];
            if (mark[r] != j) { mark[r] = j; Sj.push_back(r); }
         }
         for (int c = first_child[j]; c >= 0; c = next_sib[c])
            for (int r : S[c])
               if (r != j && mark[r] != j) { mark[r] = j; Sj.push_back(r); }
         std::sort(Sj.begin(), Sj.end());
         if (!Sj.empty() && Sj[0] < n_head) {
            parent[j] = Sj[0];
            next_sib[j] = first_child[Sj[0]];
            first_child[Sj[0]] = j;
         }
      }
   };
   build_structures();

   // ---- postorder the head forest so that every parent/child chain is contiguous (AMD does not guarantee it); this
   //      relabels columns inside the head only: the fill, the cut and the primal-before-dual constraint are unchanged
   //      (a dual row is an ancestor of all its primal neighbours, and a postorder keeps descendants first)
   if (n_head > 0) {
      std::vector<int> first_child(n_head, -1), next_sib(n_head, -1), post;
      post.reserve(n_head);
      for (int j = n_head - 1; j >= 0; --j)   // reversed so that children lists are ascending
         if (parent[j] >= 0) { next_sib[j] = first_child[parent[j]]; first_child[parent[j]] = j; }
      std::vector<int> stack;
      for (int root = 0; root < n_head; ++root) {
         if (parent[root] >= 0) continue;
         stack.push_back(root);
         while (!stack.empty()) {
            const int v = stack.back();
            const int c = first_child[v];
            if (c >= 0) { first_child[v] = next_sib[c]; stack.push_back(c); }
            else { post.push_back(v); stack.pop_back(); }
         }
      }
      bool identity = true;
      for (int k = 0; k < n_head; ++k) if (post[k] != k) { identity = false; break; }
      if (!identity) {
         std::vector<int> np(out.perm.begin(), out.perm.begin() + n_head), nc(n_head);
         for (int k = 0; k < n_head; ++k) { np[k] = out.perm[post[k]]; nc[k] = out.colcount[post[k]]; }
         std::copy(np.begin(), np.end(), out.perm.begin());
         std::copy(nc.begin(), nc.end(), out.colcount.begin());
         for (int k = 0; k < n; ++k) out.iperm[out.perm[k]] = k;
         if (n_primal >= 0)
            for (int k = 0; k < n; ++k) out.psign[k] = out.perm[k] < n_primal ? 1 : -1;
         build_structures();
      }
   }

   // ---- supernodes: fundamental chains, relaxed by a bounded share of explicit zeros (width-capped).  Along a chain
   //      (parent[e] == e+1) the below-rows of column e are contained in those of e+1, so the panel of a merged supernode
   //      has the row set of its last column and a column that lacks some of these rows just stores zeros there.
   out.sn_of_col.assign(n_head, -1);
   for (int j = 0; j < n_head;) {
      int e = j;
      int64_t true_nnz = (int64_t)S[j].size() + 1;
      while (e + 1 < n_head && e + 1 - j < opt.max_sn_width && parent[e] == e + 1) {
         const int64_t w1 = e + 2 - j;
         const int64_t stored = w1 * (w1 + 1) / 2 + w1 * (int64_t)S[e + 1].size();
         const int64_t tn = true_nnz + (int64_t)S[e + 1].size() + 1;
         if ((double)(stored - tn) > opt.relax_zeros * (double)stored) break;
         true_nnz = tn;
         ++e;
      }
      HeadSupernode sn;
      sn.c0 = j;
      sn.w = e - j + 1;
      sn.r = (int)S[e].size();
      sn.level = 0;
      sn.panel = 0;
      sn.upd = 0;
      sn.n_useg = 0;
      sn.rb = 0;
      sn.rows = (int64_t)out.rowidx.size();
      out.rowidx.insert(out.rowidx.end(), S[e].begin(), S[e].end());
      for (int c = j; c <= e; ++c) out.sn_of_col[c] = (int)out.sn.size();
      out.sn.push_back(sn);
      j = e + 1;
   }
   // levels (children precede parents in column order)
   int n_levels = 0;
   for (auto& sn : out.sn) {
      n_levels = std::max(n_levels, sn.level + 1);
      const int p = parent[sn.c0 + sn.w - 1];
      if (p >= 0) {
         auto& ps = out.sn[out.sn_of_col[p]];
         ps.level = std::max(ps.level, sn.level + 1);
      }
   }
   out.n_levels = n_levels;
   { std::vector<std::vector<int>>().swap(S); }

   // ---- row envelope of the tail at tile granularity: the first tile column that carries an entry of the tail's Schur
   //      complement in each tile row - original entries of K inside the tail, and the cliques the head supernodes leave on
   //      their tail rows.  Factorisation without pivoting fills only inside the row envelope, so tile tasks left of it are
   //      never generated (TailPlan); a dense tail simply has first = 0 everywhere.  Border rows are kept dense.
   {
      const int ntr = out.ldT / opt.tile, ntc = out.m_pad / opt.tile;
      out.tile_first.assign(ntr, 0);
      for (int t = 0; t < ntc; ++t) out.tile_first[t] = t;   // at least the diagonal tile
      auto touch = [&](int r_perm, int c_perm) {             // both >= n_head, r >= c
         const int tr = (r_perm - n_head) / opt.tile, tc = (c_perm - n_head) / opt.tile;
         if (tc < out.tile_first[tr]) out.tile_first[tr] = tc;
      };
      for (int i = 0; i < n; ++i)
         for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
            const int a = out.iperm[i], b = out.iperm[K.colidx[p]];
            if (a >= n_head && b >= n_head) touch(std::max(a, b), std::min(a, b));
         }
      for (const HeadSupernode& sn : out.sn) {
         const int* rows = out.rowidx.data() + sn.rows;
         int first_tail = -1;
         for (int a = 0; a < sn.r; ++a) {
            const int ra = rows[a];
            if (ra < n_head || ra >= n) continue;
            if (first_tail < 0) first_tail = ra;
            touch(ra, first_tail);
         }
      }
   }

   // ---- arena layout
   int64_t off = 0;
   double fl = 0;
   int64_t nnzL = 0;
   // border split (see "multifrontal metadata" below): a front keeps its border rows only in the border-row arena, its panel holds
   // the w + rb rows of K - compact, so that the solve sweeps do not drag the border rows' cache lines along
   out.mf_split = nb > 0 && nb <= opt.mf_split_nb_max;
   out.mf_konly = out.mf_split && opt.mf_konly;
   for (auto& sn : out.sn) {
      const int* rows = out.rowidx.data() + sn.rows;
      sn.rb = (int)(std::lower_bound(rows, rows + sn.r, n) - rows);
      const bool simple = sn.w == 1 && sn.r <= opt.simple_rmax && sn.level == 0;
      sn.ld = sn.w + ((out.mf_split && !simple) ? sn.rb : sn.r);
      sn.panel = off;
      off += (int64_t)sn.ld * sn.w;
      const double w = sn.w, r = sn.r;
      fl += w * w * w / 3.0 + w * w * r + w * r * r;
      nnzL += (int64_t)sn.w * (sn.w + 1) / 2 + (int64_t)sn.r * sn.w;
   }
   off = (off + 15) / 16 * 16;
   out.T_off = off;
   off += (int64_t)out.ldT * out.m_pad;
   out.arena = (off + 15) / 16 * 16;
   {
      const double dm = m, dn = nb;
      fl += dm * dm * dm / 3.0;
      nnzL += (int64_t)m * (m + 1) / 2;
      out.flops_border = dn * dm * dm + dn * dn * dm;
   }
   out.flops_factor = fl;
   out.nnzL = nnzL;

   // ---- update segments: where the Schur update of a head supernode lands inside later HEAD supernodes.  The below-rows
   //      that are head columns split into runs belonging to one target supernode each; per run one record
   //         { b0, b1, target c0, target ld, target panel offset (lo, hi), npos, 0 }  followed by npos = r - b0 ints,
   //      the position inside a target panel column of every below-row a >= b0 (all of them occur there: the structure
   //      of a column is contained in its parent's).  The device kernel then scatters without searching.
   for (auto& sn : out.sn) {
      sn.upd = (int64_t)out.upd.size();
      sn.n_useg = 0;
      const int* rows = out.rowidx.data() + sn.rows;
      sn.rb = (int)(std::lower_bound(rows, rows + sn.r, n) - rows);
      int b0 = 0;
      while (b0 < sn.r && rows[b0] < n_head) {
         const HeadSupernode& tg = out.sn[out.sn_of_col[rows[b0]]];
         int b1 = b0 + 1;
         while (b1 < sn.r && rows[b1] < tg.c0 + tg.w) ++b1;
         const int npos = sn.r - b0;
         const int hdr[8] = {b0, b1, tg.c0, tg.ld, (int)(uint32_t)(tg.panel & 0xffffffffLL), (int)(tg.panel >> 32), npos, 0};
         out.upd.insert(out.upd.end(), hdr, hdr + 8);
         const int* trows = out.rowidx.data() + tg.rows;
         int q = 0;
         for (int a = b0; a < sn.r; ++a) {
            const int ra = rows[a];
            if (ra < tg.c0 + tg.w) { out.upd.push_back(ra - tg.c0); continue; }
            while (q < tg.r && trows[q] < ra) ++q;
            if (q == tg.r || trows[q] != ra)
               PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, row %d of supernode %d missing in its ancestor %d", ra, sn.c0, tg.c0);
            out.upd.push_back(tg.w + q);
         }
         ++sn.n_useg;
         b0 = b1;
      }
   }

   // ---- multifrontal metadata.  A head supernode that is not a simple leaf is a front: the (w + r) x (w + r) lower triangle on its
   //      columns and below-rows.  Its update matrix (the r x r trailing part after the w pivots) is handed to the parent front -
   //      the supernode that holds its first below-row - which adds it at the positions recorded here ("extend-add"); simple
   //      leaves (width 1, no children, a handful of rows: every primal column of an LP block) are not fronts of their own, the
   //      parent front forms their rank-one updates itself from the leaf's column.  Nothing is scattered with atomics, the order
   //      of the additions is fixed by these lists.  Fronts without a head parent (parent column in the dense tail) scatter their
   //      update matrix into the tail / Schur complement as before; so do simple leaves without a head parent.
   {
      const int nsn = (int)out.sn.size();
      out.sn_parent.assign(nsn, -1);
      out.mf_U.assign(nsn, -1);
      out.mf_meta.assign(nsn, -1);
      out.mf_U_total = 0;
      out.mf_LV_total = 0;
      out.mf_max_front = 0;
      out.mf_ok = true;
      auto is_simple = [&](const HeadSupernode& sn) { return sn.w == 1 && sn.r <= opt.simple_rmax && sn.level == 0; };
      // Border split: the border x border part of an update matrix is needed by no front above - it is a partial sum of the block's
      // Schur contribution that would only travel from front to front (time-coupled blocks: 64 % of the update-matrix volume, fronts of
      // ~17 rows of K carrying ~100 border rows).  With the split a front keeps only the update columns that belong to rows of K
      // (uc = rb of them, each with all r rows below: K x K and border x K); the border x border part is formed once per block from the
      // finished panels, -sum_J L_b(J) D_J L_b(J)^T (k_border_schur).  Taken where that kernel's accumulator fits the LDS.
      auto ucols = [&](const HeadSupernode& sn) { return out.mf_split ? sn.rb : sn.r; };
      // rows a front holds below its pivot block: all of them, or (mf_konly) the rows of K only - the border rows of such a front are formed
      // afterwards, row-parallel, from the finished panels of its descendants (k_border_rows); what is handed from front to front is then
      // the K x K part alone, and a front is the size its rows of K give it (time-coupled blocks: ~33 rows instead of ~130)
      auto frows = [&](const HeadSupernode& sn) { return out.mf_konly ? sn.rb : sn.r; };
      std::vector<std::vector<int>> kids(nsn), leaves(nsn);
      for (int s = 0; s < nsn; ++s) {
         const HeadSupernode& sn = out.sn[s];
         if (sn.r > 0 && out.rowidx[sn.rows] < n_head) {
            const int p = out.sn_of_col[out.rowidx[sn.rows]];
            out.sn_parent[s] = p;
            (is_simple(sn) ? leaves[p] : kids[p]).push_back(s);
         }
         if (!is_simple(sn)) {
            const int64_t nf = sn.w + frows(sn);
            out.mf_max_front = std::max(out.mf_max_front, (int)nf);
            const int64_t cw = (int64_t)sn.w * nf - (int64_t)sn.w * (sn.w - 1) / 2, lt = (int64_t)sn.w * ((frows(sn) + 3) / 4 * 4);
            if (nf > MF_MAX_FRONT || std::max(cw, lt) + 8 > opt.mf_lds_doubles) out.mf_ok = false;
         }
      }
      if (out.arena >= (int64_t)INT32_MAX) out.mf_ok = false;   // leaf records keep panel offsets as int
      // position of the rows of supernode c inside the front of its parent p
      auto positions = [&](const HeadSupernode& c, const HeadSupernode& p, std::vector<int>& pos) -> bool {
         const int* rows = out.rowidx.data() + c.rows;
         const int* prow = out.rowidx.data() + p.rows;
         int q = 0;
         for (int a = 0; a < frows(c); ++a) {
            const int ra = rows[a];
            if (ra < p.c0 + p.w) { pos.push_back(ra - p.c0); continue; }
            while (q < p.r && prow[q] < ra) ++q;
            if (q == p.r || prow[q] != ra) return false;
            pos.push_back(p.w + q);
         }
         return true;
      };
      // update-matrix offsets first (a parent's record refers to its children's and to its own)
      for (int s = 0; s < nsn; ++s) {
         const HeadSupernode& sn = out.sn[s];
         if (is_simple(sn)) continue;
         out.mf_U[s] = out.mf_U_total;
         const int64_t uc = ucols(sn);
         out.mf_U_total += uc * frows(sn) - uc * (uc - 1) / 2;   // update columns 0 .. uc - 1, packed: column b holds rows b .. r - 1
      }
      if (out.mf_U_total >= (int64_t)INT32_MAX || 18 * (int64_t)nsn >= (int64_t)INT32_MAX) out.mf_ok = false;
      // entries of K and of the border that fall into the panel of a front: (position in the packed panel, index into the block's
      // value array; border entries as -1 - index) - k_front adds them to its zeroed LDS panel instead of reading the panel from the
      // arena, which holds these few entries among zeros (an LP's dual columns: the diagonal and two border entries in ~100 rows)
      std::vector<std::vector<int>> kent(nsn), kb_ent(nsn);
      if (out.mf_ok) {
         auto add_entry = [&](int c, int r, int src) {
            const int si = out.sn_of_col[c];
            const HeadSupernode& sn = out.sn[si];
            if (is_simple(sn)) return;
            if (out.mf_konly && r >= n) {                // a border entry: k_border_rows starts the border rows of the front from it
               const int* bq = out.rowidx.data() + sn.rows;
               const int a = (int)(std::lower_bound(bq + sn.rb, bq + sn.r, r) - (bq + sn.rb));
               kb_ent[si].push_back(a | ((c - sn.c0) << 16));
               kb_ent[si].push_back(-1 - src);            // (the caller passes -1 - p for border entries)
               return;
            }
            const int k = c - sn.c0, nf = sn.w + frows(sn);
            int fi;
            if (r < sn.c0 + sn.w) fi = r - sn.c0;
            else {
               const int* b = out.rowidx.data() + sn.rows;
               fi = sn.w + (int)(std::lower_bound(b, b + sn.r, r) - b);
            }
            kent[si].push_back(k * nf - k * (k - 1) / 2 + fi - k);
            kent[si].push_back(src);
         };
         for (int i = 0; i < n; ++i)
            for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
               const int a = out.iperm[i], b = out.iperm[K.colidx[p]];
               if (std::min(a, b) < n_head) add_entry(std::min(a, b), std::max(a, b), p);
            }
         for (int sc = 0; sc < border.nrows; ++sc)
            for (int p = border.rowptr[sc]; p < border.rowptr[sc + 1]; ++p) {
               const int c = out.iperm[border.colidx[p]];
               if (c < n_head) add_entry(c, n + bidx[sc], -1 - p);
            }
      }
      std::vector<int> pos;
      for (int s = 0; s < nsn && out.mf_ok; ++s) {
         const HeadSupernode& sn = out.sn[s];
         if (is_simple(sn)) continue;
         const int64_t base = (int64_t)out.mf_int.size();
         out.mf_meta[s] = base;
         const int nf = sn.w + frows(sn), n_leaf = (int)leaves[s].size();
         int sum_rc = 0;
         for (int c : kids[s]) sum_rc += frows(out.sn[c]);
         int hdr[MF_HDR] = {(int)kids[s].size(), n_leaf, out.sn_parent[s] >= 0 ? 1 : 0, 0, 0, 0, sum_rc, 0};
         const size_t hpos = out.mf_int.size();
         out.mf_int.insert(out.mf_int.end(), hdr, hdr + MF_HDR);
         for (int c : kids[s]) {
            out.mf_int.push_back((int)(out.mf_U[c] - out.mf_U[s]));
            out.mf_int.push_back(frows(out.sn[c]) | (ucols(out.sn[c]) << 16));   // r_c, and the number of update columns the child hands over
            out.mf_int.push_back(0);   // (reserved: the record keeps three integers per child)
         }
         for (int c : kids[s]) {
            pos.clear();
            if (!positions(out.sn[c], sn, pos))
               PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, rows of supernode %d missing in its parent front", out.sn[c].c0);
            out.mf_int.insert(out.mf_int.end(), pos.begin(), pos.end());
         }
         // behind the (optional) leaf part: the panel's entries of K / the border: count, then (position, source) pairs
         auto append_entries = [&]() {
            out.mf_int.push_back((int)(kent[s].size() / 2));
            out.mf_int.insert(out.mf_int.end(), kent[s].begin(), kent[s].end());
            out.mf_int[hpos + 2] |= (int)(kent[s].size() / 2) << 1;   // (the count also in the header: the kernel requests the entries at once)
            std::vector<int>().swap(kent[s]);
         };
         if (n_leaf == 0) { append_entries(); continue; }
         // leaf part: colptr | items | leaf table | position lists
         std::vector<int> lists, item_col, item_a, item_b, tab;
         int n_vals = 0, n_items = 0;
         for (int c : leaves[s]) n_items += ucols(out.sn[c]);   // (a leaf's border rows are no front columns under the border split)
         const int list_base = (nf + 1) + 2 * n_items + 4 * n_leaf;   // offset of the first position list inside the leaf part
         out.mf_int[hpos + 7] = (int)out.mf_LV_total;
         for (int c : leaves[s]) {
            const HeadSupernode& lf = out.sn[c];
            pos.clear();
            if (!positions(lf, sn, pos))
               PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, rows of leaf column %d missing in its parent front", lf.c0);
            const int loff = list_base + (int)lists.size();
            out.mf_U[c] = out.mf_LV_total + n_vals;
            tab.push_back(lf.c0);
            tab.push_back(n_vals);
            tab.push_back(frows(lf));
            tab.push_back(loff);
            lists.insert(lists.end(), pos.begin(), pos.end());
            for (int b = 0; b < ucols(lf); ++b) { item_col.push_back(pos[b]); item_a.push_back((n_vals << 9) | (frows(lf) << 4) | b); item_b.push_back(loff); }
            n_vals += 1 + lf.r;
         }
         out.mf_LV_total += n_vals;
         if (n_vals >= (1 << 22)) { out.mf_ok = false; break; }
         std::vector<int> colptr(nf + 1, 0);
         for (int q : item_col) ++colptr[q + 1];
         for (int q = 0; q < nf; ++q) colptr[q + 1] += colptr[q];
         std::vector<int> items(2 * item_col.size()), fill(colptr.begin(), colptr.end() - 1);
         for (size_t i = 0; i < item_col.size(); ++i) {   // stable: leaves stay in order
            const int at = fill[item_col[i]]++;
            items[2 * at] = item_a[i];
            items[2 * at + 1] = item_b[i];
         }
         out.mf_int.insert(out.mf_int.end(), colptr.begin(), colptr.end());
         out.mf_int.insert(out.mf_int.end(), items.begin(), items.end());
         out.mf_int.insert(out.mf_int.end(), tab.begin(), tab.end());
         out.mf_int.insert(out.mf_int.end(), lists.begin(), lists.end());
         out.mf_int[hpos + 3] = list_base + (int)lists.size();
         out.mf_int[hpos + 4] = n_items;
         out.mf_int[hpos + 5] = n_vals;
         append_entries();
      }
      if (out.mf_ok && out.mf_konly) {
         // gather-form metadata (k_border_rows / k_border_tail): for every supernode C with border rows that takes part in the multifrontal
         // scheme, the runs of its below-rows inside one target supernode's columns -> a pair of that target; its tail rows -> k_border_tail
         std::vector<std::vector<int>> pairs(nsn);
         for (int c = 0; c < nsn; ++c) {
            const HeadSupernode& sn = out.sn[c];
            if (sn.rb >= sn.r || (is_simple(sn) && out.sn_parent[c] < 0)) continue;
            const int* rows = out.rowidx.data() + sn.rows;
            int b0 = 0;
            while (b0 < sn.rb && rows[b0] < n_head) {
               const int tgi = out.sn_of_col[rows[b0]];
               const HeadSupernode& tg = out.sn[tgi];
               int b1 = b0 + 1;
               while (b1 < sn.rb && rows[b1] < tg.c0 + tg.w) ++b1;
               pairs[tgi].push_back(c);
               pairs[tgi].push_back(b0 | (b1 << 16));
               b0 = b1;
            }
            for (int q0 = b0; q0 < sn.rb; q0 += 16) {        // (rows b0 .. rb - 1: the tail rows)
               out.kb_tail.push_back(c);
               out.kb_tail.push_back(q0 | (std::min(q0 + 16, sn.rb) << 16));
            }
         }
         out.kb_off.assign(nsn, -1);
         for (int s2 = 0; s2 < nsn; ++s2) {
            const HeadSupernode& sn = out.sn[s2];
            if (is_simple(sn) || sn.rb >= sn.r) {
               if (!pairs[s2].empty()) PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, supernode %d without border rows below one with", sn.c0);
               continue;
            }
            out.kb_off[s2] = (int64_t)out.kb_rec.size();
            out.kb_rec.push_back((int)(pairs[s2].size() / 2));
            out.kb_rec.push_back((int)(kb_ent[s2].size() / 2));
            out.kb_rec.insert(out.kb_rec.end(), pairs[s2].begin(), pairs[s2].end());
            for (size_t e = 0; e < kb_ent[s2].size(); e += 2) {
               out.kb_rec.push_back(kb_ent[s2][e]);
               out.kb_rec.push_back(kb_ent[s2][e + 1]);        // index of the border value inside the block's border values
            }
         }
      }
      if (!out.mf_ok && out.mf_split) {   // compact panels need the multifrontal head: analyse again with full panels
         AnalyzeOptions full = opt;
         full.mf_split_nb_max = 0;
         out = BlockSym();
         return analyze_block(K, border, n_primal, full, out);
      }
      if (!out.mf_ok) {
         std::vector<int>().swap(out.mf_int);
         std::vector<int64_t>().swap(out.mf_fix);
         out.mf_U.assign(nsn, -1);
         out.mf_meta.assign(nsn, -1);
         out.mf_U_total = 0;
         out.mf_LV_total = 0;
      }
   }

   // ---- scatter maps
   auto head_dst = [&](int c, int r) -> int64_t {
      const HeadSupernode& sn = out.sn[out.sn_of_col[c]];
      const int ld = sn.ld;   // (entries of a front are taken from the value arrays by k_front: their offsets here are never used)
      int pos;
      if (r < sn.c0 + sn.w) {
         pos = r - sn.c0;
      } else {
         const int* b = out.rowidx.data() + sn.rows;
         const int* it = std::lower_bound(b, b + sn.r, r);
         if (it == b + sn.r || *it != r) return -1;
         pos = sn.w + (int)(it - b);
      }
      return sn.panel + pos + (int64_t)(c - sn.c0) * ld;
   };
   out.a_dst.resize(K.rowptr[n]);
   out.a_front.assign(K.rowptr[n], 0);
   auto in_front = [&](int c) {   // (meaningful when the block is multifrontal: every head supernode that is not a simple leaf is a front)
      if (c >= n_head) return false;
      const HeadSupernode& sn = out.sn[out.sn_of_col[c]];
      return !(sn.w == 1 && sn.r <= opt.simple_rmax && sn.level == 0);
   };
   for (int i = 0; i < n; ++i)
      for (int p = K.rowptr[i]; p < K.rowptr[i + 1]; ++p) {
         const int a = out.iperm[i], b = out.iperm[K.colidx[p]];
         const int c = std::min(a, b), r = std::max(a, b);
         int64_t d;
         if (c < n_head)
            d = head_dst(c, r);
         else
            d = out.T_off + (r - n_head) + (int64_t)(c - n_head) * out.ldT;
         if (d < 0) PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, entry (%d,%d) not in the symbolic structure", i, K.colidx[p]);
         out.a_dst[p] = d;
         out.a_front[p] = in_front(c) ? 1 : 0;
      }
   if (border.nrows > 0) {
      out.b_dst.resize(border.rowptr[border.nrows]);
      out.b_front.assign(border.rowptr[border.nrows], 0);
      for (int s = 0; s < border.nrows; ++s)
         for (int p = border.rowptr[s]; p < border.rowptr[s + 1]; ++p) {
            const int c = out.iperm[border.colidx[p]];
            int64_t d;
            if (c < n_head)
               d = head_dst(c, n + bidx[s]);
            else
               d = out.T_off + (out.m_pad + bidx[s]) + (int64_t)(c - n_head) * out.ldT;
            if (d < 0) PIPS_FAIL(PIPS_ERR_STATE, "analyze_block: internal error, border entry (%d,%d) not in the structure", s, border.colidx[p]);
            out.b_dst[p] = d;
            out.b_front[p] = in_front(c) ? 1 : 0;
         }
   }
   return PIPS_OK;
}

}  // namespace pips
