// Constrained approximate-minimum-degree ordering on a quotient graph.
//
// Replaces the fill-reducing ordering that PARDISO (METIS, iparm[1]=2, PardisoProjectSolver.C:68-77) or MA27/MA57
// compute internally for the leaf KKT blocks.  The pattern of K_i never changes across IPM iterations
// (DistributedLeafLinearSystem.C:52-72), so this runs once per block on the host.
//
// Algorithm: classic quotient-graph minimum degree with approximate external degrees and aggressive element
// absorption (Amestoy/Davis/Duff style), without supervariables.  Extension: "dual" rows (index >= n_primal) only
// become eligible once every primal neighbour has been eliminated; with W of full row rank this makes every leading
// principal submatrix of [D W^T; W -E] (D>0, E>=0) nonsingular, so a static (no-pivoting) LDL^T exists with the
// pivot signs known in advance.
#include <algorithm>
#include <set>
#include <vector>
#include <utility>

#include "common.h"

namespace pips {

void constrained_amd(int n, const std::vector<int>& ap, const std::vector<int>& ai, int n_primal,
                     std::vector<int>& perm, std::vector<int>& colcount) {
   perm.assign(n, -1);
   colcount.assign(n, 0);
   if (n == 0) return;
   const bool constrained = n_primal >= 0 && n_primal < n;

   std::vector<std::vector<int>> A(n), E(n), Lv(n);  // variable neighbours, element neighbours, element members
   std::vector<char> alive(n, 1), ealive(n, 0);
   std::vector<int> deg(n), need(n, 0), mark(n, 0), w(n, 0), wstamp(n, 0);
   for (int i = 0; i < n; ++i) {
      A[i].assign(ai.begin() + ap[i], ai.begin() + ap[i + 1]);
      deg[i] = (int)A[i].size();
      if (constrained && i >= n_primal)
         for (int v : A[i])
            if (v < n_primal) ++need[i];
   }
   std::set<std::pair<int, int>> pq;  // (degree, node) of eligible live variables
   auto eligible = [&](int i) { return !constrained || i < n_primal || need[i] == 0; };
   for (int i = 0; i < n; ++i)
      if (eligible(i)) pq.insert({deg[i], i});

   int stamp = 0;
   std::vector<int> Lp;
   int k = 0;
   while (k < n) {
      int p;
      if (pq.empty()) {
         // cannot happen when every dual row has a primal neighbour chain; fall back to any live variable
         p = -1;
         for (int i = 0; i < n; ++i)
            if (alive[i]) { p = i; break; }
      } else {
         p = pq.begin()->second;
         pq.erase(pq.begin());
      }
      // ---- form the new element Lp = adjacency of p in the elimination graph
      ++stamp;
      Lp.clear();
      mark[p] = stamp;
      for (int v : A[p])
         if (alive[v] && mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); }
      for (int e : E[p]) {
         if (!ealive[e]) continue;
         for (int v : Lv[e])
            if (mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); }
         ealive[e] = 0;
         std::vector<int>().swap(Lv[e]);
      }
      alive[p] = 0;
      perm[k] = p;
      colcount[k] = (int)Lp.size();
      ++k;
      std::vector<int>().swap(A[p]);
      std::vector<int>().swap(E[p]);
      const int nleft = n - k;
      if (nleft == 0) break;

      // primal pivot gone: its dual neighbours (original graph) get closer to eligibility
      if (constrained && p < n_primal) {
         for (int q = ap[p]; q < ap[p + 1]; ++q) {
            const int v = ai[q];
            if (v >= n_primal && alive[v]) {
               if (--need[v] == 0) pq.insert({deg[v], v});  // deg refreshed below if v is in Lp
            }
         }
      }

      // ---- dense termination: the new element covers every remaining variable
      if ((int)Lp.size() == nleft) {
         std::vector<int> rest(Lp);
         // primal first (keeps the eligibility invariant), then by current degree, then by index (deterministic)
         std::sort(rest.begin(), rest.end(), [&](int a, int b) {
            const bool pa = !constrained || a < n_primal, pb = !constrained || b < n_primal;
            if (pa != pb) return pa;
            return a < b;
         });
         for (int v : rest) {
            perm[k] = v;
            colcount[k] = n - k - 1;
            ++k;
         }
         break;
      }

      ealive[p] = 1;
      // ---- pass 1: w[e] = |Le \ Lp| for every live element adjacent to a member of Lp
      for (int i : Lp) {
         auto& Ei = E[i];
         size_t o = 0;
         for (size_t t = 0; t < Ei.size(); ++t) {
            const int e = Ei[t];
            if (!ealive[e] || e == p) continue;
            Ei[o++] = e;
            if (wstamp[e] != stamp) { wstamp[e] = stamp; w[e] = (int)Lv[e].size(); }
            --w[e];
         }
         Ei.resize(o);
      }
      // ---- pass 2: prune, absorb, approximate degree
      const int lp = (int)Lp.size();
      for (int i : Lp) {
         auto& Ai = A[i];
         size_t o = 0;
         for (size_t t = 0; t < Ai.size(); ++t) {
            const int v = Ai[t];
            if (alive[v] && mark[v] != stamp) Ai[o++] = v;
         }
         Ai.resize(o);
         auto& Ei = E[i];
         long d = (long)Ai.size() + (lp - 1);
         o = 0;
         for (size_t t = 0; t < Ei.size(); ++t) {
            const int e = Ei[t];
            if (w[e] == 0) {  // e is a subset of Lp: absorbed into p
               if (ealive[e]) { ealive[e] = 0; std::vector<int>().swap(Lv[e]); }
               continue;
            }
            Ei[o++] = e;
            d += w[e];
         }
         Ei.resize(o);
         Ei.push_back(p);
         long dnew = std::min<long>(d, nleft - 1);
         dnew = std::min<long>(dnew, (long)deg[i] + lp - 1);
         if (eligible(i)) {
            pq.erase({deg[i], i});
            deg[i] = (int)dnew;
            pq.insert({deg[i], i});
         } else {
            deg[i] = (int)dnew;
         }
      }
      Lv[p] = Lp;
   }
}

// ------------------------------------------------------------------------------------------------------------------
// Partial nested dissection for structured (time-coupled, banded) KKT blocks.
//
// Minimum degree orders a band as one long chain: the elimination tree has no width, and the numeric phase on the GPU is
// then a latency chain (one supernode per block at a time).  A few levels of dissection cut each block into independent
// segments (leaves) joined by small separators, so the segments of all blocks are factorised side by side; inside a leaf
// the constrained minimum degree above is used unchanged.  Only a bounded depth is dissected: full nested dissection
// multiplies the fill of a band by log n for parallelism nobody can use.
//
// The static pivot order needs every dual row after all its primal neighbours.  A separator made of DUAL rows only keeps
// that for free: the primal neighbours of a dual row of one side cannot lie on the other side, and none lies in the
// separator.  Separators are therefore sought in the dual-row graph G_y (two dual rows adjacent iff they share a primal
// column or are adjacent in K), by breadth-first level structures from a pseudo-peripheral row.  Random sparsity has a
// diameter of three or four with a middle level holding most rows: the first separator is rejected and the block is
// ordered by minimum degree exactly as before.
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct Segment {
   bool leaf;
   std::vector<int> duals;   // dual rows (indices in K) of the leaf / separator
   int first_leaf;           // index (into the segment list) of the first leaf ordered below a separator
};

struct Dissector {
   int n, n_primal, nd;
   const std::vector<int>&ap, &ai;
   std::vector<int> gp, gi;          // G_y, CSR over dual rows (local index y - n_primal)
   std::vector<int> label, level;    // work arrays
   std::vector<Segment> out;
   int max_depth, min_size;
   int min_levels = 5;   // a level structure with fewer levels is not cut (3 = one separator level between two single levels)

   Dissector(int n_, int np_, const std::vector<int>& ap_, const std::vector<int>& ai_) : n(n_), n_primal(np_), nd(n_ - np_), ap(ap_), ai(ai_) {}

   bool build_dual_graph(long long cap) {
      long long work = 0;
      for (int j = 0; j < n_primal; ++j) {
         long long c = 0;
         for (int q = ap[j]; q < ap[j + 1]; ++q) c += ai[q] >= n_primal;
         work += c * c;
         if (work > cap) return false;
      }
      gp.assign(nd + 1, 0);
      std::vector<int> mark(nd, -1);
      std::vector<std::vector<int>> adj(nd);
      for (int y = 0; y < nd; ++y) {
         mark[y] = y;
         for (int q = ap[n_primal + y]; q < ap[n_primal + y + 1]; ++q) {
            const int v = ai[q];
            if (v >= n_primal) {
               if (mark[v - n_primal] != y) { mark[v - n_primal] = y; adj[y].push_back(v - n_primal); }
            } else {
               for (int t = ap[v]; t < ap[v + 1]; ++t) {
                  const int u = ai[t] - n_primal;
                  if (u >= 0 && mark[u] != y) { mark[u] = y; adj[y].push_back(u); }
               }
            }
         }
      }
      for (int y = 0; y < nd; ++y) gp[y + 1] = gp[y] + (int)adj[y].size();
      gi.resize(gp[nd]);
      for (int y = 0; y < nd; ++y) std::copy(adj[y].begin(), adj[y].end(), gi.begin() + gp[y]);
      return true;
   }

   // BFS inside the node set carrying label id; returns the level structure (nodes in BFS order, level starts)
   void bfs(int start, int id, std::vector<int>& order, std::vector<int>& lstart) {
      order.clear();
      lstart.clear();
      order.push_back(start);
      level[start] = 0;
      lstart.push_back(0);
      size_t head = 0;
      int cur = 0;
      while (head < order.size()) {
         const int v = order[head];
         if (level[v] != cur) { cur = level[v]; lstart.push_back((int)head); }
         ++head;
         for (int q = gp[v]; q < gp[v + 1]; ++q) {
            const int u = gi[q];
            if (label[u] == id && level[u] < 0) { level[u] = level[v] + 1; order.push_back(u); }
         }
      }
      lstart.push_back((int)order.size());
   }

   // returns the index of the first leaf segment emitted for this node set
   int dissect(std::vector<int>& nodes, int depth, int& next_id) {
      const int size = (int)nodes.size();
      auto emit_leaf = [&]() {
         Segment sg{true, {}, (int)out.size()};
         sg.duals.reserve(size);
         for (int y : nodes) sg.duals.push_back(y + n_primal);
         out.push_back(std::move(sg));
         return (int)out.size() - 1;
      };
      if (depth >= max_depth || size < min_size) return emit_leaf();
      const int id = next_id++;
      for (int y : nodes) { label[y] = id; level[y] = -1; }
      std::vector<int> order, lstart;
      bfs(nodes[0], id, order, lstart);
      std::vector<int> A, B, S;
      if ((int)order.size() < size) {
         // disconnected: split the components into two groups, no separator needed
         std::vector<std::vector<int>> comps;
         comps.push_back(order);
         for (int y : nodes)
            if (level[y] < 0) { bfs(y, id, order, lstart); comps.push_back(order); }
         std::sort(comps.begin(), comps.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
         for (auto& c : comps) {
            auto& dst = A.size() <= B.size() ? A : B;
            dst.insert(dst.end(), c.begin(), c.end());
         }
         if (B.empty()) return emit_leaf();
      } else {
         // pseudo-peripheral start: restart twice from a node of the last level
         for (int pass = 0; pass < 2; ++pass) {
            const int far = order.back();
            for (int y : nodes) level[y] = -1;
            bfs(far, id, order, lstart);
         }
         const int nlev = (int)lstart.size() - 1;
         if (nlev < min_levels) return emit_leaf();
         // thinnest level in the middle third (by node count)
         int best = -1;
         for (int k = 1; k + 1 < nlev; ++k) {
            const int before = lstart[k], after = size - lstart[k + 1];
            if (before < size / 3 || after < size / 3) continue;
            if (best < 0 || lstart[k + 1] - lstart[k] < lstart[best + 1] - lstart[best]) best = k;
         }
         if (best < 0) return emit_leaf();
         // only the rows of that level with a neighbour in the next level separate anything
         for (int t = lstart[best]; t < lstart[best + 1]; ++t) {
            const int v = order[t];
            bool cut = false;
            for (int q = gp[v]; q < gp[v + 1] && !cut; ++q) cut = label[gi[q]] == id && level[gi[q]] == best + 1;
            (cut ? S : A).push_back(v);
         }
         if ((int)S.size() > std::max(48, size / 16)) return emit_leaf();
         A.insert(A.end(), order.begin(), order.begin() + lstart[best]);
         B.assign(order.begin() + lstart[best + 1], order.end());
      }
      const int first = dissect(A, depth + 1, next_id);
      dissect(B, depth + 1, next_id);
      if (!S.empty()) {
         std::sort(S.begin(), S.end());
         Segment sg{false, {}, first};
         for (int y : S) sg.duals.push_back(y + n_primal);
         out.push_back(std::move(sg));
      }
      return first;
   }
};

// exact column counts of L for a given order (explicit symbolic elimination; used for dissected = low-fill blocks only)
void exact_colcounts(int n, const std::vector<int>& ap, const std::vector<int>& ai, const std::vector<int>& perm,
                     std::vector<int>& colcount) {
   std::vector<int> iperm(n);
   for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
   std::vector<std::vector<int>> S(n);
   std::vector<int> first_child(n, -1), next_sib(n, -1), mark(n, -1);
   colcount.assign(n, 0);
   for (int j = 0; j < n; ++j) {
      auto& Sj = S[j];
      const int oj = perm[j];
      for (int q = ap[oj]; q < ap[oj + 1]; ++q) {
         const int r = iperm[ai[q]];
         if (r > j && mark[r] != j) { mark[r] = j; Sj.push_back(r); }
      }
      for (int c = first_child[j]; c >= 0; c = next_sib[c]) {
         for (int r : S[c])
            if (r != j && mark[r] != j) { mark[r] = j; Sj.push_back(r); }
         std::vector<int>().swap(S[c]);
      }
      colcount[j] = (int)Sj.size();
      if (!Sj.empty()) {
         const int par = *std::min_element(Sj.begin(), Sj.end());
         next_sib[j] = first_child[par];
         first_child[par] = j;
      }
   }
}

}  // namespace

bool dissected_order(int n, const std::vector<int>& ap, const std::vector<int>& ai, int n_primal, int max_depth,
                     std::vector<int>& perm, std::vector<int>& colcount, int min_size) {
   if (max_depth <= 0 || n_primal <= 0 || n_primal >= n) return false;
   Dissector D(n, n_primal, ap, ai);
   D.max_depth = max_depth;
   D.min_size = min_size > 0 ? min_size : 512;
   if (D.nd < 2 * D.min_size) return false;
   if (!D.build_dual_graph(64LL << 20)) return false;
   D.label.assign(D.nd, -1);
   D.level.assign(D.nd, -1);
   std::vector<int> all(D.nd);
   for (int y = 0; y < D.nd; ++y) all[y] = y;
   int next_id = 0;
   D.dissect(all, 0, next_id);
   int n_leaves = 0;
   for (auto& sg : D.out) n_leaves += sg.leaf;
   if (n_leaves < 2) return false;   // nothing to gain: the caller keeps the plain minimum-degree order

   // ---- segment of every dual row, leaf of every primal column
   const int nseg = (int)D.out.size();
   std::vector<int> seg_of(n, -1);
   for (int s = 0; s < nseg; ++s)
      for (int y : D.out[s].duals) seg_of[y] = s;
   int leaf0 = 0;
   while (!D.out[leaf0].leaf) ++leaf0;
   std::vector<std::vector<int>> leaf_primals(nseg);
   for (int j = 0; j < n_primal; ++j) {
      int leaf = -1, sep = -1;
      for (int q = ap[j]; q < ap[j + 1]; ++q) {
         const int v = ai[q];
         if (v < n_primal) continue;
         const int s = seg_of[v];
         if (D.out[s].leaf) { leaf = s; break; }
         if (sep < 0 || s < sep) sep = s;   // the separator ordered first
      }
      if (leaf < 0) leaf = sep >= 0 ? D.out[sep].first_leaf : leaf0;   // only separator rows (or nothing) touch it
      leaf_primals[leaf].push_back(j);
   }
   // ---- order: per leaf the constrained minimum degree of its induced subgraph, separators after their subtrees
   perm.clear();
   perm.reserve(n);
   std::vector<int> local(n, -1);
   for (int s = 0; s < nseg; ++s) {
      const Segment& sg = D.out[s];
      if (!sg.leaf) {
         perm.insert(perm.end(), sg.duals.begin(), sg.duals.end());
         continue;
      }
      std::vector<int> nodes(leaf_primals[s]);
      const int np = (int)nodes.size();
      nodes.insert(nodes.end(), sg.duals.begin(), sg.duals.end());
      const int m = (int)nodes.size();
      for (int t = 0; t < m; ++t) local[nodes[t]] = t;
      std::vector<int> sp(m + 1, 0), si;
      for (int t = 0; t < m; ++t) {
         const int v = nodes[t];
         for (int q = ap[v]; q < ap[v + 1]; ++q)
            if (local[ai[q]] >= 0) si.push_back(local[ai[q]]);
         sp[t + 1] = (int)si.size();
      }
      std::vector<int> lperm, lcc;
      constrained_amd(m, sp, si, np, lperm, lcc);
      for (int t = 0; t < m; ++t) perm.push_back(nodes[lperm[t]]);
      for (int t = 0; t < m; ++t) local[nodes[t]] = -1;
   }
   if ((int)perm.size() != n) return false;   // defensive: every node exactly once
   exact_colcounts(n, ap, ai, perm, colcount);
   return true;
}

// Nested dissection of a graph with a few "hub" vertices that touch (nearly) everything - the Schur complement of a 2-link arrowhead
// problem: the first-stage variables x0 (and root equality rows) are the hubs, the linking rows without them form a chain of cliques.
// The hubs are ordered LAST, in the given order; the rest is dissected by breadth-first level structures exactly as the dual-row
// graph of a time-coupled leaf block (Dissector above), minimum degree inside the leaves.  The order carries no constraint: the
// caller uses it only for quasi-definite matrices (any symmetric permutation has an LDL^T with the pivot signs known in advance).
bool hub_dissected_order(int n, const std::vector<int>& ap, const std::vector<int>& ai, const std::vector<int>& hubs_last, int min_size,
                         std::vector<int>& perm, std::vector<int>& colcount) {
   std::vector<int> local(n, 0);
   for (int h : hubs_last) {
      if (h < 0 || h >= n || local[h] < 0) return false;
      local[h] = -1;
   }
   std::vector<int> orig;
   for (int v = 0; v < n; ++v)
      if (local[v] >= 0) { local[v] = (int)orig.size(); orig.push_back(v); }
   const int m = (int)orig.size();
   if (m < 2 * std::max(min_size, 1)) return false;
   static const std::vector<int> none_p(1, 0), none_i;
   Dissector D(m, 0, none_p, none_i);   // (only its graph, bfs and dissect are used: gp / gi are filled here)
   D.max_depth = 40;
   D.min_levels = 3;
   D.min_size = std::max(min_size, 1);
   D.gp.assign(m + 1, 0);
   for (int y = 0; y < m; ++y) {
      const int v = orig[y];
      for (int q = ap[v]; q < ap[v + 1]; ++q)
         if (local[ai[q]] >= 0) D.gi.push_back(local[ai[q]]);
      D.gp[y + 1] = (int)D.gi.size();
   }
   D.label.assign(m, -1);
   D.level.assign(m, -1);
   std::vector<int> all(m);
   for (int y = 0; y < m; ++y) all[y] = y;
   int next_id = 0;
   D.dissect(all, 0, next_id);
   int n_leaves = 0;
   for (auto& sg : D.out) n_leaves += sg.leaf;
   if (n_leaves < 2) return false;
   perm.clear();
   perm.reserve(n);
   std::vector<int> loc2(m, -1);
   for (const Segment& sg : D.out) {
      if (!sg.leaf) {
         for (int y : sg.duals) perm.push_back(orig[y]);
         continue;
      }
      const std::vector<int>& nodes = sg.duals;
      const int k = (int)nodes.size();
      for (int t = 0; t < k; ++t) loc2[nodes[t]] = t;
      std::vector<int> sp(k + 1, 0), si;
      for (int t = 0; t < k; ++t) {
         for (int q = D.gp[nodes[t]]; q < D.gp[nodes[t] + 1]; ++q)
            if (loc2[D.gi[q]] >= 0) si.push_back(loc2[D.gi[q]]);
         sp[t + 1] = (int)si.size();
      }
      std::vector<int> lperm, lcc;
      constrained_amd(k, sp, si, -1, lperm, lcc);
      for (int t = 0; t < k; ++t) perm.push_back(orig[nodes[lperm[t]]]);
      for (int t = 0; t < k; ++t) loc2[nodes[t]] = -1;
   }
   perm.insert(perm.end(), hubs_last.begin(), hubs_last.end());
   if ((int)perm.size() != n) return false;
   exact_colcounts(n, ap, ai, perm, colcount);
   return true;
}

}  // namespace pips
