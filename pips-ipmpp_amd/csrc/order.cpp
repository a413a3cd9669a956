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

}  // namespace pips
