// Synthetic N-block arrowhead LP generator and leaf-KKT assembly.
//
// This is the host-harness counterpart of the callback driver the reference ships as an example
// (Drivers/CallbackExample/callbackExample.cpp:206-347: CSR, row-major, 0-based blocks handed over through
// FNNZ/FMAT/FVEC callbacks).  The matrices produced here use the same conventions, so they can be fed to the reference
// through a DistributedInputTree as well as to the HIP backend.
//
//   block i:  W_i (my_i x n_i)  = [I | P] column-permuted + random fill, k_w = max(2, round(rho * n_i)) entries per row
//             T_i (my_i x n_0)  2 entries per row          (couples to first-stage variables; "A" in the callbacks)
//             F_i (myl  x n_i)  4 entries per row          (linking equalities; "Bl" in the callbacks)
//   root:     F_0 (myl  x n_0)  2 entries per row, my_0 = mz_0 = 0
//   entries U(-1,1) (identity entries 1), c ~ U(0.5,1.5), x* ~ U(0.5,1.5), all variables x >= 0, no upper bounds.
//
// RNG: splitmix64, one independent stream per (seed, block, matrix tag).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "pips_hip.h"

namespace {

struct SplitMix64 {
   uint64_t s;
   explicit SplitMix64(uint64_t seed) : s(seed) {}
   uint64_t next() {
      uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      return z ^ (z >> 31);
   }
   double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
   int below(int n) { return (int)(next() % (uint64_t)n); }
};

uint64_t stream_seed(uint64_t seed, int block, int tag) {
   SplitMix64 h(seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(block + 1)) ^ (0xD1B54A32D192ED03ULL * (uint64_t)(tag + 1)));
   h.next();
   return h.next();
}

// rows x ncols CSR with `per_row` distinct sorted columns per row; optional forced column per row (identity part)
void random_rows(SplitMix64& rng, int rows, int ncols, int per_row, const int* forced, int* rowptr, int* colidx,
                 double* val) {
   per_row = std::min(per_row, ncols);
   std::vector<int> cols(per_row);
   rowptr[0] = 0;
   for (int r = 0; r < rows; ++r) {
      int cnt = 0;
      if (forced) cols[cnt++] = forced[r];
      while (cnt < per_row) {
         const int c = rng.below(ncols);
         bool dup = false;
         for (int t = 0; t < cnt; ++t) dup |= cols[t] == c;
         if (!dup) cols[cnt++] = c;
      }
      const int fc = forced ? forced[r] : -1;
      std::sort(cols.begin(), cols.end());
      for (int t = 0; t < per_row; ++t) {
         colidx[rowptr[r] + t] = cols[t];
         const double u = 2.0 * rng.uniform() - 1.0;
         val[rowptr[r] + t] = cols[t] == fc ? 1.0 : u;
      }
      rowptr[r + 1] = rowptr[r] + per_row;
   }
}

}  // namespace

extern "C" {

int pips_gen_row_nnz(int n_i, double rho) { return std::max(2, (int)std::lround(rho * n_i)); }

int pips_gen_block(uint64_t seed, int block, int n_i, int my_i, int n0, int myl, double rho,
                   int* W_rowptr, int* W_colidx, double* W_val, int* T_rowptr, int* T_colidx, double* T_val,
                   int* F_rowptr, int* F_colidx, double* F_val, double* c, double* xstar) {
   if (block < 1 || n_i < 2 || my_i < 0 || my_i > n_i) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_gen_block: bad sizes");
   {  // W
      SplitMix64 rng(stream_seed(seed, block, 0));
      std::vector<int> pi(n_i);
      for (int j = 0; j < n_i; ++j) pi[j] = j;
      for (int j = n_i - 1; j > 0; --j) std::swap(pi[j], pi[rng.below(j + 1)]);
      random_rows(rng, my_i, n_i, pips_gen_row_nnz(n_i, rho), pi.data(), W_rowptr, W_colidx, W_val);
   }
   if (n0 > 0) {  // T
      SplitMix64 rng(stream_seed(seed, block, 1));
      random_rows(rng, my_i, n0, 2, nullptr, T_rowptr, T_colidx, T_val);
   }
   if (myl > 0) {  // F
      SplitMix64 rng(stream_seed(seed, block, 2));
      random_rows(rng, myl, n_i, 4, nullptr, F_rowptr, F_colidx, F_val);
   }
   SplitMix64 rng(stream_seed(seed, block, 3));
   for (int j = 0; j < n_i; ++j) c[j] = 0.5 + rng.uniform();
   for (int j = 0; j < n_i; ++j) xstar[j] = 0.5 + rng.uniform();
   return 0;
}

int pips_gen_root(uint64_t seed, int n0, int myl, int* F0_rowptr, int* F0_colidx, double* F0_val, double* c0,
                  double* xstar0) {
   if (n0 < 0 || myl < 0) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_gen_root: bad sizes");
   if (myl > 0 && n0 > 0) {
      SplitMix64 rng(stream_seed(seed, 0, 2));
      random_rows(rng, myl, n0, 2, nullptr, F0_rowptr, F0_colidx, F0_val);
   }
   SplitMix64 rng(stream_seed(seed, 0, 3));
   for (int j = 0; j < n0; ++j) c0[j] = 0.5 + rng.uniform();
   for (int j = 0; j < n0; ++j) xstar0[j] = 0.5 + rng.uniform();
   return 0;
}

// log-uniform frozen IPM diagonal D_x = 10^{U(lo,hi)} (roofline runs)
int pips_gen_diagonal(uint64_t seed, int block, int n, double lo, double hi, double* d) {
   SplitMix64 rng(stream_seed(seed, block, 4));
   for (int j = 0; j < n; ++j) d[j] = std::pow(10.0, lo + (hi - lo) * rng.uniform());
   return 0;
}

// Lower-triangular CSR of  K_i = [ Q+Dx  B^T  D^T ; B 0 0 ; D 0 0 ]  with an explicit diagonal entry in every row,
// exactly the pattern DistributedLeafLinearSystem::create_kkt builds (DistributedLeafLinearSystem.C:44-72).
// Q (lower CSR, may be null), B = W_i (my x nx), D (mz x nx, may be null).  Two-pass: call with K_colidx == NULL to get
// nnz in K_rowptr[n].  diag_pos[i] receives the position of the diagonal entry of row i (may be NULL).
int pips_kkt_leaf_assemble(int nx, int my, int mz, const int* Q_rowptr, const int* Q_colidx, const double* Q_val,
                           const int* B_rowptr, const int* B_colidx, const double* B_val, const int* D_rowptr,
                           const int* D_colidx, const double* D_val, int* K_rowptr, int* K_colidx, double* K_val,
                           int* diag_pos) {
   const int n = nx + my + mz;
   int nnz = 0;
   K_rowptr[0] = 0;
   for (int i = 0; i < n; ++i) {
      const int* rp = nullptr;
      const int* ci = nullptr;
      const double* v = nullptr;
      int row = 0;
      if (i < nx) { rp = Q_rowptr; ci = Q_colidx; v = Q_val; row = i; }
      else if (i < nx + my) { rp = B_rowptr; ci = B_colidx; v = B_val; row = i - nx; }
      else { rp = D_rowptr; ci = D_colidx; v = D_val; row = i - nx - my; }
      bool has_diag = false;
      if (rp)
         for (int p = rp[row]; p < rp[row + 1]; ++p) {
            if (i < nx && ci[p] > i) continue;  // Q given as full or lower: keep lower
            if (K_colidx) { K_colidx[nnz] = ci[p]; if (K_val) K_val[nnz] = v ? v[p] : 0.0; }
            if (ci[p] == i) { has_diag = true; if (diag_pos) diag_pos[i] = nnz; }
            ++nnz;
         }
      if (!has_diag) {
         if (K_colidx) { K_colidx[nnz] = i; if (K_val) K_val[nnz] = 0.0; }
         if (diag_pos) diag_pos[i] = nnz;
         ++nnz;
      }
      K_rowptr[i + 1] = nnz;
   }
   return 0;
}

// Pattern/values of Br_i^T (S rows = Schur column ids, N_i columns) from the five border blocks, following
// BorderBiBlock {R,A,C,n_empty,F^T,G^T} (RACFG_BLOCK.h:13-53, DistributedLeafLinearSystem.C:214-252):
//   column ids  [0,n0): R^T | A^T | C^T  (R: nx x n0, A: my x n0, C: mz x n0),  [n0,n0+n_empty): empty,
//               then myl rows of F (myl x nx), then mzl rows of G (mzl x nx).
// Two-pass like pips_kkt_leaf_assemble.  All inputs CSR; any block may be NULL.
int pips_border_assemble(int nx, int my, int mz, int n0, int n_empty, int myl, int mzl, const int* R_rowptr,
                         const int* R_colidx, const double* R_val, const int* A_rowptr, const int* A_colidx,
                         const double* A_val, const int* C_rowptr, const int* C_colidx, const double* C_val,
                         const int* F_rowptr, const int* F_colidx, const double* F_val, const int* G_rowptr,
                         const int* G_colidx, const double* G_val, int* Bt_rowptr, int* Bt_colidx, double* Bt_val) {
   const int S = n0 + n_empty + myl + mzl;
   std::vector<int> cnt(S + 1, 0);
   auto count_T = [&](const int* rp, const int* ci, int rows) {
      if (!rp) return;
      for (int r = 0; r < rows; ++r)
         for (int p = rp[r]; p < rp[r + 1]; ++p) ++cnt[ci[p] + 1];
   };
   count_T(R_rowptr, R_colidx, nx);
   count_T(A_rowptr, A_colidx, my);
   count_T(C_rowptr, C_colidx, mz);
   if (F_rowptr) for (int l = 0; l < myl; ++l) cnt[n0 + n_empty + l + 1] += F_rowptr[l + 1] - F_rowptr[l];
   if (G_rowptr) for (int l = 0; l < mzl; ++l) cnt[n0 + n_empty + myl + l + 1] += G_rowptr[l + 1] - G_rowptr[l];
   Bt_rowptr[0] = 0;
   for (int s = 0; s < S; ++s) Bt_rowptr[s + 1] = Bt_rowptr[s] + cnt[s + 1];
   if (!Bt_colidx) return 0;
   std::vector<int> fill(Bt_rowptr, Bt_rowptr + S);
   auto put_T = [&](const int* rp, const int* ci, const double* v, int rows, int off) {
      if (!rp) return;
      for (int r = 0; r < rows; ++r)
         for (int p = rp[r]; p < rp[r + 1]; ++p) {
            const int q = fill[ci[p]]++;
            Bt_colidx[q] = off + r;
            if (Bt_val) Bt_val[q] = v[p];
         }
   };
   put_T(R_rowptr, R_colidx, R_val, nx, 0);
   put_T(A_rowptr, A_colidx, A_val, my, nx);
   put_T(C_rowptr, C_colidx, C_val, mz, nx + my);
   auto put_rows = [&](const int* rp, const int* ci, const double* v, int rows, int s0) {
      if (!rp) return;
      for (int l = 0; l < rows; ++l)
         for (int p = rp[l]; p < rp[l + 1]; ++p) {
            const int q = fill[s0 + l]++;
            Bt_colidx[q] = ci[p];
            if (Bt_val) Bt_val[q] = v[p];
         }
   };
   put_rows(F_rowptr, F_colidx, F_val, myl, n0 + n_empty);
   put_rows(G_rowptr, G_colidx, G_val, mzl, n0 + n_empty + myl);
   return 0;
}

// Contiguous, monotone, balanced (+-1) mapping of the root's children (scenario blocks) to ranks/GPUs — the contract
// DistributedTree::assignProcesses asserts for mapChildrenToNSubTrees (Readers/Distributed/DistributedTree.C:35-90,437+):
// map[i] <= map[i+1], loads differ by at most one, leftovers spread over the ranks.  n_ranks > n_children is an error
// there (":57-60 too many MPI processes") and here.
int pips_map_children_to_ranks(int n_children, int n_ranks, int* map) {
   if (n_children < 0 || n_ranks <= 0 || n_ranks > n_children || !map)
      PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_map_children_to_ranks: need 0 < n_ranks <= n_children (got %d ranks, %d children)", n_ranks, n_children);
   for (int r = 0; r < n_ranks; ++r) {
      const long long b = (long long)r * n_children / n_ranks, e = (long long)(r + 1) * n_children / n_ranks;
      for (long long i = b; i < e; ++i) map[i] = r;
   }
   return 0;
}

}  // extern "C"
