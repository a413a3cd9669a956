/* Plain-C use of the C ABI (include/pips_hip.h): what the DoubleLinearSolver adapter of INTEGRATION.md does for one leaf.
 * Generates one synthetic KKT block K = [D W^T; W -E], factorises it on the GPU, solves two right-hand sides, checks the
 * residual on the host and prints the inertia.  Then the leaf's Schur term  -Br^T K^-1 Br  twice: by the reference's K4-K6 chunk
 * loop on the host (addBiTLeftKiBiRightToResBlockedParallelSolvers, DistributedLinearSystem.C:766-1047: 20 border columns
 * dense-ified per chunk with their colSparsity flags, solve(nrhs, ...), sparse product back - INTEGRATION.md level 1) and by
 * pips_hip_ldl_set_border + pips_hip_ldl_factor_schur (level 1.5: the CSR border goes up, the S x S term comes down).
 *   gcc -std=c11 -O2 -Iinclude examples/leaf_solve.c -Lpips-ipmpp_amd -lpipship -Wl,-rpath,$PWD/pips-ipmpp_amd -lm -o leaf_solve */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pips_hip.h"

#define CHECK(call)                                                                  \
   do {                                                                              \
      const int rc_ = (call);                                                        \
      if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pips_hip_last_error()); return 1; } \
   } while (0)

int main(void) {
   const int nx = 2000, my = 1000, n0 = 8, myl = 8, n = nx + my;
   const double rho = 10.0 / nx;
   const int kw = pips_gen_row_nnz(nx, rho);
   int* Wrp = malloc((my + 1) * sizeof(int)); int* Wci = malloc((size_t)my * kw * sizeof(int)); double* Wv = malloc((size_t)my * kw * sizeof(double));
   int* Trp = malloc((my + 1) * sizeof(int)); int* Tci = malloc((size_t)my * 2 * sizeof(int)); double* Tv = malloc((size_t)my * 2 * sizeof(double));
   int* Frp = malloc((myl + 1) * sizeof(int)); int* Fci = malloc((size_t)myl * 4 * sizeof(int)); double* Fv = malloc((size_t)myl * 4 * sizeof(double));
   double* c = malloc(nx * sizeof(double)); double* xs = malloc(nx * sizeof(double));
   CHECK(pips_gen_block(7, 1, nx, my, n0, myl, rho, Wrp, Wci, Wv, Trp, Tci, Tv, Frp, Fci, Fv, c, xs));

   /* two passes like create_kkt: sizes first, then entries */
   int* Krp = malloc((n + 1) * sizeof(int)); int* dpos = malloc(n * sizeof(int));
   CHECK(pips_kkt_leaf_assemble(nx, my, 0, NULL, NULL, NULL, Wrp, Wci, Wv, NULL, NULL, NULL, Krp, NULL, NULL, NULL));
   const int nnz = Krp[n];
   int* Kci = malloc(nnz * sizeof(int)); double* Kv = calloc(nnz, sizeof(double));
   CHECK(pips_kkt_leaf_assemble(nx, my, 0, NULL, NULL, NULL, Wrp, Wci, Wv, NULL, NULL, NULL, Krp, Kci, Kv, dpos));
   double* d = malloc(n * sizeof(double));
   CHECK(pips_gen_diagonal(7, 1, nx, -4.0, 4.0, d));          /* primal diagonal 10^U(-4,4) */
   for (int i = 0; i < nx; ++i) Kv[dpos[i]] = d[i];
   for (int i = nx; i < n; ++i) Kv[dpos[i]] = -1e-8;           /* dual regularisation */

   if (pips_hip_device_count() <= 0) { fprintf(stderr, "no GPU: %s\n", pips_hip_last_error()); return 2; }
   void* h = NULL;
   CHECK(pips_hip_ldl_create(&h, n, Krp, Kci, -1, 0));
   CHECK(pips_hip_ldl_set_inertia_hint(h, nx));
   CHECK(pips_hip_ldl_analyze(h));
   CHECK(pips_hip_ldl_factor(h, Kv));
   int pos, neg, zero;
   CHECK(pips_hip_ldl_inertia(h, &pos, &neg, &zero));

   double* rhs = malloc((size_t)2 * n * sizeof(double)); double* x = malloc((size_t)2 * n * sizeof(double));
   for (int i = 0; i < 2 * n; ++i) x[i] = rhs[i] = sin(0.37 * i) + 0.1;
   CHECK(pips_hip_ldl_solve(h, 2, x, n));                       /* one right-hand side per row, in place */

   double worst = 0.0;
   for (int r = 0; r < 2; ++r) {                                /* residual with the lower-triangular CSR */
      double* y = calloc(n, sizeof(double));
      const double *xr = x + (size_t)r * n, *br = rhs + (size_t)r * n;
      for (int i = 0; i < n; ++i)
         for (int p = Krp[i]; p < Krp[i + 1]; ++p) {
            const int j = Kci[p];
            y[i] += Kv[p] * xr[j];
            if (j != i) y[j] += Kv[p] * xr[i];
         }
      double num = 0.0, den = 0.0;
      for (int i = 0; i < n; ++i) { num += (y[i] - br[i]) * (y[i] - br[i]); den += br[i] * br[i]; }
      if (sqrt(num / den) > worst) worst = sqrt(num / den);
      free(y);
   }
   printf("inertia (%d, %d, %d), relative residual %.2e\n", pos, neg, zero, worst);

   /* ---- border Br^T: S rows (Schur column ids) over the n rows of K; rows 0 .. n0-1 = columns of T, rows n0 .. = rows of F */
   const int S = n0 + myl;
   int* Brp = malloc((S + 1) * sizeof(int));
   CHECK(pips_border_assemble(nx, my, 0, n0, 0, myl, 0, NULL, NULL, NULL, Trp, Tci, Tv, NULL, NULL, NULL, Frp, Fci, Fv, NULL, NULL, NULL, Brp, NULL, NULL));
   const int bnnz = Brp[S];
   int* Bci = malloc((size_t)(bnnz > 0 ? bnnz : 1) * sizeof(int)); double* Bv = malloc((size_t)(bnnz > 0 ? bnnz : 1) * sizeof(double));
   CHECK(pips_border_assemble(nx, my, 0, n0, 0, myl, 0, NULL, NULL, NULL, Trp, Tci, Tv, NULL, NULL, NULL, Frp, Fci, Fv, NULL, NULL, NULL, Brp, Bci, Bv));

   /* level 1: the host's chunk loop around solve(nrhs, rhss, colSparsity) - columns "lie as rows" (:895-901) */
   double* SC1 = calloc((size_t)S * S, sizeof(double));
   const int chunk = 20;
   double* dense = malloc((size_t)chunk * n * sizeof(double));
   int* ids = malloc(chunk * sizeof(int));
   for (int s0 = 0; s0 < S;) {
      int cnt = 0;
      memset(dense, 0, (size_t)chunk * n * sizeof(double));
      for (; s0 < S && cnt < chunk; ++s0) {
         if (Brp[s0 + 1] == Brp[s0]) continue;                 /* empty border columns are skipped (:870-874) */
         for (int p = Brp[s0]; p < Brp[s0 + 1]; ++p) dense[(size_t)cnt * n + Bci[p]] = Bv[p];
         ids[cnt++] = s0;
      }
      if (cnt == 0) break;
      CHECK(pips_hip_ldl_solve(h, cnt, dense, n));
      for (int q = 0; q < cnt; ++q)                            /* K6: SC[id][:] -= Br^T x (addLeftBorderTimesDenseColsToResTranspDense) */
         for (int t = 0; t < S; ++t) {
            double acc = 0.0;
            for (int p = Brp[t]; p < Brp[t + 1]; ++p) acc += Bv[p] * dense[(size_t)q * n + Bci[p]];
            SC1[(size_t)ids[q] * S + t] -= acc;
         }
   }
   pips_hip_ldl_destroy(h);

   /* level 1.5: border declared to the handle, Schur term formed on the device */
   void* h2 = NULL;
   CHECK(pips_hip_ldl_create(&h2, n, Krp, Kci, -1, 0));
   CHECK(pips_hip_ldl_set_inertia_hint(h2, nx));
   CHECK(pips_hip_ldl_set_border(h2, S, Brp, Bci));
   double* SC2 = calloc((size_t)S * S, sizeof(double));
   CHECK(pips_hip_ldl_factor_schur(h2, Kv, Bv, SC2, S));
   double diff = 0.0, big = 0.0;
   for (int i = 0; i < S; ++i)
      for (int j = 0; j <= i; ++j) {                            /* lower triangle (DenseSymmetricMatrix) */
         const double a1 = SC1[(size_t)i * S + j], a2 = SC2[(size_t)i * S + j];
         if (fabs(a1 - a2) > diff) diff = fabs(a1 - a2);
         if (fabs(a1) > big) big = fabs(a1);
      }
   printf("Schur term: chunk loop vs pips_hip_ldl_factor_schur, max difference %.2e of %.2e\n", diff, big);
   pips_hip_ldl_destroy(h2);
   return (pos == nx && neg == my && zero == 0 && worst < 1e-10 && diff <= 1e-9 * big) ? 0 : 3;
}
