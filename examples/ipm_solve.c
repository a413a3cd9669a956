/* Plain-C use of the whole device-resident path (include/pips_hip.h): a small arrowhead LP of the synthetic family
 *     min c^T x,  [F0 F_1 .. F_N; T_i W_i] x = b,  x >= 0        (N blocks, linking variables x0, linking rows)
 * is generated block by block, handed to pips_ipm_create and solved by the interior-point harness - every iteration
 * factorises the KKT system (leaf LDL^T, Schur complement, root LDL^T) and runs solveCompressed on the GPU.  The program then
 * checks A x = b, x >= 0 and the duality gap on the host.
 *   gcc -std=c11 -O2 -Iinclude examples/ipm_solve.c -Lpips-ipmpp_amd -lpipship -Wl,-rpath,$PWD/pips-ipmpp_amd -lm -o ipm_solve */
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

/* y += M x for a CSR block whose row pointers start at rp[0] */
static void csr_axpy(int rows, const int* rp, const int* ci, const double* v, const double* x, double* y) {
   for (int r = 0; r < rows; ++r)
      for (int p = rp[r] - rp[0]; p < rp[r + 1] - rp[0]; ++p) y[r] += v[p] * x[ci[p]];
}

int main(void) {
   enum { N = 3 };
   const int n_i = 400, my_i = 200, n0 = 10, myl = 8;
   const double rho = 10.0 / n_i;
   const unsigned long long seed = 2026;
   const int kw = pips_gen_row_nnz(n_i, rho), kf0 = n0 < 2 ? n0 : 2;
   const int nx = n0 + N * n_i, ny = myl + N * my_i;

   /* concatenated CSR arrays of the N blocks, as pips_ipm_create takes them: row pointers of block i start where block i-1 ended */
   int* Wrp = malloc((size_t)N * (my_i + 1) * sizeof(int)); int* Wci = malloc((size_t)N * my_i * kw * sizeof(int)); double* Wv = malloc((size_t)N * my_i * kw * sizeof(double));
   int* Trp = malloc((size_t)N * (my_i + 1) * sizeof(int)); int* Tci = malloc((size_t)N * my_i * 2 * sizeof(int)); double* Tv = malloc((size_t)N * my_i * 2 * sizeof(double));
   int* Frp = malloc((size_t)N * (myl + 1) * sizeof(int)); int* Fci = malloc((size_t)N * myl * 4 * sizeof(int)); double* Fv = malloc((size_t)N * myl * 4 * sizeof(double));
   int* F0rp = malloc((myl + 1) * sizeof(int)); int* F0ci = malloc((size_t)myl * kf0 * sizeof(int)); double* F0v = malloc((size_t)myl * kf0 * sizeof(double));
   double* c = calloc(nx, sizeof(double)); double* b = calloc(ny, sizeof(double)); double* xstar = calloc(nx, sizeof(double));
   int n_of[N], my_of[N];

   CHECK(pips_gen_root(seed, n0, myl, F0rp, F0ci, F0v, c, xstar));
   csr_axpy(myl, F0rp, F0ci, F0v, xstar, b);                                        /* b_link = F0 x0* + sum F_i x_i* */
   size_t wp = 0, tp = 0, fp = 0;
   for (int i = 0; i < N; ++i) {
      int* wr = Wrp + (size_t)i * (my_i + 1); int* tr = Trp + (size_t)i * (my_i + 1); int* fr = Frp + (size_t)i * (myl + 1);
      double* xi = xstar + n0 + (size_t)i * n_i;
      CHECK(pips_gen_block(seed, i + 1, n_i, my_i, n0, myl, rho, wr, Wci + wp, Wv + wp, tr, Tci + tp, Tv + tp, fr, Fci + fp, Fv + fp,
                           c + n0 + (size_t)i * n_i, xi));
      double* bi = b + myl + (size_t)i * my_i;                                      /* b_i = T_i x0* + W_i x_i* */
      csr_axpy(my_i, tr, Tci + tp, Tv + tp, xstar, bi);
      csr_axpy(my_i, wr, Wci + wp, Wv + wp, xi, bi);
      csr_axpy(myl, fr, Fci + fp, Fv + fp, xi, b);
      wp += wr[my_i] - wr[0]; tp += tr[my_i] - tr[0]; fp += fr[myl] - fr[0];
      n_of[i] = n_i; my_of[i] = my_i;
   }

   if (pips_hip_device_count() <= 0) { fprintf(stderr, "no GPU: %s\n", pips_hip_last_error()); return 2; }
   void* ipm = NULL;
   CHECK(pips_ipm_create(&ipm, N, n0, myl, n_of, my_of, Wrp, Wci, Wv, Trp, Tci, Tv, Frp, Fci, Fv, F0rp, F0ci, F0v, c, b, 0.0, -1));
   CHECK(pips_ipm_set_option(ipm, "GONDZIO_MAX_CORRECTORS", 2));
   double res[7];
   CHECK(pips_ipm_solve(ipm, 100, 1e-8, 1e-8, 0, res));
   double* x = malloc(nx * sizeof(double)); double* y = malloc(ny * sizeof(double));
   CHECK(pips_ipm_get_solution(ipm, x, y));
   long long stats[4];
   CHECK(pips_ipm_get_stats(ipm, stats));

   /* host check: A x - b, min x, c^T x - b^T y */
   double* r = malloc(ny * sizeof(double));
   for (int k = 0; k < ny; ++k) r[k] = -b[k];
   csr_axpy(myl, F0rp, F0ci, F0v, x, r);
   wp = tp = fp = 0;
   for (int i = 0; i < N; ++i) {
      int* wr = Wrp + (size_t)i * (my_i + 1); int* tr = Trp + (size_t)i * (my_i + 1); int* fr = Frp + (size_t)i * (myl + 1);
      const double* xi = x + n0 + (size_t)i * n_i;
      csr_axpy(my_i, tr, Tci + tp, Tv + tp, x, r + myl + (size_t)i * my_i);
      csr_axpy(my_i, wr, Wci + wp, Wv + wp, xi, r + myl + (size_t)i * my_i);
      csr_axpy(myl, fr, Fci + fp, Fv + fp, xi, r);
      wp += wr[my_i] - wr[0]; tp += tr[my_i] - tr[0]; fp += fr[myl] - fr[0];
   }
   double rmax = 0.0, xmin = x[0], pobj = 0.0, dobj = 0.0;
   for (int k = 0; k < ny; ++k) { if (fabs(r[k]) > rmax) rmax = fabs(r[k]); dobj += b[k] * y[k]; }
   for (int j = 0; j < nx; ++j) { if (x[j] < xmin) xmin = x[j]; pobj += c[j] * x[j]; }
   printf("status %d after %d iterations (%lld factorisations, %lld solveCompressed): objective %.10f, gap %.2e, ||Ax-b||inf %.2e, min x %.2e\n",
          (int)res[4], (int)res[1], stats[0], stats[2], pobj, fabs(pobj - dobj), rmax, xmin);
   pips_ipm_destroy(ipm);
   return ((int)res[4] == 0 && rmax < 1e-6 && xmin > -1e-9 && fabs(pobj - dobj) < 1e-5 * fmax(1.0, fabs(pobj))) ? 0 : 3;
}
