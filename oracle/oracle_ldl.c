/* oracle_ldl.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the leaf factorisation the reference delegates to a third-party sparse symmetric-indefinite
 * solver: PARDISO (Schenk, ">= 7.2", README.md:50; binary libpardiso.so, not vendored) called with mtype = -2 in
 * phases 12 / 33 (PardisoSolver.C:141-205, 207-352), or HSL MA27/MA57.  Neither library is under /root/reference, so
 * what is restated here is the published algorithm class they implement: sparse LDL^T of P K P^T with a fill-reducing
 * permutation P, static pivoting with perturbation of unacceptable pivots (PARDISO: eps * ||A||, the value its
 * Schur variant spells out at PardisoProjectSchurSolver.C:145), inertia from the signs of D (PardisoSolver.C:361-367
 * reads iparm[21..22]), and forward / diagonal / backward substitution.  The elimination is the classic up-looking
 * row-by-row LDL^T driven by the elimination tree.
 *
 * Input convention = the reference's: lower-triangular CSR (krowM/jcolM/M, 0-based; SparseSymmetricMatrix isLower),
 * which is the upper-triangular CSC the up-looking algorithm wants.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
   int n;
   int* Lp;      /* n+1 column pointers of L (strictly lower, CSC) */
   int* Li;
   double* Lx;
   double* D;
   int* perm;    /* perm[k] = original index of the k-th pivot */
   int* iperm;
   int* parent;
   int inertia[3]; /* positive, negative, perturbed */
   /* permuted upper-triangular CSC of the matrix pattern + map from input entries */
   int* Up;
   int* Ui;
   int* Umap;    /* input entry p goes to U position Umap[p] */
   int nnz;
} oracle_ldl;

static void* xmalloc(size_t s) { return malloc(s ? s : 1); }

void oracle_ldl_free(oracle_ldl* f) {
   if (!f) return;
   free(f->Lp); free(f->Li); free(f->Lx); free(f->D); free(f->perm); free(f->iperm); free(f->parent);
   free(f->Up); free(f->Ui); free(f->Umap);
   free(f);
}

/* symbolic phase: permuted pattern, elimination tree, column counts of L */
oracle_ldl* oracle_ldl_analyze(int n, const int* krow, const int* jcol, const int* perm) {
   oracle_ldl* f = (oracle_ldl*)calloc(1, sizeof(oracle_ldl));
   const int nnz = krow[n];
   f->n = n;
   f->nnz = nnz;
   f->perm = (int*)xmalloc(sizeof(int) * n);
   f->iperm = (int*)xmalloc(sizeof(int) * n);
   for (int k = 0; k < n; ++k) f->perm[k] = perm ? perm[k] : k;
   for (int k = 0; k < n; ++k) f->iperm[f->perm[k]] = k;
   /* upper CSC of P K P^T: entry (i,j) -> column max(pi,pj), row min(pi,pj) */
   f->Up = (int*)calloc(n + 1, sizeof(int));
   f->Ui = (int*)xmalloc(sizeof(int) * nnz);
   f->Umap = (int*)xmalloc(sizeof(int) * nnz);
   for (int i = 0; i < n; ++i)
      for (int p = krow[i]; p < krow[i + 1]; ++p) {
         const int a = f->iperm[i], b = f->iperm[jcol[p]];
         f->Up[(a > b ? a : b) + 1]++;
      }
   for (int k = 0; k < n; ++k) f->Up[k + 1] += f->Up[k];
   int* fill = (int*)xmalloc(sizeof(int) * n);
   memcpy(fill, f->Up, sizeof(int) * n);
   for (int i = 0; i < n; ++i)
      for (int p = krow[i]; p < krow[i + 1]; ++p) {
         const int a = f->iperm[i], b = f->iperm[jcol[p]];
         const int c = a > b ? a : b, r = a > b ? b : a;
         f->Ui[fill[c]] = r;
         f->Umap[p] = fill[c]++;
      }
   /* elimination tree and column counts */
   f->parent = (int*)xmalloc(sizeof(int) * n);
   int* flag = (int*)xmalloc(sizeof(int) * n);
   int* lnz = (int*)calloc(n, sizeof(int));
   for (int k = 0; k < n; ++k) {
      f->parent[k] = -1;
      flag[k] = k;
      for (int p = f->Up[k]; p < f->Up[k + 1]; ++p) {
         int i = f->Ui[p];
         while (i < k && flag[i] != k) {
            if (f->parent[i] == -1) f->parent[i] = k;
            lnz[i]++;
            flag[i] = k;
            i = f->parent[i];
         }
      }
   }
   f->Lp = (int*)xmalloc(sizeof(int) * (n + 1));
   f->Lp[0] = 0;
   for (int k = 0; k < n; ++k) f->Lp[k + 1] = f->Lp[k] + lnz[k];
   f->Li = (int*)xmalloc(sizeof(int) * f->Lp[n]);
   f->Lx = (double*)xmalloc(sizeof(double) * f->Lp[n]);
   f->D = (double*)xmalloc(sizeof(double) * n);
   free(fill); free(flag); free(lnz);
   return f;
}

long oracle_ldl_nnzL(const oracle_ldl* f) { return f->Lp[f->n]; }

/* numeric phase.  psign (original order, may be NULL): expected pivot sign +1/-1/0.  Static pivoting rule: with
 * pref = |original diagonal entry| a pivot d is accepted iff sign*d > thr_rel*pref (|d| > thr_rel*pref when the sign is
 * unknown); otherwise it is replaced by sign*repl_rel*pref (repl_abs if pref == 0) and counted in inertia[2].
 * pref_in (original order, may be NULL): caller-supplied reference magnitudes (oracle.py passes |a_kk| + sum_j K_kj^2/|K_jj|
 * for dual rows, the normal-equation diagonal, so that a rank-deficient pivot is caught whatever sign its noise has). */
int oracle_ldl_factor(oracle_ldl* f, const double* val, const signed char* psign, double thr_rel, double repl_rel,
                      double repl_abs, const double* pref_in) {
   const int n = f->n;
   double* Ux = (double*)xmalloc(sizeof(double) * f->nnz);
   for (int p = 0; p < f->nnz; ++p) Ux[f->Umap[p]] = val[p];
   double* Y = (double*)calloc(n, sizeof(double));
   int* pattern = (int*)xmalloc(sizeof(int) * n);
   int* flag = (int*)xmalloc(sizeof(int) * n);
   int* lnz = (int*)calloc(n, sizeof(int));
   f->inertia[0] = f->inertia[1] = f->inertia[2] = 0;
   for (int k = 0; k < n; ++k) {
      int top = n;
      flag[k] = k;
      for (int p = f->Up[k]; p < f->Up[k + 1]; ++p) {
         int i = f->Ui[p];
         Y[i] += Ux[p];
         int len = 0;
         while (i < k && flag[i] != k) {
            pattern[len++] = i;
            flag[i] = k;
            i = f->parent[i];
         }
         while (len > 0) pattern[--top] = pattern[--len];
      }
      double d = Y[k];
      const double pref = pref_in ? pref_in[f->perm[k]] : fabs(d);
      const double thr = thr_rel * pref, repl = pref > 0.0 ? repl_rel * pref : repl_abs;
      Y[k] = 0.0;
      for (; top < n; ++top) {
         const int i = pattern[top];
         const double yi = Y[i];
         Y[i] = 0.0;
         const int p2 = f->Lp[i] + lnz[i];
         for (int p = f->Lp[i]; p < p2; ++p) Y[f->Li[p]] -= f->Lx[p] * yi;
         const double lki = yi / f->D[i];
         d -= lki * yi;
         f->Li[p2] = k;
         f->Lx[p2] = lki;
         lnz[i]++;
      }
      const int s = psign ? psign[f->perm[k]] : 0;
      int pert = 0;
      if (s > 0) { if (!(d > thr)) { d = repl; pert = 1; } }
      else if (s < 0) { if (!(d < -thr)) { d = -repl; pert = 1; } }
      else { if (!(fabs(d) > thr)) { d = d < 0 ? -repl : repl; pert = 1; } }
      if (pert) f->inertia[2]++; else if (d > 0) f->inertia[0]++; else f->inertia[1]++;
      f->D[k] = d;
   }
   free(Ux); free(Y); free(pattern); free(flag); free(lnz);
   return 0;
}

/* x := K^-1 x for nrhs contiguous right-hand sides of length ld (one RHS per row, PardisoSolver.C:276-352) */
void oracle_ldl_solve(const oracle_ldl* f, int nrhs, double* x, int ld) {
   const int n = f->n;
   double* y = (double*)xmalloc(sizeof(double) * n);
   for (int r = 0; r < nrhs; ++r) {
      double* b = x + (size_t)r * ld;
      for (int k = 0; k < n; ++k) y[k] = b[f->perm[k]];
      for (int j = 0; j < n; ++j) {
         const double yj = y[j];
         for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) y[f->Li[p]] -= f->Lx[p] * yj;
      }
      for (int j = 0; j < n; ++j) y[j] /= f->D[j];
      for (int j = n - 1; j >= 0; --j) {
         double s = y[j];
         for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) s -= f->Lx[p] * y[f->Li[p]];
         y[j] = s;
      }
      for (int k = 0; k < n; ++k) b[f->perm[k]] = y[k];
   }
   free(y);
}

void oracle_ldl_inertia(const oracle_ldl* f, int* out3) { out3[0] = f->inertia[0]; out3[1] = f->inertia[1]; out3[2] = f->inertia[2]; }

/* r := b - K x for a symmetric matrix given as lower CSR (SparseStorage::multSym, SparseStorage.C:846-865) */
void oracle_sym_residual(int n, const int* krow, const int* jcol, const double* val, const double* x, const double* b,
                         double* r) {
   for (int i = 0; i < n; ++i) r[i] = b[i];
   for (int i = 0; i < n; ++i)
      for (int p = krow[i]; p < krow[i + 1]; ++p) {
         const int j = jcol[p];
         r[i] -= val[p] * x[j];
         if (j != i) r[j] -= val[p] * x[i];
      }
}
