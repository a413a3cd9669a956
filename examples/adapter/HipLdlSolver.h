// The leaf-solver adapter of INTEGRATION.md section 1 as a file: what a PIPS-IPM++ maintainer adds as
// Core/LinearSolvers/HipSolver/HipLdlSolver.h.  It includes the reference's own headers by name (nothing of the reference is
// copied here); tests/test_adapter_compiles.py syntax-checks it against the reference tree when that tree is present.
#ifndef HIP_LDL_SOLVER_H
#define HIP_LDL_SOLVER_H
#include <cstdio>
#include <tuple>
#include "DenseMatrix.h"
#include "DoubleLinearSolver.h"
#include "SparseSymmetricMatrix.h"
#include "SparseMatrix.h"
#include "DenseVector.hpp"
#include "pipsdef.h"
#include "pips_hip.h"

class HipLdlSolver : public DoubleLinearSolver {
   const SparseSymmetricMatrix& mat;   // non-owning, like PardisoSolver.h:49-50
   const SparseMatrix* border{};       // non-owning (level 1.5)
   void* h{};
   static void check(int rc, const char* what) {       // reference convention: print + abort (PardisoSolver.C:201-204)
      if (rc) { printf("HipLdlSolver - ERROR in %s: %s\n", what, pips_hip_last_error()); MPI_Abort(MPI_COMM_WORLD, -1); }
   }
public:
   HipLdlSolver(const SparseSymmetricMatrix& K, int n_primal /* locnx */, int device = -1) : mat(K) {
      const auto& st = K.getStorage();                  // lower-triangular CSR, krowM/jcolM/M (SparseStorage.h:45-50)
      check(pips_hip_ldl_create(&h, st.n, st.krowM, st.jcolM, device, 0), "create");
      check(pips_hip_ldl_set_inertia_hint(h, n_primal), "hint");   // expected inertia (locnx, locmy+locmz)
      check(pips_hip_ldl_analyze(h), "analyze");        // once: the pattern never changes (DistributedLeafLinearSystem.C:10-42)
   }
   ~HipLdlSolver() override { pips_hip_ldl_destroy(h); }
   void diagonalChanged(int, int) override { matrixChanged(); }
   void matrixChanged() override { check(pips_hip_ldl_factor(h, mat.getStorage().M), "factor"); }
   void solve(Vector<double>& x) override {
      auto& v = dynamic_cast<DenseVector<double>&>(x);
      check(pips_hip_ldl_solve(h, 1, v.elements(), v.length()), "solve");
   }
   // nrhss contiguous right-hand sides, one per row of length n (PardisoSolver.C:276-352); empty ones are skipped by the library, colSparsity is not needed
   void solve(int nrhss, double* rhss, int*) override { check(pips_hip_ldl_solve(h, nrhss, rhss, mat.size()), "solve(nrhs)"); }
   void solve(GeneralMatrix& rhs_in) override {
      auto& rhs = dynamic_cast<DenseMatrix&>(rhs_in);
      const auto [nrows, ncols] = rhs.n_rows_columns();
      check(pips_hip_ldl_solve(h, (int)nrows, &rhs[0][0], (int)ncols), "solve(matrix)");
   }
   // ---- level 1.5: the leaf's Schur term formed where the factor lives.  Not part of DoubleLinearSolver:
   //      DistributedLeafLinearSystem::addTermToSchurComplBlocked (:214-252) calls these two instead of the K4-K6 chunk loop when its
   //      solver is a HipLdlSolver.  border_left_transp is the S x N_i CSR matrix the loop walks (rows = Schur column ids).
   void declare_border(const SparseMatrix& border_left_transp) {      // once, before the first factorisation
      const auto& st = border_left_transp.getStorage();
      check(pips_hip_ldl_set_border(h, st.m, st.krowM, st.jcolM), "set_border");
      border = &border_left_transp;
   }
   // = matrixChanged() followed by addTermToSchurComplBlocked(): SC is the S x S DenseSymmetricMatrix storage (row-major, lower triangle)
   void matrixChanged_and_add_schur_term(double* SC, int ldSC) {
      check(pips_hip_ldl_factor_schur(h, mat.getStorage().M, border->getStorage().M, SC, ldSC), "factor_schur");
   }
   [[nodiscard]] bool reports_inertia() const override { return true; }
   [[nodiscard]] std::tuple<unsigned, unsigned, unsigned> get_inertia() const override {
      int p, n, z; check(pips_hip_ldl_inertia(h, &p, &n, &z), "inertia"); return {unsigned(p), unsigned(n), unsigned(z)};
   }
};
#endif
