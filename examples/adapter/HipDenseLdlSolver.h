// The dense-root adapter of INTEGRATION.md section 2 as a file: what a PIPS-IPM++ maintainer adds next to DeSymIndefSolver
// (Core/LinearSolvers/DenseSymmetricIndefinitSolver/DeSymIndefSolver.{h,C}) and selects in
// DistributedRootLinearSystem::createDenseSolver (DistributedRootLinearSystem.C:146-158).  It includes the reference's own
// headers by name (nothing of the reference is copied here); tests/test_adapter_compiles.py syntax-checks it against the
// reference tree when that tree is present.
#ifndef HIP_DENSE_LDL_SOLVER_H
#define HIP_DENSE_LDL_SOLVER_H
#include <cstdio>
#include <tuple>
#include "DoubleLinearSolver.h"
#include "DenseSymmetricMatrix.h"
#include "DenseMatrix.h"
#include "DenseVector.hpp"
#include "pipsdef.h"
#include "pips_hip.h"

class HipDenseLdlSolver : public DoubleLinearSolver {
   const DenseSymmetricMatrix& mat;   // non-owning (DeSymIndefSolver.h:44); row-major n x n, lower triangle authoritative
   const int n;
   void* h{};
   static void check(int rc, const char* what) {
      if (rc) { printf("HipDenseLdlSolver - ERROR in %s: %s\n", what, pips_hip_last_error()); MPI_Abort(MPI_COMM_WORLD, -1); }
   }
public:
   // n_primal = locnx of the root: the number of positive pivots the Schur complement is expected to have
   HipDenseLdlSolver(const DenseSymmetricMatrix& SC, int n_primal, int device = -1) : mat(SC), n(static_cast<int>(SC.size())) {
      check(pips_hip_dense_ldl_create(&h, n, n_primal, device), "create");
   }
   ~HipDenseLdlSolver() override { pips_hip_dense_ldl_destroy(h); }
   void diagonalChanged(int, int) override { matrixChanged(); }
   void matrixChanged() override { check(pips_hip_dense_ldl_factor(h, mat.getStorage().M[0], n), "factor"); }
   using DoubleLinearSolver::solve;
   void solve(Vector<double>& x) override {
      auto& v = dynamic_cast<DenseVector<double>&>(x);
      check(pips_hip_dense_ldl_solve(h, 1, v.elements(), n), "solve");
   }
   void solve(GeneralMatrix& rhs_in) override {      // one right-hand side per row (DeSymIndefSolver.C:112-129)
      auto& rhs = dynamic_cast<DenseMatrix&>(rhs_in);
      const auto [nrows, ncols] = rhs.n_rows_columns();
      check(pips_hip_dense_ldl_solve(h, static_cast<int>(nrows), &rhs[0][0], static_cast<int>(ncols)), "solve(matrix)");
   }
   [[nodiscard]] bool reports_inertia() const override { return true; }
   [[nodiscard]] std::tuple<unsigned, unsigned, unsigned> get_inertia() const override {
      int p, ng, z; check(pips_hip_dense_ldl_inertia(h, &p, &ng, &z), "inertia"); return {unsigned(p), unsigned(ng), unsigned(z)};
   }
};
#endif
