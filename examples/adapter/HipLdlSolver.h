// The leaf-solver adapter of INTEGRATION.md section 1 as a file: what a PIPS-IPM++ maintainer adds as
// Core/LinearSolvers/HipSolver/HipLdlSolver.h.  It includes the reference's own headers by name (nothing of the reference is
// copied here); tests/test_adapter_compiles.py syntax-checks it against the reference tree when that tree is present.
#ifndef HIP_LDL_SOLVER_H
#define HIP_LDL_SOLVER_H
#include <cstdio>
#include <tuple>
#include <vector>
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
      // refinement like PARDISO's: at most two steps, taken only where the first solve misses the backward error (iparm[7] = 2,
      // PardisoProjectSolver.C:72); solve(nrhs) decides per right-hand side
      check(pips_hip_ldl_set_refinement_backward_error(h, 2, 1e-15), "refinement");
      // the symbolic phase runs once, with the first factorisation (the pattern never changes, DistributedLeafLinearSystem.C:10-42): a
      // border declared before that (declare_border) becomes part of it
   }
   ~HipLdlSolver() override { pips_hip_ldl_destroy(h); }
   void diagonalChanged(int, int) override { matrixChanged(); }
   void matrixChanged() override { check(pips_hip_ldl_factor(h, mat.getStorage().M), "factor"); }
   void solve(Vector<double>& x) override {
      auto& v = dynamic_cast<DenseVector<double>&>(x);
      check(pips_hip_ldl_solve(h, 1, v.elements(), v.length()), "solve");
   }
   // nrhss contiguous right-hand sides, one per row of length n (PardisoSolver.C:276-352); empty ones are skipped by the library, and of the
   // others only the rows colSparsity marks travel to the device (the caller's border pattern, DistributedLinearSystem.C:903; may be null)
   void solve(int nrhss, double* rhss, int* colSparsity) override {
      check(pips_hip_ldl_solve_sparse(h, nrhss, rhss, mat.size(), colSparsity), "solve(nrhs)");
   }
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
   // ---- level 1.5b: the leaf solvers of a rank as ONE batch.  sLinsysRootAug::assembleLocalKKT (:210-227) hands its loop over the children
   //      over: instead of children[it]->addTermToSchurComplBlocked(...) one after the other, collect the children's solvers and call this
   //      once - all leaves run in every launch, one S x S term comes back.  Likewise Lsolve / Ltsolve (:323-365) for their per-child solves.
   static void matrixChanged_and_add_schur_terms(const std::vector<HipLdlSolver*>& leaves, double* SC, int ldSC) {
      std::vector<void*> hs; std::vector<const double*> kv, bv;
      for (const HipLdlSolver* l : leaves) { hs.push_back(l->h); kv.push_back(l->mat.getStorage().M); bv.push_back(l->border ? l->border->getStorage().M : nullptr); }
      check(pips_hip_ldl_factor_schur_batch(hs.data(), (int)hs.size(), kv.data(), bv.data(), SC, ldSC), "factor_schur_batch");
   }
   static void solve_all(const std::vector<HipLdlSolver*>& leaves, const std::vector<double*>& rhs /* one per leaf, nullptr: none */) {
      std::vector<void*> hs;
      for (const HipLdlSolver* l : leaves) hs.push_back(l->h);
      check(pips_hip_ldl_solve_batch(hs.data(), (int)hs.size(), rhs.data()), "solve_batch");
   }
   [[nodiscard]] bool reports_inertia() const override { return true; }
   [[nodiscard]] std::tuple<unsigned, unsigned, unsigned> get_inertia() const override {
      int p, n, z; check(pips_hip_ldl_inertia(h, &p, &n, &z), "inertia"); return {unsigned(p), unsigned(n), unsigned(z)};
   }
};
#endif
