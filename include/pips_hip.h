/* pips_hip.h — C ABI of the MI355X-native KKT linear-system backend for PIPS-IPM++ (libpipship.so).
 *
 * Every entry point returns 0 on success and a non-zero PIPS_ERR_* code on failure; pips_hip_last_error() gives the
 * message.  The C++ adapter maps non-zero onto the reference's convention of printing and calling
 * MPI_Abort (PardisoSolver.C:201-204,222-225).  No call throws, none takes or returns a C++/torch type.
 *
 * "host" pointers are ordinary memory, "dev" pointers are HIP device memory of the device the handle was created on.
 * All matrices are CSR, row-major, 0-based, int32 indices, fp64 values (SparseStorage.h:45-50).
 */
#ifndef PIPS_HIP_H
#define PIPS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* pips_hip_last_error(void);
/* how often this process has made the host wait for the device inside the library so far (stream / event / device synchronisations and
 * blocking copies, all counted at one place: csrc/common.h) - a diagnostic: the difference around a call sequence is its number of host stops */
long long pips_hip_host_wait_count(void);
/* the same per call site, as lines "file:line count" written into buf (at most cap bytes, NUL-terminated); returns the bytes written */
int pips_hip_host_wait_sites(char* buf, int cap);
/* number of visible HIP devices (0 without a GPU; never fails) */
int pips_hip_device_count(void);

/* ---------------------------------------------------------------------------------------------------------------
 * 1. Leaf solver handle: drop-in for DoubleLinearSolver behind DistributedFactory::make_leaf_solver
 *    (DoubleLinearSolver.h:24-72, DistributedFactory.cpp:66-112).  Replaces PardisoSolver / Ma27Solver / Ma57Solver.
 * ------------------------------------------------------------------------------------------------------------- */
/* pattern of the lower-triangular CSR matrix the solver will factorise (SparseSymmetricMatrix, isLower;
 * PardisoSolver.C:51-135 reads the same krowM/jcolM).  The arrays are copied.  device < 0: current device. */
int pips_hip_ldl_create(void** handle, int n, const int* krow, const int* jcol, int device, int flags);
/* optional: the leading n_primal rows are expected to yield positive pivots, the rest negative
 * (= the inertia the regularisation loop asks for: DistributedLeafLinearSystem.C:22, LinearSystem.C:296-325) */
int pips_hip_ldl_set_inertia_hint(void* handle, int n_primal);
/* relative pivot threshold / replacement (times max|K|); PARDISO's counterpart is the 1e-8 pivot perturbation */
int pips_hip_ldl_set_pivot_rule(void* handle, double thr_rel, double repl_rel);
/* iterative refinement of every solve: at most max_steps steps; tol > 0 stops as soon as ||r||inf <= tol*||rhs||inf
 * (PARDISO: iparm[7]=2, PardisoProjectSolver.C:72), tol = 0 always does max_steps steps.  Default (1, 0). */
int pips_hip_ldl_set_refinement(void* handle, int max_steps, double tol);
/* same, the stopping test being the normwise backward error ||r||inf / (max|K| ||x||inf + ||rhs||inf) <= tol - what PARDISO's adaptive
 * refinement looks at; solve(nrhs) decides per right-hand side (the measures of a chunk read back with one copy, the correction solve only for
 * the columns whose first solve was not accurate enough).  The adapters use (2, 1e-15): iparm[7] = 2, PardisoProjectSolver.C:72 */
int pips_hip_ldl_set_refinement_backward_error(void* handle, int max_steps, double tol);
/* deterministic mode of this leaf (see pips_hip_batch_set_deterministic): factor(), solve() and solve(nrhs) repeat to the bit - no FP64
 * atomics on their path; several right-hand sides go panel by panel through the slot / gather forward substitution.  Before the first
 * factorisation.  (The reference's breakdown tests compare bitwise between runs: pipsdef.h:35,108.) */
int pips_hip_ldl_set_deterministic(void* handle, int on);
/* symbolic phase (ordering, supernodes, device allocation); pattern-only, done once */
int pips_hip_ldl_analyze(void* handle);
/* = DoubleLinearSolver::matrixChanged(): numeric LDL^T of the current values (host array of length nnz, CSR order) */
int pips_hip_ldl_factor(void* handle, const double* vals_host);
/* = DoubleLinearSolver::solve(int nrhss, double* rhss, int* colSparsity) (PardisoSolver.C:276-352): nrhs contiguous
 * right-hand sides of length ld >= n, overwritten by the solutions; all-zero right-hand sides are left out of the transfer and
 * of the solve, as the reference's packing does.  nrhs = 1 is DoubleLinearSolver::solve(Vector&). */
int pips_hip_ldl_solve(void* handle, int nrhs, double* rhs_inout_host, int ld);
/* = DoubleLinearSolver::get_inertia(): (positive, negative, zero/perturbed) pivots of the last factorisation */
int pips_hip_ldl_inertia(void* handle, int* pos, int* neg, int* zero);
/* The leaf's Schur term computed where the factor lives (INTEGRATION.md level 1.5): replaces the host's K4-K6 chunk loop
 * (addBiTLeftKiBiRightToResBlockedParallelSolvers, DistributedLinearSystem.C:766-1047; caller DistributedLeafLinearSystem.C:214-252),
 * which ships every border column dense over PCIe and back.
 *   set_border     before the analysis: pattern of Br_i^T, CSR with S rows (Schur column ids) over the n rows of K_i
 *                  (border_left_transp); empty rows are skipped as the reference skips empty border columns (:870-874)
 *   factor_schur   = matrixChanged() + addTermToSchurComplBlocked(): factorises K_i (values as in pips_hip_ldl_factor) and adds
 *                  -Br_i^T K_i^-1 Br_i to SC_host, the S x S row-major DenseSymmetricMatrix storage (lower triangle, ldSC >= S);
 *                  solves and inertia queries work afterwards as after pips_hip_ldl_factor */
int pips_hip_ldl_set_border(void* handle, int S, const int* Bt_rowptr, const int* Bt_colidx);
int pips_hip_ldl_factor_schur(void* handle, const double* K_vals_host, const double* Bt_vals_host, double* SC_host, int ldSC);
/* Variants of the solve.  solve_dev: the right-hand sides are device memory (nrhs vectors at distance ld >= n; no PCIe, asynchronous on
 * the handle's stream).  solve_sparse = DoubleLinearSolver::solve(int nrhss, double* rhss, int* colSparsity) with the third argument
 * honoured: col_sparsity[i] != 0 marks the rows that can be non-zero in any right-hand side (the border's row pattern the caller builds,
 * DistributedLinearSystem.C:903); only those rows of the non-zero right-hand sides travel to the device.  NULL: same as pips_hip_ldl_solve. */
int pips_hip_ldl_solve_dev(void* handle, int nrhs, double* rhs_inout_dev, long long ld);
int pips_hip_ldl_solve_sparse(void* handle, int nrhs, double* rhs_inout_host, int ld, const int* col_sparsity);
/* Array-of-handles entries (INTEGRATION.md level 1.5b): the leaf solvers of a rank run as ONE batch from the host's loop over its children
 * (sLinsysRootAug::assembleLocalKKT :210-227; Lsolve / Ltsolve :323-365).  The first call binds the handles (created, borders set, same
 * Schur dimension and device; the same array in the same order afterwards) into one batch engine analysed over all of them.
 *   factor_schur_batch  = for every leaf matrixChanged() + addTermToSchurComplBlocked(): K_vals_host[i] / Bt_vals_host[i] are leaf i's
 *                         values (CSR order); adds sum_i -Br_i^T K_i^-1 Br_i to SC_host (row-major, lower triangle, ldSC >= S) with one
 *                         S x S buffer on the device and one transfer; SC_host == NULL: factorise only
 *   solve_batch         one right-hand side per leaf (host pointers of length n_i, NULL = none for that leaf), overwritten
 *   solve_batch_dev     the flat device vector of all leaves, block after block
 *   inertia_batch       per leaf (positive, negative, zero / perturbed)
 * pips_hip_ldl_inertia / pips_hip_ldl_solve on a bound handle go through the batch (a single-leaf solve then costs a batch solve). */
int pips_hip_ldl_factor_schur_batch(void* const* handles, int n, const double* const* K_vals_host, const double* const* Bt_vals_host,
                                    double* SC_host, int ldSC);
int pips_hip_ldl_solve_batch(void* const* handles, int n, double* const* rhs_inout_host);
int pips_hip_ldl_solve_batch_dev(void* const* handles, int n, double* x_dev);
int pips_hip_ldl_inertia_batch(void* const* handles, int n, int* pos, int* neg, int* zero);
/* diagnostics of the symbolic phase: what[0]=nnz(L) what[1]=n_head what[2]=tail m what[3]=#head supernodes
 * what[4]=#levels what[5]=factor flops (rounded); and what[6]=refinement steps the last solve took, what[7]=how the last solve(nrhs) went
 * (0 one sweep per right-hand side, 1 interleaved panels on the matrix pipe, 2 the same with the deterministic forward substitution) */
int pips_hip_ldl_info(void* handle, int64_t* what, int n_what);
/* copies the fill-reducing permutation (perm[k] = original index eliminated k-th) */
int pips_hip_ldl_get_perm(void* handle, int* perm);
void pips_hip_ldl_destroy(void* handle);

/* ---------------------------------------------------------------------------------------------------------------
 * 2. Dense root solver: drop-in for DeSymIndefSolver (DeSymIndefSolver.C:56-168; dsytrf_/dsytrs_).
 * ------------------------------------------------------------------------------------------------------------- */
/* n_primal >= 0: inertia hint (rows [0, n_primal) are expected to give positive pivots, the rest negative ones) - the caller vouches
 * for a quasi-definite order and the factorisation keeps it (static pivots); n_primal < 0: no hint, the matrix is pivoted like
 * dsytrf does: Bunch-Kaufman 1 x 1 / 2 x 2 pivots, the search bounded to the 128 x 128 diagonal tiles.
 * pips_hip_dense_ldl_set_pivoting overrides: 0 static order, 1 Bunch-Kaufman. */
int pips_hip_dense_ldl_create(void** handle, int n, int n_primal, int device);
int pips_hip_dense_ldl_set_pivoting(void* handle, int mode);
/* With the static order on one rank the factorisation is ONE launch of tile tasks that wait for each other through flags
 * (csrc/rootkernel.hip.h), in the order of a list schedule the host builds once per dimension (csrc/rootplan.cpp).  This entry returns
 * that schedule for ntc tile columns of 128 (no device involved; tests): (kind 0 update / 1 trsm / 2 diagonal tile, tile row, tile column,
 * k0 | k1 << 16) quadruples - the bulk list, then the chain's list, each in ticket order.  workers, qmin <= 0, urgent, chain_slots < 0:
 * the defaults.  out may be NULL (counts only). */
int pips_root_plan_build(int ntc, int workers, int qmin, int urgent, int chain_slots, int* out, long long cap, long long* n_tasks,
                         long long* n_chain_tasks, double* makespan_us);
/* Several ranks hold the same matrix (the reduced Schur complement): factorise it column-cyclically over the ranks instead of
 * redundantly on each (DistributedRootLinearSystem.C:1436-1464 does the latter).  Tile column j belongs to rank j mod n_ranks; the
 * owner's panel reaches every rank through the communicator (pips_hip_comm_create / _create_external), every rank ends with the
 * complete factor and solves locally.  Collective: every rank must call factor with the same matrix.  comm == NULL or n_ranks <= 1
 * switches it off.  Fused path: PIPS_HIP_ROOT_DISTRIBUTED=1. */
int pips_hip_dense_ldl_set_distributed(void* handle, void* comm, int rank, int n_ranks);
/* = DeSymIndefSolver::matrixChanged(): A is the n x n row-major DenseSymmetricMatrix storage (DenseStorage.C:64-83,
 * lower triangle authoritative, lda = n); it is copied to the device and factorised */
int pips_hip_dense_ldl_factor(void* handle, const double* A_host, int lda);
/* factorise a matrix that already lives on the device in column-major-lower form (fused path, see section 3) */
int pips_hip_dense_ldl_factor_dev(void* handle, const double* A_dev, int lda);
int pips_hip_dense_ldl_solve(void* handle, int nrhs, double* rhs_inout_host, int ld);
int pips_hip_dense_ldl_solve_dev(void* handle, double* rhs_inout_dev);
int pips_hip_dense_ldl_inertia(void* handle, int* pos, int* neg, int* zero);
void pips_hip_dense_ldl_destroy(void* handle);

/* ---------------------------------------------------------------------------------------------------------------
 * 3. Batched, device-resident fast path for all leaves owned by one GPU.  Replaces the host loops of
 *    DistributedRootLinearSystem::factor2 (:206-243), sLinsysRootAug::assembleLocalKKT (:210-227),
 *    DistributedLeafLinearSystem::addTermToSchurComplBlocked (:214-252) with
 *    addBiTLeftKiBiRightToResBlockedParallelSolvers / addLeftBorderTimesDenseColsToResTranspDense
 *    (DistributedLinearSystem.C:766-1047,1115-1175), and the leaf parts of Lsolve/Ltsolve
 *    (addLniziLinkCons DistributedLeafLinearSystem.C:171-212, LniTransMult DistributedLinearSystem.C:430-483).
 * ------------------------------------------------------------------------------------------------------------- */
/* S = dimension of the root Schur complement (n0 + my0 + myl + mzl); stream: hipStream_t or NULL (default stream) */
int pips_hip_batch_create(void** handle, int n_blocks, int S, int device, void* stream);
/* block b: lower CSR pattern of K_b (n x n), leading primal rows, and Br_b^T as CSR with S rows over n columns
 * (build it with pips_border_assemble); Bt_* may be NULL for a block without border.  Arrays are copied. */
int pips_hip_batch_set_block(void* handle, int b, int n, int n_primal, const int* K_rowptr, const int* K_colidx,
                             const int* Bt_rowptr, const int* Bt_colidx, const double* Bt_val);
int pips_hip_batch_set_options(void* handle, int force_n_head, int refine_steps, double thr_rel, double repl_rel);
/* How the Schur contribution SC -= Br^T K^-1 Br is formed.  1: augmented partial factorisation (border columns ride as rows
 * of every panel; dense MFMA work - the PardisoSchurSolver idea, PardisoSchurSolver.C:83-389).  2: blocked solves with the
 * plain factor, 32 border columns at a time over all blocks (addTermToSchurComplBlocked, DistributedLeafLinearSystem.C:
 * 214-252 + DistributedLinearSystem.C:766-1175: densify, multi-RHS solve, sparse border times dense columns).  0 (default):
 * chosen at analyze time from the symbolic flop / byte counts - dense factors take 1, sparse structured ones 2.
 * Call before analyze; get_schur_mode reports the choice. */
int pips_hip_batch_set_schur_mode(void* handle, int mode);
int pips_hip_batch_get_schur_mode(void* handle, int* mode);
/* Deterministic mode (call before analyze): bit-identical factors, Schur complement, inertia and solutions from run to run, at a
 * cost in time and memory ("deterministic mode").  By default the head scatter, the Schur accumulation over the
 * blocks and the head substitution use hardware FP64 atomics, whose order of arrival decides the last bits - the reference's
 * breakdown tests (PIPSisZero, pipsdef.h:35,108) and any bitwise comparison between runs see that. */
int pips_hip_batch_set_deterministic(void* handle, int on);
/* add_regularization_local_kkt (DistributedLeafLinearSystem.C:108-143): K diagonal += primal on the leading n_primal rows
 * of every block, -= dual on the remaining rows; used by the inertia-correcting loop (LinearSystem.C:296-325) */
int pips_hip_batch_add_regularization(void* handle, double primal, double dual);
/* iterative refinement policy of pips_hip_batch_solve* (see pips_hip_ldl_set_refinement); default (1, 0) */
int pips_hip_batch_set_refinement(void* handle, int max_steps, double tol);
/* same, but the stopping test is the normwise backward error  ||r_b||inf / (max|K_b| ||x_b||inf + ||rhs_b||inf) <= tol
 * for every block b — the criterion PARDISO's iparm[7] refinement uses ("stops if a satisfactory level of accuracy of the
 * solution in terms of backward error is achieved") */
int pips_hip_batch_set_refinement_backward_error(void* handle, int max_steps, double tol);
/* error measure the last adaptive solve saw at its final check */
double pips_hip_batch_last_refinement_measure(void* handle);
/* refinement steps the last solve actually took */
int pips_hip_batch_last_refinement_steps(void* handle);
/* symbolic phase for all blocks (n_threads host threads) + device setup */
int pips_hip_batch_analyze(void* handle, int n_threads);
/* upload all values of K_b (CSR order, host) — needed once; afterwards only diagonals change (a2) */
int pips_hip_batch_set_values(void* handle, int b, const double* K_val_host);
/* K diagonals of all blocks from one flat device vector (block after block, rows in [x|y|z] order):
 * put_primal_diagonal / clear_dual_equality_diagonal / put_dual_inequalites_diagonal / add_regularization_local_kkt
 * (DistributedLeafLinearSystem.C:88-143) collapsed into one scatter */
int pips_hip_batch_set_diagonals_dev(void* handle, const double* diag_dev);
int pips_hip_batch_set_diagonals(void* handle, const double* diag_host);
/* factor every K_b and accumulate  SC -= sum_b Br_b^T K_b^-1 Br_b  into SC_dev: S x S, column-major with leading
 * dimension ldSC, lower triangle written (== the upper triangle of the reference's row-major layout's transpose; the
 * matrix is symmetric).  SC_dev may be NULL to factor only.  Asynchronous on the handle's stream. */
int pips_hip_batch_factor(void* handle, double* SC_dev, int ldSC);
/* x_b := K_b^-1 x_b for every block; x_dev is the flat vector of all blocks (sum of n).  Asynchronous. */
int pips_hip_batch_solve_dev(void* handle, double* x_dev);
int pips_hip_batch_solve(void* handle, double* x_host);
/* b0 += alpha * sum_b Br_b^T z_b   (addLniziLinkCons uses alpha = -1) */
int pips_hip_batch_border_tmult_dev(void* handle, const double* z_dev, double* b0_dev, double alpha);
/* t_b += alpha * Br_b x0 for every block   (LniTransMult) */
int pips_hip_batch_border_mult_dev(void* handle, const double* x0_dev, double* t_dev, double alpha);
int pips_hip_batch_inertia(void* handle, int b, int* pos, int* neg, int* zero);
/* what[0]=sum nnz(L) what[1]=sum n what[2]=sum n_head what[3]=sum tail m what[4]=#head supernodes what[5]=max levels
 * what[6]=factor flops what[7]=border (TRSM+SYRK) flops what[8]=arena bytes what[9]=max tail tile columns
 * what[10]=bytes of the head-to-head update position tables what[11]=sum of non-empty border columns what[12]=sum nnz(K lower)
 * what[13]=1 if solveCompressed takes its Ltsolve from the augmented factor (one backward sweep) while no pivot is perturbed
 * what[14]=1 multifrontal head ... what[24]=blocks analysed with the border split of the fronts (python/capi.py LeafBatch.info names all) */
int pips_hip_batch_info(void* handle, int64_t* what, int n_what);
int pips_hip_batch_sync(void* handle);
/* per-phase device time in ms (HIP events on the handle's stream) of the last pips_hip_batch_factor
 *   ms[0]=scatter ms[1]=head ms[2]=tail update GEMM ms[3]=tail diag ms[4]=tail trsm ms[5]=Schur SYRK ms[6]=total
 * and, summed over the solves since that factorisation,
 *   ms[7]=permute in/out ms[8]=head forward ms[9]=tail sweeps + D ms[10]=head backward ms[11]=refinement residual + norms
 * cnt[i] = number of records of phase i (n <= 16).  Enable with pips_hip_batch_set_timing(handle, 1). */
int pips_hip_batch_set_timing(void* handle, int on);
int pips_hip_batch_get_timing(void* handle, double* ms, int64_t* cnt, int n);
void pips_hip_batch_destroy(void* handle);

/* ---------------------------------------------------------------------------------------------------------------
 * 3b. Fused two-level system of one rank: leaves + replicated dense root + Schur reduction.
 *     factorize      = DistributedRootLinearSystem::factor2 (:206-243): initializeKKT, children factor2, assembleLocalKKT,
 *                      reduceKKT, finalizeKKT (sLinsysRootAug::finalizeKKTdense :1769-1796), factorizeKKT
 *     solve_compressed = DistributedLinearSystem::solveCompressed (:409-420): Lsolve / Dsolve / Ltsolve
 *                      (sLinsysRootAug.C:323-365); root vector layout [x0 | y0 | ylink | zlink] (mz0 = 0)
 * ------------------------------------------------------------------------------------------------------------- */
/* batch: analyzed leaf batch (created with the same S).  A0 (my0 x n0), F0 (myl x n0), G0 (mzl x n0) are the root's own
 * constraint blocks (CSR, may be NULL).  comm: handle from pips_hip_comm_create or NULL when n_ranks == 1. */
int pips_hip_kkt_create(void** handle, void* batch, int n0, int my0, int myl, int mzl, const int* A0_rowptr,
                        const int* A0_colidx, const double* A0_val, const int* F0_rowptr, const int* F0_colidx,
                        const double* F0_val, const int* G0_rowptr, const int* G0_colidx, const double* G0_val, void* comm,
                        int rank, int n_ranks);
/* leaf_diag_dev: flat K diagonals of this rank's blocks (NULL: keep); xdiag0_dev: n0 primal diagonal of the root
 * (xDiag); zdiag_link_dev: mzl entries or NULL.  Asynchronous on the batch's stream. */
/* Sparse-root variant of pips_hip_kkt_create (createSchurCompSymbSparseUpper DistributedProblem.cpp:2235+,
 * finalizeKKTsparse sLinsysRootAug.C:1629-1739, PardisoIndefSolver as root solver): the Schur complement is kept as the value
 * array of a lower-triangular CSR pattern - dense x0 block, per block the clique on its non-empty border columns, the root
 * rows A0 / F0 / G0, full diagonal - and factorised / solved by the same sparse LDL^T machinery as the leaves.  Meant for
 * 2-link structure (a linking row touches two blocks), where SC is sparse and S may be far beyond what S x S storage allows.
 * blk_cols_ptr / blk_cols: border column sets (ascending Schur column ids) of ALL n_blocks_global blocks of the problem -
 * required with n_ranks > 1 so that every rank reduces the same value array; NULL = the blocks of this batch.
 * The batch must be analyzed and use Schur mode 1 (set it explicitly for structured blocks). */
int pips_hip_kkt_create_sparse(void** handle, void* batch, int n0, int my0, int myl, int mzl, const int* A0_rowptr,
                               const int* A0_colidx, const double* A0_val, const int* F0_rowptr, const int* F0_colidx,
                               const double* F0_val, const int* G0_rowptr, const int* G0_colidx, const double* G0_val,
                               int n_blocks_global, const int* blk_cols_ptr, const int* blk_cols, void* comm, int rank,
                               int n_ranks);
/* pattern and values (host copies) of the sparse Schur complement; any output may be NULL; *nnz = number of entries */
int pips_hip_kkt_get_schur_sparse(void* handle, int* nnz, int* rowptr, int* colidx, double* val_host);
/* what[0] = elimination order of the sparse root: 0 minimum degree, 1 linking rows then x0 as a band of dense tiles, 2 linking rows
 * dissected around x0 / the root equality rows (a chain-like Schur complement becomes a tree of small fronts: default where the tile
 * envelope is thin; PIPS_HIP_SPARSE_ROOT_BAND=0|1|2 forces one); what[1 + i] = pips_hip_batch_info entry i of the root's engine */
int pips_hip_kkt_sparse_root_info(void* handle, int64_t* what, int n_what);
int pips_hip_kkt_factorize(void* handle, const double* leaf_diag_dev, const double* xdiag0_dev, const double* zdiag_link_dev);
/* sLinsysRootAug::add_regularization_local_kkt (sLinsysRootAug.C:1545-1600) as a setting of the following factorizations: the
 * x0 diagonal of the Schur complement gets + primal, the diagonal of its dual rows (y0, y_link, z_link) - dual; (0, 0) = off */
int pips_hip_kkt_set_root_regularization(void* handle, double primal, double dual);
/* root inequality rows C0 (mz0 x n0, CSR): adds -C0^T diag(zdiag0)^-1 C0 to SC at every factorize (sLinsysRootAug.C:
 * 1276-1338) and the z0 elimination to solve_compressed (:384-466); zdiag0 (< 0, = nOmegaInv of the root, caller-owned
 * device vector of mz0 entries) must be set before pips_hip_kkt_factorize */
int pips_hip_kkt_set_root_inequalities(void* handle, int mz0, const int* C0_rowptr, const int* C0_colidx, const double* C0_val);
int pips_hip_kkt_set_zdiag0_dev(void* handle, const double* zdiag0_dev);
/* in place: b0 (replicated on every rank; [x0 | y0 | ylink | zlink], or [x0 | y0 | z0 | ylink | zlink] when mz0 > 0) and
 * the flat leaf vector of this rank's blocks */
int pips_hip_kkt_solve_compressed(void* handle, double* b0_dev, double* b_leaf_dev);
int pips_hip_kkt_get_schur(void* handle, double** SC_dev, int* ld);
int pips_hip_kkt_root_inertia(void* handle, int* pos, int* neg, int* zero);
/* dense root: 0 static pivot order (default: the Schur complement the leaves build is quasi-definite in the order x0, duals),
 * 1 Bunch-Kaufman inside the diagonal tiles (default once pips_hip_kkt_set_root_inequalities folds -C0^T Omega^-1 C0 into the x0 block) */
int pips_hip_kkt_set_root_pivoting(void* handle, int mode);
/* phase times (ms, HIP events on the streams the work runs on) of the last pips_hip_kkt_factorize and of the solveCompressed calls
 * since; on while the batch's timing switch is on:  ms[0]=diagonals + zero SC  ms[1]=leaf factorisation  ms[2]=Schur reduction
 * ms[3]=finalize  ms[4]=root factorisation  ms[5]=Lsolve leaf solves  ms[6]=Lsolve border product + b0 reduction  ms[7]=Dsolve
 * ms[8]=Ltsolve  ms[9]=x_i = z_i - u_i  ms[10]=panel-wise Schur reduction on its own stream, summed over the panels (ms[2] is
 * then what the main stream waited for it: the exposed part).  What bench.py's phase table is made of. */
int pips_hip_kkt_get_timing(void* handle, double* ms, int64_t* cnt, int n);
/* solveCompressed as a captured and replayed HIP graph : the launch sequence of a call is fixed between
 * factorisations, so it is captured once per (b0, b_leaf pointers, Ltsolve path) and replayed - what pays on launch-bound problems and
 * inside the outer BiCGStab, whose preconditioner this call is (LinearSystem.C:550-798).  Only with a fixed number of refinement steps
 * (pips_hip_batch_set_refinement with tol = 0: the adaptive variant reads norms on the host between steps), one rank, the dense root;
 * otherwise the call is issued launch by launch as before.  stats: captures made / replays so far. */
/* The dense root is factorised on a stream of its own so that the leaf solves of the next solveCompressed run beside it (default on;
 * PIPS_HIP_ROOT_SYNC=1 switches it off for the process).  own_stream = 0: on the main stream - for callers that ask for the root's
 * inertia right after every factorisation (the order of LinearSystem::factorize_with_correct_inertia), where nothing can overlap. */
int pips_hip_kkt_set_root_stream(void* handle, int own_stream);
int pips_hip_kkt_set_solve_graph(void* handle, int on);
int pips_hip_kkt_solve_graph_stats(void* handle, int64_t* captures, int64_t* replays);
/* 1 if the last pips_hip_kkt_solve_compressed took its Ltsolve from the augmented factor (one unrefined backward sweep: only while no
 * pivot is perturbed and the refined leaf solve of the same call needed no refinement step), 0 if by border product + refined solve */
int pips_hip_kkt_last_ltsolve_from_factor(void* handle, int* flag);
/* Which way the last pips_hip_kkt_solve_compressed went (Lsolve / Ltsolve of sLinsysRootAug.C:323-365 with addLniziLinkCons and
 * LniTransMult, DistributedLeafLinearSystem.C:171-212, DistributedLinearSystem.C:430-483):
 *   0  two leaf solves with adaptive refinement (K_i^-1 b_i, then K_i^-1 Br_i x0) and the two sparse border products;
 *   1  refined Lsolve, Ltsolve by one backward sweep of the augmented factor;
 *   2  one forward and one backward sweep of the augmented factor [L 0; L_b I]: the forward sweep leaves -Br_i^T K_i^-1 b_i in the
 *      border rows, the backward sweep started from D^-1 y with the border rows at x0 gives K_i^-1 (b_i - Br_i x0);
 *   3  the sweeps of 2 with their result checked: r_i = (b_i - Br_i x0) - K_i x_i within the refinement tolerance, else the result is
 *      discarded and the call is repeated the refined way (one rank only).
 * 1 and 2 carry no refinement; they are taken only while no pivot of the factorisation is perturbed, and 2 only after a witness on
 * the same factors: a solveCompressed that went way 3, or way 0 / 1 with its refined solves meeting the backward-error tolerance
 * without a step (the first solveCompressed after every factorisation; needs adaptive refinement, pips_hip_batch_set_refinement*). */
int pips_hip_kkt_last_solve_path(void* handle, int* path);
/* How often a solveCompressed that goes by sweeps (way 2) is measured against the leaf rows like the witness (way 3): every `every`-th
 * one; 1 (the default) = every solve is measured, as the reference's PARDISO measures and refines every leaf solve (iparm[7] = 2,
 * PardisoProjectSolver.C:72); 0 = only the first after a factorisation.  A solve whose measure exceeds the refinement tolerance is
 * repeated the refined way from the saved right-hand side and the sweeps stay off for these factors; with several ranks the ranks decide
 * together (one number all-reduced per solveCompressed).  _counts: measured solves and failed measures so far. */
int pips_hip_kkt_set_solve_check(void* handle, int every);
int pips_hip_kkt_solve_check_counts(void* handle, long long* checked, long long* failed);
void pips_hip_kkt_destroy(void* handle);

/* plain device buffers for hosts that do not bring their own allocator */
int pips_hip_malloc(void** dev_ptr, size_t bytes);
int pips_hip_free(void* dev_ptr);
int pips_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int pips_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int pips_hip_memset(void* dst_dev, int value, size_t bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * 4. Reduction of the Schur complement and of b0 across the GPUs of a node (replaces MPI_Allreduce in
 *    DistributedRootLinearSystem::reduceKKTdense :860-881 / submatrixAllReduce* :1614-1707 and
 *    PIPS_MPIsumArrayInPlace in sLinsysRootAug::Lsolve :340-341).  RCCL over xGMI.
 * ------------------------------------------------------------------------------------------------------------- */
int pips_hip_comm_unique_id(void* id128);                 /* 128-byte ncclUniqueId, produced on rank 0 */
int pips_hip_comm_create(void** comm, const void* id128, int n_ranks, int rank, int device);
int pips_hip_allreduce_sum(void* comm, double* buf_dev, size_t n, void* stream);
void pips_hip_comm_destroy(void* comm);
/* Communicator backed by the host's own collective (what an MPI build of the reference would pass: a GPU-aware
 * MPI_Allreduce(MPI_IN_PLACE, buf_dev, n, MPI_DOUBLE, MPI_SUM, comm) as in PIPS_MPIsumArrayInPlace, pipsdef.h).  The
 * callback must return 0 after buf_dev (device memory, n doubles) holds the sum over all ranks and is safe to read
 * from any stream; the library synchronises its own stream before calling it. */
typedef int (*pips_hip_allreduce_cb)(void* user, double* buf_dev, size_t n);
int pips_hip_comm_create_external(void** comm, pips_hip_allreduce_cb allreduce, void* user);
/* The same sum as reduce-scatter + all-gather (ncclReduceScatter + ncclAllGather; the slice a rank owns after the first phase is
 * what a distributed root would keep): buf_dev must have room for ceil(n / n_ranks) * n_ranks doubles.  Host-supplied pair for an
 * external communicator (MPI_Reduce_scatter_block / MPI_Allgather, in place on buf_dev: reduce_scatter leaves the sum of slice
 * `rank` at buf_dev + rank * chunk, all_gather replicates every slice); without it the external all-reduce is used.
 * PIPS_HIP_SC_REDUCE=rsag makes pips_hip_kkt_factorize reduce the Schur complement this way. */
typedef int (*pips_hip_reduce_scatter_cb)(void* user, double* buf_dev, size_t chunk);
typedef int (*pips_hip_all_gather_cb)(void* user, double* buf_dev, size_t chunk);
int pips_hip_comm_set_external_rsag(void* comm, int n_ranks, int rank, pips_hip_reduce_scatter_cb reduce_scatter, pips_hip_all_gather_cb all_gather);
/* in-place all-gather of n_parts (= number of ranks) equal parts (rank r's part at buf_dev + r * chunk on entry); without an all-gather in
 * the communicator: the all-reduce, for which the other ranks' parts must be zero on entry */
int pips_hip_all_gather(void* comm, double* buf_dev, size_t chunk, int n_parts, void* stream);
/* Broadcast of n device doubles from rank `root` (the panel of the distributed root factorisation, pips_hip_dense_ldl_set_distributed;
 * the reference has every rank factorise the same matrix instead, DistributedRootLinearSystem.C:1436-1464).  The library's RCCL
 * communicator uses ncclBroadcast; a host-supplied communicator its broadcast callback (MPI_Bcast on device pointers) when one was
 * registered, else an all-reduce in which the other ranks contribute zeros - exact, twice the bytes on the wire; the caller clears the
 * buffer on those ranks first (pips_hip_comm_has_broadcast tells which way it goes). */
typedef int (*pips_hip_broadcast_cb)(void* user, double* buf_dev, size_t n, int root);
int pips_hip_comm_set_external_broadcast(void* comm, int n_ranks, int rank, pips_hip_broadcast_cb broadcast);
int pips_hip_broadcast(void* comm, double* buf_dev, size_t n, int root, void* stream);
int pips_hip_comm_has_broadcast(void* comm);
int pips_hip_allreduce_sum_rsag(void* comm, double* buf_dev, size_t n, void* stream);
int pips_hip_comm_size(void* comm);

/* ---------------------------------------------------------------------------------------------------------------
 * 4b. Flat-arena vector kernels: DistributedVector<T>/DenseVector<T> operations used around the path
 *     (DistributedVector.C:406-460,1160-1340; DenseVector.cpp:281-516).  n = arena length; skip_root = number of leading
 *     (replicated root) entries a non-special rank must not count in sums (iAmSpecial, DistributedVector.C:1293-1303).
 *     Reductions return one host scalar (the caller all-reduces it across ranks where the reference does).
 * ------------------------------------------------------------------------------------------------------------- */
int pips_hip_vec_axpy(long long n, double a, const double* x_dev, double* y_dev, void* stream);           /* add            */
int pips_hip_vec_axpby(long long n, double a, const double* x_dev, double b, double* y_dev, void* stream);
int pips_hip_vec_scale(long long n, double a, double* y_dev, void* stream);                               /* scale / negate */
int pips_hip_vec_copy(long long n, const double* x_dev, double* y_dev, void* stream);                     /* copyFrom       */
int pips_hip_vec_set(long long n, double a, double* y_dev, void* stream);                                 /* setToConstant  */
int pips_hip_vec_add_const(long long n, double a, double* y_dev, void* stream);                           /* add_constant   */
int pips_hip_vec_mul(long long n, const double* x_dev, double* y_dev, void* stream);                      /* componentMult  */
int pips_hip_vec_div(long long n, const double* x_dev, double* y_dev, void* stream);                      /* componentDiv   */
int pips_hip_vec_add_product(long long n, double a, const double* x_dev, const double* z_dev, double* y_dev, void* stream);
int pips_hip_vec_add_quotient(long long n, double a, const double* x_dev, const double* z_dev, const double* mask_dev,
                              double* y_dev, void* stream);
int pips_hip_vec_divide_some(long long n, const double* x_dev, const double* mask_dev, double* y_dev, void* stream);
int pips_hip_vec_select_nonzeros(long long n, const double* mask_dev, double* y_dev, void* stream);
int pips_hip_vec_safe_invert(long long n, double* y_dev, void* stream);
/* y_i = rmin - y_i if y_i < rmin, rmax - y_i if y_i > rmax, else 0; then y_i = max(y_i, -rmax)   (Vector::gondzioProjection,
 * DenseVector.cpp:405-420, used by Residuals::project_r3, Residuals.cpp:262-290) */
int pips_hip_vec_gondzio_projection(long long n, double rmin, double rmax, double* y_dev, void* stream);
int pips_hip_vec_dot(long long n, long long skip_root, const double* x_dev, const double* y_dev, double* result, void* stream);
int pips_hip_vec_one_norm(long long n, long long skip_root, const double* x_dev, double* result, void* stream);
int pips_hip_vec_inf_norm(long long n, const double* x_dev, double* result, void* stream);
int pips_hip_vec_min(long long n, const double* x_dev, double* result, void* stream);
/* two_norm = s*sqrt(sum (x/s)^2) with s = inf_norm (DistributedVector.C:424-437): returns sum (x*scale_inv)^2 */
int pips_hip_vec_sumsq_scaled(long long n, long long skip_root, double scale_inv, const double* x_dev, double* result, void* stream);
/* min over {dx_i < 0, mask_i != 0} of -x_i/dx_i (fraction_to_boundary / stepbound, Variables.C:191-225) */
int pips_hip_vec_stepbound(long long n, const double* x_dev, const double* dx_dev, const double* mask_dev, double* result, void* stream);
/* Variables::find_blocking (Variables.C:227-308) for one pair of complementary vectors: out5 = [min ratio -x_i/dx_i over
 * dx_i < 0 (inf if no entry blocks), then x, dx, y, dy at the blocking index (smallest index on ties)] */
/* the step bounds of nw <= 16 blended directions in one pass: out[k] = largest alpha with x + alpha (dx + w_k cx) >= 0 (inf if
 * unbounded), out[nw + k] the same for (y, dy, cy), w_k = min(1, wmin + (1 - wmin) k / (nw - 1)) - the corrector weight search
 * calculate_alpha_pd_weight_candidate (InteriorPointMethod.cpp:486-523); out2nw is a host array */
int pips_hip_vec_weighted_stepbounds(long long n, const double* x_dev, const double* dx_dev, const double* cx_dev, const double* y_dev,
                                     const double* dy_dev, const double* cy_dev, double wmin, int nw, double* out2nw, void* stream);
int pips_hip_vec_find_blocking(long long n, const double* x_dev, const double* dx_dev, const double* y_dev, const double* dy_dev,
                               double* out5, void* stream);
/* sum (x + a dx)(y + b dy)  (mustep_pd, Variables.C:109) */
int pips_hip_vec_dot_shifted(long long n, long long skip_root, const double* x_dev, double a, const double* dx_dev,
                             const double* y_dev, double b, const double* dy_dev, double* result, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * 4c. Host harness: Mehrotra predictor-corrector IPM with Gondzio correctors for the reference's full problem class
 *        min c^T x   s.t.  A x = b,  clow <= C x <= cupp,  xlow <= x <= xupp      (every bound optional per row / entry)
 *     with block-angular A and C, driving the fused KKT path (counterpart of PIPSIPMppSolver::solve / InteriorPointMethod /
 *     LinearSystem::computeDiagonals, solve, solveXYZS, solveCompressedBiCGStab, system_mult / Residuals::evaluate /
 *     DistributedMatrix::mult).  One or several ranks.
 *     Input = the reader's per-block layout (GMSPIPSBlockData_t, Drivers/gams/gmspips/gmspipsio.h:5-58): block 0 is the root
 *     (n0 variables; A = A0, C = C0, BL = F0, DL = G0; B, D absent), block k >= 1 has n variables, A (my x n0) and B (my x n)
 *     for its equality rows, C (mz x n0) and D (mz x n) for its inequality rows, BL (myl x n) / DL (mzl x n) its part of the
 *     linking rows.  Row pointers are block-local and start at 0.  Indicators are 0.0 / 1.0.
 *     Vector orders: x = [x0 | x_1 .. x_N]; equality rows / y = [root rows | linking rows | block 1 .. N];
 *     inequality rows / s, z, t, u, lambda, pi = [root rows | linking rows | block 1 .. N].
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
   int rows, cols;
   const int* rowptr;      /* NULL: matrix absent (all zero) */
   const int* colidx;
   const double* val;
} pips_csr_view;
typedef struct {
   int n, my, mz;                                        /* variables, own equality rows, own inequality rows */
   pips_csr_view A, B, C, D, BL, DL;
   const double *c, *xlow, *xupp, *ixlow, *ixupp;        /* n entries each */
   const double* b;                                      /* my */
   const double *clow, *cupp, *iclow, *icupp;            /* mz */
} pips_ipm_block;
/* n_blocks counts the root (block 0).  bL: right-hand side of the myl linking equalities; dlow / dupp / idlow / idupp: bounds
 * of the mzl linking inequalities.  Several ranks: blocks[1..] are this rank's blocks, the root block and
 * the linking data are identical on every rank; comm as for pips_hip_kkt_create.  Scalars are reduced over the ranks
 * (replicated parts counted on rank 0 only, DistributedVector.C:1293-1303), replicated rows of the SpMVs are summed
 * (DistributedMatrix.C:224-326); every rank returns the same result and holds the root part and its blocks' part of the
 * solution. */
int pips_ipm_create_general(void** handle, int n_blocks, const pips_ipm_block* blocks, int myl, int mzl, const double* bL,
                            const double* dlow, const double* dupp, const double* idlow, const double* idupp, double dual_reg,
                            int device, void* comm, int rank, int n_ranks);
/* the generator's class  min c^T x, A x = b, x >= 0   as a shorthand for the general entry: W/T/F rowptr
 * arrays hold N block-local row pointers back to back; x order [x_0 | x_1 .. x_N], b/y order [linking rows | block rows] */
int pips_ipm_create(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                    const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                    const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                    const double* F0_val, const double* c, const double* b, double dual_reg, int device);
int pips_ipm_create_rank(void** handle, int N, int n0, int myl, const int* n_i, const int* my_i, const int* W_rowptr,
                         const int* W_colidx, const double* W_val, const int* T_rowptr, const int* T_colidx, const double* T_val,
                         const int* F_rowptr, const int* F_colidx, const double* F_val, const int* F0_rowptr, const int* F0_colidx,
                         const double* F0_val, const double* c, const double* b, double dual_reg, int device, void* comm, int rank,
                         int n_ranks);
/* result7: [0] primal objective [1] iterations [2] mu [3] residual inf-norm [4] status (0 converged, 1 max iterations, 2 numerical
 * breakdown before any usable iterate, 3 numerical troubles: the best iterate so far is returned - its mu / residual are in
 * [2] / [3], 4 probably infeasible: phi = (||r|| + |gap|) / dnorm is >= 1e-8 and 1e4 times its best value after ten iterations,
 * PIPSIPMppSolver.cpp:128-170) [5] dual objective b^T y + clow^T lambda - cupp^T pi + xlow^T gamma - xupp^T phi [6] data norm.
 * Termination as PIPSIPMppSolver.cpp:143-149: mu <= mutol and ||r||inf <= artol * dnorm. */
int pips_ipm_solve(void* handle, int max_iter, double mutol, double artol, int verbose, double* result7);
/* Gondzio multiple centrality correctors per iteration (InteriorPointMethod.cpp:236-358): 0 = plain Mehrotra predictor-
 * corrector; default 2, i.e. 4 solves per iteration like the work unit of bench.py */
int pips_ipm_set_gondzio(void* handle, int max_correctors);
/* harness settings under the reference's option identifiers (Options.C:18-73, PIPSIPMppOptions.C:170-264,303-310):
 * GONDZIO_MAX_CORRECTORS, OUTER_SOLVE (1 iterative refinement, 2 BiCGStab), OUTER_BICG_MAX_ITER,
 * OUTER_BICG_MAX_NORMR_DIVERGENCES, OUTER_BICG_MAX_STAGNATIONS, REGULARIZATION (0/1: the inertia-correcting loop); anything else
 * returns an error */
int pips_ipm_set_option(void* handle, const char* name, double value);
/* shorthand for the x >= 0 class: bounded_mask (nx host doubles) is 1 for x_j >= 0 and 0 for a free x_j (ixlow = ixupp = 0, whose
 * computeDiagonals leaves dd_j = 0: LinearSystem.C:262-294).  Free entries carry no complementarity pair; the preconditioner gets
 * a proximal term on their diagonal (1e-6, option FREE_VARIABLE_PROXIMAL_TERM), the outer solve none.  Call before
 * pips_ipm_solve; with several ranks on every rank.  (The general entry takes free variables through ixlow = ixupp = 0.) */
int pips_ipm_set_free_variables(void* handle, const double* bounded_mask_host);
int pips_ipm_get_solution(void* handle, double* x_host, double* y_host);
/* dims4 = {nx, my, mz, complementarity pairs over all ranks}; iterate: any pointer may be NULL (sizes nx: x v w gamma phi,
 * my: y, mz: s z t u lambda pi) */
int pips_ipm_get_dims(void* handle, long long* dims4);
int pips_ipm_get_iterate(void* handle, double* x, double* s, double* y, double* z, double* t, double* u, double* v, double* w,
                         double* lambda, double* pi, double* gamma, double* phi);
/* history of the last pips_ipm_solve, one row of 7 doubles per iterate: mu, ||r||inf, primal objective, dual objective, and the
 * step taken from it: sigma, alpha_primal, alpha_dual (zeros in the final row).  rows7 may be NULL to query *n_rows. */
int pips_ipm_get_trace(void* handle, double* rows7, int max_rows, int* n_rows);
/* counters of the last pips_ipm_solve: [0] KKT factorisations, [1] of which repeats with added regularisation (the inertia
 * loop, LinearSystem.C:295-325), [2] solveCompressed calls (preconditioner applications), [3] Gondzio correctors accepted;
 * stats2: [0] BiCGStab iterations, [1] host synchronisations (scalar read-backs) */
int pips_ipm_get_stats(void* handle, long long* stats4);
int pips_ipm_get_stats2(void* handle, long long* stats2);
/* direct entries to the rows either side of the path, for parity tests.  pips_ipm_mult: out = J in (transposed = 0; in: nx, out:
 * my + mz = [A x | C x]) or out = J^T in (transposed = 1) - DistributedMatrix::mult / transpose_mult; with several ranks the
 * replicated rows are summed.  pips_ipm_outer_solve: factorises with the diagonals of the pair vectors G = [t|u|v|w],
 * L = [lambda|pi|gamma|phi] (host, 2 mz + 2 nx entries each) and runs the outer solve (OUTER_SOLVE option) on
 * [dd J^T; J diag(0, nOmegaInv)] sol = rhs ([x | y | z] order) to tolerance tol * ||rhs||; info6 = {status (1 converged, 2 skipped,
 * 3 max iterations, 4 breakdown, 5 diverged, 6 stagnation), iterations, ||r||_2, ||rhs||_2, preconditioner applications, host
 * synchronisations} */
int pips_ipm_mult(void* handle, int transposed, const double* in_host, double* out_host);
int pips_ipm_outer_solve(void* handle, const double* G_host, const double* L_host, const double* rhs_host, double tol, double* sol_host,
                         double* info6);
void pips_ipm_destroy(void* handle);

/* ---------------------------------------------------------------------------------------------------------------
 * 5. Host harness helpers (no GPU needed): synthetic arrowhead LP of SURVEY.md §8d, leaf KKT / border assembly.
 * ------------------------------------------------------------------------------------------------------------- */
int pips_gen_row_nnz(int n_i, double rho);
int pips_gen_block(uint64_t seed, int block, int n_i, int my_i, int n0, int myl, double rho, int* W_rowptr,
                   int* W_colidx, double* W_val, int* T_rowptr, int* T_colidx, double* T_val, int* F_rowptr,
                   int* F_colidx, double* F_val, double* c, double* xstar);
int pips_gen_root(uint64_t seed, int n0, int myl, int* F0_rowptr, int* F0_colidx, double* F0_val, double* c0,
                  double* xstar0);
int pips_gen_diagonal(uint64_t seed, int block, int n, double lo, double hi, double* d);
int pips_kkt_leaf_assemble(int nx, int my, int mz, const int* Q_rowptr, const int* Q_colidx, const double* Q_val,
                           const int* B_rowptr, const int* B_colidx, const double* B_val, const int* D_rowptr,
                           const int* D_colidx, const double* D_val, int* K_rowptr, int* K_colidx, double* K_val,
                           int* diag_pos);
int pips_border_assemble(int nx, int my, int mz, int n0, int n_empty, int myl, int mzl, const int* R_rowptr,
                         const int* R_colidx, const double* R_val, const int* A_rowptr, const int* A_colidx,
                         const double* A_val, const int* C_rowptr, const int* C_colidx, const double* C_val,
                         const int* F_rowptr, const int* F_colidx, const double* F_val, const int* G_rowptr,
                         const int* G_colidx, const double* G_val, int* Bt_rowptr, int* Bt_colidx, double* Bt_val);
/* block -> rank map (DistributedTree::assignProcesses, DistributedTree.C:35-90): contiguous, monotone, balanced */
int pips_map_children_to_ranks(int n_children, int n_ranks, int* map);
/* symbolic analysis only (CPU): fills what[] like pips_hip_ldl_info and optionally perm/colcount (may be NULL) */
int pips_symbolic_probe(int n, int n_primal, const int* krow, const int* jcol, int S, const int* Bt_rowptr,
                        const int* Bt_colidx, int force_n_head, int64_t* what, int n_what, int* perm, int* colcount);

/* the same under the order the sparse root takes for chain-like Schur complements: the hubs (x0, root equality rows) last, the rest
 * dissected by level structures, minimum degree in segments below min_size rows (order.cpp hub_dissected_order); PIPS_ERR_STATE when
 * the graph without the hubs has no separators.  CPU only. */
int pips_symbolic_probe_hubs(int n, int n_primal, const int* krow, const int* jcol, int n_hubs, const int* hubs, int min_size,
                             int64_t* what, int n_what, int* perm, int* colcount);

/* ---- 6. input files ---------------------------------------------------------------------------------------------------
 * One block of a block-structured LP from a "jacobian" GDX file, the format gmspips_reader opens per block
 * (Drivers/gams/gmspips/gmspips_reader.cpp:30-60; extraction rules of readBlock, gmspipsio.c:1357-2033; fields of
 * GMSPIPSBlockData_t, gmspipsio.h:5-58).  Uncompressed GDX version 7 files.  offset = stage number of block 0 (the
 * reference passes 1).  Host only. */
int pips_gdx_read_block(void** block, const char* path, int num_blocks, int act_block, int offset);
/* counts14 = {n0, ni, mA, mC, mBL, mDL, nnzA, nnzB, nnzC, nnzD, nnzBL, nnzDL, numBlocks, blockID} */
int pips_gdx_block_counts(void* block, long long* counts14);
/* which: 0 c, 1 xlow, 2 xupp, 3 ixlow, 4 ixupp, 5 b, 6 clow, 7 cupp, 8 iclow, 9 icupp, 10 bL, 11 dlow, 12 dupp, 13 idlow,
 * 14 idupp (indicators as 0.0 / 1.0); out may be NULL to query *length */
int pips_gdx_block_vector(void* block, int which, double* out, int capacity, int* length);
/* which: 0 A, 1 B, 2 C, 3 D, 4 BL, 5 DL; *present = 0 where the reference leaves the matrix pointers NULL; the arrays (any may
 * be NULL) must hold rows + 1 / nnz / nnz entries (sizes from pips_gdx_block_counts) */
int pips_gdx_block_matrix(void* block, int which, int* present, int* rows, int* cols, int* rowptr, int* colidx, double* val);
void pips_gdx_block_destroy(void* block);

#ifdef __cplusplus
}
#endif
#endif /* PIPS_HIP_H */
