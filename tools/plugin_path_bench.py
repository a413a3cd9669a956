"""What a stock PIPS-IPM++ gets from the drop-in DoubleLinearSolver adapters alone (INTEGRATION.md level 1), timed beside the fused
path: per block pips_hip_ldl_factor with the values coming from host memory, the reference's own K4-K6 loop on the host
(DistributedLinearSystem.C:766-1175: chunks of 20 x T border columns dense-ified on the host, multi-RHS pips_hip_ldl_solve through
host pointers, sparse product back into the host Schur complement), the dense root through pips_hip_dense_ldl_factor / _solve with
host pointers, and solveCompressed as host loops around single solves.  Config: BASELINE.json configs[1] (64 x 10 000, S = 2000);
--blocks B of the 64 are run and scaled by 64 / B.   usage: plugin_path_bench.py [--blocks 4] [--chunk 160]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import pips_ipmpp_amd as pa

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--chunk", type=int, default=160, help="border columns per multi-RHS solve (the reference: 20 x OMP threads)")
ap.add_argument("--level", choices=["1", "1.5", "1.5b"], default="1", help="1: adapters only, the host's K4-K6 loop ships dense border columns; "
                "1.5: pips_hip_ldl_set_border + pips_hip_ldl_factor_schur - CSR border up, S x S term down; "
                "1.5b: the same through the array-of-handles entries (pips_hip_ldl_factor_schur_batch / pips_hip_ldl_solve_batch): the leaves of the rank as one batch")
ap.add_argument("--dense-rhs", action="store_true", help="level 1: ship the dense-ified border columns whole (pips_hip_ldl_solve) instead of the rows colSparsity marks (pips_hip_ldl_solve_sparse)")
a = ap.parse_args()
t_parts = [0.0, 0.0, 0.0]    # level 1: host dense-ify, solve call (PCIe + device), host sparse product
seed, N_total, n_i, S, rho = 20261002, 64, 10000, 2000, 1e-3
my_i, n0, myl = n_i // 2, S // 2, S // 2
solvers, Bts = [], []
t_an = t_fac = t_schur = 0.0
SC = np.zeros((S, S))
for b in range(a.blocks):
    W, T, F, c, xs = pa.gen_block(seed, b + 1, n_i, my_i, n0, myl, rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    K.val[dpos] = np.concatenate([pa.gen_diagonal(seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F).to_scipy()
    s = pa.HipLdlSolver(K, n_primal=n_i, refine_steps=2, refine_tol=1e-15, backward_error=True)   # (the adapters' setting: examples/adapter/HipLdlSolver.h)
    if a.level in ("1.5", "1.5b"):
        s.set_border(pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F))
    if a.level == "1.5b":
        solvers.append(s); Bts.append(Bt)
        continue
    t0 = time.perf_counter(); s.analyze(); t_an += time.perf_counter() - t0
    if a.level == "1.5":
        s.matrixChanged_with_schur_term(np.zeros((S, S)))   # warm-up (first-touch allocations)
        t0 = time.perf_counter(); s.matrixChanged_with_schur_term(SC); t_schur += time.perf_counter() - t0   # factor + Schur term in one call
    else:
        s.matrixChanged()                                   # warm-up (first-touch allocations)
        t0 = time.perf_counter(); s.matrixChanged(); t_fac += time.perf_counter() - t0
        t0 = time.perf_counter()
        cols = np.nonzero(np.diff(Bt.indptr) > 0)[0]
        # the reference keeps ONE dense buffer and one sparsity pattern per leaf system and refills them chunk by chunk (colsBlockDense / colSparsity:
        # DistributedLinearSystem.C:840-853, std::fill + fromGetColsBlock :895-900, solve in place :903) - the same pages go up and down every time
        colsBlockDense = np.zeros((a.chunk, Bt.shape[1])); colSparsity = np.zeros(Bt.shape[1], np.int32)
        # (warm-up of the solve path like the factorisation's above: the solver objects live as long as the linear system, the device buffers of
        # solve(nrhs) are allocated at its first call of a run, not per factorisation)
        colsBlockDense[:, :] = np.random.default_rng(b).standard_normal(colsBlockDense.shape); s.solve(colsBlockDense)
        t0 = time.perf_counter()
        for k in range(0, len(cols), a.chunk):               # K4: dense-ify, K5: multi-RHS solve, K6: sparse product
            ids = cols[k:k + a.chunk]
            t1 = time.perf_counter()
            sub = Bt[ids]
            dense = colsBlockDense[:len(ids)]
            dense[:] = 0.0
            dense[np.repeat(np.arange(len(ids)), np.diff(sub.indptr)), sub.indices] = sub.data
            colSparsity[:] = 0; colSparsity[sub.indices] = 1
            t2 = time.perf_counter()
            if a.dense_rhs:
                s.solve(dense)
            else:       # the adapter's solve(nrhss, rhss, colSparsity) with the pattern the reference builds (DistributedLinearSystem.C:903)
                s.solve_sparse(dense, colSparsity)
            t3 = time.perf_counter()
            SC[ids, :] -= (Bt @ dense.T).T
            t_parts[0] += t2 - t1; t_parts[1] += t3 - t2; t_parts[2] += time.perf_counter() - t3
        t_schur += time.perf_counter() - t0
    solvers.append(s); Bts.append(Bt)
    print(f"block {b}: factor {t_fac / (b + 1) * 1e3:.1f} ms, Schur term {t_schur / (b + 1):.2f} s (running means)", file=sys.stderr, flush=True)
if a.level == "1.5b":
    t0 = time.perf_counter(); pa.HipLdlSolver.factor_schur_batch(solvers, np.zeros((S, S))); t_an = time.perf_counter() - t0   # binds + analyses + warm-up
    t0 = time.perf_counter(); pa.HipLdlSolver.factor_schur_batch(solvers, SC); t_schur = time.perf_counter() - t0
F0, c0, x0s = pa.gen_root(seed, n0, myl)
from oracle import oracle as orc   # only finalize_kkt_dense: host-side assembly of the root rows, as the reference's host does
SCf = orc.finalize_kkt_dense(np.tril(SC) * (N_total / a.blocks), n0, 0, myl, 0, pa.gen_diagonal(seed, 0, n0), F0=F0.to_scipy())
root = pa.HipDenseLdlSolver(S, n_primal=n0)
A = np.tril(SCf); A = A + np.tril(A, -1).T
root.matrixChanged(np.ascontiguousarray(A))
t0 = time.perf_counter(); root.matrixChanged(np.ascontiguousarray(A)); t_root = time.perf_counter() - t0
rng = np.random.default_rng(0)
t_sc = []
for r in range(4):
    b0 = rng.standard_normal(S); bs = [rng.standard_normal(n_i + my_i) for _ in range(a.blocks)]
    t0 = time.perf_counter()
    if a.level == "1.5b":      # the host's loops over the children handed over: one batch solve per half
        pa.HipLdlSolver.solve_batch(solvers, bs)
        for bi, Bt in zip(bs, Bts): b0 -= Bt @ bi
        root.solve(b0)
        ts = [Bt.T @ b0 for Bt in Bts]
        pa.HipLdlSolver.solve_batch(solvers, ts)
        for bi, t in zip(bs, ts): bi -= t
    else:
        for bi, sol, Bt in zip(bs, solvers, Bts):
            sol.solve(bi); b0 -= Bt @ bi
        root.solve(b0)
        for bi, sol, Bt in zip(bs, solvers, Bts):
            t = Bt.T @ b0; sol.solve(t); bi -= t
    t_sc.append(time.perf_counter() - t0)
scale = N_total / a.blocks
unit = (t_fac + t_schur) * scale + t_root + 4 * np.median(t_sc) * scale
print(json.dumps({"path": ("drop-in DoubleLinearSolver adapters only (host pointers, reference K4-K6 host loop; " + ("dense right-hand sides up" if a.dense_rhs else "colSparsity honoured: only the marked rows go up") + ")") if a.level == "1" else
                          "level 1.5: adapters + pips_hip_ldl_factor_schur (CSR border up, S x S Schur term down; solves through host pointers)" if a.level == "1.5" else
                          "level 1.5b: array-of-handles entries - all leaves of the rank as one batch (pips_hip_ldl_factor_schur_batch, pips_hip_ldl_solve_batch; host pointers)", "blocks_run": a.blocks, "chunk_columns": a.chunk,
                  "seconds_per_block": {"analyze_once": t_an / a.blocks, "factor": t_fac / a.blocks, "schur_term": t_schur / a.blocks},
                  "schur_term_parts_per_unit": {"host_densify": t_parts[0] * scale, "solve_calls": t_parts[1] * scale, "host_sparse_product": t_parts[2] * scale},
                  "seconds_per_unit": {"leaf_factor": t_fac * scale, "leaf_schur": t_schur * scale, "root_factor": t_root, "solve_compressed_x4": 4 * float(np.median(t_sc)) * scale,
                                       "total": unit}, "units_per_s": 1.0 / unit}))
