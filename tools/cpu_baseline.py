"""CPU side of BASELINE.md section 3: the reference's algorithm for the path (PARDISO leaf factorisation phase 12 on every block,
blocked Schur complement K4-K6 over ALL border columns in chunks of 20, dense root dsytrf, R = 4 solveCompressed) timed on
this machine's host cores with MKL PARDISO under the reference's iparm settings (oracle/pardiso_mkl.py, the restatement of
PardisoSolver.C / PardisoProjectSolver.C) and LAPACK - no extrapolation over border columns; whole blocks only.

usage: cpu_baseline.py [--config 0|1] [--blocks B] [--threads T]
  config 0 = BASELINE.json configs[0]: 4 blocks x 1000 vars, S = 200 (complete work unit)
  config 1 = BASELINE.json configs[1]: 64 blocks x 10 000 vars, S = 2000; --blocks B of the 64 are run completely (default 4)
             and the leaf part is scaled by 64 / B (the blocks are i.i.d. and independent)
Prints one JSON object; tools/cpu_baseline.py is how the CPU rows of BASELINE.md were produced."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=0)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
ap.add_argument("--seed", type=int, default=20261002)
a = ap.parse_args()
os.environ.setdefault("OMP_NUM_THREADS", str(a.threads))

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from oracle import pardiso_mkl as pm

if a.config == 0:
    N_total, n_i, S, rho = 4, 1000, 200, 0.01
    B = 4
else:
    N_total, n_i, S, rho = 64, 10000, 2000, 1e-3
    B = min(a.blocks, N_total)
my_i, n0, myl = n_i // 2, S // 2, S // 2
R_SOLVES = 4
assert pm.available(), "MKL PARDISO (libmkl_rt.so) not loadable"

t_factor, t_schur, t_solve1 = [], [], []
nnzL, solvers, Bts = [], [], []
SC = np.zeros((S, S))
for b in range(B):
    W, T, F, c, xs = pa.gen_block(a.seed, b + 1, n_i, my_i, n0, myl, rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    K.val[dpos] = np.concatenate([pa.gen_diagonal(a.seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
    Ks = sp.csr_matrix((K.val, K.colidx, K.rowptr), shape=(K.nrows, K.ncols))
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F).to_scipy()
    s = pm.MklPardisoSolver(Ks, num_threads=a.threads)
    if b == 0:
        s.matrixChanged()                 # untimed: thread pool start-up and first-touch of the library
    t0 = time.perf_counter()
    s.matrixChanged()                     # phase 12: analysis + numerical factorisation, as the reference does every time
    t_factor.append(time.perf_counter() - t0)
    nnzL.append(int(s.iparm[17]))
    t0 = time.perf_counter()
    orc.add_term_to_schur_compl_blocked(SC, s, Bt, blocksize=20)   # K4-K6, every non-empty border column
    t_schur.append(time.perf_counter() - t0)
    solvers.append(s)
    Bts.append(Bt)
    print(f"block {b}: factor {t_factor[-1]:.2f} s, Schur term ({int((np.diff(Bt.indptr) > 0).sum())} columns) {t_schur[-1]:.2f} s, "
          f"nnz(L) {nnzL[-1]:,}", file=sys.stderr, flush=True)

F0, c0, x0s = pa.gen_root(a.seed, n0, myl)
SCf = orc.finalize_kkt_dense(np.tril(SC) * (N_total / B), n0, 0, myl, 0, pa.gen_diagonal(a.seed, 0, n0), F0=F0.to_scipy())
root = orc.DenseRootSolver(S)
t0 = time.perf_counter()
root.matrixChanged(np.tril(SCf))
t_root = time.perf_counter() - t0

rng = np.random.default_rng(0)
t_sc = []
for r in range(R_SOLVES):
    b0 = rng.standard_normal(S)
    bs = [rng.standard_normal(n_i + my_i) for _ in range(B)]
    t0 = time.perf_counter()
    orc.solve_compressed(b0, bs, solvers, Bts, root, n0, 0, 0, myl, 0)
    t_sc.append(time.perf_counter() - t0)

scale = N_total / B
leaf_factor = float(np.sum(t_factor)) * scale
leaf_schur = float(np.sum(t_schur)) * scale
solve_c = float(np.median(t_sc)) * scale          # leaf solves dominate solveCompressed; the root solve is in there once per call
unit = leaf_factor + leaf_schur + t_root + R_SOLVES * solve_c
cpu = open("/proc/cpuinfo").read()
model = [l.split(":")[1].strip() for l in cpu.splitlines() if l.startswith("model name")]
print(json.dumps({
    "config": a.config, "workload": f"{N_total} blocks x {n_i} vars, S = {S}", "blocks_run_completely": B, "threads": a.threads,
    "cpu": model[0] if model else "?", "logical_cpus": os.cpu_count(),
    "seconds_per_block": {"factor_phase12": float(np.mean(t_factor)), "schur_term_all_columns": float(np.mean(t_schur)),
                          "solve_compressed_share": float(np.median(t_sc)) / B},
    "seconds_per_unit": {"leaf_factor": leaf_factor, "leaf_schur": leaf_schur, "root_dsytrf": t_root,
                         "solve_compressed_x4": R_SOLVES * solve_c, "total": unit},
    "units_per_s": 1.0 / unit, "nnzL_per_block_mkl_pardiso_metis": int(np.mean(nnzL)),
    "solver": "MKL PARDISO mtype -2, iparm of PardisoProjectSolver.C:68-77 (METIS, 2 refinement steps, scaling, matching), LAPACK dsytrf root",
}))
