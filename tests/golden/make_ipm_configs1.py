"""Generates tests/golden/ipm_configs1.json: final objective, primal / dual residual, complementarity and iteration count of the interior-
point solve of the BASELINE.json configs[1] LP (64 blocks x 10 000 variables, 5000 equality rows per block, Schur dimension 2000: 641 000
variables, 321 000 constraints - the LP bench.py's `ipm_end_to_end` solves: same generator, same seed, same tolerances) computed on the
CPU: oracle/ipm_oracle.py (the restatement of the reference's Mehrotra + Gondzio loop) with every KKT system solved by MKL PARDISO
(oracle/pardiso_mkl.py: mtype -2, METIS, scaling, Bunch-Kaufman, two refinement steps; the reference's iparm of PardisoProjectSolver.C:68-77
except the matching, which triples the fill of the whole arrowhead matrix and is switched off here - 20 M instead of 45 M factor entries per
block, one analysis for all iterates) on the assembled global matrix [D A^T; A 0] - the "CPU PARDISO path" of BASELINE.json's north_star
without any of the product's Schur-complement code.  north_star: "final objective + primal/dual residuals matching the CPU PARDISO path to 1e-8 relative";
tests/test_ipm_gpu.py::test_configs1_matches_the_cpu_pardiso_path holds the device harness to that.

    python tests/golden/make_ipm_configs1.py [--threads 6] [--small]      (full size: about 2.5 hours on 6 cores, ~15 GB)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ipm_oracle as io  # noqa: E402
from oracle import pardiso_mkl as pm  # noqa: E402
from tests.test_ipm_gpu import build_lp  # noqa: E402

SEED, SHAPE, MUTOL, ARTOL = 20261002, (64, 10000, 5000, 1000, 1000, 1e-3), 1e-8, 1e-8     # bench.py: --seed default, configs[1], ipm_end_to_end


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--small", action="store_true", help="4 x 1000 (configs[0]) - a dry run of this script in a minute")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ipm_configs1.json"))
    a = ap.parse_args()
    shape = (4, 1000, 500, 100, 100, 0.01) if a.small else SHAPE
    N, n_i, my_i, n0, myl, rho = shape
    t0 = time.time()
    blocks, F0, c, b, A = build_lp(SEED, N, n_i, my_i, n0, myl, rho)
    print(f"LP: {A.shape[1]} variables, {A.shape[0]} constraints, nnz(A) {A.nnz}  ({time.time() - t0:.0f} s)", flush=True)
    state = {"solver": None, "n_fact": 0, "t_fact": 0.0}

    def kkt_solver(K):
        t = time.time()
        low = sp.tril(sp.csr_matrix(K), format="csr")
        low.sort_indices()
        if state["solver"] is None:
            state["solver"] = pm.MklPardisoSolver(low, num_threads=a.threads, matching=False, reuse_analysis=True)
        else:
            state["solver"].K = low          # same pattern every iterate
        state["solver"].matrixChanged()
        state["n_fact"] += 1
        state["t_fact"] += time.time() - t
        print(f"  factorisation {state['n_fact']}: {time.time() - t:.1f} s, inertia {state['solver'].get_inertia()}", flush=True)
        s = state["solver"]
        return lambda rhs: s.solve(np.array(rhs, dtype=np.float64))

    trace = []
    o = io.solve_lp(A, b, c, 150, MUTOL, ARTOL, trace, kkt_solver=kkt_solver)
    x, y = o["x"], o["y"]
    res = {"what": "oracle/ipm_oracle.py over MKL PARDISO (reference iparm without the matching) on the global KKT matrix; see tests/golden/make_ipm_configs1.py",
           "seed": SEED, "shape": list(shape), "mutol": MUTOL, "artol": ARTOL, "status": int(o["status"]), "iterations": int(o["iterations"]),
           "objective": float(o["objective"]), "dual_objective": float(o["dual_objective"]), "mu": float(o["mu"]), "rnorm": float(o["rnorm"]),
           "dnorm": float(o["dnorm"]), "primal_residual_inf": float(np.abs(A @ x - b).max()),
           "dual_residual_inf": float(np.abs(c - A.T @ y - o["gamma"]).max()), "x_min": float(x.min()),
           "variables": int(A.shape[1]), "constraints": int(A.shape[0]), "factorizations": state["n_fact"],
           "seconds_total": round(time.time() - t0, 1), "seconds_in_factorisations": round(state["t_fact"], 1), "threads": a.threads,
           "trace_mu_rnorm_pobj_dobj": [[float(v) for v in t[1:5]] for t in trace]}
    json.dump(res, open(a.out, "w"), indent=1)
    print({k: v for k, v in res.items() if k != "trace_mu_rnorm_pobj_dobj"})


if __name__ == "__main__":
    main()
