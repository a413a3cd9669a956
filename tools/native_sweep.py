"""Development aid: the native general-form device IPM over many seeded block LPs with free variables, against HiGHS.
usage: native_sweep.py [n_seeds] [follows_mu 0|1] [min]"""
import sys, os, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pips_ipmpp_amd as pa
from oracle import ipm_oracle as io
from general_lp_gen import random_block_lp
from scipy.optimize import linprog
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
follows = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fmin = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-10
regmax = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2
eager = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
emax = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-2
bad, its = [], []
for seed in range(n):
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(2, 5))
    blocks = random_block_lp(1000 + seed, nb, int(rng.integers(4, 9)), int(rng.integers(8, 20)), int(rng.integers(2, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 4)), free_fraction=0.15)
    d = io.assemble(blocks)
    C = d["C"]; up, lo = d["icupp"] > 0, d["iclow"] > 0
    ref = linprog(d["c"], A_ub=sp.vstack([C[up], -C[lo]]), b_ub=np.concatenate([d["cupp"][up], -d["clow"][lo]]), A_eq=d["A"], b_eq=d["b"],
                  bounds=[(l if il else None, u if iu else None) for l, il, u, iu in zip(d["xlow"], d["ixlow"], d["xupp"], d["ixupp"])], method="highs")
    ipm = pa.GeneralIpmSolver(blocks)
    ipm.set_option("FREE_VARIABLE_PROXIMAL_FOLLOWS_MU", follows); ipm.set_option("FREE_VARIABLE_PROXIMAL_MIN", fmin); ipm.set_option("REGULARIZATION_MAX", regmax); ipm.set_option("INERTIA_LOOP", eager); ipm.set_option("REGULARIZATION_EAGER_MAX", emax)
    res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    err = abs(res["objective"] - ref.fun) / max(1.0, abs(ref.fun))
    its.append(res["iterations"])
    if res["status"] != 0 or err > 1e-6: bad.append((seed, res["status"], res["iterations"], err))
    ipm.close()
print(f"follows_mu={follows} min={fmin} regmax={regmax} eager={eager} eager_max={emax}: {n - len(bad)}/{n} ok, mean iterations {np.mean(its):.1f}; bad: {bad}")
