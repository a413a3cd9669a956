import sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
import pips_ipmpp_amd as pa
from tests.test_ipm_gpu import build_lp
case = int(sys.argv[1])
rng = np.random.default_rng(31000 + case)
N = int(rng.integers(1, 5)); n_i = int(rng.choice([24, 60, 150, 400])); my_i = int(n_i * rng.choice([0.3, 0.5]))
n0, myl = int(rng.integers(2, 12)), int(rng.integers(1, 10))
rho = max(4.0 / n_i, float(rng.choice([0.02, 0.1])))
blocks, F0, c, b, A = build_lp(4000 + case, N, n_i, my_i, n0, myl, rho)
for rep in range(40):
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    res = ipm.solve(max_iter=150, mutol=1e-9, artol=1e-8, verbose=False)
    tr = ipm.trace()
    if res["status"] != 0:
        print("rep", rep, "FAILED", res)
        np.set_printoptions(linewidth=200, precision=4)
        print(tr[-6:])
        break
else:
    print("no failure in 40 repetitions; last:", res["iterations"], res["status"])
