"""Development aid: the device IPM harness on a seeded family of small arrowhead LPs against HiGHS."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import linprog
import pips_ipmpp_amd as pa
from tests.test_ipm_gpu import build_lp
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(31000 + case)
    N = int(rng.integers(1, 5)); n_i = int(rng.choice([24, 60, 150, 400])); my_i = int(n_i * rng.choice([0.3, 0.5]))
    n0, myl = int(rng.integers(2, 12)), int(rng.integers(1, 10))
    rho = max(4.0 / n_i, float(rng.choice([0.02, 0.1])))
    blocks, F0, c, b, A = build_lp(4000 + case, N, n_i, my_i, n0, myl, rho)
    t0 = time.time()
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    res = ipm.solve(max_iter=150, mutol=1e-9, artol=1e-8)
    dt = time.time() - t0
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    err = abs(res["objective"] - ref.fun) / max(1.0, abs(ref.fun)) if ref.status == 0 else float("nan")
    # status 3 = numerical troubles below the reference's default accuracy, best iterate returned: judged by its objective
    ok = res["status"] in (0, 3) and ref.status == 0 and err < 1e-7
    bad += not ok
    print(f"case {case}: N={N} n_i={n_i} my_i={my_i} n0={n0} myl={myl}  status {res['status']} it {res['iterations']}  rel.obj.err {err:.1e}  {dt:.1f}s {'' if ok else '<-- CHECK'}", flush=True)
print("failures:", bad)
