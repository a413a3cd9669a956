import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pips_ipmpp_amd as pa
from tests.test_ipm_gpu import build_lp
from scipy.optimize import linprog
for shape in [(3, 60, 30, 6, 5, 0.1), (4, 1000, 500, 100, 100, 0.01)]:
    N, n_i, my_i, n0, myl, rho = shape
    blocks, F0, c, b, A = build_lp(2026, N, n_i, my_i, n0, myl, rho)
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs").fun
    for mutol in (1e-6, 1e-8, 1e-9):
        ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b, dual_reg=0.0)
        res = ipm.solve(max_iter=60, mutol=mutol, artol=1e-8, verbose=False)
        print(shape[:2], mutol, res['status'], res['iterations'], "rel obj err %.2e" % (abs(res['objective']-ref)/abs(ref)), "mu %.2e r %.2e" % (res['mu'], res['rnorm']))
