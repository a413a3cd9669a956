"""Development aid: residual of the leaf solve with 0/1/2 refinement steps for several diagonal ranges."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
from tests.util import Problem
for (lo, hi, reg) in [(-4, 4, 1e-8), (-8, 8, 1e-8), (-8, 8, 0.0), (-10, 10, 0.0)]:
    prob = Problem(5, 2, 10000, 5000, 100, 100, 1e-3, dual_reg=reg, diag_lo=lo, diag_hi=hi)
    for steps in (0, 1, 2):
        bt = pa.LeafBatch(prob.N, 0)
        for b in range(prob.N): bt.set_block(b, prob.blocks[b]["K"], prob.n_i)
        bt.set_options(refine_steps=steps)
        bt.analyze(8)
        for b in range(prob.N): bt.set_values(b, prob.blocks[b]["K"].val)
        bt.factor()
        rhs = np.random.default_rng(0).standard_normal(prob.N * prob.n_leaf)
        x = rhs.copy(); bt.solve(x)
        res = []
        for b in range(prob.N):
            r = rhs.reshape(prob.N, -1)[b]; xb = x.reshape(prob.N, -1)[b]
            res.append(np.linalg.norm(prob.K_full(b) @ xb - r) / np.linalg.norm(r))
        print(f"diag 10^[{lo},{hi}] reg {reg:g} refine {steps}: rel residual {max(res):.2e} inertia {bt.inertia(0)}", flush=True)
        bt.close()
