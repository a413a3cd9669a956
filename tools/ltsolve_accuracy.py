"""Arrowhead residual of solveCompressed with the border-backward Ltsolve and with border product + refined solve (PIPS_HIP_BORDER_BACKWARD=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as sp
import pips_ipmpp_amd as pa
from tests.util import Problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = Problem(42, N, 10000, 5000, 1000, 1000, 1e-3, diag_lo=float(sys.argv[2]) if len(sys.argv) > 2 else -4.0, diag_hi=float(sys.argv[3]) if len(sys.argv) > 3 else 4.0)
bt = pa.LeafBatch(N, prob.S)
for b in range(N):
    bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
bt.analyze(8)
for b in range(N):
    bt.set_values(b, prob.blocks[b]["K"].val)
bt.set_refinement_backward_error(2, 1e-15)
kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
kkt.factorize(None, torch.tensor(prob.x_diag0, device="cuda"))
rng = np.random.default_rng(1)
b0, bl = rng.standard_normal(prob.S), rng.standard_normal(N * prob.n_leaf)
x0d, xld = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
kkt.solve_compressed(x0d, xld); bt.sync()
x0, xl = x0d.cpu().numpy(), xld.cpu().numpy()
F0s = prob.F0.to_scipy()
K0 = sp.bmat([[sp.diags(prob.x_diag0), F0s.T], [F0s, None]], format="csr")
r0 = K0 @ x0 - b0
nl = prob.n_leaf
rl = 0.0
for b in range(N):
    Bt = prob.Bt_scipy(b); xb = xl[b * nl:(b + 1) * nl]
    rl = max(rl, np.abs(prob.K_full(b) @ xb + Bt.T @ x0 - bl[b * nl:(b + 1) * nl]).max())
    r0 += Bt @ xb
print(f"PIPS_HIP_BORDER_BACKWARD={os.environ.get('PIPS_HIP_BORDER_BACKWARD', 'auto')}: leaf rows residual {rl:.2e}, root rows residual {np.abs(r0).max():.2e} (|b| ~ {np.abs(bl).max():.1f}), |x| max {max(np.abs(x0).max(), np.abs(xl).max()):.2e}")
