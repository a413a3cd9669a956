import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import pips_ipmpp_amd as pa
N, n_i, my_i, n0, myl = 64, 10000, 5000, 1000, 1000
bt = pa.LeafBatch(N, 0, device=0); vals=[]
for b in range(N):
    W, T, F, c, xs = pa.gen_block(20261002, b + 1, n_i, my_i, n0, myl, 1e-3)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    K.val[dpos] = np.concatenate([pa.gen_diagonal(20261002, b + 1, n_i), -1e-8 * np.ones(my_i)])
    bt.set_block(b, K, n_i); vals.append(K.val)
bt.analyze(16)
for b in range(N): bt.set_values(b, vals[b])
bt.factor()
for tol in (1e-30, 1e-16, 1e-15, 1e-14):
    bt.set_refinement_backward_error(2, tol)
    x = torch.randn(N * 15000, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.time(); bt.solve(x); bt.sync(); dt = time.time() - t0
    print(f"backward-error tol {tol:g}: refinement steps taken {bt.last_refinement_steps()} last measure {bt.last_refinement_measure():.2e}  {dt*1e3:.2f} ms")
