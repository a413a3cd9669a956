"""Development aid: DoubleLinearSolver::solve(nrhs, ...) of the drop-in handle on one config-2 block (host buffers, PCIe
included) - the call the reference's own blocked Schur loop makes (DistributedLinearSystem.C:737,903,956,1008)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
n_i = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
my_i = n_i // 2
W, T, F, c, xs = pa.gen_block(1, 1, n_i, my_i, 1000, 1000, 10.0 / n_i)
K, dpos = pa.kkt_leaf_assemble(n_i, W)
K.val[dpos] = np.concatenate([pa.gen_diagonal(1, 1, n_i), -1e-8 * np.ones(my_i)])
s = pa.HipLdlSolver(K, n_primal=n_i, refine_steps=0)
s.matrixChanged()
print(s.info(), flush=True)
import scipy.sparse as sp
low = sp.csr_matrix((K.val, K.colidx, K.rowptr), shape=(K.nrows, K.ncols)); Kf = low + sp.tril(low, -1).T
for nrhs in (1, 32, 256):
    X = np.random.default_rng(0).standard_normal((nrhs, K.nrows)); R = X.copy()
    s.solve(X.copy())
    t0 = time.time(); s.solve(X); dt = time.time() - t0
    res = np.linalg.norm(Kf @ X[-1] - R[-1]) / np.linalg.norm(R[-1])
    print(f"nrhs {nrhs}: {dt*1e3:.1f} ms = {dt/nrhs*1e3:.3f} ms per right-hand side, residual {res:.1e}", flush=True)
