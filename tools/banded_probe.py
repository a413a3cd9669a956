"""Development aid: time-coupled (banded) leaf blocks instead of random sparsity — exercises the sparse head at scale."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import pips_ipmpp_amd as pa
N, n_i, my_i, bw = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 10000, 5000, 20)
rng = np.random.default_rng(0)
bt = pa.LeafBatch(N, 0, device=0)
Ks = []
for b in range(N):
    rows, cols, vals = [], [], []
    for r in range(my_i):
        center = int(r * n_i / my_i)
        cs = np.unique(np.clip(center + rng.integers(-bw, bw + 1, 9), 0, n_i - 1))
        cs = np.union1d(cs, [center])
        rows += [r] * len(cs); cols += list(cs); vals += list(rng.uniform(-1, 1, len(cs)))
    W = sp.csr_matrix((vals, (rows, cols)), shape=(my_i, n_i)); W.sum_duplicates(); W.sort_indices()
    Wp = pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)
    K, dpos = pa.kkt_leaf_assemble(n_i, Wp)
    K.val[dpos] = np.concatenate([10 ** rng.uniform(-4, 4, n_i), -1e-8 * np.ones(my_i)])
    bt.set_block(b, K, n_i); Ks.append(K)
t0 = time.time(); bt.analyze(16); print(f"analyze {time.time()-t0:.2f}s", bt.info(), flush=True)
for b in range(N): bt.set_values(b, Ks[b].val)
bt.set_timing(True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time(); bt.factor(); bt.sync(); dt = time.time() - t0
    print(f"factor {dt*1e3:.2f} ms", {k: (round(v[0], 2), v[1]) for k, v in bt.get_timing().items()}, flush=True)
info = bt.info()
print(f"nnzL {info['nnzL']:,}  -> factor streams {(info['nnzL']*8*2)/dt/1e9:.1f} GB/s of L (write+read once)")
x = np.random.default_rng(1).standard_normal(N * (n_i + my_i)); rhs = x.copy()
xd = torch.tensor(x, device="cuda"); bt.solve(xd); bt.sync()
xd = torch.tensor(x, device="cuda"); torch.cuda.synchronize(); t0 = time.time(); bt.solve(xd); bt.sync()
print(f"solve {1e3*(time.time()-t0):.2f} ms (device-resident, incl. one refinement step)")
x = xd.cpu().numpy()
K = Ks[0]; low = sp.csr_matrix((K.val, K.colidx, K.rowptr), shape=(K.nrows, K.ncols)); Kf = low + sp.tril(low, -1).T
r0 = rhs[:K.nrows]; print("residual block 0:", np.linalg.norm(Kf @ x[:K.nrows] - r0) / np.linalg.norm(r0), "inertia", bt.inertia(0))
