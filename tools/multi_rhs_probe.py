"""Device time of DoubleLinearSolver::solve(nrhs, ...) on ONE configs[1] leaf block (15 000 rows: head 10 520, dense tail 4 480), right-hand
sides resident on the device: pips_hip_ldl_solve_dev for 40 (the reference's default chunk, 20 x OMP_NUM_THREADS 2) and 160 columns,
interleaved matrix-pipe sweeps against the per-right-hand-side sweeps (PIPS_HIP_MULTI=0).  usage: multi_rhs_probe.py [nrhs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
seed, n_i, S, rho = 20261002, 10000, 2000, 1e-3
my_i, n0, myl = n_i // 2, S // 2, S // 2
W, T, F, c, xs = pa.gen_block(seed, 1, n_i, my_i, n0, myl, rho)
K, dpos = pa.kkt_leaf_assemble(n_i, W)
K.val[dpos] = np.concatenate([pa.gen_diagonal(seed, 1, n_i), -1e-8 * np.ones(my_i)])
n = n_i + my_i
low = K.to_scipy() if hasattr(K, "to_scipy") else None
import scipy.sparse as sp
low = sp.csr_matrix((K.val, K.colidx, K.rowptr), shape=(n, n))
Kf = (low + sp.tril(low, -1).T).tocsr()
for mode in ("", "0"):
    if mode:
        os.environ["PIPS_HIP_MULTI"] = mode
    else:
        os.environ.pop("PIPS_HIP_MULTI", None)
    s = pa.HipLdlSolver(K, n_primal=n_i, refine_steps=2, refine_tol=1e-15, backward_error=True)   # (the adapters' setting: examples/adapter/HipLdlSolver.h)
    s.analyze(); s.matrixChanged()
    for nrhs in [int(v) for v in sys.argv[1:]] or [40, 160]:
        g = torch.Generator(device="cuda").manual_seed(1)
        B = torch.randn((nrhs, n), dtype=torch.float64, device="cuda", generator=g)
        X = B.clone()
        s.solve_dev(X, nrhs, n); torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            X.copy_(B); torch.cuda.synchronize(); t0 = time.perf_counter()
            s.solve_dev(X, nrhs, n); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        xh, bh = X.cpu().numpy(), B.cpu().numpy()
        res = max(np.abs(Kf @ xh[r] - bh[r]).max() / np.abs(bh[r]).max() for r in (0, nrhs // 2, nrhs - 1))
        print(f"{'interleaved, matrix pipe' if not mode else 'per right-hand side (PIPS_HIP_MULTI=0)'}: {nrhs} rhs: {min(ts) * 1e3:.2f} ms per solve(nrhs) (adaptive refinement, at most 2 steps), residual {res:.1e}", flush=True)
    s.close() if hasattr(s, "close") else None
