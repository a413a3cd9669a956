"""Development aid: the whole pipeline on an energy-system-like instance - time-coupled blocks (banded W_i), first-stage
variables, and 2-link coupling rows between neighbouring blocks - solved end to end by the device IPM harness with the sparse
root (PIPS_IPM_SPARSE_ROOT=1) and, for comparison, with the dense root.  usage: energy_like_demo.py [N n_i L n0 bw]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
N, n_i, L, n0, bw = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 4000, 40, 8, 12)
import pips_ipmpp_amd as pa
my_i, myl = n_i // 2, (N - 1) * L
rng = np.random.default_rng(0)
blocks, cs, bs = [], [], []
x0s = rng.uniform(0.5, 1.5, n0)
F0 = sp.random(myl, n0, density=min(1.0, 2.0 / n0), random_state=1, format="csr"); F0.sort_indices()
blink = F0 @ x0s
cs.append(rng.uniform(0.5, 1.5, n0))
for i in range(N):
    rows = np.repeat(np.arange(my_i), 5)
    center = (np.arange(my_i) * n_i // my_i)[:, None]
    cols = np.clip(center + rng.integers(-bw, bw + 1, (my_i, 5)), 0, n_i - 1); cols[:, 0] = center[:, 0]
    W = sp.csr_matrix((rng.uniform(-1, 1, rows.size), (rows, cols.ravel())), shape=(my_i, n_i)); W.sum_duplicates(); W.sort_indices()
    T = sp.random(my_i, n0, density=2.0 / n0, random_state=10 + i, format="csr"); T.sort_indices()
    fr, fc, fv = [], [], []
    for pair in (i - 1, i):
        if 0 <= pair < N - 1:
            r = np.repeat(np.arange(pair * L, (pair + 1) * L), 3)
            fr.append(r); fc.append(rng.integers(0, n_i, r.size)); fv.append(rng.uniform(-1, 1, r.size))
    F = sp.csr_matrix((np.concatenate(fv), (np.concatenate(fr), np.concatenate(fc))), shape=(myl, n_i)); F.sum_duplicates(); F.sort_indices()
    xs = rng.uniform(0.5, 1.5, n_i)
    blocks.append(tuple(pa.Csr(M.shape[0], M.shape[1], M.indptr, M.indices, M.data) for M in (W, T, F)))
    cs.append(rng.uniform(0.5, 1.5, n_i)); bs.append(T @ x0s + W @ xs); blink = blink + F @ xs
c = np.concatenate(cs); b = np.concatenate([blink] + bs)
F0p = pa.Csr(myl, n0, F0.indptr, F0.indices, F0.data)
print(f"{N} blocks x {n_i} vars ({my_i} time-coupled rows, band {bw}), {L} linking rows per neighbouring pair: S = {n0 + myl}, {c.size:,} variables", flush=True)
for sparse in ((1,) if os.environ.get("PIPS_DEMO_ONLY_SPARSE") else (1, 0)):
    os.environ["PIPS_IPM_SPARSE_ROOT"] = str(sparse)
    t0 = time.time(); ipm = pa.IpmSolver(n0, myl, blocks, F0p, c, b); ts = time.time() - t0
    t0 = time.time(); res = ipm.solve(max_iter=150, mutol=1e-8, artol=1e-8); dt = time.time() - t0
    print(f"{'sparse' if sparse else 'dense '} root: setup {ts:.1f} s, {res['iterations']} iterations in {dt:.2f} s ({res['iterations']/dt:.1f} it/s), status {res['status']}, "
          f"objective {res['objective']:.10e}, mu {res['mu']:.1e}, ||r|| {res['rnorm']:.1e}", flush=True)
    del ipm
