"""Device-resident timing of factorize / solveCompressed on the energy-like family (BASELINE configs[3] per-GPU share):
config3_probe.py [blocks] [n_i] [sparse_root 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
import families
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_i = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
sparse_root = len(sys.argv) > 3 and sys.argv[3] == "1"
L, n0, bw, nnz_row, seed = 31, 95, 12, 10, 20261004
blocks, F0, my_i, myl = families.time_coupled_blocks(N, n_i, L, n0, bw, nnz_row, seed)
S, nleaf = n0 + myl, n_i + my_i
bt = pa.LeafBatch(N, S)
diags, vals = [], []
for b, (W, T, F) in enumerate(blocks):
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
    d = np.concatenate([pa.gen_diagonal(seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
    K.val[dpos] = d
    bt.set_block(b, K, n_i, Bt)
    diags.append(d); vals.append(K.val)
t0 = time.time(); bt.analyze(32); t_an = time.time() - t0
for b in range(N):
    bt.set_values(b, vals[b])
info = bt.info()
kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0, sparse_root=sparse_root)
leaf_diag = torch.tensor(np.concatenate(diags), device="cuda")
xd0 = torch.tensor(pa.gen_diagonal(seed, 0, n0), device="cuda")
b0 = torch.randn(S, dtype=torch.float64, device="cuda"); bl = torch.randn(N * nleaf, dtype=torch.float64, device="cuda")
def timed(f, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); bt.sync(); t0 = time.perf_counter(); f(); bt.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
tf = timed(lambda: kkt.factorize(leaf_diag, xd0))
x0, xl = b0.clone(), bl.clone()
ts = timed(lambda: kkt.solve_compressed(x0, xl))
xs = bl.clone()
tl = timed(lambda: bt.solve(xs))
print(f"{N} x {n_i}, S = {S}: analyze {t_an:.1f} s, factorize {tf:.1f} ms, solveCompressed {ts:.1f} ms, leaf solve {tl:.1f} ms; info {info}")
if os.environ.get("PROBE_PHASES"):   # phase table of one factorize + one solveCompressed (HIP events)
    bt.set_timing(True)
    kkt.factorize(leaf_diag, xd0); x0, xl = b0.clone(), bl.clone(); kkt.solve_compressed(x0, xl); bt.sync(); torch.cuda.synchronize()
    print("kkt phases:", {k: round(v, 3) if not isinstance(v, tuple) else (round(v[0], 3), v[1]) for k, v in kkt.get_timing().items()})
    print("leaf phases:", {k: (round(v[0], 3), v[1]) for k, v in bt.get_timing().items()})
