"""Ad-hoc timing of the batched factor/solve on one GPU (development aid; bench.py is the contract)."""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=64)
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--S", type=int, default=2000)
ap.add_argument("--rho", type=float, default=1e-3)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
N, n_i, my_i, n0, myl = a.blocks, a.n, a.n // 2, a.S // 2, a.S // 2
S = n0 + myl
t0 = time.time()
bt = pa.LeafBatch(N, S, device=0)
vals = []
for b in range(N):
    W, T, F, c, xs = pa.gen_block(42, b + 1, n_i, my_i, n0, myl, a.rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
    K.val[dpos] = np.concatenate([pa.gen_diagonal(42, b + 1, n_i), -1e-8 * np.ones(my_i)])
    bt.set_block(b, K, n_i, Bt)
    vals.append(K.val)
print(f"generate {time.time()-t0:.1f}s", flush=True)
t0 = time.time(); bt.analyze(8); print(f"analyze {time.time()-t0:.1f}s", bt.info(), flush=True)
for b in range(N): bt.set_values(b, vals[b])
SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
bt.set_timing(True)
for r in range(a.reps):
    SC.zero_(); torch.cuda.synchronize(); t0 = time.time()
    bt.factor(SC, S); bt.sync(); dt = time.time() - t0
    tm = bt.get_timing()
    print(f"factor {dt*1e3:.1f} ms", {k: (round(v[0], 2), v[1]) for k, v in tm.items()}, flush=True)
info = bt.info()
fl = info["flops_factor"] + info["flops_border"]
print(f"flops {fl/1e12:.2f} TF -> {fl/dt/1e12:.1f} TFLOP/s")
print("inertia b0", bt.inertia(0))
x = torch.randn(info["n"], dtype=torch.float64, device="cuda"); rhs = x.clone()
bt.set_timing(False)
for r in range(a.reps):
    x.copy_(rhs); torch.cuda.synchronize(); t0 = time.time()
    bt.solve(x); bt.sync(); print(f"solve(all blocks, 1 refinement) {(time.time()-t0)*1e3:.1f} ms", flush=True)
