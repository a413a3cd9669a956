"""solveCompressed launch by launch vs as a replayed HIP graph (pips_hip_kkt_set_solve_graph), fixed refinement (one unconditional step):
solve_graph_bench.py [blocks n_i S]   default: BASELINE configs[0] (4 x 1000, S = 200) - the launch-bound end"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
N, n_i, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 1000, 200)
rho = 10.0 / n_i
seed, my_i, n0, myl = 20261002, n_i // 2, S // 2, S // 2
bt = pa.LeafBatch(N, S)
diags = []
for b in range(N):
    W, T, F, c, xs = pa.gen_block(seed, b + 1, n_i, my_i, n0, myl, rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    d = np.concatenate([pa.gen_diagonal(seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
    K.val[dpos] = d
    bt.set_block(b, K, n_i, pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)); diags.append((K.val, d))
bt.analyze(4)
for b in range(N): bt.set_values(b, diags[b][0])
F0, c0, x0s = pa.gen_root(seed, n0, myl)
kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
kkt.factorize(torch.tensor(np.concatenate([d for _, d in diags]), device="cuda"), torch.tensor(pa.gen_diagonal(seed, 0, n0), device="cuda"))
b0 = torch.randn(S, dtype=torch.float64, device="cuda"); bl = torch.randn(N * (n_i + my_i), dtype=torch.float64, device="cuda")
r0, rl = b0.clone(), bl.clone()
def run(reps):
    bt.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        kkt.solve_compressed(b0, bl)     # (in place: the vectors just keep being overwritten - timing only)
    bt.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
out = {}
for mode in ("direct", "graph"):
    kkt.set_solve_graph(mode == "graph")
    b0.copy_(r0); bl.copy_(rl); run(5)
    out[mode] = min(run(200) for _ in range(3))
print(f"{N} x {n_i}, S = {S}: solveCompressed {out['direct']:.3f} ms launch by launch, {out['graph']:.3f} ms replayed graph "
      f"(captures, replays) = {kkt.solve_graph_stats()}")
