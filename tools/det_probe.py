"""Development aid: bitwise reproducibility of the factorisation / Schur complement / solve in deterministic mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
det = int(sys.argv[2]) if len(sys.argv) > 2 else 1
banded = len(sys.argv) > 3
prob = Problem(7, N, 600, 300, 30, 20, 0.02)
S = prob.S
def run():
    bt = pa.LeafBatch(prob.N, S)
    bt.set_deterministic(bool(det))
    for b in range(prob.N): bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(prob.N): bt.set_values(b, prob.blocks[b]["K"].val)
    out = []
    for rep in range(3):
        SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
        bt.factor(SC, S)
        x = torch.tensor(np.random.default_rng(0).standard_normal(prob.N * prob.n_leaf), device="cuda")
        bt.solve(x); bt.sync()
        out.append((SC.cpu().numpy().copy(), x.cpu().numpy().copy(), [bt.inertia(b) for b in range(prob.N)]))
    return out, bt.info()
a, info = run()
b, _ = run()
runs = a + b
print("info", info)
print("SC bitwise equal over 6 factorisations (2 handles):", all(np.array_equal(runs[0][0], r[0]) for r in runs))
print("solve bitwise equal:", all(np.array_equal(runs[0][1], r[1]) for r in runs))
print("max SC diff", max(np.abs(runs[0][0] - r[0]).max() for r in runs), "max x diff", max(np.abs(runs[0][1] - r[1]).max() for r in runs))
want = np.tril(prob.oracle_schur())
got = hip_lower_as_rowmajor(runs[0][0], S)
print("SC vs oracle", np.abs(got - want).max() / np.abs(want).max(), "inertia", runs[0][2][:2])
