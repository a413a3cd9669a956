"""Development aid: bitwise reproducibility of factorize / Schur complement / solveCompressed in deterministic mode.
usage: det_probe.py [blocks] [deterministic 0|1] [banded 0|1]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as sp
import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
det = int(sys.argv[2]) if len(sys.argv) > 2 else 1
banded = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if banded:
    from tests.test_sparse_root_gpu import TwoLinkProblem
    prob = TwoLinkProblem(93, N, 2400, 1200, 5, 4, 5.0 / 2400)
else:
    prob = Problem(7, N, 600, 300, 30, 20, 0.02)
S = prob.S
def run():
    bt = pa.LeafBatch(prob.N, S)
    bt.set_deterministic(bool(det))
    for b in range(prob.N): bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    t0 = time.time(); bt.analyze(4); ta = time.time() - t0
    for b in range(prob.N): bt.set_values(b, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    out = []
    for rep in range(3):
        kkt.factorize(diag, xd0)
        SC = kkt.schur_to_host().copy()
        rng = np.random.default_rng(0)
        b0 = torch.tensor(rng.standard_normal(S), device="cuda"); bl = torch.tensor(rng.standard_normal(prob.N * prob.n_leaf), device="cuda")
        kkt.solve_compressed(b0, bl); bt.sync()
        out.append((SC, np.concatenate([b0.cpu().numpy(), bl.cpu().numpy()]), [bt.inertia(b) for b in range(prob.N)], kkt.root_inertia()))
    return out, bt.info(), ta
a, info, ta = run()
b, _, _ = run()
runs = a + b
print("info", info, "analyze %.2f s" % ta)
print("SC bitwise equal over 6 factorisations (2 handles):", all(np.array_equal(runs[0][0], r[0]) for r in runs))
print("solveCompressed bitwise equal:", all(np.array_equal(runs[0][1], r[1]) for r in runs), " inertia equal:", all(runs[0][2] == r[2] and runs[0][3] == r[3] for r in runs))
print("max SC diff", max(np.abs(runs[0][0] - r[0]).max() for r in runs), "max x diff", max(np.abs(runs[0][1] - r[1]).max() for r in runs))
