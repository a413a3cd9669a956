"""Stress of the single-launch sweeps: many repeated solves, every result compared bit for bit with the launch-per-column path
(deterministic mode).  A flag overtaking its data or a stale read shows up as a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["PIPS_HIP_DETERMINISTIC"] = "1"
import pips_ipmpp_amd as pa
from tests.util import Problem

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def batch(prob, launches):
    if launches:
        os.environ["PIPS_HIP_SWEEP_LAUNCHES"] = "1"
    else:
        os.environ.pop("PIPS_HIP_SWEEP_LAUNCHES", None)
    bt = pa.LeafBatch(prob.N, prob.S)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(prob.N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    bt.factor()
    return bt


big = len(sys.argv) > 2 and sys.argv[2] == "big"
shapes = [(70, 400, 200, 10, 10, 0.03), (5, 1200, 600, 30, 30, 0.01), (64, 2000, 1000, 20, 20, 0.004), (300, 300, 150, 8, 8, 0.04)]
if big:
    shapes = [(64, 10000, 5000, 100, 100, 0.001), (128, 4000, 2000, 50, 50, 0.0025)]
bad_total = 0
for shape in shapes:
    prob = Problem(77, *shape)
    rng = np.random.default_rng(5)
    rhs = torch.tensor(rng.standard_normal(prob.N * prob.n_leaf), device="cuda")
    ref_bt = batch(prob, True)
    x = rhs.clone(); ref_bt.solve(x); ref_bt.sync(); ref = x.clone()
    bt = batch(prob, False)
    bad = 0
    for r in range(reps):
        x = rhs.clone(); bt.solve(x); bt.sync()
        if not torch.equal(x, ref):
            bad += 1
    print(shape, "tail", bt.info()["m"], "mismatches", bad, "of", reps, flush=True)
    bad_total += bad
    bt.close(); ref_bt.close()
for n in ((16000,) if big else (1500, 6000)):
    rng = np.random.default_rng(n)
    npr = n // 2
    A = np.zeros((n, n))
    A[:npr, :npr] = np.diag(10 ** rng.uniform(-1, 2, npr))
    A[npr:, :npr] = rng.standard_normal((n - npr, npr))
    A[npr:, npr:] = -np.diag(10 ** rng.uniform(-3, 0, n - npr))
    rhs = torch.tensor(rng.standard_normal(n), device="cuda")
    sols = []
    for launches in (True, False):
        if launches:
            os.environ["PIPS_HIP_SWEEP_LAUNCHES"] = "1"
        else:
            os.environ.pop("PIPS_HIP_SWEEP_LAUNCHES", None)
        s = pa.HipDenseLdlSolver(n, npr)
        s.matrixChanged(np.tril(A))
        if launches:
            x = rhs.clone(); s.solve_dev(x); torch.cuda.synchronize(); ref = x.clone()
        else:
            bad = 0
            for r in range(reps):
                x = rhs.clone(); s.solve_dev(x); torch.cuda.synchronize()
                if not torch.equal(x, ref):
                    bad += 1
            print("dense root", n, "mismatches", bad, "of", reps, flush=True)
            bad_total += bad
        s.close()
sys.exit(1 if bad_total else 0)
