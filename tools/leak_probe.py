"""Development aid: device memory across 30 create / analyze / 20x(factorize + solveCompressed) / destroy cycles (no leak)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pips_ipmpp_amd as pa
from tests.util import Problem
prob = Problem(3, 3, 400, 200, 12, 10, 0.02)
def once():
    bt = pa.LeafBatch(prob.N, prob.S)
    for b in range(prob.N): bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for b in range(prob.N): bt.set_values(b, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    xd = torch.tensor(prob.x_diag0, device="cuda")
    for _ in range(20):
        kkt.factorize(diag, xd)
        b0 = torch.randn(prob.S, dtype=torch.float64, device="cuda"); bl = torch.randn(prob.N * prob.n_leaf, dtype=torch.float64, device="cuda")
        kkt.solve_compressed(b0, bl)
    bt.sync()
    kkt.close() if hasattr(kkt, "close") else None
    bt.close() if hasattr(bt, "close") else None
    del kkt, bt
free0 = None
for rep in range(30):
    once()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if rep == 2: free0 = free
    if rep in (2, 10, 29): print(rep, "free MiB", free // 2**20, flush=True)
print("leak over 27 create/destroy cycles (MiB):", (free0 - free) / 2**20)
