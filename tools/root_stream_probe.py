"""Development aid: time of (factorize + root inertia query + one solveCompressed) per call with the dense root on its own stream (default)
and on the main stream (PIPS_HIP_ROOT_SYNC=1) - the order of calls the IPM harness makes.  usage: python tools/root_stream_probe.py [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_i, my_i, n0, myl = 10000, 5000, 1000, 1000
bt, diag_h = bench.build_rank_problem(pa, 5, list(range(nb)), n_i, my_i, n0, myl, 0.001, 0)
F0, c0, x0s = pa.gen_root(5, n0, myl)
kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
diag = torch.tensor(diag_h, device="cuda"); xd0 = torch.tensor(pa.gen_diagonal(5, 0, n0), device="cuda")
b0 = torch.randn(n0 + myl, dtype=torch.float64, device="cuda"); bl = torch.randn(diag.numel(), dtype=torch.float64, device="cuda")
def loop(query, n=8):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        kkt.factorize(diag, xd0)
        if query: kkt.root_inertia()
        x0, xl = b0.clone(), bl.clone()
        kkt.solve_compressed(x0, xl)
    bt.sync(); torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
loop(True, 2)
print(f"ROOT_SYNC={os.environ.get('PIPS_HIP_ROOT_SYNC')}: with inertia query {loop(True):.2f} ms per call, without {loop(False):.2f} ms", flush=True)
