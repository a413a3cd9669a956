"""Time of the dense root solve (two tail sweeps) for a given order; PIPS_HIP_SWEEP_LAUNCHES=1 selects the launch-per-column kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
npr = n // 2
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.zeros((n, n), dtype=torch.float64, device="cuda")
A[npr:, :npr] = torch.randn((n - npr, npr), dtype=torch.float64, device="cuda", generator=g) / np.sqrt(n)
A += torch.diag(torch.cat([torch.ones(npr, dtype=torch.float64, device="cuda") * 2.0, -torch.ones(n - npr, dtype=torch.float64, device="cuda") * 0.5]))
s = pa.HipDenseLdlSolver(n, npr)
s.matrixChanged_dev(A, n)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
for _ in range(3):
    s.solve_dev(x.clone())
torch.cuda.synchronize()
xs = [x.clone() for _ in range(20)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for y in xs:
    s.solve_dev(y)
torch.cuda.synchronize()
print(f"n={n} mode={'launches' if os.environ.get('PIPS_HIP_SWEEP_LAUNCHES') else 'rows'}: {(time.perf_counter()-t0)/20*1e3:.3f} ms per solve ({n//128+1} tile columns)")
