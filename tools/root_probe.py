"""Development aid: time the dense root LDL^T (DeSymIndefSolver replacement) for growing Schur dimensions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pips_ipmpp_amd as pa
sizes = [int(a) for a in sys.argv[1:]] or [2000, 4000, 8000, 16000]
for S in sizes:
    n_primal = S // 2
    g = torch.Generator(device="cuda").manual_seed(0)
    # quasi-definite [H A^T; A -G]: H, G SPD (diagonally dominant), generated on the device
    M = torch.rand((S, S), dtype=torch.float64, device="cuda", generator=g) - 0.5
    M = M + M.T
    d = torch.full((S,), float(S), dtype=torch.float64, device="cuda")
    d[n_primal:] = -float(S)
    M += torch.diag(d)
    s = pa.HipDenseLdlSolver(S, n_primal)
    work = M.clone()
    s.matrixChanged_dev(work, S); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        work.copy_(M); torch.cuda.synchronize(); t0 = time.perf_counter()
        s.matrixChanged_dev(work, S); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    x = torch.randn(S, dtype=torch.float64, device="cuda", generator=g); b = x.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_dev(x); torch.cuda.synchronize(); tsol = time.perf_counter() - t0
    res = float(torch.linalg.norm(M @ x - b) / torch.linalg.norm(b))
    print(f"S={S}: factor {t*1e3:.2f} ms = {S**3/3/t/1e12:.1f} TFLOP/s, solve {tsol*1e3:.2f} ms, residual {res:.1e}, inertia {s.get_inertia()}", flush=True)
    s.close(); del M, work
