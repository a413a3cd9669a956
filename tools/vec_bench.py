"""Achieved HBM bandwidth of the flat-arena vector kernels (pips_hip_vec_*) on one MI355X:
2^27 doubles per vector (1 GiB each, beyond the 256 MiB Infinity Cache), 20 repetitions after a warm-up."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pips_ipmpp_amd as pa
n = 1 << 27
x, y, z = (torch.rand(n, dtype=torch.float64, device="cuda") + 0.5 for _ in range(3))
m = (torch.rand(n, device="cuda") > 0.5).double()
V = pa.vec
cases = [("axpy", 3, lambda: V.axpy(0.5, x, y)), ("axpby", 3, lambda: V.axpby(0.5, x, 0.999, y)), ("scale", 2, lambda: V.scale(1.0000001, y)),
         ("mul (componentMult)", 3, lambda: V.mul(x, y)), ("add_product", 4, lambda: V.add_product(1e-3, x, z, y)),
         ("add_quotient (masked)", 5, lambda: V.add_quotient(1e-3, x, z, m, y)), ("dot", 2, lambda: V.dot(x, y)), ("inf_norm", 1, lambda: V.inf_norm(x)),
         ("stepbound", 2, lambda: V.stepbound(x, y)), ("dot_shifted", 4, lambda: V.dot_shifted(x, 0.5, z, y, 0.5, z))]
out = {}
for name, nvec, fn in cases:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out[name] = {"bytes": nvec * n * 8, "us": round(dt * 1e6, 1), "TB_per_s": round(nvec * n * 8 / dt / 1e12, 2)}
    print(f"{name:24s} {nvec} vectors x 1 GiB  {dt*1e6:8.1f} us  {nvec * n * 8 / dt / 1e12:5.2f} TB/s", flush=True)
print(json.dumps(out))
