"""End-to-end IPM of the host harness on a synthetic arrowhead LP (development / reporting aid).
usage: python tools/ipm_run.py [N n_i S]   (default config 2: 64 10000 2000)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pips_ipmpp_amd as pa
N, n_i, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 10000, 2000)
my_i, n0, myl = n_i // 2, S // 2, S // 2
rho = 10.0 / n_i
seed = 20261002
t0 = time.time()
F0, c0, x0s = pa.gen_root(seed, n0, myl)
blocks, cs, bs = [], [c0], []
blink = F0.to_scipy() @ x0s
for b in range(1, N + 1):
    W, T, F, c, xs = pa.gen_block(seed, b, n_i, my_i, n0, myl, rho)
    blocks.append((W, T, F)); cs.append(c)
    bs.append(T.to_scipy() @ x0s + W.to_scipy() @ xs)
    blink = blink + F.to_scipy() @ xs
c = np.concatenate(cs); bvec = np.concatenate([blink] + bs)
print(f"generated {N} blocks x {n_i} vars, S={S} in {time.time()-t0:.1f}s", flush=True)
t0 = time.time()
ipm = pa.IpmSolver(n0, myl, blocks, F0, c, bvec)
if os.environ.get("GONDZIO"):
    ipm.set_gondzio(int(os.environ["GONDZIO"]))
print(f"setup (symbolic analysis, device upload) {time.time()-t0:.1f}s", flush=True)
t0 = time.time()
res = ipm.solve(max_iter=100, mutol=1e-8, artol=1e-8, verbose=True)
dt = time.time() - t0
res["seconds"] = dt; res["iterations_per_second"] = res["iterations"] / dt
res["config"] = f"{N} blocks x {n_i} vars, Schur dim {S}"
print(json.dumps({k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in res.items()}))
