"""Development aid: replay one case of tests/test_fuzz_gpu.py and compare the Schur complement of the HIP path, of the oracle
and of numpy against a reference refined in long double (usage: python tools/fuzz_debug.py <case>)."""
import sys, os
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, torch
import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor
from tests.test_fuzz_gpu import _banded_W
case = int(sys.argv[1])
rng = np.random.default_rng(1000 + case)
N = int(rng.integers(1, 4)); n_i = int(rng.choice([37, 90, 160, 333, 520])); my_i = max(1, int(n_i * rng.choice([0.25, 0.5, 0.8])))
n0, myl = int(rng.integers(0, 9)), int(rng.integers(0, 9))
if n0 + myl == 0: n0 = 3
rho = float(rng.choice([2.0, 5.0, 12.0])) / n_i
structured = bool(case % 3 == 0)
os.environ["PIPS_HIP_RELAX_ZEROS"] = str(rng.choice([0.0, 0.4, 0.7])); os.environ["PIPS_HIP_SPINE"] = str(int(rng.integers(0, 2))); os.environ["PIPS_HIP_MULTI"] = str(int(rng.integers(0, 2)))
prob = Problem(500 + case, N, n_i, my_i, n0, myl, rho, diag_lo=float(rng.choice([-2, -4])), diag_hi=float(rng.choice([2, 4])))
if structured:
    for blk in prob.blocks:
        Wp = _banded_W(rng, my_i, n_i, int(rng.integers(2, 9)))
        K, dpos = pa.kkt_leaf_assemble(n_i, Wp); K.val[dpos] = blk["diag"]; blk.update(W=Wp, K=K, dpos=dpos)
S = prob.S
mode = int(rng.integers(0, 3)); cut = rng.choice(["model", "all_head", "all_tail", "half"])
print("case", case, N, n_i, my_i, n0, myl, rho, "structured", structured, "mode", mode, "cut", cut)
import mpmath
def sc_with(dtype_solver):
    SC = np.zeros((S, S))
    for b in range(N):
        Kf = prob.K_full(b).toarray(); Bt = prob.Bt_scipy(b).toarray()
        SC -= Bt @ dtype_solver(Kf, Bt.T)
    return SC
sc_np = sc_with(lambda K, B: np.linalg.solve(K, B))
def solve_ld(K, B):
    Kl = K.astype(np.longdouble); Bl = B.astype(np.longdouble)
    # long double LU via numpy is not available: iterative refinement in long double around the double solve
    X = np.linalg.solve(K, B).astype(np.longdouble)
    for _ in range(5):
        R = Bl - Kl @ X
        X = X + np.linalg.solve(K, R.astype(np.float64)).astype(np.longdouble)
    return X.astype(np.float64)
sc_ref = sc_with(solve_ld)
bt = pa.LeafBatch(N, S); bt.set_schur_mode(mode)
force = {"model": -1, "all_head": prob.n_leaf, "all_tail": 0, "half": prob.n_leaf // 2}[cut]
bt.set_options(force_n_head=force)
for b in range(N): bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
bt.analyze(2)
for b in range(N): bt.set_values(b, prob.blocks[b]["K"].val)
SCd = torch.zeros(S * S, dtype=torch.float64, device="cuda"); bt.factor(SCd, S); bt.sync()
got = hip_lower_as_rowmajor(SCd.cpu().numpy(), S)
orc_sc = np.tril(prob.oracle_schur())
ref = np.tril(sc_ref); scale = np.abs(ref).max()
print("inertia", [bt.inertia(b) for b in range(N)], "cond(K0) ~", np.linalg.cond(prob.K_full(0).toarray()))
print("rel diff  hip-ref %.2e   oracle-ref %.2e   numpy-ref %.2e   hip-oracle %.2e" % (np.abs(got - ref).max() / scale, np.abs(orc_sc - ref).max() / scale, np.abs(np.tril(sc_np) - ref).max() / scale, np.abs(got - orc_sc).max() / scale))
