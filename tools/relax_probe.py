"""Development aid (CPU only): effect of supernode amalgamation (PIPS_HIP_RELAX_ZEROS) on the head's level count."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, scipy.sparse as sp
    import pips_ipmpp_amd as pa
    kind = sys.argv[2]
    rng = np.random.default_rng(0)
    if kind == "banded":
        n_i, my_i, bw = 10000, 5000, 20
        rows, cols = [], []
        for r in range(my_i):
            center = int(r * n_i / my_i)
            cs = np.union1d(np.clip(center + rng.integers(-bw, bw + 1, 9), 0, n_i - 1), [center])
            rows += [r] * len(cs); cols += list(cs)
        W = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(my_i, n_i)); W.sum_duplicates(); W.sort_indices()
        Wp = pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)
        K, _ = pa.kkt_leaf_assemble(n_i, Wp)
        info = pa.symbolic_probe(K, n_i)
    else:
        n_i, my_i, n0, myl = 10000, 5000, 1000, 1000
        W, T, F, c, xs = pa.gen_block(1, 1, n_i, my_i, n0, myl, 1e-3)
        K, _ = pa.kkt_leaf_assemble(n_i, W)
        Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
        info = pa.symbolic_probe(K, n_i, Bt)
    print(json.dumps({k: v for k, v in info.items() if k in ("nnzL", "n_head", "m", "n_sn", "n_levels", "flops_factor", "upd_bytes", "arena_bytes")}))
else:
    for kind in ("banded", "random"):
        for nd in ("0", "2", "4", "6"):
            env = dict(os.environ, PIPS_HIP_ND_DEPTH=nd)
            out = subprocess.run([sys.executable, __file__, "child", kind], env=env, capture_output=True, text=True)
            print(kind, 'nd_depth', nd, out.stdout.strip() or out.stderr[-300:], flush=True)
