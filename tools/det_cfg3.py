import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pips_ipmpp_amd as pa
import families
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_i = 50000
c3 = families.CONFIG3_SHARE
blocks, F0, my_i, myl = families.time_coupled_blocks(N, n_i, c3["L"], c3["n0"], c3["bw"], c3["nnz_row"], c3["seed"])
n0 = c3["n0"]; S = n0 + myl
res = {}
for det in (1, 0):
    bt = pa.LeafBatch(N, S)
    bt.set_deterministic(bool(det))
    diags = []
    for b, (W, T, F) in enumerate(blocks):
        K, dpos = pa.kkt_leaf_assemble(n_i, W)
        Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
        d = np.concatenate([pa.gen_diagonal(c3["seed"], b + 1, n_i), -1e-8 * np.ones(my_i)])
        K.val[dpos] = d
        bt.set_block(b, K, n_i, Bt); diags.append((K.val, d))
    t0 = time.time(); bt.analyze(16); ta = time.time() - t0
    for b in range(N): bt.set_values(b, diags[b][0])
    info = bt.info()
    kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
    leaf_diag = torch.tensor(np.concatenate([d for _, d in diags]), device="cuda")
    xd0 = torch.tensor(pa.gen_diagonal(c3["seed"], 0, n0), device="cuda")
    outs = []
    rng = np.random.default_rng(1)
    b0h, blh = rng.standard_normal(S), rng.standard_normal(N * (n_i + my_i))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        kkt.factorize(leaf_diag, xd0); bt.sync(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
        b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
        t0 = time.perf_counter(); kkt.solve_compressed(b0, bl); bt.sync(); ts = time.perf_counter() - t0
        outs.append((kkt.schur_to_host().copy(), b0.cpu().numpy(), bl.cpu().numpy()))
    same = all(np.array_equal(outs[0][i], outs[r][i]) for r in (1, 2) for i in range(3))
    print(f"deterministic={det}: multifrontal {info['multifrontal_head']}, analyze {ta:.1f} s, factorize {tf*1e3:.1f} ms, solveCompressed {ts*1e3:.1f} ms, bit-identical over 3 runs: {same}", flush=True)
    kkt.close(); bt.close()
