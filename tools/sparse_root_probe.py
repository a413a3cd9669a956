"""Development aid: 2-link arrowhead (every linking row couples two neighbouring blocks), dense root vs sparse root
(pips_hip_kkt_create_sparse).  usage: python tools/sparse_root_probe.py [N n_i L n0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import pips_ipmpp_amd as pa
N, n_i, L, n0 = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 2000, 100, 8)
my_i, myl = n_i // 2, (N - 1) * L
S = n0 + myl
rng = np.random.default_rng(0)
blocks, diags = [], []
for i in range(N):
    W, T, _, c, xs = pa.gen_block(5, i + 1, n_i, my_i, n0, 1, 10.0 / n_i)
    rows, cols, vals = [], [], []
    for pair in (i - 1, i):
        if 0 <= pair < N - 1:
            r = np.repeat(np.arange(pair * L, (pair + 1) * L), 3)
            rows.append(r); cols.append(rng.integers(0, n_i, r.size)); vals.append(rng.uniform(-1, 1, r.size))
    F = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(myl, n_i)); F.sum_duplicates(); F.sort_indices()
    Fp = pa.Csr(myl, n_i, F.indptr, F.indices, F.data)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    d = np.concatenate([pa.gen_diagonal(5, i + 1, n_i), -1e-8 * np.ones(my_i)]); K.val[dpos] = d
    blocks.append((K, pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=Fp))); diags.append(d)
F0 = sp.random(myl, n0, density=2.0 / n0, random_state=1, format="csr"); F0.sort_indices()
F0p = pa.Csr(myl, n0, F0.indptr, F0.indices, F0.data)
diag = torch.tensor(np.concatenate(diags), device="cuda"); xd0 = torch.tensor(pa.gen_diagonal(5, 0, n0), device="cuda")
b0 = np.random.default_rng(1).standard_normal(S); bl = np.random.default_rng(2).standard_normal(N * (n_i + my_i))
sols = {}
for sparse in (True, False):
    if not sparse and S > 30000:
        print(f"dense root skipped: S = {S} would need {S*S*8*3/2**30:.1f} GiB"); continue
    free0 = torch.cuda.mem_get_info()[0]
    bt = pa.LeafBatch(N, S); bt.set_schur_mode(1)
    for b, (K, Bt) in enumerate(blocks): bt.set_block(b, K, n_i, Bt)
    bt.analyze(16)
    for b, (K, Bt) in enumerate(blocks): bt.set_values(b, K.val)
    t0 = time.time(); kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0p, sparse_root=sparse); tc = time.time() - t0
    kkt.factorize(diag, xd0); bt.sync()
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time(); kkt.factorize(diag, xd0); bt.sync(); ts.append(time.time() - t0)
    x0, xl = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
    torch.cuda.synchronize(); t0 = time.time(); kkt.solve_compressed(x0, xl); bt.sync(); tsol = time.time() - t0
    used = (free0 - torch.cuda.mem_get_info()[0]) / 2**30
    extra = f", nnz(SC) = {kkt.schur_sparse_to_host().nnz:,} of {S*(S+1)//2:,}" if sparse else ""
    print(f"{'sparse' if sparse else 'dense '} root: S = {S}, create {tc:.2f} s, factorize {min(ts)*1e3:.1f} ms, solveCompressed {tsol*1e3:.1f} ms, "
          f"device memory {used:.2f} GiB, root inertia {kkt.root_inertia()}{extra}", flush=True)
    sols[sparse] = (x0.cpu().numpy(), xl.cpu().numpy())
    del kkt, bt
if len(sols) == 2:
    print("sparse vs dense: rel. diff x0 %.1e, leaves %.1e" % (np.linalg.norm(sols[True][0] - sols[False][0]) / np.linalg.norm(sols[False][0]),
                                                             np.linalg.norm(sols[True][1] - sols[False][1]) / np.linalg.norm(sols[False][1])))
