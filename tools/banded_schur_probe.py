"""Development aid: time-coupled blocks WITH a border - Schur contribution by the augmented partial factorisation (mode 1)
versus the reference's blocked multi-RHS solves (mode 2), and what the cost model picks (mode 0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import pips_ipmpp_amd as pa
N, n_i, S, bw = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 10000, 2000, 20)
my_i, n0, myl = n_i // 2, S // 2, S // 2
rng = np.random.default_rng(0)
blocks = []
for b in range(N):
    _, T, F, c, xs = pa.gen_block(7, b + 1, n_i, my_i, n0, myl, 10.0 / n_i)
    rows, cols = [], []
    for r in range(my_i):
        center = int(r * n_i / my_i)
        cs = np.union1d(np.clip(center + rng.integers(-bw, bw + 1, 9), 0, n_i - 1), [center])
        rows += [r] * len(cs); cols += list(cs)
    W = sp.csr_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(my_i, n_i)); W.sum_duplicates(); W.sort_indices()
    Wp = pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)
    K, dpos = pa.kkt_leaf_assemble(n_i, Wp)
    K.val[dpos] = np.concatenate([10 ** rng.uniform(-4, 4, n_i), -1e-8 * np.ones(my_i)])
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
    blocks.append((K, Bt))
ref = None
for mode in (2, 1, 0):
    bt = pa.LeafBatch(N, S, device=0)
    bt.set_schur_mode(mode)
    for b, (K, Bt) in enumerate(blocks): bt.set_block(b, K, n_i, Bt)
    t0 = time.time(); bt.analyze(16); ta = time.time() - t0
    for b, (K, Bt) in enumerate(blocks): bt.set_values(b, K.val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S); bt.sync()
    bt.set_timing(True)
    SC.zero_(); torch.cuda.synchronize(); t0 = time.time(); bt.factor(SC, S); bt.sync(); dt = time.time() - t0
    tm = {k: (round(v[0], 2), v[1]) for k, v in bt.get_timing().items()}
    got = np.tril(SC.cpu().numpy().reshape(S, S).T)
    if ref is None: ref = got
    info = bt.info()
    print(f"requested mode {mode} -> used {bt.schur_mode()}: analyze {ta:.2f}s, factor+schur {dt*1e3:.1f} ms, nnzL {info['nnzL']:,}, "
          f"arena {info['arena_bytes']/2**20:.0f} MiB, vs mode 2: {np.abs(got-ref).max()/np.abs(ref).max():.1e}\n   {tm}", flush=True)
    del bt, SC
