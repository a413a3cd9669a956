"""Development aid: the configs[3] share's leaf factorisation with the border taken away (T_i = 0, no linking rows): what the head costs for the
rows of K alone - the lower bound of any scheme that treats the border rows apart.  usage: python tools/cfg3_konly_probe.py [blocks n_i]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import pips_ipmpp_amd as pa
import families
import bench
nb, n_i = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (256, 50000)
c3 = families.CONFIG3_SHARE
blocks, F0, my_i, myl = families.time_coupled_blocks(nb, n_i, c3["L"], c3["n0"], c3["bw"], c3["nnz_row"], c3["seed"])
for variant in ("with border", "K only"):
    def data(b):
        W, T, F = blocks[b]
        if variant == "K only":
            T = pa.Csr(T.nrows, T.ncols, np.zeros(T.nrows + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
            F = pa.Csr(F.nrows, F.ncols, np.zeros(F.nrows + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
        return W, T, F
    bt, diag_h = bench.build_rank_problem(pa, 5, list(range(nb)), n_i, my_i, c3["n0"], myl, 0.0, 0, block_data=data)
    diag = torch.tensor(diag_h, device="cuda")
    S = c3["n0"] + myl
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.set_diagonals(diag)
    for rep in range(2): bt.factor(SC, S); bt.sync()
    bt.set_timing(True); bt.factor(SC, S); bt.sync(); tm = bt.get_timing(); bt.set_timing(False)
    torch.cuda.synchronize(); t0 = time.time()
    for rep in range(3): bt.factor(SC, S)
    bt.sync(); dt = (time.time() - t0) / 3 * 1e3
    info = bt.info()
    if variant == "with border" and os.environ.get("PROBE_WIDE"):
        # primal diagonals over sixteen decades, as late interior-point iterations have them: does the head's time depend on the values?
        rngw = np.random.default_rng(3)
        dw = diag_h.copy().reshape(nb, -1)
        dw[:, :n_i] = 10.0 ** rngw.uniform(-8, 8, (nb, n_i))
        bt.set_diagonals(torch.tensor(dw.reshape(-1), device="cuda"))
        for rep in range(2): bt.factor(SC, S); bt.sync()
        bt.set_timing(True); bt.factor(SC, S); bt.sync(); tw = bt.get_timing(); bt.set_timing(False)
        print(f"   primal diagonals 1e-8 .. 1e8: head {tw['head'][0]:.2f} ms, perturbed pivots {sum(bt.inertia(b)[2] for b in range(nb))}", flush=True)
    print(f"{variant}: factor {dt:.2f} ms, head {tm['head'][0]:.2f} ms, nnzL {info['nnzL']:,}, levels {info['n_levels']}, max front {info['max_front']}, nb {info['nb']}", flush=True)
    bt.close()
