"""BASELINE configs[3] on the generator of SURVEY 8d (uniformly random W_i, ~10 non-zeros per row): how large can n_i get at 256 blocks per
GPU in 288 GB?  Symbolic analysis of one block per size (CPU only): stored entries of L, arena bytes, factor flops."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pips_ipmpp_amd as pa

HBM = 288e9
BLOCKS = 256
S = 8000
out = []
for n_i in [int(a) for a in sys.argv[1:]] or [5000, 10000, 15000, 20000, 25000, 30000, 50000]:
    my_i, n0, myl = n_i // 2, S // 2, S // 2
    rho = 10.0 / n_i                       # ~10 non-zeros per row of W_i ("cfg4 75k + 0.25M (10 nnz/row)")
    W, T, F, c, xs = pa.gen_block(20261002, 1, n_i, my_i, n0, myl, rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
    info = pa.symbolic_probe(K, n_i, Bt=Bt)
    per_block = info["arena_bytes"] + 8.0 * info["m"] ** 2       # + the scaled copy U = L D of the tail
    rec = dict(n_i=n_i, nnzL=info["nnzL"], n_head=info["n_head"], tail_m=info["m"], arena_bytes_per_block=info["arena_bytes"],
               bytes_per_block_with_U=per_block, bytes_256_blocks=per_block * BLOCKS, fits_288GB=bool(per_block * BLOCKS < 0.9 * HBM),
               factor_flops_per_block=info["flops_factor"] + info["flops_border"])
    out.append(rec)
    print(json.dumps(rec), flush=True)
