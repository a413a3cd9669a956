"""One seed of tools/native_sweep.py with the verbose log (development aid): native_one.py seed [option value]..."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pips_ipmpp_amd as pa
from general_lp_gen import random_block_lp
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
nb = int(rng.integers(2, 5))
blocks = random_block_lp(1000 + seed, nb, int(rng.integers(4, 9)), int(rng.integers(8, 20)), int(rng.integers(2, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 4)), free_fraction=0.15)
ipm = pa.GeneralIpmSolver(blocks)
for k in range(2, len(sys.argv) - 1, 2):
    ipm.set_option(sys.argv[k], float(sys.argv[k + 1]))
res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8, verbose=2)
print(res, ipm.stats(), ipm.stats2())
