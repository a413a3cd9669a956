"""Generates tests/golden/ipm_config1_trace.npz: the per-iteration history of the CPU IPM restatement (oracle/ipm_oracle.py,
KKT systems solved by SuperLU) on a config-1-sized LP of the synthetic family (BASELINE.json configs[0]: 4 blocks x 1000
variables, 500 equality rows per block, Schur dimension 200), SURVEY.md §8 a18: "pinned by oracle captures for config 1:
per-iteration mu, ||r||inf, duality gap, step lengths, #iterations, final objective".

    python tests/golden/make_ipm_config1.py        (about 16 minutes: SuperLU fills in heavily on the random sparsity)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ipm_oracle as io  # noqa: E402
from tests.test_ipm_gpu import build_lp  # noqa: E402

SEED, SHAPE, MUTOL, ARTOL = 2026, (4, 1000, 500, 100, 100, 0.01), 1e-9, 1e-8


def main():
    N, n_i, my_i, n0, myl, rho = SHAPE
    blocks, F0, c, b, A = build_lp(SEED, N, n_i, my_i, n0, myl, rho)
    trace = []
    o = io.solve_lp(A, b, c, 100, MUTOL, ARTOL, trace)
    rows = np.array([list(t[1:]) + [0.0] * (7 - len(t[1:])) for t in trace])   # mu, rnorm, pobj, dobj, sigma, alpha_p, alpha_d
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ipm_config1_trace.npz"), trace=rows, objective=o["objective"],
                        dual_objective=o["dual_objective"], iterations=o["iterations"], status=o["status"], dnorm=o["dnorm"],
                        seed=SEED, shape=np.array(SHAPE), mutol=MUTOL, artol=ARTOL, x0=o["x"][:n0])
    print(o["status"], o["iterations"], o["objective"])


if __name__ == "__main__":
    main()
