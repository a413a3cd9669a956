"""Generates tests/golden/config1_block_pardiso.npz (run in the build container, where libmkl_rt.so is available).

One leaf block of BASELINE configs[1] at full size (10 000 variables, 5 000 equality rows, rho = 1e-3, border to a Schur complement of
dimension 2000) against MKL PARDISO (mtype -2, the reference's iparm - PardisoProjectSolver.C:68-77; oracle/pardiso_mkl.py is only
the ctypes binding):  K_i^-1 b for two seeded right-hand sides, the inertia, and the action of the block's Schur contribution on
two seeded vectors,  Br_i^T K_i^-1 Br_i v  (two more PARDISO solves - the 2000 x 2000 matrix itself would be 32 MB).
Inputs come from the bit-reproducible generator (generator_v1.npz pins it); stored are the parameters, the vectors and PARDISO's
answers.  north_star: solution parity 1e-8 against the CPU PARDISO path - here at the size the metric is quoted on."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pardiso_mkl as pm  # noqa: E402
from tests.util import Problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = dict(seed=42, N=1, n_i=10000, my_i=5000, n0=1000, myl=1000, rho=1e-3, dual_reg=1e-8)


def vectors(seed, n_leaf, S):
    """the seeded right-hand sides and Schur test vectors (PCG64 stream: stable across numpy versions)"""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((2, n_leaf)), rng.standard_normal((2, S))


def main():
    assert pm.available()
    prob = Problem(PARAMS["seed"], PARAMS["N"], PARAMS["n_i"], PARAMS["my_i"], PARAMS["n0"], PARAMS["myl"], PARAMS["rho"],
                   dual_reg=PARAMS["dual_reg"])
    K, Bt = prob.K_scipy(0), prob.Bt_scipy(0)          # lower CSR of K_1 ; Br_1^T (S x n_leaf)
    s = pm.MklPardisoSolver(K, num_threads=8)
    s.matrixChanged()
    rhs, v = vectors(PARAMS["seed"], prob.n_leaf, prob.S)
    sol = rhs.copy()
    for k in range(2):
        s.solve(sol[k])
    w = np.zeros_like(v)
    for k in range(2):
        t = Bt.T @ v[k]
        s.solve(t)
        w[k] = Bt @ t
    np.savez_compressed(os.path.join(HERE, "config1_block_pardiso.npz"), **{k: np.array(x) for k, x in PARAMS.items()},
                        inertia=np.array(s.get_inertia()), sol=sol, schur_action=w)   # rhs and v: vectors(seed, ...) below
    print("written; inertia", s.get_inertia())


if __name__ == "__main__":
    main()
