"""Generates tests/golden/*.npz (run in the build container, where libmkl_rt.so is available).

Vectors:
  generator_v1.npz   checksums of the synthetic generator so that every machine feeds identical inputs
  arrowhead_small.npz  a 3-block arrowhead system: per-block K_i (CSR lower incl. IPM diagonals), Br_i^T, right-hand sides,
                     and the outputs of the reference's third-party arithmetic on them — MKL PARDISO driven with the
                     reference's call sequence and iparm (oracle/pardiso_mkl.py, PardisoSolver.C:141-352,
                     PardisoProjectSolver.C:68-77) for K_i^-1 b and the Schur complement, LAPACK dsytrf/dsytrs
                     (DeSymIndefSolver.C:78,112) for the root — plus the solution of the full arrowhead system.
Data only: inputs and expected outputs, no reference source text.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pips_ipmpp_amd as pa  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import pardiso_mkl as pm  # noqa: E402
from tests.util import Problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    # ---- generator checksums
    out = {}
    for (seed, blk, n_i, my_i, n0, myl, rho) in [(1, 1, 100, 50, 10, 8, 0.05), (20261002, 7, 1000, 500, 100, 100, 0.01),
                                                  (42, 64, 10000, 5000, 1000, 1000, 1e-3)]:
        W, T, F, c, xs = pa.gen_block(seed, blk, n_i, my_i, n0, myl, rho)
        key = f"blk_{seed}_{blk}_{n_i}"
        out[key] = np.array([digest(W.rowptr, W.colidx, W.val), digest(T.rowptr, T.colidx, T.val),
                             digest(F.rowptr, F.colidx, F.val), digest(c, xs)])
    F0, c0, x0 = pa.gen_root(42, 1000, 1000)
    out["root_42"] = np.array([digest(F0.rowptr, F0.colidx, F0.val), digest(c0, x0)])
    out["diag_42_3"] = np.array([digest(pa.gen_diagonal(42, 3, 1000, -4, 4))])
    np.savez(os.path.join(HERE, "generator_v1.npz"), **out)

    # ---- small arrowhead system with third-party reference outputs
    assert pm.available(), "libmkl_rt.so needed to generate the golden vectors"
    prob = Problem(20261002, 3, 120, 60, 10, 8, 0.05)
    S, N = prob.S, prob.N
    rng = np.random.default_rng(20261002)
    g = {"N": N, "n_i": prob.n_i, "my_i": prob.my_i, "n0": prob.n0, "myl": prob.myl}
    SC = np.zeros((S, S))
    solvers = []
    for b in range(N):
        K = prob.K_scipy(b)
        Bt = prob.Bt_scipy(b)
        s = pm.MklPardisoSolver(K)
        s.matrixChanged()
        solvers.append(s)
        rhs = rng.standard_normal(prob.n_leaf)
        x = rhs.copy()
        s.solve(x)
        g[f"K{b}_rowptr"], g[f"K{b}_colidx"], g[f"K{b}_val"] = K.indptr, K.indices, K.data
        g[f"Bt{b}_rowptr"], g[f"Bt{b}_colidx"], g[f"Bt{b}_val"] = Bt.indptr, Bt.indices, Bt.data
        g[f"rhs{b}"], g[f"x{b}"] = rhs, x
        g[f"inertia{b}"] = np.array(s.get_inertia())
        orc.add_term_to_schur_compl_blocked(SC, s, Bt)      # reference's K4-K6 loop around PARDISO solves
    g["SC_assembled"] = np.tril(SC)
    SCf = prob.oracle_finalize(SC.copy())
    g["SC_finalized"] = np.tril(SCf)
    g["x_diag0"] = prob.x_diag0
    g["F0_rowptr"], g["F0_colidx"], g["F0_val"] = prob.F0.rowptr, prob.F0.colidx, prob.F0.val
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SCf))
    g["root_inertia"] = np.array(root.get_inertia())
    b0 = rng.standard_normal(S)
    bs = [rng.standard_normal(prob.n_leaf) for _ in range(N)]
    g["b_root"] = b0
    for b in range(N):
        g[f"b{b}"] = bs[b]
    x0, xs = b0.copy(), [v.copy() for v in bs]
    orc.solve_compressed(x0, xs, solvers, [prob.Bt_scipy(b) for b in range(N)], root, prob.n0, 0, 0, prob.myl, 0)
    g["sol_root"] = x0
    for b in range(N):
        g[f"sol{b}"] = xs[b]
    np.savez_compressed(os.path.join(HERE, "arrowhead_small.npz"), **g)
    print("golden vectors written")


if __name__ == "__main__":
    main()
