"""Generates tests/golden/config0_global_pardiso.npz and general_global_pardiso.npz (run in the build container, where libmkl_rt.so is available).

An independent pin for the whole path at the size of BASELINE configs[0] (4 blocks x 1000 variables / 500 equality rows,
Schur dimension 200): the GLOBAL arrowhead KKT matrix

      [ K_1            B_1 ]            K_i = [ D_i  W_i^T ; W_i  -reg ]          B_i = [ T_i^T  0 ; 0 ... F_i^T ] (border)
      [      ...       ... ]            K_0 = [ D_0  F_0^T ; F_0   0   ]
      [ B_1^T  ...     K_0 ]

is assembled here from the generator's raw blocks (W_i, T_i, F_i, the diagonals) with scipy.sparse.bmat - none of the
restatement's Schur / solveCompressed code is involved - and factorised ONCE, as one sparse symmetric indefinite matrix, by MKL
PARDISO driven with the reference's settings (mtype -2, iparm of PardisoProjectSolver.C:68-77; oracle/pardiso_mkl.py is only
the ctypes binding).  Stored: the problem parameters (the generator is bit-reproducible: generator_v1.npz), seeded right-hand
sides, PARDISO's solutions and inertia.  The fused device path (leaf LDL^T -> Schur complement -> dense root -> solveCompressed)
and the CPU restatement must both reproduce the solutions to 1e-8 (north_star's tolerance against the CPU PARDISO path).
Data only: inputs and expected outputs."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pardiso_mkl as pm  # noqa: E402
from tests.util import Problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = dict(seed=20261003, N=4, n_i=1000, my_i=500, n0=100, myl=100, rho=0.01, dual_reg=1e-8)


def global_matrix(prob):
    """lower triangle of the global KKT matrix, CSR, unknown order [leaf 1 .. leaf N | x0 | y_link]"""
    N, n_i, my_i, n0, myl = prob.N, prob.n_i, prob.my_i, prob.n0, prob.myl
    rows = [[None] * (N + 1) for _ in range(N + 1)]
    for b in range(N):
        blk = prob.blocks[b]
        W, T, F = blk["W"].to_scipy(), blk["T"].to_scipy(), blk["F"].to_scipy()
        d = sp.diags(blk["diag"])
        Z = sp.csr_matrix((my_i, my_i))
        Kb = sp.bmat([[sp.csr_matrix((n_i, n_i)), None], [W, Z]], format="csr") + d              # lower: D_i on the diagonal, W_i below, -reg
        rows[b][b] = Kb
        # border rows (x0 | linking rows) against the leaf's (x_i | y_i): T_i couples y_i with x0, F_i couples x_i with y_link
        rows[N][b] = sp.bmat([[sp.csr_matrix((n0, n_i)), T.T], [F, sp.csr_matrix((myl, my_i))]], format="csr")
    F0 = prob.F0.to_scipy()
    rows[N][N] = sp.bmat([[sp.diags(prob.x_diag0), None], [F0, sp.csr_matrix((myl, myl))]], format="csr")
    return sp.csr_matrix(sp.bmat(rows, format="csr"))


GENERAL = dict(seed=17, dims=(3, 200, 80, 40, 20, 6, 7, 9, 5), rho=0.03)


def general_global_matrix(gp):
    """the general block structure (leaf inequality rows, root equality / inequality rows, linking rows of both kinds) as one
    matrix; unknown order [leaf 1 .. N : (x_i | y_i | z_i)] [x0 | y0 | z0 | y_link | z_link] = the order solveCompressed takes"""
    N, nx, my, mz, n0, my0, mz0, myl, mzl = gp.dims
    nr = n0 + my0 + mz0 + myl + mzl
    rows = [[None] * (N + 1) for _ in range(N + 1)]
    Z = sp.csr_matrix
    for b in range(N):
        blk = gp.blocks[b]
        W, Dm, T, Cb, F, G = (blk[k].to_scipy() for k in ("W", "Dm", "T", "Cb", "F", "G"))
        rows[b][b] = sp.bmat([[Z((nx, nx)), None, None], [W, Z((my, my)), None], [Dm, None, Z((mz, mz))]], format="csr") + sp.diags(blk["diag"])
        rows[N][b] = sp.bmat([[Z((n0, nx)), T.T, Cb.T], [Z((my0 + mz0, nx)), None, None], [F, Z((myl, my)), Z((myl, mz))],
                              [G, Z((mzl, my)), Z((mzl, mz))]], format="csr")
    d0 = np.concatenate([gp.x_diag0, np.zeros(my0), gp.z_diag0, np.zeros(myl), gp.z_diag_link])
    low = sp.bmat([[Z((n0, n0)), None], [sp.vstack([gp.A0.to_scipy(), gp.C0.to_scipy(), gp.F0.to_scipy(), gp.G0.to_scipy()]), Z((nr - n0, nr - n0))]],
                  format="csr")
    rows[N][N] = low + sp.diags(d0)
    return sp.csr_matrix(sp.bmat(rows, format="csr"))


def make_general():
    from tests.test_general_gpu import GeneralProblem
    gp = GeneralProblem(GENERAL["seed"], *GENERAL["dims"], GENERAL["rho"])
    N, nx, my, mz, n0, my0, mz0, myl, mzl = gp.dims
    nleaf = nx + my + mz
    Kg = general_global_matrix(gp)
    keep = np.r_[0:n0 + my0, n0 + my0 + mz0:n0 + my0 + mz0 + myl + mzl] + N * nleaf      # the border has no z0 rows
    for b in range(N):
        lo, hi = b * nleaf, (b + 1) * nleaf
        assert abs(Kg[lo:hi, lo:hi] - gp.K_scipy(b)).max() == 0.0
        assert abs(Kg[keep, lo:hi] - gp.blocks[b]["Bt"].to_scipy()).max() == 0.0
    s = pm.MklPardisoSolver(Kg, num_threads=1)
    s.matrixChanged()
    rng = np.random.default_rng(GENERAL["seed"])
    rhs = rng.standard_normal((3, Kg.shape[0]))
    sol = rhs.copy()
    Kfull = Kg + sp.tril(Kg, -1).T
    for k in range(3):
        s.solve(sol[k])
        r = np.abs(Kfull @ sol[k] - rhs[k]).max() / (np.abs(rhs[k]).max() + abs(Kfull).max() * np.abs(sol[k]).max())
        assert r < 1e-13, r
    np.savez_compressed(os.path.join(HERE, "general_global_pardiso.npz"), seed=GENERAL["seed"], dims=np.array(GENERAL["dims"]),
                        rho=GENERAL["rho"], inertia=np.array(s.get_inertia()), rhs=rhs, sol=sol)
    print("general: written; inertia", s.get_inertia(), "n", Kg.shape[0])


def main():
    assert pm.available(), "libmkl_rt.so needed to generate the golden vectors"
    prob = Problem(PARAMS["seed"], PARAMS["N"], PARAMS["n_i"], PARAMS["my_i"], PARAMS["n0"], PARAMS["myl"], PARAMS["rho"],
                   dual_reg=PARAMS["dual_reg"])
    Kg = global_matrix(prob)
    # the pieces the product is fed with are the same matrix (guards the fixture against a layout misunderstanding)
    for b in range(prob.N):
        lo, hi = b * prob.n_leaf, (b + 1) * prob.n_leaf
        assert abs(Kg[lo:hi, lo:hi] - prob.K_scipy(b)).max() == 0.0
        assert abs(Kg[prob.N * prob.n_leaf:, lo:hi] - prob.Bt_scipy(b)).max() == 0.0
    ntot = Kg.shape[0]
    s = pm.MklPardisoSolver(Kg, num_threads=1)
    s.matrixChanged()
    g = {k: np.array(v) for k, v in PARAMS.items()}
    g["inertia"] = np.array(s.get_inertia())
    rng = np.random.default_rng(PARAMS["seed"])
    rhs = rng.standard_normal((3, ntot))
    rhs[2, :prob.N * prob.n_leaf] = 0.0            # third right-hand side: root part only -> x0 = SC^-1 b0
    sol = rhs.copy()
    for k in range(3):
        s.solve(sol[k])
    Kfull = Kg + sp.tril(Kg, -1).T
    for k in range(3):
        r = np.abs(Kfull @ sol[k] - rhs[k]).max() / (np.abs(rhs[k]).max() + abs(Kfull).max() * np.abs(sol[k]).max())
        assert r < 1e-14, r
    g["rhs"], g["sol"] = rhs, sol
    np.savez_compressed(os.path.join(HERE, "config0_global_pardiso.npz"), **g)
    print("written; inertia", g["inertia"], "n", ntot, "nnz", Kg.nnz)
    make_general()


if __name__ == "__main__":
    main()
