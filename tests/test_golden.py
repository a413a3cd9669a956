"""Golden vectors (tests/golden/, made by tests/golden/make_golden.py): generator determinism, the oracle against the
reference's third-party arithmetic (MKL PARDISO with the reference's iparm + LAPACK dsytrf), and — on the GPU — the
HIP path against the same vectors."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.golden.make_golden import digest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_generator_is_bit_reproducible():
    g = np.load(os.path.join(HERE, "generator_v1.npz"))
    for (seed, blk, n_i, my_i, n0, myl, rho) in [(1, 1, 100, 50, 10, 8, 0.05), (20261002, 7, 1000, 500, 100, 100, 0.01),
                                                  (42, 64, 10000, 5000, 1000, 1000, 1e-3)]:
        W, T, F, c, xs = pa.gen_block(seed, blk, n_i, my_i, n0, myl, rho)
        want = g[f"blk_{seed}_{blk}_{n_i}"]
        assert digest(W.rowptr, W.colidx, W.val) == want[0]
        assert digest(T.rowptr, T.colidx, T.val) == want[1]
        assert digest(F.rowptr, F.colidx, F.val) == want[2]
        assert digest(c, xs) == want[3]
    F0, c0, x0 = pa.gen_root(42, 1000, 1000)
    assert digest(F0.rowptr, F0.colidx, F0.val) == g["root_42"][0] and digest(c0, x0) == g["root_42"][1]
    assert digest(pa.gen_diagonal(42, 3, 1000, -4, 4)) == g["diag_42_3"][0]


def _load():
    g = np.load(os.path.join(HERE, "arrowhead_small.npz"))
    N, n_i, my_i, n0, myl = (int(g[k]) for k in ("N", "n_i", "my_i", "n0", "myl"))
    n = n_i + my_i
    Ks = [sp.csr_matrix((g[f"K{b}_val"], g[f"K{b}_colidx"], g[f"K{b}_rowptr"]), shape=(n, n)) for b in range(N)]
    Bts = [sp.csr_matrix((g[f"Bt{b}_val"], g[f"Bt{b}_colidx"], g[f"Bt{b}_rowptr"]), shape=(n0 + myl, n)) for b in range(N)]
    return g, N, n_i, my_i, n0, myl, Ks, Bts


def test_oracle_reproduces_third_party_golden_vectors():
    g, N, n_i, my_i, n0, myl, Ks, Bts = _load()
    S = n0 + myl
    SC = np.zeros((S, S))
    solvers = []
    for b in range(N):
        s = orc.OracleLdl(Ks[b], n_primal=n_i)
        s.matrixChanged()
        solvers.append(s)
        x = g[f"rhs{b}"].copy()
        s.solve(x)
        assert np.linalg.norm(x - g[f"x{b}"]) / np.linalg.norm(g[f"x{b}"]) < 1e-9
        assert s.get_inertia()[:2] == tuple(g[f"inertia{b}"][:2])
        orc.add_term_to_schur_compl_blocked(SC, s, Bts[b])
    assert np.abs(np.tril(SC) - g["SC_assembled"]).max() / np.abs(g["SC_assembled"]).max() < 1e-9
    F0 = sp.csr_matrix((g["F0_val"], g["F0_colidx"], g["F0_rowptr"]), shape=(myl, n0))
    SCf = orc.finalize_kkt_dense(SC, n0, 0, myl, 0, g["x_diag0"], F0=F0)
    assert np.abs(np.tril(SCf) - g["SC_finalized"]).max() / np.abs(g["SC_finalized"]).max() < 1e-9
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SCf))
    x0 = g["b_root"].copy()
    xs = [g[f"b{b}"].copy() for b in range(N)]
    orc.solve_compressed(x0, xs, solvers, Bts, root, n0, 0, 0, myl, 0)
    assert np.linalg.norm(x0 - g["sol_root"]) / np.linalg.norm(g["sol_root"]) < 1e-8
    for b in range(N):
        assert np.linalg.norm(xs[b] - g[f"sol{b}"]) / np.linalg.norm(g[f"sol{b}"]) < 1e-8


@pytest.mark.gpu
def test_hip_path_reproduces_third_party_golden_vectors():
    import torch
    g, N, n_i, my_i, n0, myl, Ks, Bts = _load()
    S = n0 + myl
    bt = pa.LeafBatch(N, S)
    for b in range(N):
        bt.set_block(b, pa.Csr(Ks[b].shape[0], Ks[b].shape[1], Ks[b].indptr, Ks[b].indices, Ks[b].data), n_i,
                     pa.Csr(S, Bts[b].shape[1], Bts[b].indptr, Bts[b].indices, Bts[b].data))
    bt.analyze(2)
    for b in range(N):
        bt.set_values(b, Ks[b].data)
    F0 = pa.Csr(myl, n0, g["F0_rowptr"], g["F0_colidx"], g["F0_val"])
    kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0)
    kkt.factorize(None, torch.tensor(g["x_diag0"], device="cuda"))
    from tests.util import hip_lower_as_rowmajor
    got = hip_lower_as_rowmajor(kkt.schur_to_host(), S)
    assert np.abs(got - g["SC_finalized"]).max() / np.abs(g["SC_finalized"]).max() < 1e-9
    for b in range(N):
        assert bt.inertia(b)[:2] == tuple(g[f"inertia{b}"][:2])
    assert kkt.root_inertia()[:2] == tuple(g["root_inertia"][:2])
    rhs = np.concatenate([g[f"rhs{b}"] for b in range(N)])
    x = torch.tensor(rhs, device="cuda")
    bt.solve(x)
    xs = x.cpu().numpy().reshape(N, -1)
    for b in range(N):
        assert np.linalg.norm(xs[b] - g[f"x{b}"]) / np.linalg.norm(g[f"x{b}"]) < 1e-9
    b0 = torch.tensor(g["b_root"], device="cuda")
    bl = torch.tensor(np.concatenate([g[f"b{b}"] for b in range(N)]), device="cuda")
    kkt.solve_compressed(b0, bl)
    bt.sync()
    assert np.linalg.norm(b0.cpu().numpy() - g["sol_root"]) / np.linalg.norm(g["sol_root"]) < 1e-8
    sol = bl.cpu().numpy().reshape(N, -1)
    for b in range(N):
        assert np.linalg.norm(sol[b] - g[f"sol{b}"]) / np.linalg.norm(g[f"sol{b}"]) < 1e-8


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[0] at full size against ONE MKL-PARDISO factorisation of the global arrowhead matrix
# (tests/golden/make_config0_pardiso.py: assembled from the generator's raw blocks, no Schur / solveCompressed code of the
# restatement involved).  north_star: residual / solution parity 1e-8 against the CPU PARDISO path.
# ----------------------------------------------------------------------------------------------------------------------
def _config0():
    from tests.util import Problem
    g = np.load(os.path.join(HERE, "config0_global_pardiso.npz"))
    prob = Problem(int(g["seed"]), int(g["N"]), int(g["n_i"]), int(g["my_i"]), int(g["n0"]), int(g["myl"]), float(g["rho"]),
                   dual_reg=float(g["dual_reg"]))
    return g, prob


def test_oracle_matches_global_pardiso_solution_config0():
    g, prob = _config0()
    N, S, nl = prob.N, prob.S, prob.n_leaf
    leaf = [prob.oracle_leaf(b) for b in range(N)]
    SC = prob.oracle_finalize(prob.oracle_schur())
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SC))
    pos = sum(s.get_inertia()[0] for s in leaf) + root.get_inertia()[0]
    neg = sum(s.get_inertia()[1] for s in leaf) + root.get_inertia()[1]
    assert (pos, neg) == tuple(int(v) for v in g["inertia"][:2])          # Sylvester: inertia adds up over the Schur complement
    Bts = [prob.Bt_scipy(b) for b in range(N)]
    for k in range(g["rhs"].shape[0]):
        rhs, want = g["rhs"][k], g["sol"][k]
        x0 = rhs[N * nl:].copy()
        xs = [rhs[b * nl:(b + 1) * nl].copy() for b in range(N)]
        orc.solve_compressed(x0, xs, leaf, Bts, root, prob.n0, 0, 0, prob.myl, 0)
        got = np.concatenate(xs + [x0])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-8


@pytest.mark.gpu
def test_hip_path_matches_global_pardiso_solution_config0():
    import torch
    g, prob = _config0()
    N, S, nl = prob.N, prob.S, prob.n_leaf
    bt = pa.LeafBatch(N, S)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
    kkt.factorize(None, torch.tensor(prob.x_diag0, device="cuda"))
    pos = sum(bt.inertia(b)[0] for b in range(N)) + kkt.root_inertia()[0]
    neg = sum(bt.inertia(b)[1] for b in range(N)) + kkt.root_inertia()[1]
    assert (pos, neg) == tuple(int(v) for v in g["inertia"][:2])
    for k in range(g["rhs"].shape[0]):
        rhs, want = g["rhs"][k], g["sol"][k]
        b0 = torch.tensor(rhs[N * nl:].copy(), device="cuda")
        bl = torch.tensor(rhs[:N * nl].copy(), device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        got = np.concatenate([bl.cpu().numpy(), b0.cpu().numpy()])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-8
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-8


# the general block structure (leaf inequality rows, root equality / inequality rows, both kinds of linking rows) against one
# MKL-PARDISO factorisation of the global matrix (make_config0_pardiso.py: general_global_matrix)
def _general():
    from tests.test_general_gpu import GeneralProblem
    g = np.load(os.path.join(HERE, "general_global_pardiso.npz"))
    gp = GeneralProblem(int(g["seed"]), *[int(v) for v in g["dims"]], float(g["rho"]))
    return g, gp


def test_oracle_matches_global_pardiso_solution_general_structure():
    g, gp = _general()
    N, nx, my, mz, n0, my0, mz0, myl, mzl = gp.dims
    S, nleaf = gp.S, nx + my + mz
    leaf, Bts = [], []
    SC = np.zeros((S, S))
    for b in range(N):
        s = orc.OracleLdl(gp.K_scipy(b), n_primal=nx)
        s.matrixChanged()
        Bt = gp.blocks[b]["Bt"].to_scipy()
        orc.add_term_to_schur_compl_blocked(SC, s, Bt)
        leaf.append(s)
        Bts.append(Bt)
    SCf = orc.finalize_kkt_dense(SC, n0, my0, myl, mzl, gp.x_diag0, A0=gp.A0.to_scipy(), F0=gp.F0.to_scipy(), G0=gp.G0.to_scipy(),
                                 C0=gp.C0.to_scipy(), z_diag=gp.z_diag0, z_diag_link=gp.z_diag_link)
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SCf))
    pos = sum(s.get_inertia()[0] for s in leaf) + root.get_inertia()[0]
    neg = sum(s.get_inertia()[1] for s in leaf) + root.get_inertia()[1] + mz0      # the eliminated z0 rows: negative diagonal
    assert (pos, neg) == tuple(int(v) for v in g["inertia"][:2])
    for k in range(g["rhs"].shape[0]):
        rhs, want = g["rhs"][k], g["sol"][k]
        b0 = rhs[N * nleaf:].copy()
        bs = [rhs[b * nleaf:(b + 1) * nleaf].copy() for b in range(N)]
        orc.solve_compressed(b0, bs, leaf, Bts, root, n0, my0, mz0, myl, mzl, C0=gp.C0.to_scipy(), z_diag_reg=gp.z_diag0)
        got = np.concatenate(bs + [b0])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("schur_mode", [1, 2])
def test_hip_path_matches_global_pardiso_solution_general_structure(schur_mode):
    import torch
    g, gp = _general()
    N, nx, my, mz, n0, my0, mz0, myl, mzl = gp.dims
    S, nleaf = gp.S, nx + my + mz
    bt = pa.LeafBatch(N, S)
    bt.set_schur_mode(schur_mode)
    for b in range(N):
        bt.set_block(b, gp.blocks[b]["K"], nx, gp.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(N):
        bt.set_values(b, gp.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, n0, my0, myl, mzl, A0=gp.A0, F0=gp.F0, G0=gp.G0)
    kkt.set_root_inequalities(gp.C0)
    kkt.set_zdiag0(torch.tensor(gp.z_diag0, device="cuda"))
    kkt.factorize(torch.tensor(np.concatenate([b["diag"] for b in gp.blocks]), device="cuda"),
                  torch.tensor(gp.x_diag0, device="cuda"), torch.tensor(gp.z_diag_link, device="cuda"))
    pos = sum(bt.inertia(b)[0] for b in range(N)) + kkt.root_inertia()[0]
    neg = sum(bt.inertia(b)[1] for b in range(N)) + kkt.root_inertia()[1] + mz0
    assert (pos, neg) == tuple(int(v) for v in g["inertia"][:2])
    for k in range(g["rhs"].shape[0]):
        rhs, want = g["rhs"][k], g["sol"][k]
        b0 = torch.tensor(rhs[N * nleaf:].copy(), device="cuda")
        bl = torch.tensor(rhs[:N * nleaf].copy(), device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        got = np.concatenate([bl.cpu().numpy(), b0.cpu().numpy()])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-8


# one leaf block of BASELINE configs[1] at full size (10 000 variables, border to S = 2000) against MKL PARDISO
# (tests/golden/make_config1_block_pardiso.py): K^-1 b, inertia and the action of the block's Schur contribution
def _config1_block():
    from tests.util import Problem
    from tests.golden.make_config1_block_pardiso import vectors
    g = np.load(os.path.join(HERE, "config1_block_pardiso.npz"))
    prob = Problem(int(g["seed"]), 1, int(g["n_i"]), int(g["my_i"]), int(g["n0"]), int(g["myl"]), float(g["rho"]), dual_reg=float(g["dual_reg"]))
    rhs, v = vectors(int(g["seed"]), prob.n_leaf, prob.S)
    return g, prob, rhs, v


def test_oracle_matches_pardiso_on_a_full_size_config1_block():
    g, prob, rhs, v = _config1_block()
    s = prob.oracle_leaf(0)
    assert s.get_inertia()[:2] == tuple(int(x) for x in g["inertia"][:2])
    for k in range(2):
        x = rhs[k].copy()
        s.solve(x)
        assert np.linalg.norm(x - g["sol"][k]) / np.linalg.norm(g["sol"][k]) < 1e-8
    Bt = prob.Bt_scipy(0)
    t = Bt.T @ v[0]
    s.solve(t)
    w = Bt @ t
    assert np.linalg.norm(w - g["schur_action"][0]) / np.linalg.norm(g["schur_action"][0]) < 1e-8


@pytest.mark.gpu
def test_hip_path_matches_pardiso_on_a_full_size_config1_block():
    import torch
    from tests.util import hip_lower_as_rowmajor
    g, prob, rhs, v = _config1_block()
    S = prob.S
    bt = pa.LeafBatch(1, S)
    bt.set_block(0, prob.blocks[0]["K"], prob.n_i, prob.blocks[0]["Bt"])
    bt.analyze(4)
    bt.set_values(0, prob.blocks[0]["K"].val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    assert bt.inertia(0)[:2] == tuple(int(x) for x in g["inertia"][:2])
    for k in range(2):
        x = torch.tensor(rhs[k].copy(), device="cuda")
        bt.solve(x)
        bt.sync()
        got = x.cpu().numpy()
        assert np.linalg.norm(got - g["sol"][k]) / np.linalg.norm(g["sol"][k]) < 1e-8
        assert np.abs(got - g["sol"][k]).max() / np.abs(g["sol"][k]).max() < 1e-8
    # the leaf's contribution to the Schur complement is -Br^T K^-1 Br (lower triangle authoritative): its action on v
    L = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    full = L + np.tril(L, -1).T
    for k in range(2):
        w = -(full @ v[k])
        assert np.linalg.norm(w - g["schur_action"][k]) / np.linalg.norm(g["schur_action"][k]) < 1e-8
