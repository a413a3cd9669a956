"""Array-of-handles entries of the leaf plug-in (INTEGRATION.md level 1.5b: pips_hip_ldl_factor_schur_batch / solve_batch / inertia_batch),
the device-pointer solve and the solve that honours colSparsity - each against the oracle's restatement of the reference's per-leaf loop
(addTermToSchurComplBlocked, DistributedLeafLinearSystem.C:214-252; PardisoSolver::solve(nrhss, rhss, colSparsity), PardisoSolver.C:276-352)."""
import numpy as np
import pytest
import torch

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem
from tests.test_leaf_gpu import _TimeCoupledProblem

pytestmark = pytest.mark.gpu


def _solvers(prob):
    out = []
    for b in range(prob.N):
        s = pa.HipLdlSolver(prob.blocks[b]["K"], n_primal=prob.n_i)
        s.set_border(prob.blocks[b]["Bt"])
        out.append(s)
    return out


@pytest.mark.parametrize("shape", ["random", "time_coupled"])
def test_handles_as_one_batch_match_the_per_leaf_loop(shape):
    prob = Problem(11, 4, 700, 350, 30, 20, 0.01) if shape == "random" else _TimeCoupledProblem(5, 3, 900, 450, 10, 8, 6)
    S, N = prob.S, prob.N
    solvers = _solvers(prob)
    got = np.zeros((S, S))
    pa.HipLdlSolver.factor_schur_batch(solvers, got)
    want = np.zeros((S, S))
    for b in range(N):
        want = orc.add_term_to_schur_compl_blocked(want, prob.oracle_leaf(b), prob.Bt_scipy(b))
    scale = np.abs(want).max()
    assert np.abs(np.tril(got) - np.tril(want)).max() / scale < 1e-9
    assert np.abs(np.triu(got, 1)).max() == 0.0
    assert pa.HipLdlSolver.inertia_batch(solvers) == [(prob.n_i, prob.my_i, 0)] * N
    assert solvers[1].get_inertia() == (prob.n_i, prob.my_i, 0)       # a bound handle answers through the batch
    # one right-hand side per leaf (Lsolve / Ltsolve hand their loop over the children over); a leaf without one this time
    rng = np.random.default_rng(0)
    rhs = [rng.standard_normal(prob.n_leaf) for _ in range(N)]
    rhs[N - 1] = None
    sol = [None if r is None else r.copy() for r in rhs]
    pa.HipLdlSolver.solve_batch(solvers, sol)
    for b in range(N):
        if rhs[b] is None:
            continue
        xo = rhs[b].copy()
        prob.oracle_leaf(b).solve(xo)
        assert np.linalg.norm(sol[b] - xo) / np.linalg.norm(xo) < 1e-9
    # the flat device vector
    flat = np.concatenate([rng.standard_normal(prob.n_leaf) for _ in range(N)])
    x = torch.tensor(flat, device="cuda")
    pa.HipLdlSolver.solve_batch_dev(solvers, x)
    xs = x.cpu().numpy().reshape(N, -1)
    for b in range(N):
        xo = flat.reshape(N, -1)[b].copy()
        prob.oracle_leaf(b).solve(xo)
        assert np.linalg.norm(xs[b] - xo) / np.linalg.norm(xo) < 1e-9
    # a single-leaf solve on a bound handle still works (through the batch)
    one = flat.reshape(N, -1)[0].copy()
    solvers[0].solve(one)
    assert np.linalg.norm(one - xs[0]) / np.linalg.norm(xs[0]) < 1e-12
    # second factorisation with other values: the binding is reused
    for b in range(N):
        prob.blocks[b]["K"].val[prob.blocks[b]["dpos"]] = prob.blocks[b]["diag"] * 1.5
    got2 = np.zeros((S, S))
    pa.HipLdlSolver.factor_schur_batch(solvers, got2)
    assert np.abs(got2 - got).max() > 1e-6 * scale
    for s in solvers:
        s.close()


def test_solve_dev_and_col_sparsity():
    prob = Problem(11, 2, 700, 350, 30, 20, 0.01)
    blk = prob.blocks[0]
    s = pa.HipLdlSolver(blk["K"], n_primal=prob.n_i)
    s.matrixChanged()
    Bt = prob.Bt_scipy(0)
    cols = np.nonzero(np.diff(Bt.indptr) > 0)[0][:12]
    dense = np.ascontiguousarray(Bt[cols].toarray())
    dense[3] = 0.0                                              # an all-zero right-hand side stays out, as in the reference's packing
    want = dense.copy()
    s.solve(want)
    # (a) colSparsity = the rows any border column touches (DistributedLinearSystem.C:903)
    cs = np.zeros(prob.n_leaf, np.int32)
    cs[np.unique(Bt[cols].indices)] = 1
    got = dense.copy()
    s.solve_sparse(got, cs)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    assert np.all(got[3] == 0.0)
    # (b) device pointers: one and several right-hand sides
    xd = torch.tensor(dense[0], device="cuda")
    s.solve_dev(xd)
    torch.cuda.synchronize()
    assert np.abs(xd.cpu().numpy() - want[0]).max() <= 1e-12 * np.abs(want[0]).max()
    Xd = torch.tensor(np.ascontiguousarray(dense[[0, 1, 2, 4]]), device="cuda")
    s.solve_dev(Xd, nrhs=4, ld=prob.n_leaf)
    torch.cuda.synchronize()
    assert np.abs(Xd.cpu().numpy() - want[[0, 1, 2, 4]]).max() <= 1e-10 * np.abs(want).max()
    s.close()
