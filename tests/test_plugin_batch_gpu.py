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


def test_single_and_batch_factorisations_mixed_never_answer_from_stale_factors():
    """A handle goes through matrixChanged() on its own, later through the array-of-handles entry with other values, later on its own again:
    solve / solve_sparse / solve_dev / get_inertia always answer from the NEWEST factorisation of that leaf, whichever engine holds it
    (round-4 advisor finding: the private factors stayed 'factored' after a batch factorisation and were used, silently)."""
    prob = Problem(11, 3, 500, 250, 24, 16, 0.02)
    N = prob.N
    solvers = _solvers(prob)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(prob.n_leaf)

    def expect(b):
        xo = rhs.copy()
        prob.oracle_leaf(b).solve(xo)
        return xo

    def check(b, tol=1e-9):
        want = expect(b)
        x = rhs.copy()
        solvers[b].solve(x)
        assert np.linalg.norm(x - want) / np.linalg.norm(want) < tol
        cs = np.ones(prob.n_leaf, np.int32)
        X = np.stack([rhs, 2.0 * rhs])
        solvers[b].solve_sparse(X, cs)
        assert np.linalg.norm(X[0] - want) / np.linalg.norm(want) < tol and np.linalg.norm(X[1] - 2 * want) / np.linalg.norm(want) < 2 * tol
        xd = torch.tensor(rhs, device="cuda")
        solvers[b].solve_dev(xd)
        torch.cuda.synchronize()
        assert np.linalg.norm(xd.cpu().numpy() - want) / np.linalg.norm(want) < tol

    def rescale(b, f):
        prob.blocks[b]["K"].val[prob.blocks[b]["dpos"]] = prob.blocks[b]["diag"] * f
        if hasattr(prob, "_leaf_cache"):
            prob._leaf_cache.pop(b, None)

    # 1. every handle alone
    for b in range(N):
        solvers[b].matrixChanged()
    x_old = rhs.copy()
    solvers[0].solve(x_old)
    check(0)
    # 2. other values, through the batch: the private factors are stale now
    for b in range(N):
        rescale(b, 3.0)
    SC = np.zeros((prob.S, prob.S))
    pa.HipLdlSolver.factor_schur_batch(solvers, SC)
    x_new = rhs.copy()
    solvers[0].solve(x_new)
    assert np.linalg.norm(x_new - x_old) > 1e-3 * np.linalg.norm(x_old)       # (the two factorisations do differ)
    for b in range(N):
        check(b)
    # 3. one handle alone again with third values: its batch copy is the stale one; the siblings still answer from the batch
    rescale(1, 0.25)
    solvers[1].matrixChanged()
    check(1)
    check(0)
    check(2)
    assert solvers[1].get_inertia() == (prob.n_i, prob.my_i, 0)
    # ... and the array-of-handles solve refuses to mix the two generations instead of answering from the stale copy
    with pytest.raises(RuntimeError, match="factorised on its own"):
        pa.HipLdlSolver.solve_batch(solvers, [rhs.copy() for _ in range(N)])
    # 4. a sibling goes away: the others keep working through the batch
    solvers[2].close()
    check(0)
    solvers[0].close()
    solvers[1].close()
