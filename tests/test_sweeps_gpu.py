"""The single-launch tail sweeps (k_tail_rows_fwd / k_tail_rows_bwd: one workgroup per tile row / column, ticket order,
flags per solution piece) against the launch-per-tile-column kernels they replace: same arithmetic in the same order, so the
solutions must agree to the bit (deterministic mode, so that the sparse head does not blur the comparison) - dense tails, banded tails (tile envelope), the dense root, repeated solves (flag epochs,
ticket reset) - and against the checker."""
import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from tests.util import Problem

pytestmark = pytest.mark.gpu


def _batch(prob, monkeypatch, launches):
    monkeypatch.setenv("PIPS_HIP_DETERMINISTIC", "1")     # the head's atomics would differ from run to run by themselves
    if launches:
        monkeypatch.setenv("PIPS_HIP_SWEEP_LAUNCHES", "1")
    else:
        monkeypatch.delenv("PIPS_HIP_SWEEP_LAUNCHES", raising=False)
    bt = pa.LeafBatch(prob.N, prob.S)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for b in range(prob.N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    bt.factor()
    return bt


@pytest.mark.parametrize("shape", [(5, 1200, 600, 30, 30, 0.01), (3, 2600, 1300, 20, 10, 0.004), (70, 400, 200, 10, 10, 0.03)])
def test_row_sweeps_equal_column_launches_bitwise(shape, monkeypatch):
    import torch
    prob = Problem(77, *shape)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(prob.N * prob.n_leaf)
    got = []
    for launches in (True, False):
        bt = _batch(prob, monkeypatch, launches)
        assert bt.info()["m"] > 128, "the case must have more than one tail tile"
        xs = []
        for rep in range(3):           # epochs advance, tickets are reset by the last workgroup of every launch
            x = torch.tensor(rhs * (rep + 1), device="cuda")
            bt.solve(x)
            bt.sync()
            xs.append(x.cpu().numpy())
        got.append(xs)
        bt.close()
    for rep in range(3):
        assert np.array_equal(got[0][rep], got[1][rep])
        assert np.isfinite(got[1][rep]).all()
    # and the solution is right: residual of every block
    x = got[1][0].reshape(prob.N, -1)
    for b in range(prob.N):
        r = prob.K_full(b) @ x[b] - rhs.reshape(prob.N, -1)[b]
        assert np.abs(r).max() / np.abs(rhs).max() < 1e-9


def test_row_sweeps_banded_tail_envelope(monkeypatch):
    """time-coupled blocks: the tail is banded, tile rows start at their envelope (tile_first)"""
    import torch
    from tests.test_configs_gpu import energy_like_blocks
    N, n_i, L, n0, bw, nnz_row = 6, 6000, 8, 12, 12, 10
    blocks, F0, my_i, myl = energy_like_blocks(N, n_i, L, n0, bw, nnz_row, 99)
    S = n0 + myl
    rng = np.random.default_rng(3)
    diag = [np.concatenate([10 ** rng.uniform(-2, 2, n_i), -1e-8 * np.ones(my_i)]) for _ in range(N)]
    rhs = rng.standard_normal(N * (n_i + my_i))
    got = []
    monkeypatch.setenv("PIPS_HIP_DETERMINISTIC", "1")
    for launches in (True, False):
        if launches:
            monkeypatch.setenv("PIPS_HIP_SWEEP_LAUNCHES", "1")
        else:
            monkeypatch.delenv("PIPS_HIP_SWEEP_LAUNCHES", raising=False)
        bt = pa.LeafBatch(N, S)
        Ks = []
        for b in range(N):
            W, T, F = blocks[b]
            K, dpos = pa.kkt_leaf_assemble(n_i, W)
            K.val[dpos] = diag[b]
            Ks.append(K)
            bt.set_block(b, K, n_i, pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F))
        bt.analyze(4)
        for b in range(N):
            bt.set_values(b, Ks[b].val)
        bt.factor()
        assert bt.info()["m"] > 4 * 128
        x = torch.tensor(rhs, device="cuda")
        bt.solve(x)
        bt.sync()
        got.append(x.cpu().numpy())
        bt.close()
    assert np.array_equal(got[0], got[1])
    x = got[1].reshape(N, -1)
    for b in range(N):
        K = sp.csr_matrix((Ks[b].val, Ks[b].colidx, Ks[b].rowptr), shape=(Ks[b].nrows, Ks[b].ncols))
        Kf = K + sp.tril(K, -1).T
        r = Kf @ x[b] - rhs.reshape(N, -1)[b]
        assert np.abs(r).max() / np.abs(rhs).max() < 1e-8


@pytest.mark.parametrize("n", [700, 3000])
def test_row_sweeps_dense_root(n, monkeypatch):
    rng = np.random.default_rng(n)
    n_primal = n // 2
    B = rng.standard_normal((n - n_primal, n_primal))
    A = np.zeros((n, n))
    A[:n_primal, :n_primal] = np.diag(10 ** rng.uniform(-1, 2, n_primal))
    A[n_primal:, :n_primal] = B
    A[n_primal:, n_primal:] = -np.diag(10 ** rng.uniform(-3, 0, n - n_primal))
    A = np.tril(A) + np.tril(A, -1).T
    rhs = rng.standard_normal(n)
    got = []
    for launches in (True, False):
        if launches:
            monkeypatch.setenv("PIPS_HIP_SWEEP_LAUNCHES", "1")
        else:
            monkeypatch.delenv("PIPS_HIP_SWEEP_LAUNCHES", raising=False)
        s = pa.HipDenseLdlSolver(n, n_primal)
        s.matrixChanged(np.tril(A))
        xs = [s.solve((rhs * (k + 1)).copy()) for k in range(2)]
        assert s.get_inertia() == (n_primal, n - n_primal, 0)
        got.append(xs)
        s.close()
    for k in range(2):
        assert np.array_equal(got[0][k], got[1][k])
    assert np.linalg.norm(A @ got[1][0] - rhs) / np.linalg.norm(rhs) < 1e-10


@pytest.mark.parametrize("nrhs", [2, 7])
def test_row_sweeps_several_right_hand_sides(nrhs, monkeypatch):
    """DoubleLinearSolver::solve(nrhs, ...) with the per-right-hand-side scheme (grid.y = right-hand side): every right-hand side has
    its own tickets and flags; agrees with the launch-per-column kernels, and right-hand side r with the single solve of r."""
    monkeypatch.setenv("PIPS_HIP_MULTI", "0")        # the scheme that runs the single-vector kernels with a grid over the right-hand sides
    prob = Problem(31, 1, 3000, 1500, 0, 0, 0.004)
    K = prob.blocks[0]["K"]
    rng = np.random.default_rng(2)
    rhs = rng.standard_normal((nrhs, prob.n_leaf))
    got = []
    for launches in (True, False):
        if launches:
            monkeypatch.setenv("PIPS_HIP_SWEEP_LAUNCHES", "1")
        else:
            monkeypatch.delenv("PIPS_HIP_SWEEP_LAUNCHES", raising=False)
        s = pa.HipLdlSolver(K, prob.n_i)
        s.analyze()
        s.matrixChanged()
        x = s.solve(rhs.copy())
        one = s.solve(rhs[nrhs - 1].copy())
        got.append((x, one))
        s.close()
    # (the sparse head adds with atomics and the deterministic mode takes one right-hand side at a time: agreement to rounding)
    scale = np.abs(got[0][0]).max()
    assert np.abs(got[0][0] - got[1][0]).max() <= 1e-11 * scale
    assert np.abs(got[1][0][nrhs - 1] - got[1][1]).max() <= 1e-11 * scale
    Kf = prob.K_full(0)
    for r in range(nrhs):
        assert np.abs(Kf @ got[1][0][r] - rhs[r]).max() / np.abs(rhs[r]).max() < 1e-9


def test_sweep_that_gives_up_is_reported(monkeypatch):
    """A wait inside the single-launch sweeps that exceeds its poll limit poisons its output with NaN and raises an error word; the
    host must hear of it at its next synchronisation point instead of handing NaN back with PIPS_OK.  With a poll limit of zero
    every wait that is not satisfied at once gives up."""
    monkeypatch.setenv("PIPS_HIP_SWEEP_POLL_LIMIT", "0")
    rng = np.random.default_rng(0)
    n = 1500                                   # 12 tile columns: later rows wait for earlier ones
    M = rng.standard_normal((n, n))
    M = M @ M.T + n * np.eye(n)
    s = pa.HipDenseLdlSolver(n, n_primal=n)
    s.matrixChanged(np.ascontiguousarray(np.tril(M)))
    x = rng.standard_normal(n)
    with pytest.raises(pa.PipsHipError, match="gave up waiting"):
        for _ in range(20):                    # (a lucky schedule can satisfy every wait once)
            s.solve(x.copy())
    monkeypatch.delenv("PIPS_HIP_SWEEP_POLL_LIMIT")
    s2 = pa.HipDenseLdlSolver(n, n_primal=n)
    s2.matrixChanged(np.ascontiguousarray(np.tril(M)))
    y = x.copy()
    s2.solve(y)
    assert np.linalg.norm(M @ y - x) / np.linalg.norm(x) < 1e-12

