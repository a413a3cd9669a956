"""Seeded sweep over shapes, sparsity patterns and analysis settings: Schur complement, solves and inertia of the batched
leaf path against the oracle.  Every case is small enough for the CPU oracle; the settings cover the corners the fixed
tests do not combine (forced cuts x Schur mode x amalgamation x dissection x spine x multi-RHS scheme x front rows)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu


def _banded_W(rng, my_i, n_i, bw):
    rows, cols = [], []
    for r in range(my_i):
        center = int(r * n_i / my_i)
        cs = np.union1d(np.clip(center + rng.integers(-bw, bw + 1, 4), 0, n_i - 1), [center])
        rows += [r] * len(cs)
        cols += list(cs)
    W = sp.csr_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(my_i, n_i))
    W.sum_duplicates()
    W.sort_indices()
    return pa.Csr(my_i, n_i, W.indptr, W.indices, W.data)


@pytest.mark.parametrize("case", range(int(os.environ.get("PIPS_FUZZ_CASES", "40"))))
def test_random_configuration(case, monkeypatch):
    import torch
    rng = np.random.default_rng(1000 + case)
    N = int(rng.integers(1, 4))
    n_i = int(rng.choice([37, 90, 160, 333, 520]))
    my_i = max(1, int(n_i * rng.choice([0.25, 0.5, 0.8])))
    n0, myl = int(rng.integers(0, 9)), int(rng.integers(0, 9))
    if n0 + myl == 0:
        n0 = 3
    rho = float(rng.choice([2.0, 5.0, 12.0])) / n_i
    structured = bool(case % 3 == 0)
    monkeypatch.setenv("PIPS_HIP_RELAX_ZEROS", str(rng.choice([0.0, 0.4, 0.7])))
    monkeypatch.setenv("PIPS_HIP_SPINE", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("PIPS_HIP_MULTI", str(int(rng.integers(0, 2))))
    # (drawn from a generator of its own so that the cases above keep their shapes) fronts on the rows of K only where the border split applies
    monkeypatch.setenv("PIPS_HIP_MF_KONLY", str(int(np.random.default_rng(7000 + case).integers(0, 2))))
    # (likewise) the tails as the column launches on every third case: batches this small take the single launch by default
    if np.random.default_rng(9000 + case).integers(0, 3) == 0:
        monkeypatch.setenv("PIPS_HIP_TAIL_SINGLE", "0")
    prob = Problem(500 + case, N, n_i, my_i, n0, myl, rho, diag_lo=float(rng.choice([-2, -4])), diag_hi=float(rng.choice([2, 4])))
    if structured:
        for blk in prob.blocks:
            Wp = _banded_W(rng, my_i, n_i, int(rng.integers(2, 9)))
            K, dpos = pa.kkt_leaf_assemble(n_i, Wp)
            K.val[dpos] = blk["diag"]
            blk.update(W=Wp, K=K, dpos=dpos)
    S = prob.S
    bt = pa.LeafBatch(N, S)
    bt.set_schur_mode(int(rng.integers(0, 3)))
    cut = rng.choice(["model", "all_head", "all_tail", "half"])
    force = {"model": -1, "all_head": prob.n_leaf, "all_tail": 0, "half": prob.n_leaf // 2}[cut]
    bt.set_options(force_n_head=force)
    for b in range(N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for b in range(N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
    bt.factor(SC, S)
    bt.sync()
    ctx = (case, N, n_i, my_i, n0, myl, cut, structured, bt.schur_mode(), bt.info())
    got = hip_lower_as_rowmajor(SC.cpu().numpy(), S)
    want = np.tril(prob.oracle_schur())
    # Accuracy note (tools/fuzz_debug.py, docs/HISTORY_r1_r2.md section 6): Schur mode 1 forms SC from the factors without iterative
    # refinement, so its forward error is cond(K_i) * eps (1e-7 for a block of cond 3e10 in this sweep); the oracle's
    # solve-based K4-K6 refines every column and stays at 1e-14.  The fixed tests (well-posed shapes) hold 1e-9.
    assert np.abs(got - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-300), ctx
    rhs = rng.standard_normal(N * prob.n_leaf)
    x = rhs.copy()
    bt.solve(x)
    for b in range(N):
        assert bt.inertia(b) == (prob.n_i, prob.my_i, 0), ctx
        r = rhs.reshape(N, -1)[b]
        assert np.linalg.norm(prob.K_full(b) @ x.reshape(N, -1)[b] - r) <= 1e-9 * np.linalg.norm(r), ctx
    # drop-in handle, several right-hand sides at once (both device schemes are drawn above)
    s0 = pa.HipLdlSolver(prob.blocks[0]["K"], n_primal=prob.n_i)
    s0.matrixChanged()
    X = rng.standard_normal((int(rng.integers(2, 12)), prob.n_leaf))
    R = X.copy()
    s0.solve(X)
    for k in range(X.shape[0]):
        assert np.linalg.norm(prob.K_full(0) @ X[k] - R[k]) <= 1e-9 * np.linalg.norm(R[k]), ctx
