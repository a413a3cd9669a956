"""The dense tails of the leaf blocks as ONE dependency-driven launch (csrc/tailkernel.hip.h; the default for batches of up to 16 blocks,
PIPS_HIP_TAIL_SINGLE=0 / 1 forces a side): the same factors, Schur contribution and inertia as the launch-per-step driver -
with a tile envelope (all of K in the tail: the primal tile rows couple to nothing left of their diagonal, their trsm finish in any order),
with padded tile rows, over several blocks and over repeated factorisations."""
import numpy as np
import pytest

import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu


def _schur(prob, cut, monkeypatch, single):
    import torch
    if single:
        monkeypatch.setenv("PIPS_HIP_TAIL_SINGLE", "1")
    else:
        monkeypatch.setenv("PIPS_HIP_TAIL_SINGLE", "0")
    S = prob.S
    bt = pa.LeafBatch(prob.N, S)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.set_options(force_n_head={"model": -1, "all_tail": 0}[cut])
    bt.analyze(4)
    out = []
    for rep in range(2):                      # (the flags start from their template again)
        for b in range(prob.N):
            bt.set_values(b, prob.blocks[b]["K"].val)
        SC = torch.zeros(S * S, dtype=torch.float64, device="cuda")
        bt.factor(SC, S)
        bt.sync()
        out.append(hip_lower_as_rowmajor(SC.cpu().numpy(), S))
        assert [bt.inertia(b) for b in range(prob.N)] == [(prob.n_i, prob.my_i, 0)] * prob.N
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(prob.N * prob.n_leaf)
    x = rhs.copy()
    bt.solve(x)
    bt.close() if hasattr(bt, "close") else None
    return out, rhs, x


@pytest.mark.parametrize("cut", ["model", "all_tail"])
@pytest.mark.parametrize("shape", [(3, 700, 40, 30, 0.02), (4, 1000, 100, 100, 0.01)])
def test_one_launch_matches_the_column_launches(shape, cut, monkeypatch):
    N, n_i, n0, myl, rho = shape
    prob = Problem(33, N, n_i, n_i // 2, n0, myl, rho)
    want = np.tril(prob.oracle_schur())
    scale = np.abs(want).max()
    (sc_a, sc_a2), rhs, xa = _schur(prob, cut, monkeypatch, True)
    (sc_b, _), _, xb = _schur(prob, cut, monkeypatch, False)
    assert np.abs(sc_a - want).max() / scale < 1e-9 and np.abs(sc_a2 - want).max() / scale < 1e-9
    assert np.abs(sc_a - sc_b).max() / scale < 1e-11
    for b in range(N):
        Kf = prob.K_full(b)
        sl = slice(b * prob.n_leaf, (b + 1) * prob.n_leaf)
        assert np.linalg.norm(Kf @ xa[sl] - rhs[sl]) / np.linalg.norm(rhs[sl]) < 1e-10
    assert np.linalg.norm(xa - xb) / np.linalg.norm(xb) < 1e-9


@pytest.mark.parametrize("case", range(8))
def test_random_shapes_both_sides_agree(case, monkeypatch):
    """random batch sizes, block sizes, fills and cuts: one launch against the column launches"""
    rng = np.random.default_rng(4000 + case)
    N = int(rng.integers(1, 6))
    n_i = int(rng.choice([640, 900, 1300, 1800, 2300]))
    n0, myl = int(rng.integers(3, 120)), int(rng.integers(0, 90))
    prob = Problem(600 + case, N, n_i, n_i // 2, n0, myl, float(rng.choice([3.0, 6.0, 10.0])) / n_i)
    cut = str(rng.choice(["model", "all_tail"]))
    (sc_a, _), rhs, xa = _schur(prob, cut, monkeypatch, True)
    (sc_b, _), _, xb = _schur(prob, cut, monkeypatch, False)
    scale = max(np.abs(sc_b).max(), 1e-300)
    assert np.abs(sc_a - sc_b).max() / scale < 1e-10
    assert np.linalg.norm(xa - xb) / np.linalg.norm(xb) < 1e-8
