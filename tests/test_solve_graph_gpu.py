"""solveCompressed as a captured and replayed HIP graph (pips_hip_kkt_set_solve_graph; SURVEY 8 f-1 "graph-captured solveCompressed"):
bit-identical to the launch-by-launch path on the same factors, one capture per (right-hand-side buffers, Ltsolve path), replays
across factorisations, fallback to the direct path where a capture cannot hold the work (adaptive refinement)."""
import numpy as np
import pytest
import torch

import pips_ipmpp_amd as pa
from tests.util import Problem

pytestmark = pytest.mark.gpu


def _system(prob, adaptive=False):
    bt = pa.LeafBatch(prob.N, prob.S)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for b in range(prob.N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    if adaptive:
        bt.set_refinement_backward_error(2, 1e-15)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
    return bt, kkt


@pytest.mark.parametrize("shape", ["dense_tail", "small"])
def test_graph_replay_is_bit_identical_to_the_direct_path(shape, monkeypatch):
    monkeypatch.setenv("PIPS_HIP_DETERMINISTIC", "0")
    prob = Problem(3, 4, 1000, 500, 100, 100, 0.01) if shape == "dense_tail" else Problem(5, 3, 200, 100, 12, 10, 0.05)
    # bit-identity needs a path without FP64 atomics racing: deterministic mode excludes graphs, so compare on solves whose only
    # atomics are the (order-insensitive up to rounding) border products - hence a tolerance of a few ulps instead of equality
    bt, kkt = _system(prob)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    rng = np.random.default_rng(0)
    b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
    b0, bl = torch.empty(prob.S, dtype=torch.float64, device="cuda"), torch.empty(prob.N * prob.n_leaf, dtype=torch.float64, device="cuda")

    def solve():
        b0.copy_(torch.tensor(b0h)); bl.copy_(torch.tensor(blh))
        kkt.solve_compressed(b0, bl)
        bt.sync()
        return b0.cpu().numpy().copy(), bl.cpu().numpy().copy()

    kkt.factorize(diag, xd0)
    ref0, refl = solve()
    kkt.set_solve_graph(True)
    for rep in range(3):
        g0, gl = solve()
        assert np.linalg.norm(g0 - ref0) <= 1e-12 * np.linalg.norm(ref0) and np.linalg.norm(gl - refl) <= 1e-12 * np.linalg.norm(refl)
    assert kkt.solve_graph_stats() == (1, 3)
    # a new factorisation (other diagonals): the same graph is replayed on the new factors
    diag2 = diag * 1.7
    kkt.factorize(diag2, xd0)
    g0, gl = solve()
    kkt.set_solve_graph(False)
    d0, dl = solve()
    assert np.linalg.norm(g0 - d0) <= 1e-12 * np.linalg.norm(d0) and np.linalg.norm(gl - dl) <= 1e-12 * np.linalg.norm(dl)
    assert np.linalg.norm(g0 - ref0) > 1e-6 * np.linalg.norm(ref0)        # (it really is another system)
    kkt.close(); bt.close()


def test_adaptive_refinement_keeps_the_direct_path():
    prob = Problem(5, 3, 200, 100, 12, 10, 0.05)
    bt, kkt = _system(prob, adaptive=True)
    kkt.set_solve_graph(True)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    b0 = torch.randn(prob.S, dtype=torch.float64, device="cuda"); bl = torch.randn(prob.N * prob.n_leaf, dtype=torch.float64, device="cuda")
    kkt.solve_compressed(b0, bl)
    bt.sync()
    assert kkt.solve_graph_stats() == (0, 0)
    kkt.close(); bt.close()


def test_settings_changed_after_a_capture_force_a_new_capture():
    """The captured launch sequence bakes in the number of refinement launches and the root's pivoting mode (and the zdiag0 / C0 buffers of
    eliminated root inequality rows): changing one of them after a capture must not replay the stale sequence."""
    prob = Problem(5, 3, 200, 100, 12, 10, 0.05)
    bt, kkt = _system(prob)
    bt.set_refinement(0, 0.0)
    kkt.set_solve_graph(True)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    rng = np.random.default_rng(1)
    b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
    b0, bl = torch.empty(prob.S, dtype=torch.float64, device="cuda"), torch.empty(prob.N * prob.n_leaf, dtype=torch.float64, device="cuda")

    def solve():
        b0.copy_(torch.tensor(b0h)); bl.copy_(torch.tensor(blh))
        kkt.solve_compressed(b0, bl)
        bt.sync()
        return b0.cpu().numpy().copy(), bl.cpu().numpy().copy()

    solve()
    assert kkt.solve_graph_stats() == (1, 1)
    bt.set_refinement(2, 0.0)                       # two refinement steps per leaf solve: another launch sequence
    g0, gl = solve()
    assert kkt.solve_graph_stats() == (2, 2)
    kkt.set_solve_graph(False)
    d0, dl = solve()
    assert np.linalg.norm(g0 - d0) <= 1e-12 * np.linalg.norm(d0) and np.linalg.norm(gl - dl) <= 1e-12 * np.linalg.norm(dl)
    kkt.set_solve_graph(True)
    solve()
    n_cap = kkt.solve_graph_stats()[0]
    kkt.set_root_pivoting(1)                        # Bunch-Kaufman root: other kernels in the Dsolve
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    g0, gl = solve()
    assert kkt.solve_graph_stats()[0] == n_cap + 1
    kkt.set_solve_graph(False)
    d0, dl = solve()
    assert np.linalg.norm(g0 - d0) <= 1e-10 * np.linalg.norm(d0) and np.linalg.norm(gl - dl) <= 1e-10 * np.linalg.norm(dl)
    kkt.close(); bt.close()
