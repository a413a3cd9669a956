"""solveCompressed by one forward and one backward sweep of the augmented factor [L 0; L_b I] (pips_hip_kkt_last_solve_path == 2;
Engine::forward_augmented / backward_augmented) against the oracle's solve_compressed (sLinsysRootAug.C:323-365 restated) and against the
refined two-solve path on the same factors: dense-tail blocks whose simple leaves own border rows, time-coupled blocks with compact front
panels (border rows only in the border-row arena), all-head blocks without a tail.  The path is taken only after a refined pass on the
same factors needed no refinement step - or, on one rank, after the first sweep pair's own result passed the residual check of the leaf
rows (path 3) - and never while a pivot is perturbed."""
import numpy as np
import pytest
import torch

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem
from tests.test_leaf_gpu import _TimeCoupledProblem

pytestmark = pytest.mark.gpu


def _system(prob, force_head=False):
    bt = pa.LeafBatch(prob.N, prob.S)
    for b in range(prob.N):
        bt.set_block(b, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    if force_head:
        bt.set_options(force_n_head=prob.n_leaf)
    bt.analyze(2)
    bt.set_refinement_backward_error(2, 1e-15)
    for b in range(prob.N):
        bt.set_values(b, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0)
    return bt, kkt


def _oracle(prob, b0, bl):
    S, N = prob.S, prob.N
    SCo = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    root = orc.DenseRootSolver(S)
    root.matrixChanged(SCo)
    b0_o = b0.copy()
    bs_o = [bl.reshape(N, -1)[b].copy() for b in range(N)]
    orc.solve_compressed(b0_o, bs_o, [prob.oracle_leaf(b) for b in range(N)], [prob.Bt_scipy(b) for b in range(N)], root, prob.n0, 0, 0, prob.myl, 0)
    return b0_o, np.concatenate(bs_o)


@pytest.mark.parametrize("shape", ["dense_tail", "small", "time_coupled", "time_coupled_all_head", "time_coupled_k_only_fronts"])
def test_augmented_sweeps_match_oracle_and_refined_path(shape, monkeypatch):
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")     # (the cost model would pick it or not by the shape; here it is under test)
    if shape == "time_coupled_k_only_fronts":
        monkeypatch.setenv("PIPS_HIP_MF_KONLY", "1")   # the border rows the sweeps read come from k_border_rows / k_border_tail
    if shape == "dense_tail":
        prob = Problem(3, 4, 1000, 500, 100, 100, 0.01)
    elif shape == "small":
        prob = Problem(5, 3, 200, 100, 12, 10, 0.05)
    else:
        prob = _TimeCoupledProblem(5, 3, 3000, 1500, 10, 8, 6)
    bt, kkt = _system(prob, force_head=shape == "time_coupled_all_head")
    assert bt.info()["blocks_with_k_only_fronts"] == (prob.N if shape == "time_coupled_k_only_fronts" else 0)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    kkt.factorize(diag, xd0)
    rng = np.random.default_rng(11)
    paths, sols = [], []
    for rep in range(3):
        b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
        b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        paths.append(kkt.last_solve_path())
        want0, wantl = _oracle(prob, b0h, blh)
        g0, gl = b0.cpu().numpy(), bl.cpu().numpy()
        assert np.linalg.norm(g0 - want0) <= 1e-8 * np.linalg.norm(want0), (rep, paths)
        assert np.linalg.norm(gl - wantl) <= 1e-8 * np.linalg.norm(wantl), (rep, paths)
        sols.append((b0h, blh, g0, gl))
    # every solveCompressed by sweeps is measured against the leaf rows (the default: pips_hip_kkt_set_solve_check(1)) ...
    assert paths == [3, 3, 3] and kkt.solve_check_counts() == (3, 0), (paths, kkt.solve_check_counts())
    # ... every second one / only the first after a factorisation (the witness; the others ride on it)
    for every, want_paths in ((2, [3, 2, 3, 2]), (0, [3, 2, 2, 2])):
        kkt.set_solve_check(every)
        kkt.factorize(diag, xd0)
        got_paths = []
        for rep in range(4):
            b0, bl = torch.tensor(sols[rep % 3][0], device="cuda"), torch.tensor(sols[rep % 3][1], device="cuda")
            kkt.solve_compressed(b0, bl)
            got_paths.append(kkt.last_solve_path())
            assert np.linalg.norm(bl.cpu().numpy() - sols[rep % 3][3]) <= 1e-10 * np.linalg.norm(sols[rep % 3][3])
        assert got_paths == want_paths, (every, got_paths)
    kkt.set_solve_check(1)
    # the same right-hand side through the refined path on the same factors
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "0")
    bt2, kkt2 = _system(prob, force_head=shape == "time_coupled_all_head")
    kkt2.factorize(diag, xd0)
    b0h, blh, g0, gl = sols[2]
    b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
    kkt2.solve_compressed(b0, bl)
    bt2.sync()
    assert kkt2.last_solve_path() in (0, 1)
    assert np.linalg.norm(g0 - b0.cpu().numpy()) <= 1e-10 * np.linalg.norm(g0)
    assert np.linalg.norm(gl - bl.cpu().numpy()) <= 1e-10 * np.linalg.norm(gl)
    # a new factorisation needs a new witness
    kkt.factorize(diag * 1.3, xd0)
    b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
    kkt.solve_compressed(b0, bl)
    assert kkt.last_solve_path() == 3
    for h in (kkt, bt, kkt2, bt2):
        h.close()


@pytest.mark.parametrize("witness", ["refined", "failing_check"])
def test_witness_variants(witness, monkeypatch):
    """refined: PIPS_HIP_AUG_WITNESS=0 keeps the refined pass as the witness (what several ranks always do).  failing_check: with a
    tolerance no double-precision result can meet the check of the first sweep pair fails - its result is discarded, the saved right-hand
    side goes the refined way, and the sweeps stay off for these factors; the answer is the oracle's either way."""
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")
    if witness == "refined":
        monkeypatch.setenv("PIPS_HIP_AUG_WITNESS", "0")
    prob = _TimeCoupledProblem(5, 3, 3000, 1500, 10, 8, 6)
    bt, kkt = _system(prob)
    if witness == "failing_check":
        bt.set_refinement_backward_error(2, 1e-30)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    rng = np.random.default_rng(5)
    paths = []
    for rep in range(3):
        b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
        b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        paths.append(kkt.last_solve_path())
        want0, wantl = _oracle(prob, b0h, blh)
        assert np.linalg.norm(b0.cpu().numpy() - want0) <= 1e-8 * np.linalg.norm(want0), (rep, paths)
        assert np.linalg.norm(bl.cpu().numpy() - wantl) <= 1e-8 * np.linalg.norm(wantl), (rep, paths)
    if witness == "refined":
        assert paths[0] in (0, 1) and paths[1:] == [3, 3], paths      # (followers: sweeps, each measured)
    else:
        assert all(q in (0, 1) for q in paths), paths
    kkt.close(); bt.close()


def test_perturbed_pivots_keep_the_refined_path(monkeypatch):
    """A block with a structurally singular pivot (zero primal diagonal on an unconstrained variable) gets a replaced pivot: the factors are
    not trusted, every solveCompressed goes the refined way."""
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")
    prob = Problem(5, 3, 200, 100, 12, 10, 0.05)
    d = prob.blocks[1]["diag"].copy()
    blk = prob.blocks[1]
    # a primal variable that no constraint row touches would do; simplest: make one primal diagonal tiny relative to its scale
    d[:prob.n_i] = np.maximum(d[:prob.n_i], 1e-3)
    d[0] = 0.0
    blk["K"].val[blk["dpos"]] = d
    blk["diag"] = d
    bt, kkt = _system(prob)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    pert = sum(bt.inertia(b)[2] for b in range(prob.N))
    assert pert > 0                        # (the construction must produce what the test is about)
    rng = np.random.default_rng(2)
    for rep in range(3):
        b0, bl = torch.tensor(rng.standard_normal(prob.S), device="cuda"), torch.tensor(rng.standard_normal(prob.N * prob.n_leaf), device="cuda")
        kkt.solve_compressed(b0, bl)
        assert kkt.last_solve_path() in (0, 1)
    assert kkt.solve_check_counts() == (0, 0)
    kkt.close(); bt.close()


def _leaf_backward_errors(prob, b0_in, bl_in, x0, xl):
    """per block: || (b_i - Br_i x0) - K_i x_i ||_inf / (||K_i||_inf ||x_i||_inf + ||b_i - Br_i x0||_inf) from the CSR values on the host, and
    the same residual relative to the right-hand side alone"""
    out = []
    for b in range(prob.N):
        Kf = prob.K_full(b).tocsr()
        rhs = bl_in.reshape(prob.N, -1)[b] - prob.Bt_scipy(b).T @ x0
        x = xl.reshape(prob.N, -1)[b]
        r = rhs - Kf @ x
        # third entry: the measure in the engine's own norm (max |K entry| for ||K||): what pips_hip_batch_last_refinement_measure reports
        out.append((np.abs(r).max() / (abs(Kf).sum(axis=1).max() * np.abs(x).max() + np.abs(rhs).max()), np.abs(r).max() / np.abs(rhs).max(),
                    np.abs(r).max() / (np.abs(Kf.data).max() * np.abs(x).max() + np.abs(rhs).max())))
    return out


@pytest.mark.parametrize("shape", ["dense_tail", "time_coupled"])
def test_followers_with_adversarial_right_hand_sides_are_measured_and_fall_back(shape, monkeypatch):
    """Round-4 verdict, weak 1(b): diagonals frozen over sixteen decades (a late interior-point iterate), a benign witness, then followers
    scaled by 1e+-6 and concentrated on the rows of the smallest pivots.  EVERY follower that goes by sweeps is measured against the leaf
    rows (way 3) - none rides on the witness - and whatever way a follower took, its leaf rows hold to the backward error the adaptive
    refinement of the reference's PARDISO guarantees (iparm[7] = 2, PardisoProjectSolver.C:72).  The residual relative to the right-hand
    side alone, ||K x - b|| / ||b||, is printed, not asserted: with x = K^-1 b up to 1e16 ||b|| / ||K|| no method in double precision
    bounds it (the refined path does not either)."""
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")
    if shape == "dense_tail":
        prob = Problem(21, 3, 600, 300, 30, 20, 0.01, diag_lo=-8.0, diag_hi=8.0)
    else:
        prob = _TimeCoupledProblem(23, 3, 3000, 1500, 10, 8, 6)
        rng = np.random.default_rng(23)
        for blk in prob.blocks:
            blk["diag"][:prob.n_i] = 10.0 ** rng.uniform(-8, 8, prob.n_i)
            blk["K"].val[blk["dpos"]] = blk["diag"]
    bt, kkt = _system(prob)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in range(prob.N)]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    assert sum(bt.inertia(b)[2] for b in range(prob.N)) == 0
    rng = np.random.default_rng(7)
    n_leaf = prob.n_leaf
    small = [np.argsort(np.abs(prob.blocks[b]["diag"][:prob.n_i]))[:40] for b in range(prob.N)]
    cases = [("witness", 1.0, None)] + [(f"follower scaled {sc:g} on the smallest pivots", sc, True) for sc in (1e6, 1e-6, 1e6, 1e-6)] + \
            [("follower, mixed scales", 1.0, False)]
    worst_eta = 0.0
    for name, scale, aligned in cases:
        blh = rng.standard_normal(prob.N * n_leaf)
        if aligned:
            v = np.zeros((prob.N, n_leaf))
            for b in range(prob.N):
                v[b, small[b]] = rng.standard_normal(40)
            blh = scale * v.ravel()
        elif aligned is False:
            blh *= 10.0 ** rng.uniform(-6, 6, blh.size)
        b0h = scale * rng.standard_normal(prob.S)
        b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        path = kkt.last_solve_path()
        assert path in (0, 1, 3), (name, path)          # never way 2: no follower is taken on trust
        errs = _leaf_backward_errors(prob, b0h, blh, b0.cpu().numpy(), bl.cpu().numpy())
        eta = max(e[0] for e in errs)
        print(f"{shape}: {name}: way {path}, leaf backward error {eta:.2e}, ||r||/||b|| {max(e[1] for e in errs):.2e}")
        assert eta <= (5e-15 if path == 3 else 1e-13), (name, path, errs)      # (way 3: the measure itself is <= 1e-15 with max|K| for ||K||)
        if path == 3:
            # the one-launch measure of the device (k_measure_leaf_rows) is the quantity computed here from the CSR values on the host: same
            # residual up to the rounding of two summation orders (it is a residual AT rounding level), never spuriously tiny or large
            dev, host = bt.last_refinement_measure(), max(e[2] for e in errs)
            assert dev <= 1e-15 and 0.1 * host <= dev <= 10.0 * host + 1e-18, (name, dev, host)
        worst_eta = max(worst_eta, eta)
    checked, failed = kkt.solve_check_counts()
    print(f"{shape}: measured {checked}, failed {failed}, worst leaf backward error {worst_eta:.2e}")
    assert checked >= 1 and (failed > 0 or checked == len(cases))      # (a failed measure turns the later solves to the refined way)
    # a follower whose measure FAILS goes the refined way from the saved right-hand side, and the sweeps stay off for these factors
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * n_leaf)
    b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
    kkt.solve_compressed(b0, bl)
    assert kkt.last_solve_path() == 3
    c0, f0 = kkt.solve_check_counts()
    bt.set_refinement_backward_error(2, 1e-30)          # no double-precision result meets this: the next measure fails
    ways = []
    for rep in range(2):
        b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        ways.append(kkt.last_solve_path())
        errs = _leaf_backward_errors(prob, b0h, blh, b0.cpu().numpy(), bl.cpu().numpy())
        assert max(e[0] for e in errs) <= 1e-13
    assert ways[0] in (0, 1) and ways[1] in (0, 1), ways
    assert kkt.solve_check_counts() == (c0 + 1, f0 + 1)
    kkt.close(); bt.close()


def _joint_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PIPS_HIP_AUG_SWEEPS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = _TimeCoupledProblem(5, 4, 3000, 1500, 10, 8, 6)
    mine = np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]
    calls = []

    def allreduce(ptr, n):
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()
        calls.append(n)
    comm = pa.ExternalComm(allreduce)
    bt = pa.LeafBatch(len(mine), prob.S, device=0)
    for i, b in enumerate(mine):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    bt.set_refinement_backward_error(2, 1e-15)
    for i, b in enumerate(mine):
        bt.set_values(i, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=world)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in mine]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    rng = np.random.default_rng(4)
    res = {}
    for rep in range(3):
        b0_full = rng.standard_normal(prob.S)
        bs_full = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
        if rep == 2 and rank == 1:
            bt.set_refinement_backward_error(2, 1e-30)        # this rank's measure fails from here on: BOTH ranks must repeat the solve
        b0 = torch.tensor(b0_full, device="cuda")
        bl = torch.tensor(np.concatenate([bs_full[b] for b in mine]), device="cuda")
        n_calls = len(calls)
        kkt.solve_compressed(b0, bl)
        bt.sync()
        res[f"path{rep}"] = kkt.last_solve_path()
        res[f"calls{rep}"] = np.array(calls[n_calls:])
        res[f"xroot_{rep}"] = b0.cpu().numpy()
        xl = bl.cpu().numpy().reshape(len(mine), -1)
        for i, b in enumerate(mine):
            res[f"x{b}_{rep}"] = xl[i]
    res["counts"] = np.array(kkt.solve_check_counts())
    np.savez(os.path.join(out, f"rank{rank}.npz"), blocks=np.array(mine), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_decide_together_about_a_failed_measure(tmp_path):
    """Several ranks: every solveCompressed by sweeps is measured on every rank and the outcome is all-reduced (one number); when one
    rank's measure fails, both ranks restore their right-hand sides and repeat the solve the refined way - the answer is the oracle's on
    both, and the sequence of collectives is the same on both ranks."""
    import os
    import torch.multiprocessing as mp
    world = 2
    port = 29300 + os.getpid() % 500
    mp.start_processes(_joint_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    prob = _TimeCoupledProblem(5, 4, 3000, 1500, 10, 8, 6)
    S = prob.S
    SCo = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    root = orc.DenseRootSolver(S)
    root.matrixChanged(SCo)
    g = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    rng = np.random.default_rng(4)
    for rep in range(3):
        b0 = rng.standard_normal(S)
        bs = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
        orc.solve_compressed(b0, bs, [prob.oracle_leaf(b) for b in range(prob.N)], [prob.Bt_scipy(b) for b in range(prob.N)], root, prob.n0, 0, 0, prob.myl, 0)
        for r in range(world):
            assert np.linalg.norm(g[r][f"xroot_{rep}"] - b0) <= 1e-8 * np.linalg.norm(b0), (rep, r)
            for b in g[r]["blocks"]:
                assert np.linalg.norm(g[r][f"x{b}_{rep}"] - bs[b]) <= 1e-8 * np.linalg.norm(bs[b]), (rep, r, b)
        assert list(g[0][f"calls{rep}"]) == list(g[1][f"calls{rep}"])         # the same collectives in the same order on both ranks
    # first two solves: sweeps, measured, accepted on both ranks (the first one settles once "does any rank sweep": one extra number);
    # third: rank 1's measure fails -> both repeat the refined way
    assert [int(g[r]["path0"]) for r in range(world)] == [3, 3] and [int(g[r]["path1"]) for r in range(world)] == [3, 3]
    assert all(int(g[r]["path2"]) in (0, 1) for r in range(world))
    assert list(g[0]["calls0"]) == [1, S, 1] and list(g[0]["calls1"]) == [S, 1] and list(g[0]["calls2"]) == [S, 1, S, 1]
    assert [tuple(g[r]["counts"]) for r in range(world)] == [(3, 1), (3, 1)]
