"""Deterministic mode (pips_hip_batch_set_deterministic): no FP64 atomics anywhere on the path - the head scatter and the head
forward substitution go through contribution slots gathered in a fixed order, the Schur accumulation over the blocks through
group buffers added in a fixed tree, the border products through gather lists.  Asserted here: the Schur complement, every
inertia count and the solveCompressed result are BIT-identical over repeated factorisations, over two separately analysed
handles, and between one rank and 2 / 4 / 8 ranks (processes sharing the GPU, gloo all-reduce behind the external communicator).  The reference's
breakdown tests compare against 1e-40 (PIPSisZero, pipsdef.h:35,108; LinearSystem.C:640-785) - they are meaningful only when
the numbers are reproducible."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pips_ipmpp_amd as pa
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu


def _problem(kind):
    if kind == "banded":     # time-coupled blocks: three elimination-tree levels, head-to-head updates, 2-link borders
        from tests.test_sparse_root_gpu import TwoLinkProblem
        return TwoLinkProblem(93, 8, 2400, 1200, 5, 4, 5.0 / 2400)
    if kind.startswith("sparse"):   # the same structure with the sparse root: 112 linking rows (enough for the dissected root order)
        from tests.test_sparse_root_gpu import TwoLinkProblem
        return TwoLinkProblem(94, 8, 600, 300, 5, 16, 5.0 / 600)
    return Problem(7, 8, 600, 300, 30, 20, 0.02)


def _run(prob, mine, deterministic, comm=None, rank=0, world=1, reps=3, sparse=False):
    S = prob.S
    bt = pa.LeafBatch(len(mine), S)
    bt.set_deterministic(deterministic)
    for i, b in enumerate(mine):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for i, b in enumerate(mine):
        bt.set_values(i, prob.blocks[b]["K"].val)
    kw = {}
    if sparse:   # every rank needs the border column sets of all blocks (one value array is reduced)
        kw = dict(sparse_root=True, all_block_cols=[np.nonzero(np.diff(prob.blocks[b]["Bt"].rowptr) > 0)[0] for b in range(prob.N)])
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=world, **kw)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in mine]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    rng = np.random.default_rng(0)
    b0_full = rng.standard_normal(S)
    bs_full = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    out = []
    for _ in range(reps):
        kkt.factorize(diag, xd0)
        SC = kkt.schur_sparse_to_host().data.copy() if sparse else kkt.schur_to_host().copy()
        b0 = torch.tensor(b0_full, device="cuda")
        bl = torch.tensor(np.concatenate([bs_full[b] for b in mine]), device="cuda")
        kkt.solve_compressed(b0, bl)
        bt.sync()
        out.append(dict(SC=SC, x0=b0.cpu().numpy(), xl=bl.cpu().numpy().reshape(len(mine), -1),
                        inertia=[bt.inertia(i) for i in range(len(mine))] + [kkt.root_inertia()]))
    return out


_ROOT_ORDER = {"sparse_band": "1", "sparse_amd": "0", "sparse_dissected": "2"}


@pytest.mark.parametrize("kind", ["random", "banded", "sparse_band", "sparse_amd", "sparse_dissected"])
def test_bit_identical_over_runs_and_handles(kind, monkeypatch):
    """sparse_*: the Schur complement as a CSR value array (group buffers as long as that array), the root factorised by the leaf engine in
    deterministic mode too - in each of its three elimination orders."""
    sparse = kind.startswith("sparse")
    if sparse:
        monkeypatch.setenv("PIPS_HIP_SPARSE_ROOT_BAND", _ROOT_ORDER[kind])
    def _run_here(*a, **kw):
        return _run(*a, sparse=sparse, **kw)

    prob = _problem(kind)
    mine = list(range(prob.N))
    runs = _run_here(prob, mine, True) + _run_here(prob, mine, True)
    for r in runs[1:]:
        assert np.array_equal(r["SC"], runs[0]["SC"]) and np.array_equal(r["x0"], runs[0]["x0"]) and np.array_equal(r["xl"], runs[0]["xl"])
        assert r["inertia"] == runs[0]["inertia"]
    # ... and it is the same system the default (atomic) path solves
    ref = _run_here(prob, mine, False, reps=1)[0]
    assert np.abs(ref["SC"] - runs[0]["SC"]).max() <= 1e-9 * np.abs(ref["SC"]).max()
    assert np.linalg.norm(ref["xl"] - runs[0]["xl"]) <= 1e-8 * np.linalg.norm(ref["xl"])
    assert ref["inertia"] == runs[0]["inertia"]
    # the default path is NOT reproducible to the bit (that is what the mode is for); do not assert it - just report
    two = _run_here(prob, mine, False, reps=2)
    print(f"{kind}: default path, two factorisations: max |dSC| {np.abs(two[0]['SC'] - two[1]['SC']).max():.1e}")


def _worker(rank, world, port, out, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = _problem(kind)
    mine = [int(b) for b in np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]]

    def allreduce(ptr, n):
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()

    calls = []

    def all_gather(ptr, chunk):     # in place: rank r's part at r * chunk
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, chunk * world), device="cuda")
        h = t.cpu()
        parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, h[rank * chunk:(rank + 1) * chunk].clone())
        t.copy_(torch.cat(parts))
        torch.cuda.synchronize()
        calls.append(chunk)

    def reduce_scatter(ptr, chunk):
        allreduce(ptr, chunk * world)

    # a communicator WITH an all-gather (4 ranks): every rank's group slots travel once (pips_hip_all_gather) instead of an all-reduce in
    # which the other ranks hold zeros - the same bits at an eighth ... a quarter of the bytes; the others keep the all-reduce fallback
    comm = pa.ExternalComm(allreduce, reduce_scatter, all_gather, n_ranks=world, rank=rank) if world == 4 else pa.ExternalComm(allreduce)
    r = _run(prob, mine, True, comm=comm, rank=rank, world=world, reps=2, sparse=kind.startswith("sparse"))
    assert world != 4 or len(calls) >= 4        # two factorisations + two solveCompressed went through the all-gather
    assert np.array_equal(r[0]["SC"], r[1]["SC"]) and np.array_equal(r[0]["xl"], r[1]["xl"])
    np.savez(os.path.join(out, f"det{rank}.npz"), SC=r[0]["SC"], x0=r[0]["x0"], xl=r[0]["xl"], mine=np.array(mine), inertia=np.array(r[0]["inertia"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("random", 2), ("banded", 2), ("random", 4), ("banded", 4), ("random", 8), ("sparse_dissected", 2), ("sparse_band", 4),
                                        ("sparse_dissected", 8)])
def test_bit_identical_between_one_and_several_ranks(tmp_path, kind, world, monkeypatch):
    """2, 4 and 8 processes sharing the GPU: every rank's group buffers reach every rank and all eight group slots are added in one fixed
    tree (Engine::det_global), so the sums associate as on one rank - equal bits, not just reproducible ones."""
    if kind.startswith("sparse"):
        monkeypatch.setenv("PIPS_HIP_SPARSE_ROOT_BAND", _ROOT_ORDER[kind])   # (inherited by the spawned ranks)
    port = 29500 + (os.getpid() % 2000) + {"random": 41, "banded": 43}.get(kind, 47) + 3 * world
    mp.start_processes(_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True, start_method="spawn")
    prob = _problem(kind)
    one = _run(prob, list(range(prob.N)), True, reps=1, sparse=kind.startswith("sparse"))[0]
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"det{r}.npz"))
        assert np.array_equal(g["SC"], one["SC"])          # the reduced, finalised Schur complement: same bits on every rank
        assert np.array_equal(g["x0"], one["x0"])          # root part of solveCompressed
        for i, b in enumerate(g["mine"]):
            assert np.array_equal(g["xl"][i], one["xl"][b])
            assert tuple(g["inertia"][i]) == one["inertia"][b]
        assert tuple(g["inertia"][-1]) == one["inertia"][-1]


def _run_aug(prob, mine, deterministic, comm=None, rank=0, world=1, reps=2):
    """factorize + two solveCompressed with adaptive refinement on (the setting under which the sweeps of the augmented factor are taken and
    measured): results and the path of every solve"""
    S = prob.S
    bt = pa.LeafBatch(len(mine), S)
    bt.set_deterministic(deterministic)
    for i, b in enumerate(mine):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    bt.set_refinement_backward_error(2, 1e-15)
    for i, b in enumerate(mine):
        bt.set_values(i, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=world)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in mine]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    rng = np.random.default_rng(5)
    rhs = [(rng.standard_normal(S), [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]) for _ in range(2)]
    out = []
    for _ in range(reps):
        kkt.factorize(diag, xd0)
        res = dict(x0=[], xl=[], paths=[])
        for b0_full, bs_full in rhs:
            b0 = torch.tensor(b0_full, device="cuda")
            bl = torch.tensor(np.concatenate([bs_full[b] for b in mine]), device="cuda")
            kkt.solve_compressed(b0, bl)
            bt.sync()
            res["x0"].append(b0.cpu().numpy()); res["xl"].append(bl.cpu().numpy().reshape(len(mine), -1)); res["paths"].append(kkt.last_solve_path())
        res["aug"] = bt.info()["augmented_sweeps"]
        res["checks"] = kkt.solve_check_counts()
        out.append(res)
    kkt.close(); bt.close()
    return out


@pytest.mark.parametrize("kind", ["random", "banded"])
def test_augmented_sweeps_in_deterministic_mode(kind, monkeypatch):
    """Round 5: deterministic mode takes both halves of solveCompressed from the augmented factor too (Engine::forward_augmented_det: the head
    through the slot scheme, the blocks' border slots gathered target by target - k_border_gather_det - and group-wise into the root's
    right-hand side; the backward sweep is a gather anyway).  Bit-identical over runs and handles, measured like the default mode's (path 3),
    and the same solution as the default mode's sweeps and as deterministic mode's refined path."""
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")     # (the cost model decides by the shape; here the path is under test)
    prob = _problem(kind)
    mine = list(range(prob.N))
    runs = _run_aug(prob, mine, True) + _run_aug(prob, mine, True)
    assert runs[0]["aug"] == 1 and runs[0]["paths"] == [3, 3] and runs[0]["checks"][1] == 0, runs[0]
    for r in runs[1:]:
        assert r["paths"] == [3, 3]
        for q in range(2):
            assert np.array_equal(r["x0"][q], runs[0]["x0"][q]) and np.array_equal(r["xl"][q], runs[0]["xl"][q])
    ref = _run_aug(prob, mine, False, reps=1)[0]
    assert ref["paths"] == [3, 3]
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "0")
    refined = _run_aug(prob, mine, True, reps=1)[0]
    assert refined["aug"] == 0 and all(p in (0, 1) for p in refined["paths"])
    for other in (ref, refined):
        for q in range(2):
            assert np.linalg.norm(other["x0"][q] - runs[0]["x0"][q]) <= 1e-9 * np.linalg.norm(other["x0"][q])
            assert np.linalg.norm(other["xl"][q] - runs[0]["xl"][q]) <= 1e-9 * np.linalg.norm(other["xl"][q])


def _worker_aug(rank, world, port, out, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = _problem(kind)
    mine = [int(b) for b in np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]]

    def allreduce(ptr, n):
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()

    r = _run_aug(prob, mine, True, comm=pa.ExternalComm(allreduce), rank=rank, world=world, reps=1)[0]
    np.savez(os.path.join(out, f"aug{rank}.npz"), x0=np.array(r["x0"]), xl=np.array(r["xl"]), mine=np.array(mine), paths=np.array(r["paths"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("banded", 2), ("random", 4)])
def test_augmented_sweeps_in_deterministic_mode_over_ranks(tmp_path, kind, world, monkeypatch):
    """... and between one rank and 2 / 4 processes sharing the GPU: the border slots go through the same eight group slots as Br^T z."""
    monkeypatch.setenv("PIPS_HIP_AUG_SWEEPS", "1")
    port = 29500 + (os.getpid() % 2000) + 61 + 3 * world
    mp.start_processes(_worker_aug, args=(world, port, str(tmp_path), kind), nprocs=world, join=True, start_method="spawn")
    prob = _problem(kind)
    one = _run_aug(prob, list(range(prob.N)), True, reps=1)[0]
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"aug{r}.npz"))
        assert list(g["paths"]) == [3, 3] == one["paths"]
        for q in range(2):
            assert np.array_equal(g["x0"][q], one["x0"][q])
            for i, b in enumerate(g["mine"]):
                assert np.array_equal(g["xl"][q][i], one["xl"][q][b])


def test_ipm_is_bit_reproducible(monkeypatch):
    """With PIPS_HIP_DETERMINISTIC=1 (the default of pips_hip_batch_set_deterministic for batches the harness creates) the whole
    interior-point run - every iterate's mu, residual, objectives, step lengths, and the final point - repeats to the bit; the
    harness' own reductions are two-stage with a fixed order.  Instance: one of the reference's GAMSsmall LPs whose default-mode
    runs take 6 or 7 iterations depending on the order in which the atomics arrive."""
    import json
    monkeypatch.setenv("PIPS_HIP_DETERMINISTIC", "1")
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gamssmall.json")))["instances"]
    inst = [d for d in data if d["name"] == "singletonInequalityColumn_B0Bl0"][0]
    runs = []
    for _ in range(3):
        ipm = pa.GeneralIpmSolver(inst["blocks"], dual_reg=1e-9)
        res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
        assert res["status"] == 0 and abs(res["objective"] - inst["expected_objective"]) < 1e-4
        runs.append((res["iterations"], ipm.trace().tobytes(), ipm.iterate()["x"].tobytes(), ipm.iterate()["z"].tobytes()))
        ipm.close()
    assert all(r == runs[0] for r in runs)


@pytest.mark.parametrize("n_i", [300, 1500, 2400])
def test_several_right_hand_sides_in_deterministic_mode(n_i, monkeypatch):
    """DoubleLinearSolver::solve(nrhs) of a leaf handle in deterministic mode (pips_hip_ldl_set_deterministic): the interleaved panels with the slot / gather forward substitution instead of
    one right-hand side at a time - bit-identical over runs and handles, equal to the single solves of the same handle to rounding, and to
    SuperLU."""
    import scipy.sparse.linalg as spl
    prob = Problem(5, 1, n_i, n_i // 2, 4, 4, 6.0 / n_i)
    rng = np.random.default_rng(1)
    R = rng.standard_normal((70, prob.n_leaf))       # three panels, the last one partly filled
    R[13] = 0.0
    runs, levels_seen = [], set()
    for rep in range(2):
        s = pa.HipLdlSolver(prob.blocks[0]["K"], n_primal=prob.n_i)
        s.set_deterministic()
        s.matrixChanged()
        for again in range(2):
            X = R.copy()
            s.solve(X)
            assert s.info()["last_multi_path"] == 2
            levels_seen.add(s.info()["n_levels"])
            runs.append(X)
        ones = np.stack([s.solve(R[k].copy()) for k in (0, 31, 32, 69)])
        s.close()
    for X in runs[1:]:
        assert np.array_equal(X, runs[0])
    assert not runs[0][13].any()
    print("head levels", levels_seen)
    lu = spl.splu(prob.K_full(0))
    for k in range(70):
        if k != 13:
            xr = lu.solve(R[k])
            assert np.linalg.norm(runs[0][k] - xr) / np.linalg.norm(xr) < 1e-9
    for i, k in enumerate((0, 31, 32, 69)):
        assert np.linalg.norm(ones[i] - runs[0][k]) / np.linalg.norm(ones[i]) < 1e-11
