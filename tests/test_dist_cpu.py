"""world_size-2 gloo test of the N > 1 path on CPU: block sharding (assignProcesses contract), the Schur-complement
all-reduce (reduceKKTdense) and the b0 reduction of Lsolve, with the oracle doing the per-rank arithmetic."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem

SHAPE = dict(seed=91, N=5, n_i=80, my_i=40, n0=8, myl=6, rho=0.08)


def test_block_to_rank_map_contract():
    for n, r in [(5, 2), (64, 8), (7, 3), (512, 8), (9, 9)]:
        m = pa.map_children_to_ranks(n, r)
        assert (np.diff(m) >= 0).all() and m[0] == 0 and m[-1] == r - 1
        loads = np.bincount(m, minlength=r)
        assert loads.max() - loads.min() <= 1
    with pytest.raises(pa.capi.PipsHipError):
        pa.map_children_to_ranks(3, 4)   # "too many MPI processes" (DistributedTree.C:57-60)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = Problem(**SHAPE)
    mine = np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]
    S = prob.S
    # factor2: local assembleLocalKKT, then reduceKKT (sum), then finalize + factor replicated on every rank
    SC = torch.tensor(np.tril(prob.oracle_schur(mine)))
    dist.all_reduce(SC)
    SCf = prob.oracle_finalize(SC.numpy().copy())
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SCf))
    # solveCompressed: ranks > 0 zero b0 (sLinsysRootAug.C:330-331), local Lsolve, all-reduce b0, replicated Dsolve, local Ltsolve
    rng = np.random.default_rng(4)
    b0_full = rng.standard_normal(S)
    bs_full = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    b0 = b0_full.copy() if rank == 0 else np.zeros(S)
    leaf = {b: prob.oracle_leaf(b) for b in mine}
    xs = {b: bs_full[b].copy() for b in mine}
    for b in mine:
        leaf[b].solve(xs[b])
        b0 -= prob.Bt_scipy(b) @ xs[b]
    t = torch.tensor(b0)
    dist.all_reduce(t)
    x0 = t.numpy().copy()
    root.solve(x0)
    for b in mine:
        tt = prob.Bt_scipy(b).T @ x0
        leaf[b].solve(tt)
        xs[b] -= tt
    np.savez(os.path.join(out, f"rank{rank}.npz"), SC=SCf, xroot=x0, blocks=np.array(mine), **{f"x{b}": xs[b] for b in mine})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    prob = Problem(**SHAPE)
    S = prob.S
    SC1 = prob.oracle_finalize(prob.oracle_schur())
    root = orc.DenseRootSolver(S)
    root.matrixChanged(np.tril(SC1))
    rng = np.random.default_rng(4)
    b0 = rng.standard_normal(S)
    bs = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    orc.solve_compressed(b0, bs, [prob.oracle_leaf(b) for b in range(prob.N)], [prob.Bt_scipy(b) for b in range(prob.N)],
                         root, prob.n0, 0, 0, prob.myl, 0)
    seen = []
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.abs(np.tril(g["SC"]) - np.tril(SC1)).max() / np.abs(SC1).max() < 1e-12
        assert np.linalg.norm(g["xroot"] - b0) / np.linalg.norm(b0) < 1e-9
        for b in g["blocks"]:
            assert np.linalg.norm(g[f"x{b}"] - bs[b]) / np.linalg.norm(bs[b]) < 1e-9
            seen.append(int(b))
    assert sorted(seen) == list(range(prob.N))
