"""Multi-rank root without a multi-GPU node (DistributedRootLinearSystem.C:860-881,1614-1707): 2 and 4 processes
share device 0 and reduce through host-staged gloo collectives behind the external-communicator callbacks.  Covered here: the
Schur SYRK in row-panel groups with one reduction per panel issued on a second stream while later panels are still being
computed (PIPS_HIP_SC_PANELS), and the reduce-scatter + all-gather formulation (PIPS_HIP_SC_REDUCE=rsag) with its padding to a
multiple of the rank count.  Every variant must give the single-process Schur complement and solution.  No timing exists for
any of this: the GPU box has one device."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu
SHAPE = dict(seed=195, N=6, n_i=360, my_i=180, n0=520, myl=520, rho=0.02)   # S = 1040: up to four row panels; compressed border maps differ per block


def _worker(rank, world, port, out, panels, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PIPS_HIP_SC_PANELS"] = str(panels)
    os.environ["PIPS_HIP_SC_REDUCE"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = Problem(**SHAPE)
    mine = np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]
    S = prob.S
    calls = []

    def dev(ptr, n):
        return torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")

    def allreduce(ptr, n):
        t = dev(ptr, n)
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()
        calls.append(("ar", n))

    def reduce_scatter(ptr, chunk):     # gloo has no reduce-scatter: sum everything on the host, hand back ONLY this rank's slice
        t = dev(ptr, chunk * world)
        h = t.cpu()
        h = torch.nan_to_num(h, nan=0.0)   # padding entries beyond the piece: scratch
        dist.all_reduce(h)
        poisoned = torch.full_like(h, float("nan"))
        poisoned[rank * chunk:(rank + 1) * chunk] = h[rank * chunk:(rank + 1) * chunk]
        t.copy_(poisoned)
        torch.cuda.synchronize()
        calls.append(("rs", chunk))

    def all_gather(ptr, chunk):
        t = dev(ptr, chunk * world)
        mine_slice = t[rank * chunk:(rank + 1) * chunk].cpu()
        parts = [torch.empty_like(mine_slice) for _ in range(world)]
        dist.all_gather(parts, mine_slice)
        t.copy_(torch.cat(parts))
        torch.cuda.synchronize()
        calls.append(("ag", chunk))

    comm = pa.ExternalComm(allreduce, reduce_scatter, all_gather, n_ranks=world, rank=rank)
    bt = pa.LeafBatch(len(mine), S, device=0)
    bt.set_schur_mode(1)
    for i, b in enumerate(mine):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for i, b in enumerate(mine):
        bt.set_values(i, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=world)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in mine]) if len(mine) else np.zeros(0), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    for _ in range(2):       # twice: buffers and events are reused
        calls.clear()
        kkt.factorize(diag, xd0)
        bt.sync()
    SC = hip_lower_as_rowmajor(kkt.schur_to_host(), S)
    n_sc_calls = len(calls)
    rng = np.random.default_rng(4)
    b0_full = rng.standard_normal(S)
    bs_full = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    b0 = torch.tensor(b0_full, device="cuda")
    bl = torch.tensor(np.concatenate([bs_full[b] for b in mine]) if len(mine) else np.zeros(0), device="cuda")
    kkt.solve_compressed(b0, bl)
    bt.sync()
    xl = bl.cpu().numpy().reshape(len(mine), -1) if len(mine) else np.zeros((0, prob.n_leaf))
    np.savez(os.path.join(out, f"rank{rank}.npz"), SC=SC, xroot=b0.cpu().numpy(), blocks=np.array(mine), n_sc_calls=n_sc_calls,
             kinds=np.array([c[0] for c in calls]), **{f"x{b}": xl[i] for i, b in enumerate(mine)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,panels,mode", [(2, 1, "allreduce"), (2, 4, "allreduce"), (2, 1, "rsag"), (2, 3, "rsag"), (4, 4, "rsag"), (4, 2, "allreduce")])
def test_panelwise_and_rsag_reduction(tmp_path, world, panels, mode):
    port = 29500 + (os.getpid() % 2000) + 31 + 7 * panels + (3 if mode == "rsag" else 0) + world
    mp.start_processes(_worker, args=(world, port, str(tmp_path), panels, mode), nprocs=world, join=True, start_method="spawn")
    prob = Problem(**SHAPE)
    S = prob.S
    SC1 = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    root = orc.DenseRootSolver(S)
    root.matrixChanged(SC1)
    rng = np.random.default_rng(4)
    b0 = rng.standard_normal(S)
    bs = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    orc.solve_compressed(b0, bs, [prob.oracle_leaf(b) for b in range(prob.N)], [prob.Bt_scipy(b) for b in range(prob.N)],
                         root, prob.n0, 0, 0, prob.myl, 0)
    seen = []
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        # one reduction (or one reduce-scatter + all-gather pair) per row panel
        assert int(g["n_sc_calls"]) == panels * (2 if mode == "rsag" else 1), (g["n_sc_calls"], g["kinds"])
        assert np.abs(g["SC"] - SC1).max() / np.abs(SC1).max() < 1e-9
        assert np.linalg.norm(g["xroot"] - b0) / np.linalg.norm(b0) < 1e-8
        for b in g["blocks"]:
            assert np.linalg.norm(g[f"x{b}"] - bs[b]) / np.linalg.norm(bs[b]) < 1e-8
            seen.append(int(b))
    assert sorted(seen) == list(range(prob.N))
