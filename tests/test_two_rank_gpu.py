"""The N > 1 path of the product on a one-GPU box: two processes share device 0, each owns its contiguous share of the blocks
(pips_map_children_to_ranks), the two reductions of the path (packed Schur triangle, b0) go through the host-supplied
all-reduce of an ExternalComm (gloo, staged through host memory - RCCL itself cannot put two ranks on one GPU).  Checked
against the oracle's single-process result: the reduced and finalised Schur complement on every rank, x0 on every rank,
every block's solution on its owner."""
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

SHAPE = dict(seed=92, N=5, n_i=300, my_i=150, n0=12, myl=10, rho=0.03)


def _make_problem(sparse_root):
    if sparse_root:
        from tests.test_sparse_root_gpu import TwoLinkProblem
        return TwoLinkProblem(93, 6, 240, 120, 5, 4, 5.0 / 240)
    return Problem(**SHAPE)


def _worker(rank, world, port, out, sparse_root):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if sparse_root == "distributed":     # dense root factorised column-cyclically over the two ranks (DenseLdl::set_distributed)
        os.environ["PIPS_HIP_ROOT_DISTRIBUTED"] = "1"
        sparse_root = False
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = _make_problem(sparse_root)
    mine = np.nonzero(pa.map_children_to_ranks(prob.N, world) == rank)[0]
    S = prob.S
    calls = []

    def allreduce(ptr, n):          # what a host MPI would do with a device buffer it cannot touch directly
        t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()
        calls.append(n)

    comm = pa.ExternalComm(allreduce)
    bt = pa.LeafBatch(len(mine), S, device=0)
    if sparse_root:
        bt.set_schur_mode(1)
    for i, b in enumerate(mine):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(2)
    for i, b in enumerate(mine):
        bt.set_values(i, prob.blocks[b]["K"].val)
    # sparse root: every rank needs the border column sets of ALL blocks (one common pattern to reduce)
    cols = [np.nonzero(np.diff(prob.blocks[b]["Bt"].rowptr) > 0)[0] for b in range(prob.N)] if sparse_root else None
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=world, sparse_root=sparse_root,
                       all_block_cols=cols)
    diag = torch.tensor(np.concatenate([prob.blocks[b]["diag"] for b in mine]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    SC = kkt.schur_sparse_to_host().toarray() if sparse_root else hip_lower_as_rowmajor(kkt.schur_to_host(), S)
    rng = np.random.default_rng(4)
    b0_full = rng.standard_normal(S)
    bs_full = [rng.standard_normal(prob.n_leaf) for _ in range(prob.N)]
    b0 = torch.tensor(b0_full, device="cuda")     # every rank passes the full b0; ranks > 0 are zeroed inside (Lsolve)
    bl = torch.tensor(np.concatenate([bs_full[b] for b in mine]), device="cuda")
    kkt.solve_compressed(b0, bl)
    bt.sync()
    xl = bl.cpu().numpy().reshape(len(mine), -1)
    np.savez(os.path.join(out, f"rank{rank}.npz"), SC=SC, xroot=b0.cpu().numpy(), blocks=np.array(mine), calls=np.array(calls),
             **{f"x{b}": xl[i] for i, b in enumerate(mine)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse_root", [False, True, "distributed"], ids=["dense_root", "sparse_root", "distributed_dense_root"])
def test_two_processes_share_one_gpu(tmp_path, sparse_root):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if sparse_root else 0)
    mp.start_processes(_worker, args=(world, port, str(tmp_path), sparse_root), nprocs=world, join=True, start_method="spawn")
    distributed = sparse_root == "distributed"
    sparse_root = sparse_root is True
    prob = _make_problem(sparse_root)
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
        # Schur complement (packed triangle / CSR values), then b0; the distributed root adds one panel per tile column + the inertia counts
        assert len(g["calls"]) == 2 + (((S + 127) // 128 + 1) if distributed else 0) and g["calls"][-1] == S
        if not sparse_root:
            assert g["calls"][0] == S * (S + 1) // 2
        assert np.abs(g["SC"] - SC1).max() / np.abs(SC1).max() < 1e-9
        assert np.linalg.norm(g["xroot"] - b0) / np.linalg.norm(b0) < 1e-8
        for b in g["blocks"]:
            assert np.linalg.norm(g[f"x{b}"] - bs[b]) / np.linalg.norm(bs[b]) < 1e-8
            seen.append(int(b))
    assert sorted(seen) == list(range(prob.N))
