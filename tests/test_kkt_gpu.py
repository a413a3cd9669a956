"""GPU parity of the fused two-level system (factor2 + solveCompressed) against the oracle restatement."""
import numpy as np
import pytest

import pips_ipmpp_amd as pa
from oracle import oracle as orc
from tests.util import Problem, hip_lower_as_rowmajor

pytestmark = pytest.mark.gpu


def build_system(prob, blocks=None, comm=None, rank=0, n_ranks=1):
    blocks = list(range(prob.N)) if blocks is None else blocks
    bt = pa.LeafBatch(len(blocks), prob.S)
    for i, b in enumerate(blocks):
        bt.set_block(i, prob.blocks[b]["K"], prob.n_i, prob.blocks[b]["Bt"])
    bt.analyze(4)
    for i, b in enumerate(blocks):
        bt.set_values(i, prob.blocks[b]["K"].val)
    kkt = pa.KktSystem(bt, prob.n0, 0, prob.myl, 0, F0=prob.F0, comm=comm, rank=rank, n_ranks=n_ranks)
    return bt, kkt


@pytest.mark.parametrize("shape", [(3, 200, 24, 16, 0.04), (4, 1000, 100, 100, 0.01)])
def test_factorize_and_solve_compressed(shape):
    import torch
    N, n_i, n0, myl, rho = shape
    prob = Problem(77, N, n_i, n_i // 2, n0, myl, rho)
    S = prob.S
    bt, kkt = build_system(prob)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    kkt.factorize(diag, xd0)
    got = hip_lower_as_rowmajor(kkt.schur_to_host(), S)
    # oracle: assembleLocalKKT + finalizeKKTdense + dsytrf
    SCo = prob.oracle_finalize(prob.oracle_schur())
    want = np.tril(SCo)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-9
    root = orc.DenseRootSolver(S)
    root.matrixChanged(want)
    assert kkt.root_inertia() == (prob.n0, prob.myl, 0)
    assert root.get_inertia()[:2] == (prob.n0, prob.myl)
    # solveCompressed
    rng = np.random.default_rng(5)
    b0 = rng.standard_normal(S)
    bl = rng.standard_normal(N * prob.n_leaf)
    b0_d = torch.tensor(b0, device="cuda")
    bl_d = torch.tensor(bl, device="cuda")
    kkt.solve_compressed(b0_d, bl_d)
    bt.sync()
    leaf = [prob.oracle_leaf(b) for b in range(N)]
    b0_o = b0.copy()
    bs_o = [bl.reshape(N, -1)[b].copy() for b in range(N)]
    orc.solve_compressed(b0_o, bs_o, leaf, [prob.Bt_scipy(b) for b in range(N)], root, prob.n0, 0, 0, prob.myl, 0)
    x0 = b0_d.cpu().numpy()
    xl = bl_d.cpu().numpy().reshape(N, -1)
    assert np.linalg.norm(x0 - b0_o) / np.linalg.norm(b0_o) < 1e-8
    for b in range(N):
        assert np.linalg.norm(xl[b] - bs_o[b]) / np.linalg.norm(bs_o[b]) < 1e-8
    # the result solves the full arrowhead system  [K_i Br_i; Br_i^T K_0] x = b
    r0 = (prob.oracle_finalize(np.zeros((S, S))))
    K0 = np.tril(r0) + np.tril(r0, -1).T
    res0 = K0 @ x0 - b0
    num = 0.0
    for b in range(N):
        Bt = prob.Bt_scipy(b)
        ri = prob.K_full(b) @ xl[b] + Bt.T @ x0 - bl.reshape(N, -1)[b]
        res0 += Bt @ xl[b]
        num += ri @ ri
    num += res0 @ res0
    assert np.sqrt(num) / np.sqrt(b0 @ b0 + bl @ bl) < 1e-9


def test_rccl_communicator_single_rank_roundtrip():
    """dlopen'd RCCL: unique id, communicator of one rank, in-place all-reduce (the path bench.py uses for N > 1)."""
    import torch
    ident = pa.Comm.unique_id()
    assert len(ident) == 128
    comm = pa.Comm(ident, 1, 0, 0)
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    comm.close()


def test_reduction_path_with_one_rank_communicator(monkeypatch):
    """pack lower triangle -> RCCL all-reduce -> unpack, and the b0 all-reduce, driven on one GPU: results must not change."""
    import torch
    monkeypatch.setenv("PIPS_HIP_FORCE_REDUCE", "1")
    prob = Problem(78, 3, 200, 100, 24, 16, 0.04)
    comm = pa.Comm(pa.Comm.unique_id(), 1, 0, 0)
    bt, kkt = build_system(prob, comm=comm, rank=0, n_ranks=1)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
    got = hip_lower_as_rowmajor(kkt.schur_to_host(), prob.S)
    want = np.tril(prob.oracle_finalize(prob.oracle_schur()))
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-9
    rng = np.random.default_rng(5)
    b0, bl = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
    b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
    kkt.solve_compressed(b0_d, bl_d)
    bt.sync()
    root = orc.DenseRootSolver(prob.S)
    root.matrixChanged(want)
    bs_o = [bl.reshape(prob.N, -1)[b].copy() for b in range(prob.N)]
    b0_o = b0.copy()
    orc.solve_compressed(b0_o, bs_o, [prob.oracle_leaf(b) for b in range(prob.N)], [prob.Bt_scipy(b) for b in range(prob.N)], root,
                         prob.n0, 0, 0, prob.myl, 0)
    assert np.linalg.norm(b0_d.cpu().numpy() - b0_o) / np.linalg.norm(b0_o) < 1e-8
    comm.close()


def test_reduction_path_with_host_supplied_allreduce(monkeypatch):
    """External communicator: the all-reduce is a host callback (here torch.distributed over a one-rank RCCL group, in the
    reference it would be PIPS_MPIsumArrayInPlace on a GPU-aware MPI).  The callback must see every reduction of the path
    (packed Schur triangle, b0) and the results must equal the oracle's."""
    import torch
    import torch.distributed as dist
    monkeypatch.setenv("PIPS_HIP_FORCE_REDUCE", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29561")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        seen = []
        inner = pa.ExternalComm.torch_distributed()

        def counting(ptr, n):
            seen.append(n)
            t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()

        comm = pa.ExternalComm(counting)
        prob = Problem(79, 3, 200, 100, 24, 16, 0.04)
        bt, kkt = build_system(prob, comm=comm, rank=0, n_ranks=1)
        diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
        kkt.factorize(diag, torch.tensor(prob.x_diag0, device="cuda"))
        got = hip_lower_as_rowmajor(kkt.schur_to_host(), prob.S)
        want = np.tril(prob.oracle_finalize(prob.oracle_schur()))
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-9
        rng = np.random.default_rng(6)
        b0, bl = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
        b0_d, bl_d = torch.tensor(b0, device="cuda"), torch.tensor(bl, device="cuda")
        kkt.solve_compressed(b0_d, bl_d)
        bt.sync()
        assert seen == [prob.S * (prob.S + 1) // 2, prob.S]
        root = orc.DenseRootSolver(prob.S)
        root.matrixChanged(want)
        bs_o = [bl.reshape(prob.N, -1)[b].copy() for b in range(prob.N)]
        b0_o = b0.copy()
        orc.solve_compressed(b0_o, bs_o, [prob.oracle_leaf(b) for b in range(prob.N)], [prob.Bt_scipy(b) for b in range(prob.N)], root,
                             prob.n0, 0, 0, prob.myl, 0)
        assert np.linalg.norm(b0_d.cpu().numpy() - b0_o) / np.linalg.norm(b0_o) < 1e-8
        comm.close()
        inner.close()
    finally:
        dist.destroy_process_group()


def test_root_on_the_main_stream_gives_the_same_bits(monkeypatch):
    """pips_hip_kkt_set_root_stream(0): the dense root factorised on the main stream instead of a stream of its own (what a caller does that
    asks for the inertia after every factorisation) - same kernels in the same order, so the root's solution and inertia are identical
    to the bit in deterministic mode (the leaves' atomics aside, which that mode removes)."""
    import torch
    monkeypatch.setenv("PIPS_HIP_DETERMINISTIC", "1")
    prob = Problem(78, 4, 600, 300, 60, 40, 0.02)
    diag = torch.tensor(np.concatenate([b["diag"] for b in prob.blocks]), device="cuda")
    xd0 = torch.tensor(prob.x_diag0, device="cuda")
    rng = np.random.default_rng(1)
    b0h, blh = rng.standard_normal(prob.S), rng.standard_normal(prob.N * prob.n_leaf)
    res = []
    for own in (True, False, True):
        bt, kkt = build_system(prob)
        kkt.set_root_stream(own)
        for rep in range(2):       # twice: the second factorisation has to wait for the first one's root before it clears the Schur complement
            kkt.factorize(diag, xd0)
            b0, bl = torch.tensor(b0h, device="cuda"), torch.tensor(blh, device="cuda")
            kkt.solve_compressed(b0, bl)
            bt.sync()
        res.append((b0.cpu().numpy(), bl.cpu().numpy(), kkt.root_inertia()))
        kkt.close(); bt.close()
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2]
    assert res[0][2] == (prob.n0, prob.myl, 0)
