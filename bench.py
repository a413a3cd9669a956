#!/usr/bin/env python3
"""bench.py — KKT factor+solve work units per second on N MI355X (BASELINE.json metric).

One step = one work unit of SURVEY.md §8d: 1 factorize (leaf diagonal update, all leaf LDL^T factors, Schur contributions
sum_i Br_i^T K_i^-1 Br_i, SC all-reduce, root finalize, dense root LDL^T) + 4 solveCompressed (Lsolve, root solve, Ltsolve).
N = 1 runs BASELINE.json configs[1]: 64 scenario blocks x 10k vars (my_i = 5k), ~0.1 % fill, Schur dim 2000, all leaves
batched on one device.  N > 1 shards 64 blocks per GPU (weak scaling), one process per GPU, RCCL all-reduce of SC and b0.
All inputs are resident in HBM before the timed region.  `value` counts 64-block scenario groups processed per second
(= N x IPM-iteration linear-algebra units per second), so it is the whole-job aggregate.

--family time-coupled --blocks-per-gpu 256 --n 50000 runs the per-GPU share of BASELINE.json configs[3] (2048 blocks x 50 000
variables on 8 GPUs: banded W_i, 95 first-stage variables, 31 two-link rows between neighbouring blocks: S = 8000); there the
sparse head of the leaf factorisation is the dominant kernel group and the roofline object prices it against HBM.

The line accounts for itself: `phase_ms` lists every phase of the step (HIP events on the streams the work runs on, one
instrumented step after the timed region), `roofline` is the kernel group with the largest share of the step for the
configuration that was run, `roofline_all` the others; with N > 1 `collective` carries the reductions' time, bytes and overlap.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E (MI355X_MICROARCH.md)
TILE = 128
R_SOLVES = 4


def update_kernel_algorithmic_flops(m, nb):
    """Algorithmic flops of the tail update kernel (k_tile_gemm_bal<0>, or k_tile_gemm<0> with static shares) for one block with dense tail m and nb border rows.  The
    launch of tile column j (K = j*TILE already factored columns) gives every entry of the column below its diagonal tile and
    its nb border entries one length-K dot product (2K flops), and the lower triangle of the NEXT diagonal tile its dot
    products with the same K columns (engine.hip TailPlan::build, diag_ahead; the last step of a diagonal tile, with the
    column just finished, is the side stream's k_tile_gemm<4> and is not counted here).  Summed: ~ m^3/3 + nb*m^2."""
    fl = 0.0
    j = 0
    while j * TILE < m:
        K = j * TILE
        tc = min(TILE, m - K)
        below = m - K - tc
        tc_next = max(0, min(TILE, below))
        entries = tc * below + nb * tc + tc_next * (tc_next + 1) / 2.0
        fl += 2.0 * K * entries
        j += 1
    return fl


def update_kernel_traffic(n_blocks, n_i, S, world):
    """HBM bytes of the update kernel per factorize from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes over this very command, FETCH_SIZE doubled per MI355X_MICROARCH.md and tools/pmc_calib); only valid for the profiled
    workload (configs[1])."""
    import glob
    import re
    if world != 1 or n_blocks != 64 or n_i != 10000 or S != 2000:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_update_traffic.json")),
                   key=lambda q: -int(re.match(r"r(\d+)_", os.path.basename(q)).group(1)))
    for path in files:
        try:
            return json.load(open(path))["hbm_bytes_per_factorize"]
        except Exception:
            continue
    return None


def head_traffic(shape, key="hbm_bytes_per_factorize"):
    """HBM bytes of the sparse-head kernels per factorize (or, key = "solve_hbm_bytes_per_step", of the leaf solve sweeps per step)
    from the newest committed PMC summary of this very shape (tools/profile_cfg3.sh writes the shape into the file; the round-4 file
    is the 256-block chain); None for a workload that was not profiled."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg3*_head_traffic.json")),
                   key=lambda q: (-int(re.match(r"r(\d+)_", os.path.basename(q)).group(1)), q))
    for path in files:
        try:
            d = json.load(open(path))
        except Exception:
            continue
        have = d.get("shape") or shape_key("time-coupled", 256, 50000, 8000, 256)
        if have == shape and int(re.match(r"r(\d+)_", os.path.basename(path)).group(1)) >= 4:   # (round 3: the head before the border split)
            return d.get(key)
    return None


def shape_key(family, bpg, n_i, S_whole, chain_blocks):
    """what one GPU holds in a run, machine-readable (config.shape of the line): the key under which one-GPU lines are looked up"""
    return {"family": family, "blocks_per_gpu": int(bpg), "n": int(n_i), "schur_dim": int(S_whole), "chain_blocks": (int(chain_blocks) if chain_blocks else None)}


def same_shape_on_one_gpu(shape):
    """The newest committed ONE-GPU bench line of the per-GPU shape an N-GPU run uses (profiles/r*_bench_other_configs.jsonl,
    profiles/r*_cfg3*_bench_n1.json): the one-device reference of that shape for reading a scaling series whose N = 1 line is
    another configuration.  Read from the files, never pasted; None when no such line is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_other_configs.jsonl")) + glob.glob(os.path.join(ROOT, "profiles", "r*_cfg3*_bench_n1.json")),
                   key=lambda q: (-int(re.match(r"r(\d+)_", os.path.basename(q)).group(1)), q))
    for path in files:
        for raw in open(path):
            try:
                d = json.loads(raw)
            except Exception:
                continue
            if not isinstance(d, dict) or d.get("n_gpus") != 1 or "config" not in d:
                continue
            have = d["config"].get("shape")
            if have is None:     # lines of rounds 1-4 (random family): "<B> blocks x <n> vars (...), Schur dim <S>, <B> blocks/GPU"
                m = re.match(r"(\d+) blocks x (\d+) vars .*Schur dim (\d+), (\d+) blocks/GPU", d["config"].get("workload", ""))
                if m and d["config"].get("family", "random") == "random":
                    have = shape_key("random", m.group(4), m.group(2), m.group(3), None)
            if have == shape and not d["config"].get("deterministic"):
                return {"units_per_s": d["value"], "ms_per_step": d["ms_per_step"], "source": os.path.relpath(path, ROOT)}
    return None


def workload_label(family, world, bpg, n_i, my_i, rho, S, whole=None, myl=None):
    """config.workload: what was run, and which BASELINE.json configuration it is"""
    total = world * bpg
    if family == "random":
        tag = (" [BASELINE configs[1]]" if world == 1 and bpg == 64 and n_i == 10000 and S == 2000 else
               " [BASELINE configs[2]]" if world == 8 and bpg == 64 and n_i == 10000 and S == 4000 else
               f" [BASELINE configs[2] shape (Schur dim 4k, 64 blocks/GPU) on {world} of its 8 GPUs]" if world > 1 and bpg == 64 and n_i == 10000 and S == 4000 else "")
        return f"{total} blocks x {n_i} vars ({my_i} eq rows, rho={rho}), Schur dim {S}, {bpg} blocks/GPU" + tag
    c3 = whole.G == 2048 and whole.S == 8000 and n_i == 50000
    tag = (" [BASELINE configs[3]]" if c3 and total == whole.G else
           f" [BASELINE configs[3] shape on {world} of its 8 GPUs]" if c3 and bpg == 256 else
           " [the 256-block chain of rounds 3-4: 31 linking rows per pair]" if whole.G == 256 and whole.S == 8000 and n_i == 50000 and bpg == 256 else "")
    return (f"time-coupled chain of {whole.G} blocks x {n_i} vars ({my_i} banded eq rows, 10 nnz/row), {whole.n0} first-stage variables, "
            f"{whole.myl} two-link rows over its {whole.G - 1} neighbouring pairs, Schur dim {whole.S}: "
            + (f"all of it on {world} GPU(s), {bpg} blocks/GPU" if total == whole.G else
               f"blocks 0..{total - 1} on {world} GPU(s), {bpg} blocks/GPU, with the {myl} linking rows they touch (Schur dim {S})") + tag)


def build_rank_problem(pa, seed, blocks, n_i, my_i, n0, myl, rho, device, block_data=None, threads=None):
    """block_data: b -> (W, T, F) for families other than the generator's"""
    S = n0 + myl
    bt = pa.LeafBatch(len(blocks), S, device=device)
    vals, diags = [], []
    for i, b in enumerate(blocks):
        if block_data is not None:
            W, T, F = block_data(b)
        else:
            W, T, F, c, xs = pa.gen_block(seed, b + 1, n_i, my_i, n0, myl, rho)
        K, dpos = pa.kkt_leaf_assemble(n_i, W)
        Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F)
        diag = np.concatenate([pa.gen_diagonal(seed, b + 1, n_i), -1e-8 * np.ones(my_i)])
        K.val[dpos] = diag
        bt.set_block(i, K, n_i, Bt)
        vals.append(K.val)
        diags.append(diag)
    bt.analyze(threads or min(16, os.cpu_count() or 8))
    # PARDISO-style adaptive iterative refinement (iparm[7]=2 in the reference, PardisoProjectSolver.C:72: at most 2 steps,
    # stop when the backward error is satisfactory): normwise backward error <= 1e-15 for every block
    bt.set_refinement_backward_error(2, 1e-15)
    for i in range(len(blocks)):
        bt.set_values(i, vals[i])
    return bt, np.concatenate(diags)


def ipm_end_to_end(pa, seed, N, n_i, my_i, n0, myl, rho, family_blocks=None, family_F0=None):
    """Full interior-point solve (Mehrotra + Gondzio harness, termination mu <= 1e-8 and ||r||inf <= 1e-8 dnorm) of the LP the
    generator defines for this shape: b = A x*, x* ~ U(0.5, 1.5)."""
    rng = np.random.default_rng(seed)
    if family_blocks is not None:
        F0, c0, x0s = family_F0, rng.uniform(0.5, 1.5, n0), rng.uniform(0.5, 1.5, n0)
    else:
        F0, c0, x0s = pa.gen_root(seed, n0, myl)
    blocks, cs, bs = [], [c0], []
    blink = F0.to_scipy() @ x0s
    for b in range(1, N + 1):
        if family_blocks is not None:
            W, T, F = family_blocks[b - 1]
            c, xs = rng.uniform(0.5, 1.5, n_i), rng.uniform(0.5, 1.5, n_i)
        else:
            W, T, F, c, xs = pa.gen_block(seed, b, n_i, my_i, n0, myl, rho)
        blocks.append((W, T, F))
        cs.append(c)
        bs.append(T.to_scipy() @ x0s + W.to_scipy() @ xs)
        blink = blink + F.to_scipy() @ xs
    ipm = pa.IpmSolver(n0, myl, blocks, F0, np.concatenate(cs), np.concatenate([blink] + bs))
    w0, sites0 = pa.host_wait_count(), pa.host_wait_sites()
    t0 = time.perf_counter()
    res = ipm.solve(max_iter=150, mutol=1e-8, artol=1e-8)
    dt = time.perf_counter() - t0
    waits = pa.host_wait_count() - w0
    sites = {k: v - sites0.get(k, 0) for k, v in pa.host_wait_sites().items() if v - sites0.get(k, 0) > 0}
    st = ipm.stats()
    ipm.close()
    out = {"status": res["status"], "iterations": res["iterations"], "seconds": round(dt, 3), "iterations_per_s": round(res["iterations"] / dt, 3),
           "objective": res["objective"], "mu": res["mu"], "rel_residual": res["rnorm"] / res["dnorm"], "factorizations": st["factorizations"],
           "solve_compressed": st["solve_compressed"], "variables": int(n0 + N * n_i), "constraints": int(myl + N * my_i),
           # how often the host stopped for the device inside the library (csrc/common.h counts every synchronisation and blocking copy)
           "host_waits": int(waits), "host_waits_per_iteration": round(waits / max(res["iterations"], 1), 1),
           "host_wait_sites": dict(sorted(sites.items(), key=lambda kv: -kv[1])[:8])}
    # the same LP through the CPU PARDISO path (tests/golden/ipm_configs1.json, made by tests/golden/make_ipm_configs1.py in the build
    # container): north_star asks for agreement to 1e-8 relative; tests/test_ipm_gpu.py::test_configs1_matches_the_cpu_pardiso_path asserts it
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "ipm_configs1.json")))
        if family_blocks is None and int(g["seed"]) == seed and [int(v) for v in g["shape"][:5]] == [N, n_i, my_i, n0, myl] and float(g["shape"][5]) == rho:
            out["cpu_pardiso_path"] = {"objective": g["objective"], "iterations": g["iterations"], "rel_residual": g["rnorm"] / g["dnorm"],
                                       "objective_rel_diff": abs(res["objective"] - g["objective"]) / abs(g["objective"]),
                                       "source": "tests/golden/ipm_configs1.json"}
    except Exception:
        pass
    return out


def cpu_baseline(pa, seed, n_i, my_i, n0, myl, rho, n_blocks_total, bpg=64, block0=None, whole_block=False):
    """Reference-style CPU path timed on a bounded sample and extrapolated linearly (all blocks are statistically
    identical and independent): per block PARDISO phase 12 (or the oracle LDL^T) + multi-RHS solves for the border
    columns (K5) + the sparse accumulation (K6) + 2*R single solves; plus the dense root dsytrf."""
    from oracle import oracle as orc
    from oracle import pardiso_mkl as pm
    import scipy.sparse as sp
    import psutil
    cores = min(16, os.cpu_count() or 1)         # bounded: the sample must not exhaust the host
    mem_ok = psutil.virtual_memory().available > 48 * 2**30
    if block0 is not None:
        W, T, F = block0
    else:
        W, T, F, c, xs = pa.gen_block(seed, 1, n_i, my_i, n0, myl, rho)
    K, dpos = pa.kkt_leaf_assemble(n_i, W)
    K.val[dpos] = np.concatenate([pa.gen_diagonal(seed, 1, n_i), -1e-8 * np.ones(my_i)])
    Ks = sp.csr_matrix((K.val, K.colidx, K.rowptr), shape=(K.nrows, K.ncols))
    Bt = pa.border_assemble(n_i, my_i, 0, n0, 0, A=T, F=F).to_scipy()
    S = n0 + myl
    # whole_block (runs of 20 steps or more: the driver's): EVERY non-empty border column of the block goes through K4-K6 in the reference's
    # chunks (20 columns per thread, DistributedLinearSystem.C:766-1047) - no extrapolation over columns; else 32 columns, scaled
    n_rhs_sample = S if whole_block else min(32, S)
    if pm.available() and mem_ok:
        kind_detail = f"MKL PARDISO mtype -2 with the reference's iparm (PardisoProjectSolver.C:68-77), {cores} threads"
        solver = pm.MklPardisoSolver(Ks, num_threads=cores)
        used = cores
    else:
        kind_detail = "oracle_ldl.c up-looking LDL^T, 1 thread"
        info = pa.symbolic_probe(K, n_i, want_perm=True)
        solver = orc.OracleLdl(Ks, perm=info["perm"], n_primal=n_i)
        used = 1
    t0 = time.perf_counter()
    solver.matrixChanged()
    t_factor = time.perf_counter() - t0
    nonempty = np.nonzero(np.diff(Bt.indptr) > 0)[0]
    cols = nonempty[:n_rhs_sample]
    n_border = len(nonempty)          # the reference skips empty border columns (DistributedLinearSystem.C:870-874)
    if whole_block:
        chunk = 20 * max(1, used)
        t0 = time.perf_counter()
        for k0 in range(0, len(cols), chunk):
            rhs = np.ascontiguousarray(Bt[cols[k0:k0 + chunk]].toarray())   # K4: dense-ify
            solver.solve(rhs)                                                # K5
            SCrows = (Bt @ rhs.T).T  # noqa: F841                            # K6
        t_schur = time.perf_counter() - t0
    else:
        dense = np.ascontiguousarray(Bt[cols].toarray())
        # best of two / three repetitions: the first multi-RHS call pays thread start-up, and the host is shared
        t_schur = float("inf")
        for _ in range(2):
            rhs = dense.copy()
            t0 = time.perf_counter()
            solver.solve(rhs)
            SCrows = (Bt @ rhs.T).T  # noqa: F841  (K6)
            t_schur = min(t_schur, (time.perf_counter() - t0) * (n_border / max(1, len(cols))))
    t_solve = float("inf")
    for rep in range(3):
        x = np.random.default_rng(rep).standard_normal(K.nrows)
        t0 = time.perf_counter()
        solver.solve(x)
        t_solve = min(t_solve, time.perf_counter() - t0)
    M = np.random.default_rng(1).standard_normal((S, S))
    M = np.tril(M @ M.T + S * np.eye(S))
    root = orc.DenseRootSolver(S)
    t0 = time.perf_counter()
    root.matrixChanged(M)
    t_root = time.perf_counter() - t0
    per_group = bpg * (t_factor + t_schur + 2 * R_SOLVES * t_solve) + t_root
    return {
        "value": 1.0 / per_group, "unit": f"{bpg}-block work units/s", "cores": used, "kind": "port",
        "sample": (f"1 of {n_blocks_total} blocks: factor {t_factor:.2f}s, {len(cols)} of {n_border} non-empty border columns solved "
                   f"({'measured' if whole_block else 'extrapolated'} {t_schur:.2f}s), 1 single solve {t_solve*1e3:.0f}ms, root dsytrf {t_root:.2f}s; "
                   f"x{bpg} blocks per unit, {2*R_SOLVES} leaf solves per block; {kind_detail}"),
    }


def launch_ranks(n):
    """One process per GPU (the reference maps blocks to MPI ranks the same way: Readers/Distributed/DistributedTree.C:62-89).
    The children are this script with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and its own
    arguments; the parent never initialises HIP (torch.cuda.device_count() does not) and only hands on the children's exit code; rank 0's
    JSON line goes to the inherited stdout.  (torch.distributed.run is not used here: its argument parser trips over script options such
    as --n; a launch through it works as before because WORLD_SIZE is then set.)"""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("PIPS_BENCH_SHARE_GPU"):
        sys.stderr.write(f"bench.py: --gpus {n} but this node shows {have} device(s) (PIPS_BENCH_SHARE_GPU=1 lets the ranks share device 0 "
                         "for a validation run)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:       # a rank failed: the others would wait for it in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks-per-gpu", type=int, default=64)
    ap.add_argument("--n", type=int, default=10000, help="variables per block")
    ap.add_argument("--schur-dim", type=int, default=None,
                    help="S = n0 + myl of the random family; default: 2000 on one GPU (BASELINE configs[1]), 4000 on several "
                         "(BASELINE configs[2]: 512 blocks x 10k, Schur dim 4k, 64 blocks per GPU on 8 GPUs)")
    ap.add_argument("--rho", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ipm", action="store_true", help="skip the end-to-end IPM run reported next to the metric (N = 1 only)")
    ap.add_argument("--root", choices=["auto", "dense", "sparse"], default="auto",
                    help="root system: dense LDL^T of the S x S Schur complement, or the sparse root (2-link structure: the Schur complement is "
                         "kept as a CSR value array and factorised by a one-block sparse engine); auto = sparse for the time-coupled family")
    ap.add_argument("--family", choices=["random", "time-coupled"], default="random",
                    help="random: the generator of SURVEY 8d (BASELINE configs[1], [2], [4]); time-coupled: banded blocks with 2-link rows - "
                         "ONE fixed chain (configs[3]: --chain-blocks 2048, --schur-dim 8000) of which rank r holds blocks "
                         "[r * blocks-per-gpu, (r + 1) * blocks-per-gpu); --rho is ignored")
    ap.add_argument("--chain-blocks", type=int, default=None,
                    help="time-coupled family: blocks of the whole chain (default 2048 = BASELINE configs[3]; 256 = the 256-block chain of "
                         "rounds 3-4, 31 linking rows per pair).  Fewer ranks than the chain needs run its first gpus * blocks-per-gpu blocks "
                         "with the linking rows those blocks touch")
    ap.add_argument("--solve-check-every", type=int, default=None,
                    help="measure every k-th solveCompressed that goes by sweeps of the augmented factor against the leaf rows "
                         "(library default 1: every one; 0: only the first after a factorisation - rounds 4's behaviour)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as child processes BEFORE anything here touches the GPU
        raise SystemExit(launch_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); the line reports n_gpus = {world}\n")
    if a.schur_dim is None:
        a.schur_dim = 8000 if a.family == "time-coupled" else 2000 if world == 1 else 4000
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X backend has no CPU fallback")
    # PIPS_BENCH_SHARE_GPU=1 (validation only, never a measurement): all ranks of a multi-process launch use device 0 and the
    # reductions are staged through host memory over gloo - the N > 1 control flow of this script on a one-GPU box
    share_gpu = world > 1 and bool(os.environ.get("PIPS_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # PIPS_BENCH_FORCE_COMM=1 drives the whole multi-rank code path (process group, communicator bootstrap, packed Schur
    # reduction, b0 reduction) with a single rank, so that it can be checked on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("PIPS_BENCH_FORCE_COMM"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if world == 1:
            os.environ["PIPS_HIP_FORCE_REDUCE"] = "1"
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import pips_ipmpp_amd as pa
    import families

    n_i, my_i = a.n, a.n // 2
    bpg = a.blocks_per_gpu
    blocks = list(range(rank * bpg, (rank + 1) * bpg))
    n_blocks_total = bpg * world
    fam_blocks = fam_F0 = chain = whole = None
    if a.family == "time-coupled":
        # ONE fixed chain (BASELINE configs[3]: 2048 blocks, S = 8000) mapped onto the ranks as the reference maps its tree
        # (Readers/Distributed/DistributedTree.C:62-89): a rank draws ONLY its own blocks (every block has its own generator), so host
        # memory and set-up time per rank do not depend on the number of ranks.  With fewer ranks than the chain needs the problem is
        # the chain's first world * bpg blocks and the linking rows they touch (TimeCoupledChain.prefix).
        whole = families.config3_chain(n_i, a.chain_blocks, a.schur_dim)
        if n_blocks_total > whole.G:
            raise SystemExit(f"bench.py: {world} rank(s) x {bpg} blocks exceed the chain's {whole.G} blocks (--chain-blocks)")
        chain = whole.prefix(n_blocks_total)
        fam_blocks = dict(zip(blocks, chain.blocks(blocks[0], blocks[-1] + 1)))
        fam_F0, my_i, myl, n0 = chain.F0(), chain.my_i, chain.myl, chain.n0
    else:
        n0 = myl = a.schur_dim // 2
    S = n0 + myl
    shape = shape_key(a.family, bpg, n_i, whole.S if whole is not None else S, whole.G if whole is not None else None)

    comm = None
    comm_kind = "none"
    if use_dist:
        # The library's own RCCL communicator (dlopen'd librccl, bootstrapped with a unique id broadcast over the process
        # group).  PIPS_BENCH_COMM=torch selects the host-supplied all-reduce instead (torch.distributed, also RCCL); the
        # same switch is taken on every rank if any rank fails to create its communicator.
        want_own = os.environ.get("PIPS_BENCH_COMM", "rccl") != "torch" and not share_gpu
        ok = torch.ones(1, dtype=torch.int32, device="cpu" if share_gpu else "cuda")
        if want_own:
            # ncclCommInitRank is collective: a rank that fails BEFORE it (librccl not loadable, bad device) would leave the others
            # blocked inside it.  So every rank first checks locally what can be checked locally and the ranks vote; only a
            # unanimous yes enters the collective initialisation.
            try:
                my_id = pa.Comm.unique_id()          # loads librccl, touches the device
                torch.zeros(1, device="cuda").item()
            except Exception as e:
                sys.stderr.write(f"[rank {rank}] own RCCL communicator unavailable ({e}); using torch.distributed\n")
                my_id = None
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                try:
                    idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
                    if rank == 0:
                        idt.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
                    dist.broadcast(idt, 0)
                    comm = pa.Comm(bytes(idt.cpu().numpy().tobytes()), world, rank, local_rank)
                except Exception as e:
                    sys.stderr.write(f"[rank {rank}] ncclCommInitRank failed ({e}); using torch.distributed\n")
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if want_own and int(ok.item()) == 1:
            comm_kind = "rccl (library communicator)"
        else:
            if comm is not None:
                comm.close()
            if share_gpu:
                def staged(ptr, n):
                    t = torch.as_tensor(pa.capi._DeviceDoubles(ptr, n), device="cuda")
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                    torch.cuda.synchronize()
                comm = pa.ExternalComm(staged)
                comm_kind = "gloo, host-staged (ranks share one GPU: validation run, not a measurement)"
            else:
                comm = pa.ExternalComm.torch_distributed()
                comm_kind = "rccl (torch.distributed callback)"

    ranks_seen = None
    if comm is not None:
        # "did the communicator the factorisation uses see N ranks": a sum of ones through pips_hip_allreduce_sum
        one = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
        comm.allreduce_sum(one)
        torch.cuda.synchronize()
        ranks_seen = int(round(float(one.item())))
        if ranks_seen != world:
            raise SystemExit(f"[rank {rank}] the communicator sums over {ranks_seen} rank(s), the launch has {world}")

    bt, diag_h = build_rank_problem(pa, a.seed, blocks, n_i, my_i, n0, myl, a.rho, local_rank, threads=max(1, min(16, (os.cpu_count() or 8) // world)),
                                    block_data=(lambda b: fam_blocks[b]) if fam_blocks is not None else None)
    if fam_blocks is not None:
        F0 = fam_F0
    else:
        F0, c0, x0s = pa.gen_root(a.seed, n0, myl)
    # time-coupled family: 2-link rows -> sparse root, as the reference does for this class (sLinsysRootAug.C:1629-1739)
    sparse_root = a.root == "sparse" or (a.root == "auto" and fam_blocks is not None)
    all_cols = None
    if sparse_root:
        if fam_blocks is None:
            raise SystemExit("--root sparse needs --family time-coupled (the random family's Schur complement is dense)")
        # non-empty border columns of EVERY block of the problem (every rank reduces the same pattern): x0 columns T touches, rows of F.
        # A rank knows its own blocks; the lists of the others come over the process group (a few hundred integers per block)
        mine = [chain.border_columns(b, fam_blocks[b][1]) for b in blocks]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            all_cols = [c for part in gathered for c in part]
        else:
            all_cols = mine
    kkt = pa.KktSystem(bt, n0, 0, myl, 0, F0=F0, comm=comm, rank=rank, n_ranks=world, sparse_root=sparse_root, all_block_cols=all_cols)
    if a.solve_check_every is not None:
        kkt.set_solve_check(a.solve_check_every)
    dev = torch.device("cuda", local_rank)
    diag = torch.tensor(diag_h, device=dev)
    xd0 = torch.tensor(pa.gen_diagonal(a.seed, 0, n0), device=dev)
    g = torch.Generator(device="cpu").manual_seed(a.seed + rank)
    rhs_leaf = torch.randn(diag.numel(), dtype=torch.float64, generator=g).to(dev)
    g0 = torch.Generator(device="cpu").manual_seed(a.seed)
    rhs0 = torch.randn(S, dtype=torch.float64, generator=g0).to(dev)
    b_leaf = torch.empty_like(rhs_leaf)
    b0 = torch.empty_like(rhs0)

    step_paths = []

    def step():
        kkt.factorize(diag, xd0)
        step_paths.clear()
        for _ in range(R_SOLVES):
            b_leaf.copy_(rhs_leaf)
            b0.copy_(rhs0)
            kkt.solve_compressed(b0, b_leaf)
            step_paths.append(kkt.last_solve_path())   # (a host-side field: no synchronisation)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    waits0 = pa.host_wait_count()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    host_waits_per_step = (pa.host_wait_count() - waits0) / max(a.steps, 1)   # (inside the library: csrc/common.h)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- phase table + rooflines: ONE instrumented step (1 factorize + 4 solveCompressed) after the timed region, HIP events on
    #      the streams the work runs on (pips_hip_batch_get_timing / pips_hip_kkt_get_timing)
    bt.set_timing(True)
    aug0 = bt.info().get("augmented_passes", 0)
    barrier()
    t_instr = time.perf_counter()
    step()
    torch.cuda.synchronize()
    bt.sync()
    t_instr = (time.perf_counter() - t_instr) * 1e3
    aug_passes = bt.info().get("augmented_passes", 0) - aug0    # passes of this step that swept the augmented factor (border rows included)
    tm = bt.get_timing()
    tk = kkt.get_timing()
    bt.set_timing(False)
    info = bt.info()
    nb_blocks = len(blocks)
    upd_ms, upd_launches = tm["tail_update"]
    # per-block tail sizes are statistically equal; use the exact aggregate from the symbolic phase
    m_avg = info["m"] / nb_blocks
    # border rows that ride along = the NON-EMPTY border columns of a block (info["nb"]), not S: on configurations whose blocks
    # touch only part of the linking columns the difference is large (configs[4] share: S = 16000, ~3700 non-empty per block)
    nb_avg = info["nb"] / nb_blocks
    n_solve_once = max(tm["solve_tail"][1], 1)

    def group(name, kernel, bound, ms_, launches, work, note=None):
        """work: algorithmic flops (mfma) or bytes (hbm) of the group per step"""
        peak, unit, scale = (PEAK_FP64_MFMA_TFLOPS, "TFLOP/s", 1e12) if bound == "mfma" else (PEAK_HBM_GBS, "GB/s", 1e9)
        ach = work / (ms_ * 1e-3) / scale if ms_ > 0 else 0.0
        g = {"group": name, "kernel": kernel, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
             "frac": round(ach / peak, 4), "ms_per_step": round(ms_, 3), "launches_per_step": int(launches),
             "avg_launch_ms": round(ms_ / max(launches, 1), 4),
             ("algorithmic_flops_per_step" if bound == "mfma" else "algorithmic_bytes_per_step"): work,
             ("algorithmic_flops_per_launch" if bound == "mfma" else "algorithmic_bytes_per_launch"): work / max(launches, 1)}
        if note:
            g["note"] = note
        return g

    upd_flops = nb_blocks * update_kernel_algorithmic_flops(int(round(m_avg)), int(round(nb_avg)))
    groups = [
        group("tail update", "k_tile_gemm_bal<0> (v_mfma_f64_4x4x4_4b_f64)", "mfma", upd_ms, upd_launches, upd_flops,
              "sum_j 2 (128 j) [tc below + tc (tc + 1) / 2 + nb tc] per block (DESIGN.md 4)"),
        group("Schur SYRK", "k_tile_gemm_bal<2>", "mfma", tm["schur"][0], tm["schur"][1], nb_blocks * nb_avg * (nb_avg + 1) * m_avg,
              "nb (nb + 1) m per block"),
        # bytes_F = 8 (nnz K + nnz L) + 4 (nnz K + nnz_idx L), here for the part of L the sparse head produces
        group("sparse head", "k_front<*> + k_head_factor_simple" if info.get("multifrontal_head") else "k_head_factor*", "hbm", tm["head"][0], tm["head"][1],
              8.0 * (info["nnzK"] + info["nnzL_head"]) + 4.0 * (info["nnzK"] + info["head_row_indices"]),
              "8 (nnz K + nnz L_head) + 4 (nnz K + row indices of the head supernodes): read K once, write L once"),
    ] + ([group("dense root", "k_tile_gemm<3> + k_tile_diag + k_tile_gemm<1>", "mfma", tk["root_factor"][0], 1, S ** 3 / 3.0, "S^3 / 3")]
         if not sparse_root else []) + [
        # a solve with K_i reads the rows of K of the factor twice (16 bytes per entry and pass); a pass over the AUGMENTED factor also its
        # border rows (head panels / border-row arena, the tails' border rows)
        group("leaf solve sweeps", "k_leaf_fwd_gather / k_head_fwd_chain / k_tail_rows_fwd / k_tail_rows_bwd / k_head_bwd_chain / k_leaf_bwd", "hbm",
              tm["solve_head_fwd"][0] + tm["solve_tail"][0] + tm["solve_head_bwd"][0], n_solve_once,
              16.0 * (n_solve_once * (info["nnzL"] - info.get("nnzL_border", 0)) + aug_passes * (info.get("nnzL_border", 0) + info.get("tail_border_entries", 0))),
              "16 bytes per entry of L and pass (forward + backward): the rows of K in every pass, the border rows in the passes over the augmented factor "
              f"({aug_passes} of {n_solve_once})"),
    ]
    dominant = max(groups, key=lambda g: g["ms_per_step"])
    roofline = dict(dominant)
    if dominant["group"] == "tail update":
        traffic = update_kernel_traffic(n_blocks_total, n_i, S, world)
    elif dominant["group"] == "sparse head":
        traffic = head_traffic(shape)
    elif dominant["group"] == "leaf solve sweeps":
        traffic = head_traffic(shape, "solve_hbm_bytes_per_step")
    else:
        traffic = None
    # per launch, like `achieved` (the rocprofv3 average duration of the kernel is ms_per_step / launches_per_step)
    roofline["traffic"] = (traffic / dominant["launches_per_step"] if traffic and dominant["launches_per_step"] else None)
    roofline["traffic_per_step"] = traffic
    # every phase of the step.  Top level: the partition of pips_hip_kkt_factorize / pips_hip_kkt_solve_compressed (4 solves);
    # below it what the leaf engine reports for its own part.  The root factorisation runs on a stream of its own beside the first
    # Lsolve's leaf solve; only what exceeds that solve is on the critical path.
    top = {k: round(v[0], 3) for k, v in tk.items()}
    # what the main stream waited for the root: the join before the first Dsolve where the root has a stream of its own (measured by
    # events around the join, phase root_wait), the whole factorisation where it sits on the main stream
    root_exposed = tk["root_wait"][0] + tk["root_factor_main_stream"][0]
    accounted = sum(tk[k][0] for k in ("diag_zero", "leaf_factor", "reduce", "finalize", "root_factor_main_stream", "lsolve_leaf", "lsolve_border_reduce",
                                       "root_wait", "dsolve", "ltsolve", "solve_check", "combine"))
    phase_ms = {"step": top, "root_factor_exposed": round(root_exposed, 3), "accounted": round(accounted, 3),
                "host_waits_per_step": round(host_waits_per_step, 2),   # how often a timed step made the host wait for the device inside the library
                "instrumented_step_wall": round(t_instr, 3),   # the one step the phases were taken from, host clock around it (events and two extra waits inside)
                "leaf_factor": {k: round(tm[k][0], 3) for k in ("scatter", "head", "tail_update", "tail_diag", "tail_trsm", "schur")},
                "leaf_solves": {k: round(tm[k][0], 3) for k in ("solve_permute", "solve_head_fwd", "solve_tail", "solve_head_bwd", "solve_refine")},
                "leaf_solve_passes": n_solve_once, "leaf_solve_passes_augmented": aug_passes}
    collective = None
    if use_dist:
        # bytes of the Schur reduction: the packed triangle of the dense root, or the value array of the sparse root's pattern
        payload = 8.0 * kkt.schur_sparse_nnz() if sparse_root else 8.0 * S * (S + 1) / 2
        P = max(world, 1)
        tot = tk["reduce_panels"][0] if tk["reduce_panels"][1] > 0 else tk["reduce"][0]
        collective = {"ranks_seen": ranks_seen, "schur_reduce_exposed_ms": round(tk["reduce"][0], 3), "schur_reduce_total_ms": round(tot, 3),
                      "schur_panels": tk["reduce_panels"][1] or 1,
                      "overlapped_fraction": (round(1.0 - tk["reduce"][0] / tot, 3) if tk["reduce_panels"][1] > 0 and tot > 0 else 0.0),
                      "b0_reduce_ms_in": "phase_ms.step.lsolve_border_reduce (border product + all-reduce of S doubles, x4)",
                      "payload_bytes_per_step": payload + R_SOLVES * 8.0 * S,
                      "wire_bytes_per_gpu_per_step": 2.0 * (P - 1) / P * (payload + R_SOLVES * 8.0 * S)}

    if rank == 0:
        ms = dt / a.steps * 1e3
        out = {
            "metric": "KKT factor+solve per IPM iter/sec, N-block arrowhead LP",
            "value": round(world * a.steps / dt, 4), "unit": f"{bpg}-block work units/s (1 factorize + 4 solveCompressed)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(a.family, world, bpg, n_i, my_i, a.rho, S, whole, myl),
                       "shape": shape, "deterministic": bool(os.environ.get("PIPS_HIP_DETERMINISTIC") not in (None, "", "0")),
                       "family": (a.family if a.family == "random" else "time-coupled (surrogate for SURVEY 8d config 4 = BASELINE configs[3]: "
                                  "the random generator's fill at n_i = 50 000 gives dense factors, BASELINE.md)"), "root": ("sparse (CSR Schur complement, linking rows dissected around x0, one-block multifrontal engine)"
                                                    if sparse_root else "dense LDL^T"), "sparse_head": "multifrontal (k_front)" if info.get("multifrontal_head") else "scatter (FP64 atomics)",
                       # N > 1 runs the configs[2] shape, N = 1 configs[1]: per-GPU throughput of the two differs by the Schur dimension alone
                       # (the driver's efficiency against the N = 1 line mixes that in) - the same shape on ONE device, measured, for reference
                       **({"same_shape_on_one_gpu": same_shape_on_one_gpu(shape)} if world > 1 else {}),
                       "solves_per_unit": R_SOLVES, "collective": comm_kind, "leaf_refinement": "adaptive, <=2 steps, normwise backward error <= 1e-15 (steps taken in the last solve: %d)" % bt.last_refinement_steps(),
                       "solve_path": {0: "every solveCompressed: two leaf solves with adaptive refinement + the two sparse border products",
                                      1: "refined Lsolve; Ltsolve by one backward sweep of the augmented factor (no pivot perturbed, the refined Lsolve needed no step)",
                                      2: "first solveCompressed after a factorisation: refined leaf solve(s) - Lsolve, and Ltsolve too unless the border rows are thin enough for the "
                                         "backward sweep of the factor (the witness: no pivot perturbed, no refinement step needed); the others: one forward + one backward sweep of the augmented factor [L 0; L_b I] (DESIGN.md 2)",
                                      3: "every solveCompressed: one forward + one backward sweep of the augmented factor [L 0; L_b I], its result measured against the "
                                         "leaf rows (residual measure of the adaptive refinement within the tolerance, no pivot perturbed; a failed measure repeats the "
                                         "solve the refined way) - ways 2 in solve_paths_last_step are sweeps riding on an earlier measure (--solve-check-every) (DESIGN.md 2)"}[3 if step_paths and step_paths[0] == 3 else kkt.last_solve_path()],
                       "solve_paths_last_step": list(step_paths), "solve_checks": dict(zip(("measured", "failed"), kkt.solve_check_counts())),
                       "iter_per_s": round(a.steps / dt, 4),
                       "nnzL_per_gpu": info["nnzL"], "tail_dim_avg": round(m_avg, 1), "border_rows_avg": round(nb_avg, 1),
                       "factor_flops_per_gpu": info["flops_factor"] + info["flops_border"]},
            "roofline": roofline,
            "roofline_all": [{k: g[k] for k in ("group", "bound", "achieved", "peak", "unit", "frac", "ms_per_step", "launches_per_step")} for g in groups],
            "phase_ms": phase_ms,
        }
        if collective:
            out["collective"] = collective
    # The loop's handles go down HERE, before the end-to-end IPM and the CPU baseline: with the HIP runtime alive and the streams idle (not from
    # __del__ during interpreter shutdown), and so that the IPM runs in a process without the loop's second streams (measured: with the loop's
    # root stream still alive the IPM of the 256-block chain takes 2.73 - 2.75 s instead of 2.46 - 2.47 s, tools/ab_async_root.sh)
    torch.cuda.synchronize()
    kkt.close()
    bt.close()
    if comm is not None:
        comm.close()
    if rank == 0:
        # the watchdog counts per phase: the timed loop, the CPU baseline and the end-to-end IPM each get the full time (one clock over all of
        # them would drop the bench line of a large run whose optional sections are long)
        import faulthandler
        wd = float(os.environ.get("PIPS_BENCH_WATCHDOG", "3000"))
        if not a.no_cpu_baseline and world == 1:
            faulthandler.dump_traceback_later(wd, exit=True)
            try:
                out["cpu_baseline"] = cpu_baseline(pa, a.seed, n_i, my_i, n0, myl, a.rho, n_blocks_total, bpg,
                                                   fam_blocks[0] if fam_blocks is not None else None, whole_block=a.steps >= 20)
            except Exception as e:  # the baseline must never break the bench line
                out["cpu_baseline"] = {"value": None, "unit": f"{bpg}-block work units/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        if not a.no_ipm and world == 1:
            faulthandler.dump_traceback_later(wd, exit=True)
            # "report units/s, and separately end-to-end IPM iterations/s of a full solve with the host driver" - the
            # device-resident harness on the LP of the same generator and shape, outside the timed region of the metric
            try:
                if sparse_root:   # the harness takes the same root as the metric's run
                    os.environ["PIPS_IPM_SPARSE_ROOT"] = "1"
                out["ipm_end_to_end"] = ipm_end_to_end(pa, a.seed, n_blocks_total, n_i, my_i, n0, myl, a.rho, fam_blocks, fam_F0)
            except Exception as e:
                out["ipm_end_to_end"] = {"error": str(e)}
        # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a pipe: push it out
        # first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _leave():
    """The line is out and the handles are closed: the process leaves through the interpreter's and the runtimes' normal exit handlers
    (round 4 skipped them with os._exit after ONE run of this script had never returned past its JSON line; 164 runs through the normal exit in
    round 5 - tools/stress_exit.sh, default and asynchronous root, small and full size - all ended).  Kept as a net: an
    alarm with the default disposition - the kernel ends the process if the teardown stalls for a minute (nothing of the interpreter is needed
    for that, unlike a watchdog thread, which finalisation freezes)."""
    import signal
    sys.stdout.flush()
    sys.stderr.flush()
    tools = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    if "rocprof" in tools:       # (a profiler writes its results in its own exit handler, which may take long)
        return
    signal.signal(signal.SIGALRM, signal.SIG_DFL)
    signal.alarm(60)


if __name__ == "__main__":
    # a run that stops making progress says where: every thread's Python stack on stderr, then the process exits with status 1 (the driver's bench finishes
    # within minutes; PIPS_BENCH_WATCHDOG=<seconds> for the stress runs of tools/stress_exit.sh)
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("PIPS_BENCH_WATCHDOG", "3000")), exit=True)
    main()
    faulthandler.cancel_dump_traceback_later()
    _leave()
