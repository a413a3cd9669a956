"""bench.py's one-line contract on small workloads (the driver reads this line): required keys, the self-accounting objects, the root
the time-coupled family takes, and the N > 1 control flow with two processes sharing the one GPU (gloo, host-staged reductions:
validation, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "phase_ms"]


def _run(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    return json.loads(lines[0])


def _check_line(d, n_gpus):
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 / n_gpus - 1.0) < 0.02      # units/s = ranks / (s per step)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    step = d["phase_ms"]["step"]
    # 4 solveCompressed: two leaf-solve passes each, or one where the sweeps of the augmented factor serve (each of them measured against
    # the leaf rows; several ranks exchange the outcome)
    passes, aug = d["phase_ms"]["leaf_solve_passes"], d["phase_ms"]["leaf_solve_passes_augmented"]
    assert step["leaf_factor"] > 0 and step["lsolve_leaf"] > 0 and 4 <= passes <= 8 and aug <= 4
    assert set(d["config"]["solve_paths_last_step"]) <= {0, 1, 3}          # never way 2 by default: no solve rides on another one's measure
    assert len(d["config"]["solve_paths_last_step"]) == 4
    # the phases of the instrumented step are disjoint pieces of the main stream's time: their sum cannot exceed the host clock around that
    # step, and what they leave out is launch gaps and the host's own work (large on these small workloads; within 2 % on the BASELINE
    # shapes: profiles/r5_bench_other_configs.jsonl)
    ph = d["phase_ms"]
    assert 0.0 < ph["accounted"] <= 1.02 * ph["instrumented_step_wall"], ph
    assert abs(ph["root_factor_exposed"] - (ph["step"]["root_wait"] + ph["step"]["root_factor_main_stream"])) < 2e-3
    assert ph["step"]["root_wait"] <= ph["step"]["root_factor"] + 0.05      # the join waits for (part of) the root factorisation, nothing else
    # host waits inside the library per timed step (csrc/common.h counts every synchronisation and blocking copy): the measure of each of the
    # four solves reads one number back, the factorisation one or two - a per-block or per-level wait would show here
    # (one rank: a host-supplied all-reduce waits for the stream at every collective by construction)
    assert ph["host_waits_per_step"] >= 1 and (n_gpus > 1 or ph["host_waits_per_step"] <= 12), ph


def test_default_family_small():
    d = _run(["--blocks-per-gpu", "4", "--n", "1000", "--schur-dim", "200", "--rho", "0.01", "--steps", "2", "--warmup", "1", "--no-ipm"])
    _check_line(d, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and (cb["value"] is None or cb["value"] > 0)
    assert d["config"]["root"] == "dense LDL^T"
    assert any(g["group"] == "dense root" for g in d["roofline_all"])


@pytest.mark.parametrize("root", ["auto", "dense"])
def test_time_coupled_family_small(root):
    d = _run(["--family", "time-coupled", "--blocks-per-gpu", "8", "--n", "2000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--root", root])
    _check_line(d, 1)
    assert d["config"]["family"].startswith("time-coupled (surrogate")
    assert d["config"]["root"].startswith("sparse" if root == "auto" else "dense")
    ipm = d["ipm_end_to_end"]            # the harness on the same family (and the same root)
    assert ipm["status"] == 0 and ipm["rel_residual"] < 1e-7
    assert 0 < ipm["host_waits_per_iteration"] <= 60, ipm                 # (23 - 25 measured; 311 while every per-block inertia query waited on its own)


def test_two_processes_share_the_gpu():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(29700 + os.getpid() % 200)]
    d = _run(["--gpus", "2", "--family", "time-coupled", "--blocks-per-gpu", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
              "--no-ipm"], env={"PIPS_BENCH_SHARE_GPU": "1"}, launcher=launcher)
    _check_line(d, 2)
    c = d["collective"]
    assert c is not None and c["payload_bytes_per_step"] > 0
    assert d["config"]["root"].startswith("sparse") and "gloo" in d["config"]["collective"]


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher: bench.py starts one process per rank before it touches the GPU, the line says n_gpus = 2
    and the communicator the factorisation reduces through counted two ranks (here both on device 0: validation, not a measurement).
    Without --schur-dim the N > 1 default is BASELINE configs[2]'s Schur dimension."""
    d = _run(["--gpus", "2", "--blocks-per-gpu", "2", "--n", "1000", "--rho", "0.01", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-ipm"],
             env={"PIPS_BENCH_SHARE_GPU": "1"})
    _check_line(d, 2)
    assert d["collective"]["ranks_seen"] == 2
    assert "Schur dim 4000" in d["config"]["workload"]


def test_eight_ranks_hold_one_fixed_chain():
    """configs[3] as a problem that shards: `--gpus 8 --family time-coupled` on a 16-block chain with Schur dimension 200 (reduced --n; the
    eight ranks share device 0) - the Schur dimension is the chain's, not a function of the rank count, every rank holds its two blocks,
    the sparse root's pattern is assembled from the lists the ranks exchange, and the factorisation's communicator counted eight ranks."""
    args = ["--family", "time-coupled", "--chain-blocks", "16", "--schur-dim", "200", "--blocks-per-gpu", "2", "--n", "600", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-ipm"]
    d8 = _run(["--gpus", "8"] + args, env={"PIPS_BENCH_SHARE_GPU": "1"})
    _check_line(d8, 8)
    assert d8["collective"]["ranks_seen"] == 8
    assert "Schur dim 200: all of it on 8 GPU(s), 2 blocks/GPU" in d8["config"]["workload"]
    assert d8["config"]["shape"] == {"family": "time-coupled", "blocks_per_gpu": 2, "n": 600, "schur_dim": 200, "chain_blocks": 16}
    assert d8["config"]["root"].startswith("sparse")
    # four ranks run the first eight blocks of the SAME chain (its Schur dimension still 200) with the linking rows those blocks touch
    d4 = _run(["--gpus", "4"] + args, env={"PIPS_BENCH_SHARE_GPU": "1"})
    _check_line(d4, 4)
    assert d4["collective"]["ranks_seen"] == 4
    assert "Schur dim 200: blocks 0..7 on 4 GPU(s)" in d4["config"]["workload"] and "(Schur dim 151)" in d4["config"]["workload"]   # 95 + floor(8 * 105 / 15)
    # more ranks than the chain has blocks for: refused
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--family", "time-coupled", "--chain-blocks", "4", "--blocks-per-gpu", "8",
                          "--n", "600", "--steps", "1"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "exceed the chain" in out.stderr


def test_gpus_flag_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("PIPS_BENCH_SHARE_GPU", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 2 and "device(s)" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
