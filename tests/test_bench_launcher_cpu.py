"""bench.py --gpus N without a launcher (CPU side of the contract: no GPU here): the parent counts devices without initialising HIP and
refuses more ranks than devices; with PIPS_BENCH_SHARE_GPU=1 it starts N rank processes with the launcher's environment - each of which
then stops at "needs a GPU" on this box, which shows that N of them were started - and hands on a non-zero exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def test_more_ranks_than_devices_is_refused_before_any_rank_starts():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a multi-GPU box runs the ranks")
    out = _run(["--gpus", "2", "--steps", "1"], env={"PIPS_BENCH_SHARE_GPU": ""})
    assert out.returncode == 2 and "device(s)" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_ranks_are_started_with_the_launcher_environment():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: tests/test_bench_contract_gpu.py runs the real thing")
    out = _run(["--gpus", "3", "--steps", "1", "--blocks-per-gpu", "1", "--n", "200", "--schur-dim", "40"], env={"PIPS_BENCH_SHARE_GPU": "1"})
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") >= 1        # every rank gets as far as the device check (the others are stopped when one fails)
