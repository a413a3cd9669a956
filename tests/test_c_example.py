"""The C ABI is usable from plain C: examples/leaf_solve.c compiles as C11 against include/pips_hip.h, links libpipship.so
and (on the GPU box) factorises and solves a leaf block with the right inertia and residual."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pips-ipmpp_amd")


def _build(tmp_path):
    exe = str(tmp_path / "leaf_solve")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "leaf_solve.c"), "-L" + LIBDIR, "-lpipship", "-Wl,-rpath," + LIBDIR]
    if os.path.isdir("/opt/rocm/lib"):
        cmd += ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    cmd += ["-lm", "-o", exe]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_example_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    assert subprocess.run([exe]).returncode == 2   # "no GPU": no silent CPU path


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_example_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "inertia (2000, 1000, 0)" in out.stdout
