"""The C ABI is usable from plain C: examples/leaf_solve.c compiles as C11 against include/pips_hip.h, links libpipship.so
and (on the GPU box) factorises and solves a leaf block with the right inertia and residual."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pips-ipmpp_amd")


def _build(tmp_path, name="leaf_solve"):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L" + LIBDIR, "-lpipship", "-Wl,-rpath," + LIBDIR]
    if os.path.isdir("/opt/rocm/lib"):
        cmd += ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    cmd += ["-lm", "-o", exe]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_example_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe, exe2 = _build(tmp_path), _build(tmp_path, "ipm_solve")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    assert subprocess.run([exe]).returncode == 2   # "no GPU": no silent CPU path
    assert subprocess.run([exe2]).returncode == 2


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_example_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "inertia (2000, 1000, 0)" in out.stdout
    assert "Schur term: chunk loop vs pips_hip_ldl_factor_schur" in out.stdout   # level 1 (K4-K6 on the host) == level 1.5


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_ipm_example_runs_on_the_gpu(tmp_path):
    """examples/ipm_solve.c: the whole device path (pips_ipm_*: leaf and root factorisations, solveCompressed, the IPM loop)
    driven from plain C; the program checks feasibility, sign and duality gap of the returned solution itself."""
    exe = _build(tmp_path, "ipm_solve")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "status 0 after" in out.stdout
