"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side analysis (ordering incl. the partial nested dissection,
symbolic factorisation, update segments, generator / assembly helpers): tools/asan_driver.cpp analyses 180 random and
time-coupled block variants.  GPU sanitizers are not available on the pool; this covers the pointer-heavy CPU code."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pips-ipmpp_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_analysis_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_driver")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + CSRC,
                            "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "asan_driver.cpp"),
                            os.path.join(CSRC, "order.cpp"), os.path.join(CSRC, "symbolic.cpp"), os.path.join(CSRC, "gen.cpp"),
                            "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "analysed 180 block variants" in run.stdout
    assert "runtime error" not in run.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_gdx_reader_survives_damaged_files_under_asan_ubsan(tmp_path):
    """csrc/gdx.cpp parses untrusted binary files: valid jacobian files (written by the test), every truncation of one of them and
    thousands of random byte flips go through pips_gdx_read_block and all accessors under AddressSanitizer + UBSan."""
    import numpy as np
    from tests.test_gamssmall import _synthetic_jacobian
    rng = np.random.default_rng(21)
    files = []
    for k in range(3):
        path = str(tmp_path / f"j{k}.gdx")
        _synthetic_jacobian(path, rng, 3)
        files.append(path)
    exe = str(tmp_path / "asan_gdx_driver")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fsanitize=float-cast-overflow", "-fno-omit-frame-pointer", "-I" + CSRC,
                            "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "asan_gdx_driver.cpp"),
                            os.path.join(CSRC, "gdx.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("toolchain without sanitizer runtimes")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, str(tmp_path / "scratch.gdx"), "3"] + files, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "3 valid files" in run.stdout and "runtime error" not in run.stderr
