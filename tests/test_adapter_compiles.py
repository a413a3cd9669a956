"""The reference-side bindings of INTEGRATION.md (examples/adapter/HipLdlSolver.h, HipDenseLdlSolver.h: the DoubleLinearSolver
subclasses a PIPS-IPM++ maintainer adds) are syntax-checked against the reference's own headers - every `override` must match
the interface they replace (LinearSolvers/DoubleLinearSolver.h:24-72).  Needs the reference tree and an mpi.h; skipped where
they are absent (the GPU box)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CORE = "/root/reference/PIPS-IPM/Core"
MPI_INC = next((d for d in ("/opt/conda/include", "/usr/include/x86_64-linux-gnu/mpich", "/usr/lib/x86_64-linux-gnu/openmpi/include")
                if os.path.exists(os.path.join(d, "mpi.h"))), None)


@pytest.mark.skipif(not os.path.isdir(REF_CORE) or MPI_INC is None or shutil.which("g++") is None,
                    reason="needs the reference tree, mpi.h and g++")
def test_adapters_match_the_reference_interface(tmp_path):
    tu = tmp_path / "adapter_tu.cpp"
    tu.write_text('#include <type_traits>\n#include "HipLdlSolver.h"\n#include "HipDenseLdlSolver.h"\n#include "mpi_allreduce_callback.h"\n'
                  "// both are abstract-free: instantiable once a matrix exists\n"
                  "static_assert(!std::is_abstract<HipLdlSolver>::value && !std::is_abstract<HipDenseLdlSolver>::value, \"pure virtuals left\");\n")
    inc = ["-I" + d for d in sorted(p for p in glob.glob(REF_CORE + "/**/", recursive=True))]
    # -fpermissive: the reference itself needs it (two-phase lookup order in Utilities/pipsdef.h:540)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-fpermissive", "-w", "-I" + REF_CORE] + inc + ["-I" + MPI_INC, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "examples", "adapter"), str(tu)]
    run = subprocess.run(cmd, capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-4000:]


def test_integration_md_shows_the_same_adapter():
    """INTEGRATION.md quotes the leaf adapter: the quoted class body and the file must not drift apart."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "examples", "adapter", "HipLdlSolver.h")).read()
    start = md.index("class HipLdlSolver : public DoubleLinearSolver {")
    body = md[start:md.index("```", start)].strip()
    assert body in src


GMSPIPS_INC = "/root/reference/PIPS-IPM/Drivers/gams/gmspips"


@pytest.mark.skipif(not os.path.exists(os.path.join(GMSPIPS_INC, "gmspipsio.h")) or shutil.which("gcc") is None,
                    reason="needs the reference tree and gcc")
def test_readblock_shim_fills_the_reference_struct(tmp_path):
    """examples/adapter/read_block_compat.c implements `readBlock` with the reference's signature and GMSPIPSBlockData_t
    (both from the reference's own gmspipsio.h) on top of pips_gdx_read_block.  Built here against that header, it reads every
    block file of the 26 known-answer instances; each field must equal the committed fixture (= what the Python reader gives)."""
    import json
    import numpy as np
    lib = os.path.join(ROOT, "pips-ipmpp_amd")
    exe = str(tmp_path / "read_block_dump")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I" + GMSPIPS_INC, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "adapter", "read_block_compat.c"), os.path.join(ROOT, "examples", "adapter", "read_block_dump.c"),
           "-L" + lib, "-lpipship", "-Wl,-rpath," + lib]
    if os.path.isdir("/opt/rocm/lib"):
        cmd += ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd + ["-o", exe])
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "gamssmall.json")))["instances"]
    n_files = 0
    for d in data:
        for k, want in enumerate(d["blocks"]):
            out = subprocess.run([exe, f"/root/reference{d['source']}{k}.gdx", str(d["num_blocks"]), str(k)], capture_output=True, text=True)
            assert out.returncode == 0, out.stdout + out.stderr
            got = {line.split(":")[0]: [float(t) for t in line.split(":")[1].split()] for line in out.stdout.strip().splitlines()}
            nnz = {m: (len(want[m]["val"]) if want[m] else 0) for m in ("A", "B", "C", "D", "BL", "DL")}
            assert got["counts"] == [want["n0"], want["ni"], want["mA"], want["mC"], want["mBL"], want["mDL"]] + [nnz[m] for m in ("A", "B", "C", "D", "BL", "DL")]
            for f in ("c", "xlow", "xupp", "ixlow", "ixupp", "b", "clow", "cupp", "iclow", "icupp", "bL", "dlow", "dupp", "idlow", "idupp"):
                assert np.array_equal(got[f], np.asarray(want[f], dtype=float)), (d["name"], k, f)
            for m in ("A", "B", "C", "D", "BL", "DL"):
                if want[m] is None:
                    assert got["rm" + m] == [] and got["ci" + m] == [] and got["val" + m] == []
                else:
                    assert got["rm" + m] == want[m]["rowptr"] and got["ci" + m] == want[m]["colidx"] and got["val" + m] == want[m]["val"], (d["name"], k, m)
            n_files += 1
    assert n_files == sum(d["num_blocks"] for d in data) == 108
