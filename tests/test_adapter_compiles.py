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
