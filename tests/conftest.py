import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # No test of this suite needs more than a few minutes (the longest generates 256 x 50 000-variable blocks).  If the
    # pytest-timeout plugin is there, a test that hangs - a wedged device, a kernel that never returns - ends the session with a
    # stack dump after 15 minutes instead of holding the GPU box until somebody else's limit kills it.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900
        config.option.timeout_method = "thread"


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once if the toolchain is here.  The product
    itself never builds or falls back at run time - a missing library is an error there (pips_ipmpp_amd/capi.py)."""
    lib = os.path.join(ROOT, "pips-ipmpp_amd", "libpipship.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if os.path.exists(lib) and os.path.exists(orc):
        return
    if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        import __graft_entry__ as entry
        entry.build()
