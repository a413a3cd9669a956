"""Runs one GAMSsmall instance many times on the device harness and prints the verbose log of the runs that go wrong (development aid)."""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_native_general_gpu import GAMSSMALL
import pips_ipmpp_amd as pa
name = sys.argv[1] if len(sys.argv) > 1 else "hier_approach_4blocks_2by3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
inst = [d for d in GAMSSMALL if d["name"] == name][0]
c = collections.Counter()
shown_good = False
for k in range(reps):
    ipm = pa.GeneralIpmSolver(inst["blocks"], dual_reg=1e-9)
    sys.stdout.flush()
    saved = os.dup(1)
    f = os.open("/tmp/flake.log", os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(f, 1)
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8, verbose=2)
    os.dup2(saved, 1); os.close(f); os.close(saved)
    ok = res["status"] == 0 and abs(res["objective"] - inst["expected_objective"]) < 1e-4 and res["iterations"] <= 1.1 * inst["expected_iterations"] + 1
    c[(res["status"], res["iterations"], ok)] += 1
    if not ok or not shown_good:
        print("=" * 30, "run", k, "ok" if ok else "BAD", res["status"], res["iterations"], ipm.stats())
        print(open("/tmp/flake.log").read()[:6000])
        shown_good = shown_good or ok
    ipm.close()
print(c)
