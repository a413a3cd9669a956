"""Every GAMSsmall instance N times on the device harness (native route): counts the runs that miss the reference's test criteria."""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_native_general_gpu import GAMSSMALL
import pips_ipmpp_amd as pa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tot_bad = 0
for inst in GAMSSMALL:
    c = collections.Counter()
    for k in range(reps):
        ipm = pa.GeneralIpmSolver(inst["blocks"], dual_reg=1e-9)
        res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
        ok = res["status"] == 0 and abs(res["objective"] - inst["expected_objective"]) < 1e-4 and res["iterations"] <= 1.1 * inst["expected_iterations"] + 1
        c[(res["status"], res["iterations"], ok)] += 1
        tot_bad += not ok
        ipm.close()
    print(f"{inst['name']:45s} expected its {inst['expected_iterations']:3d}  {dict(c)}", flush=True)
print("bad runs:", tot_bad)
