"""Development aid: the device IPM harness on the reference's GAMSsmall known-answer instances (tests/golden/gamssmall.json)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pips_ipmpp_amd as pa
from pips_ipmpp_amd.standard_form import block_standard_form
data = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gamssmall.json")))["instances"]
only = sys.argv[1] if len(sys.argv) > 1 else None
reg = float(os.environ.get("REG", "1e-9"))
for inst in data:
    if only and only not in inst["name"]:
        continue
    native_free = bool(os.environ.get("NATIVE_FREE"))
    sf = block_standard_form(inst["blocks"], split_free=not native_free)
    ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=reg)
    if native_free:
        ipm.set_free_variables(sf["bounded_mask"])
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8, verbose=int(os.environ.get("VERB", "2")) if only else 0)
    st = ipm.stats()
    print(f"{inst['name'][:50]:50s} exp {inst['expected_objective']:8.2f} ({inst['expected_iterations']:2d})  status {res['status']} it {res['iterations']:3d} "
          f"obj {res['objective'] + sf['offset']:12.6f} mu {res['mu']:.1e} r {res['rnorm']:.1e}  fact {st['factorizations']} reg {st['regularised_repeats']} sc {st['solve_compressed']}", flush=True)
