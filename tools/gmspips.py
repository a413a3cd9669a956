"""Counterpart of the reference's `gmspips` driver (Drivers/gams/gmspips/gmspips.cpp:  gmspips <numBlocks> <file stem> ...):
reads the per-block GDX files <stem>0.gdx .. <stem>{n-1}.gdx, solves the LP with the device-resident IPM and prints the
objective.   python tools/gmspips.py <numBlocks> <file stem> [mutol] [artol]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pips_ipmpp_amd as pa  # noqa: E402
from pips_ipmpp_amd.standard_form import block_standard_form, kkt_violation, recover_duals, recover_solution  # noqa: E402


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        return 2
    nblocks, stem = int(sys.argv[1]), sys.argv[2]
    mutol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6      # the reference's termination defaults (PIPSIPMppSolver.cpp:143-149)
    artol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
    blocks = [pa.capi.gdx_read_block(f"{stem}{k}.gdx", nblocks, k) for k in range(nblocks)]
    native_free = bool(os.environ.get("PIPS_NATIVE_FREE"))   # free variables as single columns without a complementarity pair
    sf = block_standard_form(blocks, split_free=not native_free)
    ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=1e-9)
    if native_free:
        ipm.set_free_variables(sf["bounded_mask"])
    res = ipm.solve(max_iter=200, mutol=mutol, artol=artol, verbose=1)
    y, duals_std = ipm.solution()
    x = recover_solution(sf, y)
    duals = recover_duals(sf, duals_std)
    names = {0: "SUCCESSFUL_TERMINATION", 1: "MAX_ITS_EXCEEDED", 2: "NUMERICAL_BREAKDOWN", 3: "NUMERICAL_TROUBLES (best iterate)", 4: "INFEASIBLE (probably)"}
    print(f"status {names.get(res['status'], res['status'])}  iterations {res['iterations']}  objective {res['objective'] + sf['offset']:.10g}")
    print("linking variables:", np.array2string(x[0], precision=6))
    print("marginals of the linking rows: eq", np.array2string(duals[0]["link_eq"], precision=6), " ineq", np.array2string(duals[0]["link_ineq"], precision=6))
    print(f"largest violation of the optimality conditions of the original problem: {kkt_violation(blocks, x, duals, tol=1e-5):.2e}")
    return 0 if res["status"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
