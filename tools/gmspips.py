"""Counterpart of the reference's `gmspips` driver (Drivers/gams/gmspips/gmspips.cpp:  gmspips <numBlocks> <file stem> ...):
reads the per-block GDX files <stem>0.gdx .. <stem>{n-1}.gdx with the library's reader, hands the blocks to the device-resident
IPM as they are (bounds, two-sided rows, linking rows native on the device) and prints the objective.
   python tools/gmspips.py <numBlocks> <file stem> [mutol] [artol]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pips_ipmpp_amd as pa  # noqa: E402


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        return 2
    nblocks, stem = int(sys.argv[1]), sys.argv[2]
    mutol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6      # the reference's termination defaults (PIPSIPMppSolver.cpp:143-149)
    artol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
    blocks = [pa.capi.gdx_read_block(f"{stem}{k}.gdx", nblocks, k) for k in range(nblocks)]
    ipm = pa.GeneralIpmSolver(blocks, dual_reg=1e-9)
    res = ipm.solve(max_iter=200, mutol=mutol, artol=artol, verbose=1)
    itr = ipm.iterate()
    names = {0: "SUCCESSFUL_TERMINATION", 1: "MAX_ITS_EXCEEDED", 2: "NUMERICAL_BREAKDOWN", 3: "NUMERICAL_TROUBLES (best iterate)", 4: "INFEASIBLE (probably)"}
    print(f"status {names.get(res['status'], res['status'])}  iterations {res['iterations']}  objective {res['objective']:.10g}  dual objective {res['dual_objective']:.10g}")
    root = blocks[0]
    my0, myl, mz0, mzl = root["mA"], root["mBL"], root["mC"], root["mDL"]
    print("linking variables:", np.array2string(itr["x"][:root["n0"]], precision=6))
    print("multipliers of the linking rows: eq", np.array2string(itr["y"][my0:my0 + myl], precision=6), " ineq", np.array2string(itr["z"][mz0:mz0 + mzl], precision=6))
    print(f"complementarity pairs {ipm.n_pairs}, statistics {ipm.stats()} {ipm.stats2()}")
    return 0 if res["status"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
