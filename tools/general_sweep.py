"""Development aid: random general-form block LPs (tests/general_lp_gen.py: all bound kinds, ranged rows, free and fixed
variables) through standard_form + the device harness against HiGHS.   python tools/general_sweep.py [n_cases] [first_seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from scipy.optimize import linprog  # noqa: E402
import pips_ipmpp_amd as pa  # noqa: E402
from pips_ipmpp_amd.standard_form import block_standard_form, general_lp  # noqa: E402
from tests.general_lp_gen import random_block_lp  # noqa: E402


def shape(seed):
    rng = np.random.default_rng(seed)
    nb, n0, ni = int(rng.integers(2, 6)), int(rng.integers(3, 9)), int(rng.integers(6, 30))
    return nb, n0, ni, int(rng.integers(2, min(ni, 10))), int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 4))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n_cases):
        bl = random_block_lp(seed, *shape(seed))
        c, Aeq, beq, Aub, bub, bounds = general_lp(bl)
        ref = linprog(c, A_eq=Aeq, b_eq=beq, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
        if ref.status != 0:
            print(f"seed {seed}: HiGHS status {ref.status}, skipped")
            continue
        native = bool(os.environ.get("NATIVE_FREE"))
        sf = block_standard_form(bl, split_free=not native)
        ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=1e-9)
        if native:
            ipm.set_free_variables(sf["bounded_mask"])
            if os.environ.get("FREE_REG"):
                ipm.set_option("FREE_VARIABLE_PROXIMAL_TERM", float(os.environ["FREE_REG"]))
        res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8, verbose=int(os.environ.get("VERB", "0")))
        st = ipm.stats()
        err = abs(res["objective"] + sf["offset"] - ref.fun) / max(1.0, abs(ref.fun))
        ok = res["status"] == 0 and err < 1e-6
        bad += not ok
        print(f"seed {seed}: shape {shape(seed)} std {sf['A'].shape}  status {res['status']} it {res['iterations']:3d} rel.obj.err {err:.1e} "
              f"fact {st['factorizations']} reg {st['regularised_repeats']} sc {st['solve_compressed']} {'' if ok else '<-- CHECK'}", flush=True)
        ipm.close()
    print("failures:", bad)


if __name__ == "__main__":
    main()
