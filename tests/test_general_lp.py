"""Random block LPs with every bound kind (lower / upper / boxed / fixed / free variables, <= / >= / ranged / equality rows,
own and linking rows - tests/general_lp_gen.py) in the reference's reader layout: the conversion to the harness' standard
form is checked against HiGHS on the bounded form (CPU), the device harness against HiGHS (GPU; tools/general_sweep.py ran
300 seeds without a failure)."""
import numpy as np
import pytest
from scipy.optimize import linprog

from pips_ipmpp_amd.standard_form import block_standard_form, general_lp, recover_solution
from tests.general_lp_gen import random_block_lp


def _shape(seed):
    rng = np.random.default_rng(seed)
    nb, n0, ni = int(rng.integers(2, 6)), int(rng.integers(3, 9)), int(rng.integers(6, 30))
    return nb, n0, ni, int(rng.integers(2, min(ni, 10))), int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 4))


@pytest.mark.parametrize("seed", range(8))
def test_standard_form_keeps_the_optimum(seed):
    bl = random_block_lp(seed, *_shape(seed))
    c, A_eq, b_eq, A_ub, b_ub, bounds = general_lp(bl)
    ref = linprog(c, A_eq=A_eq, b_eq=b_eq, A_ub=A_ub, b_ub=b_ub, bounds=bounds, method="highs")
    assert ref.status == 0
    sf = block_standard_form(bl)
    r2 = linprog(sf["c"], A_eq=sf["A"], b_eq=sf["b"], bounds=(0, None), method="highs")
    assert r2.status == 0 and abs(r2.fun + sf["offset"] - ref.fun) <= 1e-8 * max(1.0, abs(ref.fun))
    x = np.concatenate(recover_solution(sf, r2.x))
    assert np.abs(A_eq @ x - b_eq).max() < 1e-7 and (A_ub @ x - b_ub).max() < 1e-7
    assert all((lo is None or xi >= lo - 1e-7) and (up is None or xi <= up + 1e-7) for xi, (lo, up) in zip(x, bounds))
    kinds = {(lo is None, up is None, lo is not None and lo == up) for lo, up in bounds}
    assert len(kinds) >= 3   # the instance really mixes bound kinds


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 116))
def test_device_harness_on_general_form_lps(seed):
    import pips_ipmpp_amd as pa
    bl = random_block_lp(seed, *_shape(seed))
    c, A_eq, b_eq, A_ub, b_ub, bounds = general_lp(bl)
    ref = linprog(c, A_eq=A_eq, b_eq=b_eq, A_ub=A_ub, b_ub=b_ub, bounds=bounds, method="highs")
    assert ref.status == 0
    sf = block_standard_form(bl)
    ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=1e-9)
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
    assert res["status"] == 0, res
    assert abs(res["objective"] + sf["offset"] - ref.fun) <= 1e-6 * max(1.0, abs(ref.fun)), (res, ref.fun)
    y, _ = ipm.solution()
    x = np.concatenate(recover_solution(sf, y))
    scale = max(1.0, np.abs(b_eq).max())
    assert np.abs(A_eq @ x - b_eq).max() < 1e-6 * scale and (A_ub @ x - b_ub).max() < 1e-6 * scale
