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


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [100, 104, 109])
def test_native_free_variables_on_general_form_lps(seed):
    """Free variables kept as one column (no complementarity pair, dd = 0 as in the reference's computeDiagonals,
    LinearSystem.C:262-294) on the degenerate general-form family: optimum of HiGHS, original constraints satisfied."""
    import pips_ipmpp_amd as pa
    bl = random_block_lp(seed, *_shape(seed))
    c, A_eq, b_eq, A_ub, b_ub, bounds = general_lp(bl)
    ref = linprog(c, A_eq=A_eq, b_eq=b_eq, A_ub=A_ub, b_ub=b_ub, bounds=bounds, method="highs")
    sf = block_standard_form(bl, split_free=False)
    mask = sf["bounded_mask"]
    assert (mask == 0).sum() >= 2
    ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=1e-9)
    ipm.set_free_variables(mask)
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
    assert res["status"] == 0 and abs(res["objective"] + sf["offset"] - ref.fun) <= 1e-6 * max(1.0, abs(ref.fun))
    y, _ = ipm.solution()
    assert y[mask == 1].min() > -1e-8
    x = np.concatenate(recover_solution(sf, y))
    assert np.abs(A_eq @ x - b_eq).max() < 1e-6 * max(1.0, np.abs(b_eq).max())


@pytest.mark.gpu
def test_native_free_variables_follow_the_oracle():
    """A well-posed LP of the generator family in which a tenth of the variables that are positive at the optimum are declared
    free (dropping an inactive bound keeps the optimum): the device harness and the CPU restatement, both with free entries
    outside the complementarity terms, take the same path - iteration count, objective, and mu / objectives / sigma / step
    lengths per iterate."""
    import pips_ipmpp_amd as pa
    from oracle import ipm_oracle as io
    from tests.test_ipm_gpu import build_lp
    N, n_i, my_i, n0, myl = 3, 60, 30, 6, 5
    blocks, F0, c, b, A = build_lp(2031, N, n_i, my_i, n0, myl, 0.1)
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    rng = np.random.default_rng(3)
    cand = np.nonzero(ref.x > 0.1)[0]
    free = rng.choice(cand, size=max(4, len(cand) // 10), replace=False)
    mask = np.ones(A.shape[1])
    mask[free] = 0.0
    ref2 = linprog(c, A_eq=A, b_eq=b, bounds=[(0, None) if m else (None, None) for m in mask], method="highs")
    assert ref2.status == 0 and abs(ref2.fun - ref.fun) <= 1e-9 * abs(ref.fun)
    ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
    ipm.set_free_variables(mask)
    res = ipm.solve(max_iter=100, mutol=1e-9, artol=1e-8)
    assert res["status"] == 0 and abs(res["objective"] - ref.fun) <= 1e-9 * abs(ref.fun)
    trace = []
    o = io.solve_lp(A, b, c, 100, 1e-9, 1e-8, trace, bounded=mask)
    assert o["status"] == 0 and abs(o["iterations"] - res["iterations"]) <= 1
    tr = ipm.trace()
    n_cmp = min(len(trace), tr.shape[0]) - 1
    assert n_cmp >= 8
    for k in range(n_cmp):
        it, mu, rnorm, pobj, dobj, sigma, ap, ad = trace[k]
        # the device solves each system with BiCGStab to the reference's outer tolerance (1e-8 up to iteration 3, 1e-9 up to 7,
        # then 1e-10) around a preconditioner that carries the proximal term of the free entries; the CPU side solves exactly
        tol = 2e-3 if k < n_cmp - 4 else 2e-2
        scale = max(1.0, abs(pobj), abs(dobj))
        assert abs(tr[k, 0] - mu) <= tol * mu and abs(tr[k, 2] - pobj) <= tol * scale and abs(tr[k, 3] - dobj) <= tol * scale, (k, tr[k], trace[k])
        assert abs(tr[k, 4] - sigma) <= 10 * tol and abs(tr[k, 5] - ap) <= 10 * tol and abs(tr[k, 6] - ad) <= 10 * tol, (k, tr[k], trace[k])
    x, _ = ipm.solution()
    assert (x[free] > 0.05).all() and x[mask == 1].min() > -1e-9
