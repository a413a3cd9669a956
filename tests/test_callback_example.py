"""Known-answer test on the reference's own smoke instance (Drivers/CallbackExample, README.md:76): objective 14."""
import json
import os

import numpy as np
import pytest

from tests.lp_standard_form import standard_form

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    data = json.load(open(os.path.join(HERE, "callback_example.json")))
    return data, standard_form(data)


def test_highs_and_ipm_oracle_reproduce_the_reference_objective():
    from scipy.optimize import linprog
    from oracle import ipm_oracle as io
    data, lp = _load()
    ref = linprog(lp["c"], A_eq=lp["A"], b_eq=lp["b"], bounds=(0, None), method="highs")
    assert ref.status == 0 and abs(ref.fun - data["expected_objective"]) < 1e-9
    # the instance has a redundant equality (2 x0[0] = 2 appears twice): the CPU restatement needs the dual regularisation
    o = io.solve_lp(lp["A"], lp["b"], lp["c"], 100, 1e-8, 1e-8, dual_reg=1e-9)
    assert o["status"] == 0
    assert abs(o["objective"] - data["expected_objective"]) < 1e-6


@pytest.mark.gpu
def test_device_harness_reproduces_the_reference_objective():
    import pips_ipmpp_amd as pa
    data, lp = _load()
    ipm = pa.IpmSolver(lp["n0"], lp["myl"], lp["blocks"], lp["F0"], lp["c"], lp["b"], dual_reg=1e-9)
    res = ipm.solve(max_iter=100, mutol=1e-8, artol=1e-8)
    assert res["status"] == 0, res
    assert abs(res["objective"] - data["expected_objective"]) < 1e-6, res
    x, _ = ipm.solution()
    assert np.linalg.norm(lp["A"] @ x - lp["b"], np.inf) < 1e-6
