"""Known-answer test on the reference's own smoke instance (Drivers/CallbackExample, README.md:76): objective 14."""
import json
import os

import numpy as np
import pytest

from pips_ipmpp_amd.standard_form import block_standard_form

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _blocks(data):
    """The callback-convention nodes of the fixture in the reader's block layout (GMSPIPSBlockData_t): x >= 0 everywhere,
    inequality rows bounded from above only."""
    n0 = data["nodes"][0]["n"]
    mBL, mDL = len(data["link_eq_rhs"]), len(data["link_ineq_upp"])
    out = []
    for k, nd in enumerate(data["nodes"]):
        n, mz = nd["n"], nd["mz"]
        out.append(dict(numBlocks=len(data["nodes"]), blockID=k, n0=n0, ni=n, mA=nd["my"], mC=mz, mBL=mBL, mDL=mDL, c=nd["c"],
                        xlow=[0.0] * n, xupp=[0.0] * n, ixlow=[1] * n, ixupp=[0] * n, b=nd["b"], clow=[0.0] * mz, cupp=nd["cupp"],
                        iclow=[0] * mz, icupp=[1] * mz, bL=data["link_eq_rhs"], dlow=[0.0] * mDL, dupp=data["link_ineq_upp"],
                        idlow=[0] * mDL, idupp=[1] * mDL, A=nd["A"], B=nd["B"], C=nd["C"], D=nd["D"], BL=nd["Bl"], DL=nd["Dl"]))
    return out


def _load():
    data = json.load(open(os.path.join(HERE, "callback_example.json")))
    lp = block_standard_form(_blocks(data))
    assert lp["offset"] == 0.0
    return data, lp


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
