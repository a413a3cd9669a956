"""CPU: the generalised IPM restatement (oracle/ipm_oracle.py: bounds of both kinds, inequality rows, root and linking rows) is
pinned against an independent LP solver (HiGHS) and against the reference's own known answers (the 26 GAMSsmall objectives,
Test/IntegrationTests/gamssmall_instance_data.txt) before the GPU tests use it as the checker; the BiCGStab restatement
(LinearSystem.C:550-798) returns the reference's status codes."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import ipm_oracle as io
from tests.general_lp_gen import random_block_lp

HERE = os.path.dirname(os.path.abspath(__file__))
GAMSSMALL = json.load(open(os.path.join(HERE, "golden", "gamssmall.json")))["instances"]


def _highs(d):
    from scipy.optimize import linprog
    C = d["C"]
    up, lo = d["icupp"] > 0, d["iclow"] > 0
    return linprog(d["c"], A_ub=sp.vstack([C[up], -C[lo]]) if C.shape[0] else None,
                   b_ub=np.concatenate([d["cupp"][up], -d["clow"][lo]]) if C.shape[0] else None,
                   A_eq=d["A"] if d["A"].shape[0] else None, b_eq=d["b"] if d["A"].shape[0] else None,
                   bounds=[(l if il else None, u if iu else None) for l, il, u, iu in zip(d["xlow"], d["ixlow"], d["xupp"], d["ixupp"])], method="highs")


@pytest.mark.parametrize("seed", range(6))
def test_general_restatement_against_highs(seed):
    blocks = random_block_lp(500 + seed, 3, 6, 14, 4, 3, 2, 2, free_fraction=0.15 if seed % 2 else 0.0)
    d = io.assemble(blocks)
    ref = _highs(d)
    o = io.solve_blocks(blocks, max_iter=100, mutol=1e-9, artol=1e-8, free_diag=1e-10)
    assert ref.status == 0 and o["status"] == 0
    assert abs(o["objective"] - ref.fun) < 1e-7 * max(1.0, abs(ref.fun))
    assert abs(o["objective"] - o["dual_objective"]) < 1e-6 * max(1.0, abs(ref.fun))
    # complementarity and sign conditions of the four pairs
    for gap, dual in (("t", "lam"), ("u", "pi"), ("v", "gamma"), ("w", "phi")):
        assert o[gap].min(initial=0.0) >= 0 and o[dual].min(initial=0.0) >= 0 and np.abs(o[gap] * o[dual]).max(initial=0.0) < 1e-6


@pytest.mark.parametrize("inst", GAMSSMALL, ids=[d["name"] for d in GAMSSMALL])
def test_general_restatement_reproduces_the_reference_objectives(inst):
    o = io.solve_blocks(inst["blocks"], max_iter=200, mutol=1e-8, artol=1e-8, dual_reg=1e-9, free_diag=1e-10)
    assert o["status"] == 0 and abs(o["objective"] - inst["expected_objective"]) < 1e-4


def test_bicgstab_restatement_status_codes():
    rng = np.random.default_rng(0)
    n = 120
    K = sp.random(n, n, density=0.08, random_state=3) + sp.identity(n) * 4
    K = (K + K.T).tocsc()
    b = rng.standard_normal(n)
    exact, rough = spl.splu(K), spl.splu((K + 0.5 * sp.identity(n)).tocsc())
    x, st, it, rn = io.bicgstab(lambda v: K @ v, exact.solve, b)
    assert io.BICG_STATUS[st] == "skipped" and it == 0 and rn <= 1e-10 * np.linalg.norm(b)
    x, st, it, rn = io.bicgstab(lambda v: K @ v, rough.solve, b)
    assert io.BICG_STATUS[st] == "converged" and 1 <= it <= 20 and np.linalg.norm(K @ x - b) <= 1e-9 * np.linalg.norm(b)
    x, st, it, rn = io.bicgstab(lambda v: K @ v, rough.solve, b, max_iter=1)
    assert io.BICG_STATUS[st] == "max iterations" and it == 1
    # a preconditioner that returns zero: rho = <r0, r> stays, v = K dx = 0 -> r0^T v = 0 -> breakdown
    x, st, it, rn = io.bicgstab(lambda v: K @ v, lambda v: 0.0 * v, b)
    assert io.BICG_STATUS[st] == "breakdown"
