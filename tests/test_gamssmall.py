"""The reference's own known-answer instances (Test/IntegrationTests/gamssmall_instance_data.txt: 26 active GAMSsmall LPs with
expected objective and iteration count; t_pips.cpp:115-119 checks EXPECT_NEAR(objective, expected, 1e-4) and
iterations <= 1.1 * expected) through the GDX reader, the CPU restatement of the IPM and the device harness.

tests/golden/gamssmall.json holds the block data extracted from the per-block GDX files (tests/golden/make_gamssmall.py)."""
import json
import os
import warnings

import numpy as np
import pytest

from pips_ipmpp_amd.standard_form import block_standard_form, general_lp, kkt_violation, recover_duals, recover_solution

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = json.load(open(os.path.join(HERE, "golden", "gamssmall.json")))["instances"]
IDS = [d["name"] for d in DATA]
REF = "/root/reference"
OBJ_TOL = 1e-4   # t_pips.cpp:27,116


def test_fixture_has_the_reference_instance_list():
    assert len(DATA) == 26
    assert {d["name"] for d in DATA} >= {"exampleAC_boundStrength", "hier_approach_8blocks_2by3", "example_breakSingletonRows"}
    assert all(len(d["blocks"]) == d["num_blocks"] for d in DATA)


def test_gdx_container_round_trip(tmp_path):
    """Writer and reader of the GDX container agree on every record kind the jacobian files use: scalars, sets, equation /
    variable records with special values, a 2-dimensional parameter with byte, word and integer sized keys."""
    from pips_ipmpp_amd import gdx
    rng = np.random.default_rng(5)
    for top in (200, 40000, 3000000):
        rows = np.sort(rng.choice(np.arange(1, top), size=40, replace=False))
        cols = np.sort(rng.choice(np.arange(top, 2 * top), size=60, replace=False))
        keys = sorted({(int(rng.choice(rows)), int(rng.choice(cols))) for _ in range(300)})
        avals = rng.choice([0.5, 1.0, -1.0, 2.0, 0.0, 3.25, -7e5, 1e-9], size=len(keys))
        evals = np.column_stack([rng.standard_normal(len(rows)), np.zeros(len(rows)),
                                 rng.choice([gdx.SV_MINF, 0.0, 4.0], size=len(rows)), rng.choice([gdx.SV_PINF, 9.0], size=len(rows)),
                                 rng.integers(1, 6, size=len(rows)).astype(float)])
        xvals = np.column_stack([np.zeros(len(cols)), np.zeros(len(cols)), rng.choice([gdx.SV_MINF, 0.0, 0.8], size=len(cols)),
                                 rng.choice([gdx.SV_PINF, 5.0], size=len(cols)), rng.integers(1, 5, size=len(cols)).astype(float)])
        path = str(tmp_path / f"t{top}.gdx")
        syms = [("numUel", gdx.PARAMETER, "Number of UELS", np.zeros((1, 0)), [[2.0 * top]]),
                ("i", gdx.SET, "Equation names", rows[:, None], rows[:, None].astype(float)),
                ("j", gdx.SET, "Variable names", cols[:, None], cols[:, None].astype(float)),
                ("e", gdx.EQUATION, "Equations", rows[:, None], evals),
                ("x", gdx.VARIABLE, "Variables", cols[:, None], xvals),
                ("A", gdx.PARAMETER, "Jacobian", np.array(keys), avals[:, None])]
        gdx.write_gdx(path, syms)
        g = gdx.GdxFile(path)
        assert g.names() == [s[0] for s in syms] and g.uels == []
        for name, typ, text, k, v in syms:
            s = g.symbol(name)
            assert s.type == typ and s.text == text and s.dim == np.asarray(k).shape[1]
            assert np.array_equal(s.keys, np.asarray(k).reshape(len(v), -1))
            assert np.array_equal(s.values, np.asarray(v, dtype=float).reshape(len(v), -1))
    with pytest.raises(gdx.GdxError):
        open(tmp_path / "bad.gdx", "wb").write(b"not a gdx file at all, just bytes" * 4)
        gdx.GdxFile(str(tmp_path / "bad.gdx"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box): the committed fixture stands in")
def test_fixture_is_what_the_reader_extracts_from_the_reference_files():
    from pips_ipmpp_amd import gdx
    for d in DATA:
        blocks = gdx.read_problem(REF + d["source"], d["num_blocks"])
        for got, want in zip(blocks, d["blocks"]):
            for key, w in want.items():
                gv = got[key]
                if isinstance(w, dict):
                    assert gv is not None and all(list(np.asarray(gv[f]).ravel()) == list(np.asarray(w[f]).ravel()) for f in w), (d["name"], key)
                elif w is None:
                    assert gv is None
                else:
                    assert np.array_equal(np.asarray(gv), np.asarray(w)), (d["name"], key)
    # the whole-model file of an instance (label table present, byte-sized keys) reads as well
    g = gdx.GdxFile(REF + DATA[0]["source"] + ".gdx")
    assert len(g.uels) == 23 and g.symbol("A").keys.shape[1] == 2 and g.symbol("x").values.shape[1] == 5


@pytest.mark.parametrize("inst", DATA, ids=IDS)
def test_highs_reproduces_the_reference_objective(inst):
    """Pins the extracted data (bounded two-sided form) and the conversion to the harness' standard form."""
    from scipy.optimize import linprog
    c, A_eq, b_eq, A_ub, b_ub, bounds = general_lp(inst["blocks"])
    r = linprog(c, A_eq=A_eq, b_eq=b_eq, A_ub=A_ub, b_ub=b_ub, bounds=bounds, method="highs")
    assert r.status == 0 and abs(r.fun - inst["expected_objective"]) < OBJ_TOL
    sf = block_standard_form(inst["blocks"])
    r2 = linprog(sf["c"], A_eq=sf["A"], b_eq=sf["b"], bounds=(0, None), method="highs")
    assert r2.status == 0 and abs(r2.fun + sf["offset"] - inst["expected_objective"]) < OBJ_TOL
    # back to the original variables: feasible for the bounded form, same objective
    x = np.concatenate(recover_solution(sf, r2.x))
    assert abs(c @ x - inst["expected_objective"]) < OBJ_TOL
    assert np.abs(A_eq @ x - b_eq).max() < 1e-7 and (A_ub is None or (A_ub @ x - b_ub).max() < 1e-7)
    assert all((lo is None or xi >= lo - 1e-7) and (up is None or xi <= up + 1e-7) for xi, (lo, up) in zip(x, bounds))


@pytest.mark.parametrize("inst", DATA, ids=IDS)
def test_recovered_duals_satisfy_the_original_optimality_conditions(inst):
    """Primal point and row multipliers mapped back from the standard form (HiGHS solves it here) are a KKT point of the
    bounded two-sided original: reduced-cost signs at the bounds, multiplier signs and complementarity on the rows."""
    from scipy.optimize import linprog
    sf = block_standard_form(inst["blocks"], split_free=False)
    bnds = [(0, None) if m else (None, None) for m in sf["bounded_mask"]]
    r = linprog(sf["c"], A_eq=sf["A"], b_eq=sf["b"], bounds=bnds, method="highs")
    assert r.status == 0
    assert kkt_violation(inst["blocks"], recover_solution(sf, r.x), recover_duals(sf, r.eqlin.marginals)) < 1e-7


@pytest.mark.parametrize("free", ["split", "native"])
@pytest.mark.parametrize("inst", DATA, ids=IDS)
def test_ipm_oracle_reproduces_the_reference_objective(inst, free):
    """free variables either split into x+ - x- or kept as one column without a complementarity pair (the reference's way)"""
    from oracle import ipm_oracle as io
    sf = block_standard_form(inst["blocks"], split_free=free == "split")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # several instances carry parallel / redundant rows on purpose (the reference removes them in its presolve): dual regularisation
        o = io.solve_lp(sf["A"], sf["b"], sf["c"], 200, 1e-8, 1e-8, dual_reg=1e-9, bounded=sf["bounded_mask"], free_diag=1e-10)
    assert o["status"] == 0
    assert abs(o["objective"] + sf["offset"] - inst["expected_objective"]) < OBJ_TOL
    assert o["iterations"] <= 1.1 * inst["expected_iterations"]   # t_pips.cpp:119


@pytest.mark.gpu
@pytest.mark.parametrize("free", ["split", "native"])
@pytest.mark.parametrize("inst", DATA, ids=IDS)
def test_device_harness_reproduces_the_reference_objective(inst, free):
    import pips_ipmpp_amd as pa
    sf = block_standard_form(inst["blocks"], split_free=free == "split")
    ipm = pa.IpmSolver(sf["n0"], sf["myl"], sf["blocks"], sf["F0"], sf["c"], sf["b"], dual_reg=1e-9)
    if free == "native":
        assert (sf["bounded_mask"] == 0).any() or inst["name"].startswith("exampleAC_")   # three instances have no free variable
        ipm.set_free_variables(sf["bounded_mask"])
    res = ipm.solve(max_iter=200, mutol=1e-8, artol=1e-8)
    assert res["status"] == 0, res
    assert abs(res["objective"] + sf["offset"] - inst["expected_objective"]) < OBJ_TOL, res
    assert res["iterations"] <= 1.1 * inst["expected_iterations"] + 1
    x, y = ipm.solution()
    assert x[sf["bounded_mask"] == 1].min() > -1e-8 and np.linalg.norm(sf["A"] @ x - sf["b"], np.inf) < 1e-6 * max(1.0, np.abs(sf["b"]).max())
    # primal point and multipliers in terms of the original problem
    assert kkt_violation(inst["blocks"], recover_solution(sf, x), recover_duals(sf, y), tol=1e-5) < 1e-4


def _same_block(got, want, tag):
    for key, w in want.items():
        g = got[key]
        if isinstance(w, dict):
            assert g is not None, (tag, key)
            for f in w:
                assert list(np.asarray(g[f]).ravel()) == list(np.asarray(w[f]).ravel()), (tag, key, f)
        elif w is None:
            assert g is None, (tag, key)
        else:
            assert np.array_equal(np.asarray(g, dtype=float), np.asarray(w, dtype=float)), (tag, key)


def _synthetic_jacobian(path, rng, num_blocks):
    """A jacobian GDX file of a random block-structured LP (all blocks in one file, stages as in the reference's files):
    labels 1..m are rows, m+1..m+n columns, the last column is the objective variable, the last row the objective row."""
    from pips_ipmpp_amd import gdx
    nvar = [int(rng.integers(1, 4)) for _ in range(num_blocks)]
    nrow = [int(rng.integers(1, 4)) for _ in range(num_blocks)]
    nlink = int(rng.integers(1, 3))
    m, n = sum(nrow) + nlink + 1, sum(nvar) + 1
    row_stage = sum(([k + 1] * nrow[k] for k in range(num_blocks)), []) + [num_blocks + 1] * nlink + [num_blocks + 1]
    col_stage = sum(([k + 1] * nvar[k] for k in range(num_blocks)), []) + [1]
    rows, cols = np.arange(1, m + 1), np.arange(m + 1, m + n + 1)
    evals, xvals, keys, avals = [], [], [], []
    for i in range(m - 1):
        kind = rng.integers(0, 4)
        lo = gdx.SV_MINF if kind == 1 else float(rng.integers(-3, 4))
        up = gdx.SV_PINF if kind == 2 else (lo if kind == 0 else float(rng.integers(4, 9)))
        evals.append([0.0, 0.0, lo, up, float(row_stage[i])])
    evals.append([0.0, 0.0, 0.0, 0.0, float(row_stage[-1])])
    for j in range(n - 1):
        kind = rng.integers(0, 4)
        lo = gdx.SV_MINF if kind == 1 else float(rng.integers(0, 2))
        up = gdx.SV_PINF if kind != 3 else lo + float(rng.integers(0, 5))
        xvals.append([0.0, 0.0, lo, up, float(col_stage[j])])
    xvals.append([0.0, 0.0, gdx.SV_MINF, gdx.SV_PINF, 1.0])
    for i in range(m - 1):
        blk = row_stage[i] - 1
        for j in range(n - 1):
            cb = col_stage[j] - 1
            if (blk == num_blocks or cb == 0 or cb == blk) and rng.random() < 0.6:
                keys.append((rows[i], cols[j])); avals.append(float(rng.choice([1.0, -1.0, 2.0, 0.5, 3.75, -7.0])))
    for j in range(n - 1):
        if rng.random() < 0.8:
            keys.append((rows[-1], cols[j])); avals.append(-float(rng.integers(1, 5)))
    keys.append((rows[-1], cols[-1])); avals.append(1.0)
    syms = [("numUel", gdx.PARAMETER, "Number of UELS", np.zeros((1, 0)), [[float(m + n)]]),
            ("i", gdx.SET, "Equation names", rows[:, None], rows[:, None].astype(float)),
            ("j", gdx.SET, "Variable names", cols[:, None], cols[:, None].astype(float)),
            ("jobj", gdx.SET, "Objective name", np.array([[cols[-1]]]), [[0.0]]),
            ("iobj", gdx.SET, "Objective row", np.array([[rows[-1]]]), [[0.0]]),
            ("objcoef", gdx.PARAMETER, "Objective coefficient", np.zeros((1, 0)), [[float(rng.choice([1.0, -1.0]))]]),
            ("e", gdx.EQUATION, "Equations", rows[:, None], np.array(evals)),
            ("x", gdx.VARIABLE, "Variables", cols[:, None], np.array(xvals)),
            ("A", gdx.PARAMETER, "Jacobian", np.array(keys), np.array(avals)[:, None])]
    gdx.write_gdx(path, syms)


def test_library_gdx_reader_matches_the_python_restatement(tmp_path):
    """pips_gdx_read_block (C ABI, csrc/gdx.cpp) against pips_ipmpp_amd.gdx.read_block on synthetic jacobian files and, when the
    reference tree is here, on every block file of the 26 instances."""
    import pips_ipmpp_amd as pa
    from pips_ipmpp_amd import gdx
    rng = np.random.default_rng(11)
    for case in range(12):
        nb = int(rng.integers(2, 5))
        path = str(tmp_path / f"syn{case}.gdx")
        _synthetic_jacobian(path, rng, nb)
        for k in range(nb):
            _same_block(pa.capi.gdx_read_block(path, nb, k), gdx.read_block(path, nb, k), (case, k))
    with pytest.raises(RuntimeError):
        pa.capi.gdx_read_block(str(tmp_path / "does_not_exist.gdx"), 2, 0)
    open(tmp_path / "junk.gdx", "wb").write(bytes(range(256)) * 3)
    with pytest.raises(RuntimeError):
        pa.capi.gdx_read_block(str(tmp_path / "junk.gdx"), 2, 0)
    if os.path.isdir(REF):
        whole = 0
        for d in DATA:
            for k in range(d["num_blocks"]):
                _same_block(pa.capi.gdx_read_block(f"{REF}{d['source']}{k}.gdx", d["num_blocks"], k), d["blocks"][k], (d["name"], k))
                # the unsplit model file (label table, byte-sized keys) yields the same block without the reference's split step;
                # one instance's unsplit file carries a coefficient across blocks, which both readers reject like readBlock does
                try:
                    _same_block(pa.capi.gdx_read_block(f"{REF}{d['source']}.gdx", d["num_blocks"], k), d["blocks"][k], (d["name"], k, "whole"))
                    whole += 1
                except RuntimeError as e:
                    assert "different blocks" in str(e)
                    with pytest.raises(gdx.GdxError):
                        gdx.read_block(f"{REF}{d['source']}.gdx", d["num_blocks"], k)
        assert whole >= 100
