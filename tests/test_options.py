"""PIPSIPMpp.opt parsing (AbstractOptions::load_options_from_file, AbstractOptions.C:62-135) and the mapping of the path-relevant
identifiers onto the library's settings."""
import os

import numpy as np
import pytest

from pips_ipmpp_amd.options import apply_options, load_options

REF_OPT = "/root/reference/PIPS-IPM/Drivers/gams/simple/GAMSsmall/examples_parallelRows/PIPSIPMpp.opt"


def test_parser_follows_the_reference_rules(tmp_path):
    p = tmp_path / "PIPSIPMpp.opt"
    p.write_text("# comment\n// another\n\nSC_COMPUTE_BLOCKWISE true bool\nGONDZIO_MAX_CORRECTORS 3 int\nOUTER_BICG_TOL 1e-9 double\n"
                 "OUTER_SOLVE 2 integer\nREGULARIZATION False boolean\nBROKEN_LINE 5\nBAD_INT x int\nBAD_BOOL maybe bool\nUNKNOWN_TYPE 1 float\n"
                 "PRESOLVE_PARALLEL_ROWS true bool trailing words are ignored\n")
    o = load_options(str(p))
    assert o == {"SC_COMPUTE_BLOCKWISE": True, "GONDZIO_MAX_CORRECTORS": 3, "OUTER_BICG_TOL": 1e-9, "OUTER_SOLVE": 2, "REGULARIZATION": False,
                 "PRESOLVE_PARALLEL_ROWS": True}
    applied, ignored = apply_options(o)
    assert applied == [] and set(ignored) == set(o)


@pytest.mark.skipif(not os.path.exists(REF_OPT), reason="reference tree not present")
def test_reads_an_options_file_of_the_reference():
    o = load_options(REF_OPT)
    assert o["PRESOLVE_PARALLEL_ROWS"] is True and o["PRESOLVE_MAX_ROUNDS"] == 1 and o["PRESOLVE_SINGLETON_ROWS"] is False


@pytest.mark.gpu
def test_options_reach_the_batch_and_the_harness(tmp_path):
    import pips_ipmpp_amd as pa
    from tests.test_ipm_gpu import build_lp
    blocks, F0, c, b, A = build_lp(77, 3, 60, 30, 6, 5, 0.1)
    p = tmp_path / "PIPSIPMpp.opt"
    # no Gondzio correctors, iterative refinement as the outer solve
    p.write_text("GONDZIO_MAX_CORRECTORS 0 int\nOUTER_SOLVE 1 int\nPRESOLVE_MAX_ROUNDS 1 int\n")
    base = pa.IpmSolver(6, 5, blocks, F0, c, b)
    r0 = base.solve(max_iter=100, mutol=1e-8, artol=1e-8)
    ipm = pa.IpmSolver(6, 5, blocks, F0, c, b)
    applied, ignored = apply_options(load_options(str(p)), ipm=ipm)
    assert applied == ["GONDZIO_MAX_CORRECTORS", "OUTER_SOLVE"] and ignored == ["PRESOLVE_MAX_ROUNDS"]
    r1 = ipm.solve(max_iter=100, mutol=1e-8, artol=1e-8)
    assert r0["status"] == 0 and r1["status"] == 0 and abs(r0["objective"] - r1["objective"]) < 1e-6 * abs(r0["objective"])
    assert base.stats()["gondzio_correctors"] > 0 and ipm.stats()["gondzio_correctors"] == 0
    # the reference's own default for the system the outer solve runs on (PIPSIPMppOptions.C:293) reaches the harness
    q = tmp_path / "reg.opt"
    q.write_text("OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM false bool\nOUTER_BICG_MAX_STAGNATIONS 4 int\n")
    ipm2 = pa.IpmSolver(6, 5, blocks, F0, c, b)
    applied, ignored = apply_options(load_options(str(q)), ipm=ipm2)
    assert applied == ["OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM", "OUTER_BICG_MAX_STAGNATIONS"] and ignored == []
    r2 = ipm2.solve(max_iter=100, mutol=1e-8, artol=1e-8)
    assert r2["status"] == 0 and abs(r0["objective"] - r2["objective"]) < 1e-6 * abs(r0["objective"])
    with pytest.raises(RuntimeError):
        ipm.set_option("NOT_AN_OPTION", 1)
    with pytest.raises(RuntimeError):
        ipm.set_option("OUTER_SOLVE", 7)
    # batch: SC_COMPUTE_BLOCKWISE selects the Schur route
    for flag, mode in ((True, 2), (False, 1)):
        bt = pa.LeafBatch(1, 4)
        applied, _ = apply_options({"SC_COMPUTE_BLOCKWISE": flag, "PARDISO_NITERATIVE_REFINS": 2}, batch=bt)
        assert applied == ["SC_COMPUTE_BLOCKWISE", "PARDISO_NITERATIVE_REFINS"]
        W, T, F, _, _ = pa.gen_block(5, 1, 40, 20, 2, 2, 0.2)
        K, dpos = pa.kkt_leaf_assemble(40, W)
        K.val[dpos] = np.concatenate([np.ones(40), -1e-8 * np.ones(20)])
        bt.set_block(0, K, 40, pa.border_assemble(40, 20, 0, 2, 0, A=T, F=F))
        bt.analyze(16)
        assert bt.schur_mode() == mode
