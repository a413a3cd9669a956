"""The committed config-1 capture of the CPU IPM restatement (tests/golden/ipm_config1_trace.npz) against HiGHS."""
import numpy as np

from tests.test_ipm_gpu import GOLDEN_CFG1, build_lp


def test_config1_capture_agrees_with_highs():
    """CPU: the committed config-1 capture of the IPM restatement (tests/golden/make_ipm_config1.py) ends at the
    optimum HiGHS finds for the same LP - the capture is pinned against an independent solver."""
    from scipy.optimize import linprog
    g = np.load(GOLDEN_CFG1)
    N, n_i, my_i, n0, myl = (int(v) for v in g["shape"][:5])
    blocks, F0, c, b, A = build_lp(int(g["seed"]), N, n_i, my_i, n0, myl, float(g["shape"][5]))
    ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert ref.status == 0 and int(g["status"]) == 0
    assert abs(float(g["objective"]) - ref.fun) / abs(ref.fun) < 1e-9
    assert g["trace"].shape == (int(g["iterations"]) + 1, 7)
