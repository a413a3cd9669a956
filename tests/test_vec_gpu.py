"""Flat-arena vector kernels (a15) against numpy on the same data: single-rounding ops bit-for-bit, multiply-add ops to
2 ulp (the GPU contracts a*x+y into one FMA, numpy rounds twice), sums to 1e-12 of the absolute sum."""
import numpy as np
import pytest

import pips_ipmpp_amd as pa

pytestmark = pytest.mark.gpu
N = 100003  # ragged on purpose


def _mk(seed):
    import torch
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(N)
    return a, torch.tensor(a, device="cuda")


def _ulp_close(got, want, scale):
    """|got - want| <= 2 ulp of the largest term that entered the result (FMA contraction vs two roundings)."""
    return bool(np.all(np.abs(got - want) <= 4.5e-16 * scale))


def test_elementwise_ops_match_numpy_exactly():
    import torch
    V = pa.capi.vec
    x, xd = _mk(1)
    z, zd = _mk(2)
    y, yd = _mk(3)
    mask = (np.random.default_rng(4).random(N) < 0.7).astype(np.float64)
    md = torch.tensor(mask, device="cuda")
    get = lambda: yd.cpu().numpy().copy()  # noqa: E731
    V.axpy(0.37, xd, yd)
    assert _ulp_close(get(), y + 0.37 * x, np.abs(y) + np.abs(0.37 * x))
    y = get()
    V.add_product(-1.5, xd, zd, yd)
    assert _ulp_close(get(), y + (-1.5 * x) * z, np.abs(y) + np.abs(1.5 * x * z))
    y = get()
    V.add_quotient(2.0, xd, zd, md, yd)
    assert _ulp_close(get(), np.where(mask != 0, y + 2.0 * x / z, y), np.abs(y) + np.abs(2.0 * x / z))
    y = get()
    V.divide_some(zd, md, yd)
    y = np.where(mask != 0, y / z, y)
    assert np.array_equal(get(), y)
    V.select_nonzeros(md, yd)
    y = np.where(mask != 0, y, 0.0)
    assert np.array_equal(get(), y)
    V.safe_invert(yd)
    y = np.where(y != 0, 1.0 / np.where(y != 0, y, 1.0), 0.0)
    assert np.array_equal(get(), y)
    V.scale(-1.0, yd)
    V.add_const(0.25, yd)
    V.mul(xd, yd)
    V.div(zd, yd)
    y = ((-y + 0.25) * x) / z
    assert np.array_equal(get(), y)
    V.axpby(2.0, xd, -0.5, yd)
    assert _ulp_close(get(), 2.0 * x + -0.5 * y, np.abs(2.0 * x) + np.abs(0.5 * y))
    # gondzioProjection (DenseVector.cpp:405-420): step back into [rmin, rmax], never below -rmax
    y = get()
    rmin, rmax = 0.2, 0.9
    V.gondzio_projection(rmin, rmax, yd)
    want = np.where(y < rmin, rmin - y, np.where(y > rmax, rmax - y, 0.0))
    want = np.maximum(want, -rmax)
    assert np.array_equal(get(), want)
    assert (want != 0).any() and (want == 0).any() and (want == -rmax).any()


def test_reductions_match_numpy():
    V = pa.capi.vec
    x, xd = _mk(5)
    y, yd = _mk(6)
    assert abs(V.dot(xd, yd) - x @ y) <= 1e-12 * np.abs(x * y).sum()
    assert abs(V.dot(xd, yd, skip_root=1000) - x[1000:] @ y[1000:]) <= 1e-12 * np.abs(x * y).sum()
    assert abs(V.one_norm(xd) - np.abs(x).sum()) <= 1e-12 * np.abs(x).sum()
    assert V.inf_norm(xd) == np.abs(x).max()
    assert V.min(xd) == x.min()
    assert abs(V.two_norm(xd) - np.linalg.norm(x)) <= 1e-13 * np.linalg.norm(x)
    # step bound: largest alpha with v + alpha dv >= 0
    v = np.abs(x) + 0.1
    import torch
    vd = torch.tensor(v, device="cuda")
    want = np.min(np.where(y < 0, -v / np.where(y < 0, y, -1.0), np.inf))
    assert V.stepbound(vd, yd) == want
    a, b = 0.3, 0.7
    want = ((v + a * y) * (np.abs(y) + b * x)).sum()
    wd = torch.tensor(np.abs(y), device="cuda")
    got = V.dot_shifted(vd, a, yd, wd, b, xd)
    assert abs(got - want) <= 1e-12 * np.abs((v + a * y) * (np.abs(y) + b * x)).sum()


def test_empty_and_tiny_vectors():
    import torch
    V = pa.capi.vec
    e = torch.zeros(0, dtype=torch.float64, device="cuda")
    assert V.dot(e, e) == 0.0 and V.inf_norm(e) == 0.0 and V.min(e) == float("inf")
    one = torch.tensor([-3.0], dtype=torch.float64, device="cuda")
    assert V.inf_norm(one) == 3.0 and V.two_norm(one) == 3.0


def test_find_blocking_matches_numpy():
    """DenseVector::find_blocking (DenseVector.cpp:1002-1050 region): the entry that attains the step bound, with the
    complementary pair's values at that index; ties resolve to the lowest index; no blocking entry -> inf and zeros."""
    import torch
    V = pa.capi.vec
    for seed, n in ((1, 7), (2, 1000), (3, 300001)):
        rng = np.random.default_rng(seed)
        v, g = rng.random(n) + 0.05, rng.random(n) + 0.05
        dv, dg = rng.standard_normal(n), rng.standard_normal(n)
        if n > 100:   # a tie: two entries with exactly the same (minimal) ratio
            v[[17, 90]], dv[[17, 90]] = 0.001, -8.0
        t = [torch.tensor(a, device="cuda") for a in (v, dv, g, dg)]
        out = V.find_blocking(*t)
        r = np.where(dv < 0, -v / np.where(dv < 0, dv, -1.0), np.inf)
        i = int(np.argmin(r))
        assert np.array_equal(out, [r[i], v[i], dv[i], g[i], dg[i]])
        if n > 100:
            assert i == 17
    z = torch.tensor(np.abs(dv), device="cuda")
    out = V.find_blocking(t[0], z, t[2], t[3])
    assert out[0] == np.inf and not out[1:].any()


def test_weighted_stepbounds_match_numpy():
    """One pass for the 11 blended directions of the corrector weight search (InteriorPointMethod.cpp:486-523) = 22 separate
    step bounds."""
    import torch
    V = pa.capi.vec
    for seed, n in ((1, 5), (2, 3000), (3, 400001)):
        rng = np.random.default_rng(seed)
        v, g = rng.random(n) + 0.05, rng.random(n) + 0.05
        dv, cv, dg, cg = (rng.standard_normal(n) for _ in range(4))
        if seed == 1:
            dg, cg = np.abs(dg), np.abs(cg)      # no blocking entry on the second triple: infinities
        t = [torch.tensor(a, device="cuda") for a in (v, dv, cv, g, dg, cg)]
        wmin = 0.37
        bp, bd = V.weighted_stepbounds(*t, wmin, 11)
        for k in range(11):
            w = min(1.0, wmin + (1.0 - wmin) / 10.0 * k)
            for got, x, dx, cx in ((bp[k], v, dv, cv), (bd[k], g, dg, cg)):
                s = dx + w * cx
                want = np.min(-x[s < 0] / s[s < 0]) if (s < 0).any() else np.inf
                assert got == want or abs(got - want) <= 1e-14 * want, (seed, k, got, want)
