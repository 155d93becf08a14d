"""Shade's specular term is pow(dot, shininess) on floats (Ray.cu:655).  The oracle evaluates it with the host libm's
powf; k_shade restates that algorithm (kernels.hip powf_libm: glibc 2.35 e_powf.c with the x86-64 FMA build's fused
multiply-adds) instead of calling the device's double-precision pow.  Held against the oracle's powf value by value."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import cudaraytracer_amd as crt
    c = crt.Context(0)
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_device_powf_equals_the_host_libm_on_shading_inputs(ctx):
    """dot products in [-1, 1] (a tenth of them tiny), the exponents materials use and a few that are not integers:
    bit-identical, NaNs compared as NaNs"""
    rng = np.random.default_rng(11)
    n = 8_000_000
    x = rng.uniform(-1, 1, n).astype(np.float32)
    x[::10] *= np.float32(1e-3)
    x[5::257] *= np.float32(37.5)
    y = rng.choice(np.array([2, 10, 50, 3, 7, 1.5, 0.3, 20, 100, 4.25, 12, 64, 5, -2, 33, 1, 128], np.float32), n)
    g = ctx.powf_probe(x, y)
    o = orc.powf(x, y)
    both_nan = np.isnan(g) & np.isnan(o)
    bad = (bits(g) != bits(o)) & ~both_nan
    assert int(bad.sum()) == 0, f"{int(bad.sum())} of {n} differ, first x={x[bad][0]!r} y={y[bad][0]!r}: device {g[bad][0]!r} host {o[bad][0]!r}"
    assert int(np.isnan(o).sum()) > 1000 and int((o == 0).sum()) > 0       # the NaN and the underflow branches were exercised


def test_device_powf_special_values(ctx):
    """zeros, infinities, NaN, subnormal bases, over- and underflow, signed results of odd powers"""
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3.0e38, -3.0e38, 0.5, -0.5, 2.0, -2.0, 1.17549435e-38],
                  np.float32)
    ys = np.array([0.0, -0.0, 1.0, -1.0, 2.0, 3.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 127.0, 128.0, -149.0, -150.0, 151.0, 1e10, 255.0],
                  np.float32)
    x, y = [a.ravel() for a in np.meshgrid(sp, ys)]
    g = ctx.powf_probe(x, y)
    o = orc.powf(x, y)
    both_nan = np.isnan(g) & np.isnan(o)
    bad = (bits(g) != bits(o)) & ~both_nan
    assert int(bad.sum()) == 0, [(float(a), float(b), float(c), float(d)) for a, b, c, d in zip(x[bad], y[bad], g[bad], o[bad])][:8]
