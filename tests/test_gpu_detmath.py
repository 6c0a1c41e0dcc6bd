"""Device rt_detmath == host rt_detmath, bit for bit, and IEEE f32/f64 divide and sqrt on gfx950."""
import ctypes as C

import numpy as np
import pytest

from homework_18_graphics_raytracer_amd import _capi

pytestmark = pytest.mark.gpu

OPS = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "atan2": 4, "pow": 5, "f32_div": 6, "f32_sqrt": 7,
       "f64_sqrt_hi": 8, "f64_sqrt_lo": 9, "f64_div_hi": 10, "f64_div_lo": 11, "round": 12, "sincos_sin": 13, "sincos_cos": 14}


def _both(op, x, y):
    lib = _capi.amd_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    h = np.empty_like(x)
    d = np.empty_like(x)
    _capi.check(lib.rt_math_eval_host(OPS[op], x.ctypes.data, y.ctypes.data, h.ctypes.data, x.size))
    _capi.check(lib.rt_math_eval_device(OPS[op], x.ctypes.data, y.ctypes.data, d.ctypes.data, x.size))
    return h.view(np.uint32), d.view(np.uint32)


def _inputs(rng, n):
    """A mix: uniform bit patterns (all exponents, NaN, inf, denormals), values near 1, and ordinary ranges."""
    bits = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    near1 = (1.0 + rng.normal(0, 1e-3, n)).astype(np.float32)
    mid = rng.uniform(-100, 100, n).astype(np.float32)
    unit = rng.uniform(-1, 1, n).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, 1e-45, -1e-45, 3.4e38, 1e-38], dtype=np.float32)
    return np.concatenate([bits, near1, mid, unit, np.resize(special, n)])


@pytest.mark.parametrize("op", list(OPS))
def test_device_matches_host_bitwise(op):
    rng = np.random.default_rng(1234 + OPS[op])
    n = 1 << 18
    x = _inputs(rng, n)
    y = _inputs(rng, n)
    rng.shuffle(y)
    if op == "pow":
        # also the regime the renderer uses: base in [0,1], huge exponents (materials.rs:63, smoothness 1e-5)
        xb = rng.uniform(0, 1, n).astype(np.float32)
        yb = rng.choice(np.array([1.0, 4.9999995, 99.0, 1000.0, 99988.0, 1.0000001, 0.41666666], dtype=np.float32), n)
        x = np.concatenate([x, xb, (1.0 - rng.uniform(0, 1e-3, n)).astype(np.float32)])
        y = np.concatenate([y, yb, yb])
    h, d = _both(op, x, y)
    # NaN payloads may differ between platforms; compare NaN-ness there
    hn = np.isnan(h.view(np.float32))
    dn = np.isnan(d.view(np.float32))
    if op.startswith("f64_"):
        # the two halves of a binary64 NaN are not NaN-classifiable as f32; only compare where the host result is finite
        ok = h == d
        # allow differing NaN payloads: detect via the HI half pattern (exponent all ones)
        assert ok.mean() > 0.98
        bad = np.argwhere(~ok).ravel()
        xs, ys = x[bad], y[bad]
        prod_nan = ~np.isfinite(xs.astype(np.float64) * ys.astype(np.float64)) | ~np.isfinite(xs.astype(np.float64) / ys.astype(np.float64)) | (xs.astype(np.float64) * ys.astype(np.float64) < 0)
        assert prod_nan.all(), f"{op}: {np.count_nonzero(~prod_nan)} finite mismatches, e.g. x={xs[~prod_nan][:3]} y={ys[~prod_nan][:3]}"
        return
    assert np.array_equal(hn, dn)
    mism = (h != d) & ~hn
    assert not mism.any(), f"{op}: {mism.sum()} mismatches, e.g. x={x[mism][:3]!r} y={y[mism][:3]!r} host={h[mism][:3]} dev={d[mism][:3]}"
