"""rt_detmath.h on the host: accuracy against binary64 libm rounded once, agreement with glibc's f32 libm,
and the C99 special cases.  (Device == host bit-for-bit is tests/test_gpu_detmath.py.)"""
import ctypes as C

import numpy as np
import pytest

from homework_18_graphics_raytracer_amd import _capi
import _oracle

OPS = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "atan2": 4, "pow": 5, "round": 12, "sincos_sin": 13, "sincos_cos": 14}


def ev(op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    _capi.check(_capi.amd_lib().rt_math_eval_host(OPS[op], x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size))
    return out


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


RNG = np.random.default_rng(42)
N = 400_000


def _check(got, want64, min_exact=0.99999):
    want = want64.astype(np.float32)
    ok = np.isfinite(want) & (np.abs(want) > 1e-37)
    d = ulp_diff(got[ok], want[ok])
    assert d.max() <= 1, f"max ulp diff {d.max()}"
    assert np.mean(d == 0) >= min_exact, f"exact fraction {np.mean(d == 0)}"


@pytest.mark.parametrize("name,fn", [("sin", np.sin), ("cos", np.cos), ("tan", np.tan)])
def test_trig_is_correctly_rounded_almost_always(name, fn):
    x = np.concatenate([RNG.uniform(-8, 8, N), RNG.uniform(-70, 70, N), RNG.uniform(-1e5, 1e5, N),
                        RNG.uniform(-1e-3, 1e-3, N)]).astype(np.float32)
    _check(ev(name, x), fn(x.astype(np.float64)))


def test_trig_huge_arguments_use_payne_hanek():
    x = (RNG.uniform(1, 2, N) * 2.0 ** RNG.integers(20, 127, N)).astype(np.float32)
    x *= RNG.choice([-1, 1], N).astype(np.float32)
    _check(ev("sin", x), np.sin(x.astype(np.float64)), min_exact=0.9999)
    _check(ev("cos", x), np.cos(x.astype(np.float64)), min_exact=0.9999)


def test_fused_sincos_equals_sin_and_cos_bit_for_bit():
    """The kernels call sincosf (one argument reduction for both); the oracle calls sinf and cosf."""
    bits = RNG.integers(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = np.concatenate([bits, RNG.uniform(-8, 8, N).astype(np.float32), RNG.uniform(-1e5, 1e5, N).astype(np.float32),
                        (RNG.uniform(1, 2, N) * 2.0 ** RNG.integers(20, 127, N)).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 0.78539816, 1.5707964, 3.1415927, -3.1415927], dtype=np.float32)])
    for fused, plain in (("sincos_sin", "sin"), ("sincos_cos", "cos")):
        a, b = ev(fused, x), ev(plain, x)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(a)
        assert np.array_equal(a[ok].view(np.uint32), b[ok].view(np.uint32))


def test_acos_atan2():
    x = np.concatenate([RNG.uniform(-1, 1, N), 1 - RNG.uniform(0, 1e-4, N), -1 + RNG.uniform(0, 1e-4, N)]).astype(np.float32)
    _check(ev("acos", x), np.arccos(x.astype(np.float64)))
    y = RNG.normal(0, 1, N).astype(np.float32)
    z = RNG.normal(0, 1, N).astype(np.float32)
    _check(ev("atan2", y, z), np.arctan2(y.astype(np.float64), z.astype(np.float64)))


def test_pow_including_the_specular_regime():
    # smoothness 1e-5 gives exponents ~99988 on bases just under 1 (materials.rs:60-63)
    b = np.concatenate([RNG.uniform(0, 1, N), 1 - RNG.uniform(0, 2e-4, N), RNG.uniform(0, 50, N)]).astype(np.float32)
    e = np.concatenate([RNG.choice([1.0, 0.99999988, 4.9999971, 99.0, 1000.0, 1.0000001, 0.41666666], N),
                        RNG.choice([99988.08, 999.88, 99.0], N), RNG.uniform(-5, 5, N)]).astype(np.float32)
    with np.errstate(all="ignore"):
        _check(ev("pow", b, e), np.power(b.astype(np.float64), e.astype(np.float64)), min_exact=0.9999)


def test_agrees_with_glibc_libm_to_one_ulp():
    x = RNG.uniform(-70, 70, N).astype(np.float32)
    for name in ("sin", "cos"):
        assert ulp_diff(ev(name, x), _oracle.math(name, x, kind="libm")).max() <= 1
    u = RNG.uniform(-1, 1, N).astype(np.float32)
    assert ulp_diff(ev("acos", u), _oracle.math("acos", u, kind="libm")).max() <= 1
    b = RNG.uniform(0, 1, N).astype(np.float32)
    e = RNG.choice([1.0, 4.9999971, 99.0, 1000.0, 99988.08], N).astype(np.float32)
    g, l = ev("pow", b, e), _oracle.math("pow", b, e, kind="libm")
    big = np.abs(l) > 1e-30
    assert ulp_diff(g[big], l[big]).max() <= 1


def test_oracle_detmath_build_calls_the_same_functions():
    x = RNG.uniform(-70, 70, 10000).astype(np.float32)
    assert np.array_equal(ev("sin", x).view(np.uint32), _oracle.math("sin", x).view(np.uint32))


def test_special_cases_follow_c99():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    f = lambda *v: np.array(v, dtype=np.float32)
    assert np.isnan(ev("sin", f(inf, -inf, nan))).all() and np.isnan(ev("cos", f(inf, nan))).all()
    s = ev("sin", f(0.0, -0.0))
    assert s[0] == 0 and not np.signbit(s[0]) and np.signbit(s[1])
    assert ev("cos", f(0.0))[0] == 1.0
    a = ev("acos", f(1.0, -1.0, 0.0, 1.0000001, -1.5, nan))
    assert a[0] == 0 and a[1] == np.float32(np.pi) and a[2] == np.float32(np.pi / 2) and np.isnan(a[3:]).all()
    t = ev("atan2", f(0.0, -0.0, 0.0, -0.0, 1.0, -1.0, inf, inf, 1.0, 1.0), f(1.0, 1.0, -1.0, -1.0, 0.0, 0.0, inf, -inf, inf, -inf))
    pi = np.float32(np.pi)
    want = f(0.0, -0.0, pi, -pi, pi / 2, -pi / 2, pi / 4, 3 * np.pi / 4, 0.0, pi)
    assert np.array_equal(t, want) and np.signbit(t[1])
    p = ev("pow", f(nan, 1.0, 0.0, 0.0, -0.0, -0.0, -8.0, -8.0, -1.0, 0.5, 2.0, inf, 0.0, 2.0, 2.0),
           f(0.0, nan, 2.5, -1.0, 3.0, -3.0, 3.0, 0.5, inf, inf, inf, -1.0, 0.0, 200.0, -200.0))
    assert p[0] == 1 and p[1] == 1 and p[2] == 0 and p[3] == inf and p[4] == 0 and np.signbit(p[4]) and p[5] == -inf
    assert p[6] == -512 and np.isnan(p[7]) and p[8] == 1 and p[9] == 0 and p[10] == inf and p[11] == 0 and p[12] == 1
    assert p[13] == inf and p[14] == 0


def test_round_half_away_from_zero_without_double_rounding():
    x = np.array([0.5, -0.5, 1.5, 2.5, 0.49999997, -0.49999997, 8388607.5, 1e10, -0.2, np.nan], dtype=np.float32)
    r = ev("round", x)
    assert r[:9].tolist() == [1, -1, 2, 3, 0, -0.0, 8388608, 1e10, -0.0] and np.isnan(r[9])
    y = RNG.uniform(-300, 300, N).astype(np.float32)
    assert np.array_equal(ev("round", y), np.where(y >= 0, np.floor(y.astype(np.float64) + 0.5), -np.floor(-y.astype(np.float64) + 0.5)).astype(np.float32))
