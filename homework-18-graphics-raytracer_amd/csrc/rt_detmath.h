/*
 * rt_detmath.h — deterministic f32 transcendentals shared by host and device.
 *
 * Why this exists: the reference calls f32::{sin,cos,tan,acos,atan2,powf}
 * (src/main.rs:89,311-312,508,543-549,856-857; src/materials.rs:63;
 * src/lights.rs:57,64), which lower to the platform libm.  ROCm's device libm
 * does not return the same bits as glibc, and glibc itself dispatches to
 * FMA/non-FMA variants per CPU, so "what libm returns" is not a reproducible
 * target.  These functions evaluate in binary64 using ONLY + - * / sqrt,
 * integer ops and comparisons (all IEEE-754 exact operations, no fma, no libm),
 * then round ONCE to binary32.  Compiled with -ffp-contract=off they return
 * identical bits from g++ on x86-64 and from hipcc on gfx950, and the binary64
 * intermediate (rel. error < 2^-48) makes the result the correctly rounded
 * binary32 value except when the exact value lies within ~2^-24 ulp of a
 * rounding boundary — i.e. it agrees with any "< 1 ulp" libm to within the last
 * bit and with a correctly rounded one almost always.
 *
 * All special cases (NaN, infinities, signed zeros, domain errors) follow
 * C99 Annex F, which is what Rust's f32 methods inherit from libm.
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>
#include "rt_detmath_tables.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

namespace rtdm {

RT_HD uint32_t f32_bits(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
RT_HD float    f32_from_bits(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }
RT_HD uint64_t f64_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
RT_HD double   f64_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

RT_HD bool  is_nan(float x) { return x != x; }
RT_HD bool  is_inf(float x) { return (f32_bits(x) & 0x7fffffffu) == 0x7f800000u; }
RT_HD bool  sign_bit(float x) { return (f32_bits(x) >> 31) != 0; }
RT_HD float f_abs(float x) { return f32_from_bits(f32_bits(x) & 0x7fffffffu); }
RT_HD float quiet_nan() { return f32_from_bits(0x7fc00000u); }
RT_HD float pos_inf() { return f32_from_bits(0x7f800000u); }

/* f32::is_normal (used by the reference at main.rs:751,1159): not zero,
 * subnormal, infinite or NaN. */
RT_HD bool is_normal(float x) {
    uint32_t e = (f32_bits(x) >> 23) & 0xffu;
    return e != 0u && e != 0xffu;
}

RT_HD double d_abs(double x) { return f64_from_bits(f64_bits(x) & 0x7fffffffffffffffull); }

/* IEEE square roots (sqrtss / sqrtsd on the host, correctly rounded expansions
 * on gfx950 — verified bit-for-bit by tests/test_gpu_detmath.py). */
RT_HD float  f_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD double d_sqrt(double x) { return __builtin_sqrt(x); }

/* floor(t + 0.5) for |t| < 2^31, by conversion (no libm rint/floor). */
RT_HD int32_t d_round_to_i32(double t) {
    double u = t + 0.5;
    int32_t k = (int32_t)u;        /* truncation toward zero */
    if ((double)k > u) k -= 1;     /* make it floor */
    return k;
}

/* 2^k as a double, k in [-1022, 1023]. */
RT_HD double d_exp2i(int32_t k) { return f64_from_bits((uint64_t)(int64_t)(1023 + k) << 52); }

/* ---- sin / cos kernels on |r| <= pi/4 (+ rounding slop) ------------------- */

RT_HD double sin_kernel(double r) {
    const double z = r * r;
    double p = RT_DM_INVF21;
    p = p * z - RT_DM_INVF19;
    p = p * z + RT_DM_INVF17;
    p = p * z - RT_DM_INVF15;
    p = p * z + RT_DM_INVF13;
    p = p * z - RT_DM_INVF11;
    p = p * z + RT_DM_INVF9;
    p = p * z - RT_DM_INVF7;
    p = p * z + RT_DM_INVF5;
    p = p * z - RT_DM_INVF3;
    return r + r * (z * p);
}

RT_HD double cos_kernel(double r) {
    const double z = r * r;
    double p = RT_DM_INVF20;
    p = p * z - RT_DM_INVF18;
    p = p * z + RT_DM_INVF16;
    p = p * z - RT_DM_INVF14;
    p = p * z + RT_DM_INVF12;
    p = p * z - RT_DM_INVF10;
    p = p * z + RT_DM_INVF8;
    p = p * z - RT_DM_INVF6;
    p = p * z + RT_DM_INVF4;
    return (1.0 - 0.5 * z) + (z * z) * p;
}

/* Argument reduction: returns r in about [-pi/4, pi/4] and the quadrant q
 * (0..3) such that x = r + q*pi/2 (mod 2 pi).  x must be finite. */
RT_HD double reduce_pio2(float xf, int32_t *q) {
    const double x = (double)xf;
    const uint32_t ax = f32_bits(xf) & 0x7fffffffu;
    if (ax <= 0x3f490fdau) { /* |x| <= pi/4 rounded down */
        *q = 0;
        return x;
    }
    if (ax < 0x49800000u) { /* |x| < 2^20: three-part Cody-Waite, k*PIO2_1 and k*PIO2_2 are exact */
        const int32_t k = d_round_to_i32(x * RT_DM_INV_PIO2);
        const double kd = (double)k;
        double r = x - kd * RT_DM_PIO2_1;
        r = r - kd * RT_DM_PIO2_2;
        r = r - kd * RT_DM_PIO2_3;
        *q = k & 3;
        return r;
    }
    /* |x| >= 2^20: Payne-Hanek with 96 bits of 2/pi selected by the exponent.
     * x = M * 2^(8*idx - 22) with M < 2^31; bits 32..95 of M * W (W = the 96-bit
     * window of 2/pi) are x*2/pi mod 4 in units of 2^-62. */
    const uint32_t tab[24] = RT_DM_TWO_OVER_PI_BITS;
    const uint32_t e = ax >> 23;
    const uint32_t idx = (e >> 3) - 16u;
    const uint32_t shift = e & 7u;
    const uint32_t m = ((ax & 0x007fffffu) | 0x00800000u) << shift;
    uint64_t res0 = (uint64_t)(uint32_t)(m * tab[idx]) << 32;
    const uint64_t res1 = (uint64_t)m * tab[idx + 4];
    const uint64_t res2 = (uint64_t)m * tab[idx + 8];
    res0 = (res0 | (res2 >> 32)) + res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    double r = (double)(int64_t)res0 * (RT_DM_PIO2 * 0x1p-62);
    int32_t qq = (int32_t)(n & 3u);
    if (sign_bit(xf)) { r = -r; qq = (4 - qq) & 3; }
    *q = qq;
    return r;
}

RT_HD float sinf(float x) {
    if (is_nan(x) || is_inf(x)) return quiet_nan();
    if (x == 0.0f) return x; /* sin(+-0) = +-0 */
    int32_t q;
    const double r = reduce_pio2(x, &q);
    double v;
    switch (q) {
        case 0: v = sin_kernel(r); break;
        case 1: v = cos_kernel(r); break;
        case 2: v = -sin_kernel(r); break;
        default: v = -cos_kernel(r); break;
    }
    return (float)v;
}

RT_HD float cosf(float x) {
    if (is_nan(x) || is_inf(x)) return quiet_nan();
    int32_t q;
    const double r = reduce_pio2(x, &q);
    double v;
    switch (q) {
        case 0: v = cos_kernel(r); break;
        case 1: v = -sin_kernel(r); break;
        case 2: v = -cos_kernel(r); break;
        default: v = sin_kernel(r); break;
    }
    return (float)v;
}

/* sinf(x) and cosf(x) of the same argument: one reduction and one evaluation of each kernel instead of two reductions
 * and, in a wave whose lanes fall into different quadrants, two evaluations of each kernel per function.  Same
 * operations on the same values as sinf / cosf above, so the same results bit for bit. */
RT_HD void sincosf(float x, float *s, float *c) {
    if (is_nan(x) || is_inf(x)) {
        *s = quiet_nan();
        *c = quiet_nan();
        return;
    }
    int32_t q;
    const double r = reduce_pio2(x, &q);
    const double sk = sin_kernel(r), ck = cos_kernel(r);
    const double sv = (q & 1) ? ck : sk;
    const double cv = (q & 1) ? sk : ck;
    *s = (x == 0.0f) ? x : (float)((q & 2) ? -sv : sv);
    *c = (float)(((q + 1) & 2) ? -cv : cv);
}

RT_HD float tanf(float x) {
    if (is_nan(x) || is_inf(x)) return quiet_nan();
    if (x == 0.0f) return x; /* tan(+-0) = +-0 */
    int32_t q;
    const double r = reduce_pio2(x, &q);
    const double s = sin_kernel(r), c = cos_kernel(r);
    const double v = (q & 1) ? -(c / s) : (s / c);
    return (float)v;
}

/* ---- atan on [0, inf) in binary64 ----------------------------------------- */

/* atan(t) for 0 <= t <= 1: t = c + delta with c = j/8, atan(t) = atan(c) + atan(u),
 * u = (t - c) / (1 + t*c), |u| <= 1/16, atan(u) by its Taylor series. */
RT_HD double atan_unit(double t) {
    const double tab[9] = RT_DM_ATAN_TAB;
    int32_t j = d_round_to_i32(t * 8.0);
    if (j < 0) j = 0;
    if (j > 8) j = 8;
    const double c = (double)j * 0.125;
    const double u = (t - c) / (1.0 + t * c);
    const double z = u * u;
    double p = RT_DM_INV21;
    p = p * z - RT_DM_INV19;
    p = p * z + RT_DM_INV17;
    p = p * z - RT_DM_INV15;
    p = p * z + RT_DM_INV13;
    p = p * z - RT_DM_INV11;
    p = p * z + RT_DM_INV9;
    p = p * z - RT_DM_INV7;
    p = p * z + RT_DM_INV5;
    p = p * z - RT_DM_INV3;
    return tab[j] + (u + u * (z * p));
}

/* atan2 for y >= 0 (not NaN), any finite or infinite x, not both zero/inf:
 * result in [0, pi]. */
RT_HD double atan2_upper(double y, double x) {
    const double ax = d_abs(x);
    double a;
    if (y <= ax) a = atan_unit(y / ax);
    else a = RT_DM_PIO2 - atan_unit(ax / y);
    return (x < 0.0) ? (RT_DM_PI - a) : a;
}

RT_HD float atan2f(float y, float x) {
    if (is_nan(x) || is_nan(y)) return quiet_nan();
    const bool yneg = sign_bit(y);
    const bool xneg = sign_bit(x);
    double r;
    if (y == 0.0f) {
        r = xneg ? RT_DM_PI : 0.0;                 /* atan2(+-0, -x) = +-pi ; atan2(+-0, +x) = +-0 */
    } else if (x == 0.0f) {
        r = RT_DM_PIO2;                            /* atan2(+-y, +-0) = +-pi/2 */
    } else if (is_inf(x)) {
        if (is_inf(y)) r = xneg ? (3.0 * RT_DM_PIO4) : RT_DM_PIO4;
        else r = xneg ? RT_DM_PI : 0.0;
    } else if (is_inf(y)) {
        r = RT_DM_PIO2;
    } else {
        r = atan2_upper((double)f_abs(y), (double)x);
    }
    const float f = (float)r;
    return yneg ? -f : f;
}

RT_HD float acosf(float x) {
    if (is_nan(x)) return quiet_nan();
    const float ax = f_abs(x);
    if (ax > 1.0f) return quiet_nan();
    if (x == 1.0f) return 0.0f;
    const double xd = (double)x;
    /* (1-x) and (1+x) are exact in binary64 for binary32 x */
    const double s = d_sqrt((1.0 - xd) * (1.0 + xd));
    return (float)atan2_upper(s, xd);
}

/* ---- log / exp in binary64 ------------------------------------------------- */

/* ln(x) for finite x > 0 (binary32 input widened; subnormal binary32 are normal
 * binary64).  x = m * 2^e, m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh((m-1)/(m+1)). */
RT_HD double log_pos(double x) {
    uint64_t b = f64_bits(x);
    int32_t e = (int32_t)((b >> 52) & 0x7ffu) - 1023;
    b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = f64_from_bits(b); /* [1, 2) */
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; } /* m in [0.7071.., 1.4142..] */
    const double s = (m - 1.0) / (m + 1.0); /* |s| <= 0.1716 */
    const double z = s * s;
    double p = RT_DM_INV31;
    p = p * z + RT_DM_INV29;
    p = p * z + RT_DM_INV27;
    p = p * z + RT_DM_INV25;
    p = p * z + RT_DM_INV23;
    p = p * z + RT_DM_INV21;
    p = p * z + RT_DM_INV19;
    p = p * z + RT_DM_INV17;
    p = p * z + RT_DM_INV15;
    p = p * z + RT_DM_INV13;
    p = p * z + RT_DM_INV11;
    p = p * z + RT_DM_INV9;
    p = p * z + RT_DM_INV7;
    p = p * z + RT_DM_INV5;
    p = p * z + RT_DM_INV3;
    const double lm = 2.0 * (s + s * (z * p));
    const double ed = (double)e;
    return ed * RT_DM_LN2_HI + (ed * RT_DM_LN2_LO + lm);
}

/* e^z for |z| <= 200, result a normal binary64. */
RT_HD double exp_mid(double z) {
    const int32_t k = d_round_to_i32(z * RT_DM_INV_LN2);
    const double kd = (double)k;
    const double r = (z - kd * RT_DM_LN2_HI) - kd * RT_DM_LN2_LO; /* |r| <= 0.3466 */
    double p = RT_DM_INVF16;
    p = p * r + RT_DM_INVF15;
    p = p * r + RT_DM_INVF14;
    p = p * r + RT_DM_INVF13;
    p = p * r + RT_DM_INVF12;
    p = p * r + RT_DM_INVF11;
    p = p * r + RT_DM_INVF10;
    p = p * r + RT_DM_INVF9;
    p = p * r + RT_DM_INVF8;
    p = p * r + RT_DM_INVF7;
    p = p * r + RT_DM_INVF6;
    p = p * r + RT_DM_INVF5;
    p = p * r + RT_DM_INVF4;
    p = p * r + RT_DM_INVF3;
    p = p * r + RT_DM_INVF2;
    const double v = 1.0 + (r + (r * r) * p);
    return v * d_exp2i(k);
}

/* y is an integer-valued binary32?  returns 0 = not integer, 1 = odd, 2 = even */
RT_HD int32_t f32_int_class(float y) {
    const uint32_t ay = f32_bits(y) & 0x7fffffffu;
    const uint32_t e = ay >> 23;
    if (e < 127u) return 0;          /* |y| < 1 (and y != 0 handled by caller) */
    if (e >= 151u) return 2;         /* |y| >= 2^24: even integer */
    const uint32_t frac_bits = 150u - e; /* number of fractional mantissa bits, 0..23 */
    const uint32_t mant = (ay & 0x007fffffu) | 0x00800000u;
    if (frac_bits != 0u && (mant & ((1u << frac_bits) - 1u)) != 0u) return 0;
    return ((mant >> frac_bits) & 1u) ? 1 : 2;
}

/* powf with the C99 F.9.4.4 special cases. */
RT_HD float powf(float x, float y) {
    if (y == 0.0f) return 1.0f;                       /* pow(x, +-0) = 1 even for NaN x */
    if (x == 1.0f) return 1.0f;                       /* pow(+1, y) = 1 even for NaN y  */
    if (is_nan(x) || is_nan(y)) return quiet_nan();
    const float ax = f_abs(x);
    const bool xneg = sign_bit(x);
    const bool yneg = sign_bit(y);
    const int32_t yint = f32_int_class(y);
    if (is_inf(y)) {
        if (ax == 1.0f) return 1.0f;                  /* pow(-1, +-inf) = 1 */
        return ((ax < 1.0f) == yneg) ? pos_inf() : 0.0f;
    }
    if (ax == 0.0f) {
        const bool neg_result = xneg && (yint == 1);
        if (yneg) return neg_result ? -pos_inf() : pos_inf();
        return neg_result ? -0.0f : 0.0f;
    }
    if (is_inf(x)) {
        const bool neg_result = xneg && (yint == 1);
        if (yneg) return neg_result ? -0.0f : 0.0f;
        return neg_result ? -pos_inf() : pos_inf();
    }
    if (xneg && yint == 0) return quiet_nan();        /* negative base, non-integer exponent */
    const double z = (double)y * log_pos((double)ax);
    float r;
    if (z > 100.0) r = pos_inf();                     /* > ln(FLT_MAX) = 88.7 */
    else if (z < -110.0) r = 0.0f;                    /* < ln(min subnormal)/1 = -103.3 */
    else r = (float)exp_mid(z);
    return (xneg && yint == 1) ? -r : r;
}

/* f32::round (half away from zero), exact: no "+0.5 then floor" double rounding. */
RT_HD float f_round(float x) {
    const uint32_t ax = f32_bits(x) & 0x7fffffffu;
    if (ax >= 0x4b000000u) return x;            /* |x| >= 2^23, inf, NaN: already integral */
    const float t = (float)(int32_t)x;          /* truncation toward zero, exact */
    const float d = x - t;                      /* exact */
    float r = t;
    if (d >= 0.5f) r = t + 1.0f;
    if (d <= -0.5f) r = t - 1.0f;
    /* keep the sign of zero like libm round() */
    return (r == 0.0f) ? f32_from_bits(f32_bits(x) & 0x80000000u) : r;
}

/* f32 -> i32 `as` cast with Rust's saturating semantics (NaN -> 0), used by
 * the generative materials (main.rs:849,1020). */
RT_HD int32_t f32_as_i32(float x) {
    if (is_nan(x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)x;
}

#ifdef RT_DIAG_FAKE_MATH /* timing experiment only: results are WRONG; measures what the transcendentals cost */
namespace fake {
RT_HD float sinf(float x) { return x * 0.5f; }
RT_HD float cosf(float x) { return 1.0f - x * 0.25f; }
RT_HD void sincosf(float x, float *s, float *c) { *s = x * 0.5f; *c = 1.0f - x * 0.25f; }
RT_HD float acosf(float x) { return 1.5f - x; }
RT_HD float atan2f(float y, float x) { return y + x; }
RT_HD float powf(float x, float y) { return x * 0.5f + y * 1e-6f; }
} /* namespace fake */
#endif

} /* namespace rtdm */

#endif /* RT_DETMATH_H */
