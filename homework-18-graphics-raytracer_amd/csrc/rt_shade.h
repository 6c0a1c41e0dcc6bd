/*
 * rt_shade.h — per-hit shading arithmetic of the render path (host + device).
 *
 * Restates, in the reference's floating-point operation order:
 *   approx (ColorMaterial / GenerativeMaterial)   materials.rs:33-37, 85-103; closures main.rs:848-863, 1019-1026
 *   adjust_normal                                 materials.rs:40-44  (cgmath Quaternion::from_arc + rotate)
 *   get_diffuse / get_specular                    materials.rs:46-66
 *   approximate_into_directional                  lights.rs:48-93
 *   reflect / refract closures                    main.rs:329, 344-352
 * The traversal/recursion itself lives in rt_kernels.hip.
 */
#ifndef RT_SHADE_H
#define RT_SHADE_H

#include "../../include/rt_amd.h"
#include "rt_vec.h"

namespace rt {

/* materials.rs:21-31 with the per-hit closure outputs resolved */
struct Mat {
    V3 normal;
    V3 diffuse;
    V3 specular;
    float shiness, smoothness, transparency, refraction_index, opaque_decay;
};

/* approx 0.1 `ulps_eq!` on f32 with the default epsilon (f32::EPSILON) and
 * max_ulps (4), as cgmath's from_arc uses it. */
RT_HD bool ulps_eq(float a, float b) {
    if (rtdm::f_abs(a - b) <= RT_F_EPSILON) return true;
    if (a != a || b != b) return false;                       /* signum(NaN) != signum(x) */
    if (rtdm::sign_bit(a) != rtdm::sign_bit(b)) return false; /* f32::signum is +-1 for +-0 too */
    const uint32_t ia = rtdm::f32_bits(a), ib = rtdm::f32_bits(b);
    const uint32_t d = ia <= ib ? ib - ia : ia - ib;
    return d <= 4u;
}

/* ColorMaterial::adjust_normal: rotate the tangent-space normal `mn` by the
 * arc that takes +z to `normal` (cgmath 0.16 Quaternion::from_arc(z, normal, None)
 * followed by Quaternion * Vector3). */
RT_HD V3 adjust_normal(V3 mn, V3 normal) {
    const V3 src = v3(0.0f, 0.0f, 1.0f);
    const float mag_avg = rtdm::f_sqrt(magnitude2(src) * magnitude2(normal));
    const float d = dot(src, normal);
    float qs;
    V3 qv;
    if (ulps_eq(d, mag_avg)) {
        qs = 1.0f;
        qv = v3(0.0f, 0.0f, 0.0f);
    } else if (ulps_eq(d, -mag_avg)) {
        V3 axis = cross(v3(1.0f, 0.0f, 0.0f), src);
        if (ulps_eq(axis.x, 0.0f) && ulps_eq(axis.y, 0.0f) && ulps_eq(axis.z, 0.0f)) axis = cross(v3(0.0f, 1.0f, 0.0f), src);
        axis = normalize(axis);
        const float half = RT_F_PI * 0.5f; /* Rad::turn_div_2() * 0.5 */
        float s, c;
        rtdm::sincosf(half, &s, &c);
        qs = c;
        qv = axis * s;
    } else {
        const float s0 = mag_avg + d;
        const V3 v0 = cross(src, normal);
        const float inv = 1.0f / rtdm::f_sqrt(s0 * s0 + dot(v0, v0));
        qs = s0 * inv;
        qv = v0 * inv;
    }
    const V3 tmp = cross(qv, mn) + (mn * qs);
    return (cross(qv, tmp) * 2.0f) + mn;
}

/* Material::approx at uv.  `need_uv` tells the caller whether uv is read at all
 * (sphere uv costs an acos and an atan2, main.rs:310-313, and only generative
 * materials consume it). */
RT_HD bool material_reads_uv(const rt_material &m) {
    return m.diffuse_fn != RT_DIFFUSE_CONST || m.normal_fn != RT_NORMAL_CONST;
}

RT_HD Mat material_approx(const rt_material &m, float u, float v) {
    Mat c;
    c.shiness = m.shiness;
    c.specular = v3(m.specular_color[0], m.specular_color[1], m.specular_color[2]);
    c.smoothness = m.smoothness;
    c.transparency = m.transparency;
    c.refraction_index = m.refraction_index;
    c.opaque_decay = m.opaque_decay;
    if (m.diffuse_fn == RT_DIFFUSE_CONST) {
        c.diffuse = v3(m.diffuse_color[0], m.diffuse_color[1], m.diffuse_color[2]);
    } else {
        const float arg = (m.diffuse_fn == RT_DIFFUSE_STRIPE_V) ? v * m.tex_frequency : (u + v) * m.tex_frequency;
        const int32_t cell = rtdm::f32_as_i32(arg);
        /* Rust's % keeps the sign of the dividend: only remainder 0 selects colour a */
        c.diffuse = (cell % 2 == 0) ? v3(m.tex_color_a[0], m.tex_color_a[1], m.tex_color_a[2])
                                    : v3(m.tex_color_b[0], m.tex_color_b[1], m.tex_color_b[2]);
    }
    if (m.normal_fn == RT_NORMAL_WAVE_U) {
        const float angle = u * m.normal_frequency * 2.0f * RT_F_PI;
        float wave_s, wave_c;
        rtdm::sincosf(angle, &wave_s, &wave_c);
        const V3 w = v3(wave_s, 0.0f, wave_c);
        c.normal = (dot(w, v3(0.0f, 0.0f, 1.0f)) <= 0.0f) ? -w : w;
    } else {
        c.normal = v3(m.normal[0], m.normal[1], m.normal[2]);
    }
    return c;
}

/* materials.rs:46-53 */
RT_HD V3 get_diffuse(const Mat &m, V3 normal, V3 light_direction) {
    const float cosine = dot(light_direction, normal);
    return cosine > 0.0f ? m.diffuse * cosine : v3(0.0f, 0.0f, 0.0f);
}

/* Is powf(x, y) certainly +0.0f?  For 0 <= x < 1 and finite y > 0: ln x <= x - 1, so y ln x <= y (x - 1); rtdm::powf returns
 * 0.0f as soon as its binary64 product y * ln x is below -110 (rt_detmath.h; e^-110 is far below half the smallest binary32
 * subnormal).  The test leaves a margin of 1 for the two binary32 roundings in y * (x - 1) (relative 2^-23) and the 2^-48
 * of log_pos; x == 0 gives +0 by C99's pow(+0, y > 0).  Phong exponents here reach 1e5 (smoothness 1e-5, main.rs:866, 885,
 * 935): outside the highlight the power underflows, and a wave none of whose lanes is inside one skips the binary64
 * evaluation altogether.  Same bits as evaluating it (tests: GPU == oracle, which always evaluates). */
RT_HD bool pow_underflows_to_zero(float x, float y) {
#ifdef RT_NO_POW_SHORTCUT /* A/B */
    return false;
#endif
    return x >= 0.0f && x < 1.0f && y > 0.0f && y <= 3.0e38f && (x == 0.0f || y * (x - 1.0f) < -111.0f);
}

/* materials.rs:55-66 */
RT_HD V3 get_specular(const Mat &m, V3 normal, V3 view_direction, V3 light_direction) {
    const float cosine = dot(light_direction, normal);
    if (cosine <= 0.0f) return v3(0.0f, 0.0f, 0.0f);
    const V3 reflected = 2.0f * cosine * normal - light_direction;
    const float specular = 1.0f / (m.smoothness + RT_F_EPSILON);
    const float energy_conserving = (specular + 8.0f) / (8.0f * RT_F_PI);
    const float rv = dot(reflected, view_direction);
    const float clamped = (rv > 0.0f) ? rv : 0.0f; /* f32::max(0.0): NaN -> 0.0 */
    float power = 0.0f;
    const bool zero = pow_underflows_to_zero(clamped, specular);
#if defined(__HIP_DEVICE_COMPILE__)
    /* a wave none of whose lanes needs the binary64 evaluation branches around it (a select would evaluate it regardless) */
    if (__builtin_amdgcn_ballot_w64(!zero) != 0ull) power = zero ? 0.0f : rtdm::powf(clamped, specular);
#else
    if (!zero) power = rtdm::powf(clamped, specular);
#endif
    const float amount = power * energy_conserving;
    return m.specular * amount;
}

/* lights.rs:6-11 */
struct DirLight {
    V3 direction;
    V3 color;
};

/* lights.rs:48-93.  Returns false for None (outside the spot cone). */
template <class Light> /* rt_light, or the same record in the constant address space (rt_cast.h uniform_ref: a wave-uniform light index) */
RT_HD bool approximate_into_directional(const Light &l, V3 position, DirLight *out) {
    const V3 color = v3(l.color[0], l.color[1], l.color[2]);
    if (l.kind == RT_LIGHT_DIRECTIONAL) {
        out->direction = v3(l.direction[0], l.direction[1], l.direction[2]);
        out->color = color;
        return true;
    }
    const V3 origin = v3(l.origin[0], l.origin[1], l.origin[2]);
    const V3 offset = position - origin;
    if (l.kind == RT_LIGHT_SPOT) {
        const V3 direction = v3(l.direction[0], l.direction[1], l.direction[2]);
        /* cgmath InnerSpace::angle = acos(dot / (|a| * |b|)) */
        const float angle = rtdm::f_abs(rtdm::acosf(dot(direction, offset) / (magnitude(direction) * magnitude(offset))));
        const float spread = l.angle;
        if (angle > spread) return false;
        const float angular = rtdm::powf(1.0f - angle / spread, l.softness + RT_F_EPSILON);
        const float dist_att = 1.0f / (magnitude(offset) + RT_F_EPSILON);
        out->direction = normalize(position - origin);
        out->color = color * angular * dist_att;
        return true;
    }
    /* Point */
    const float dist_att = 1.0f / (magnitude(offset) + RT_F_EPSILON);
    out->direction = normalize(offset);
    out->color = color * dist_att;
    return true;
}

/* Does get_shade's light loop (main.rs:413-433) reach the shadow cast for this light — approximate_into_directional returns a
 * directional AND its cosine with the adjusted normal is positive?  The same answer as the full evaluation, without a spot
 * light's acos (binary64: a square root, two divisions, a ten-term series) wherever the cosine of the angle is clear of the
 * cone's edge by a margin: cos_in / cos_out = cos(spread) +- 1e-4, rounded outwards, computed once per light by rt_scene_create.
 *   x > cos_in  : the angle is below the spread by ~1e-4, three orders of magnitude more than acosf's last bit: `angle > spread` is false;
 *   x < cos_out : likewise true: None;
 *   otherwise (and for NaN, which fails both compares): the reference's own expression.
 * An x above 1 by rounding makes acosf NaN and `NaN > spread` false — not None — which is also what x > cos_in says; an x below -1
 * (a point behind the light, on its axis) does the same at the other end, so the shortcut to None is only taken for x >= -1.
 * The colour is not evaluated (the consumer of the SHADE item does that, once). */
struct LightAux {
    float cos_in, cos_out;
};
template <class Light, class Aux>
RT_HD bool light_asks(const Light &l, const Aux &aux, V3 position, V3 adj_n, V3 *direction_out) {
    V3 direction;
    if (l.kind == RT_LIGHT_DIRECTIONAL) {
        direction = v3(l.direction[0], l.direction[1], l.direction[2]);
    } else {
        const V3 origin = v3(l.origin[0], l.origin[1], l.origin[2]);
        const V3 offset = position - origin;
        if (l.kind == RT_LIGHT_SPOT) {
            const V3 axis = v3(l.direction[0], l.direction[1], l.direction[2]);
            const float x = dot(axis, offset) / (magnitude(axis) * magnitude(offset)); /* cgmath InnerSpace::angle's argument */
            if (!(x > aux.cos_in)) {
                if (x < aux.cos_out && x >= -1.0f) return false; /* below -1 (by rounding, behind the light on its axis) acosf is NaN as well: the light asks */
                const float angle = rtdm::f_abs(rtdm::acosf(x));
                if (angle > l.angle) return false;
            }
            direction = normalize(position - origin);
        } else {
            direction = normalize(offset);
        }
    }
    *direction_out = direction;
    const float cosine = -dot(direction, adj_n);
    return !(cosine <= 0.0f);
}

/* closure at main.rs:329 + normalize at main.rs:333 */
RT_HD V3 reflect_dir(V3 n, V3 l) { return normalize(l - 2.0f * dot(l, n) * n); }

/* closure at main.rs:344-352 (already normalised once, as `.map(|x| x.normalize())`) */
RT_HD bool refract_dir(V3 n, V3 l, float k, V3 *out) {
    const float c = -dot(l, n);
    if (k * k >= 1.0f - c * c) {
        const V3 r = (l + n * c) / k - n * rtdm::f_sqrt(1.0f - (1.0f - c * c) / (k * k));
        *out = normalize(r);
        return true;
    }
    return false;
}

} /* namespace rt */

#endif /* RT_SHADE_H */
