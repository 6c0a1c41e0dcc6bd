/*
 * rt_oracle.cpp — CPU restatement of the reference render path.
 *
 * ============================ TEST INFRASTRUCTURE ============================
 * This file is the parity ORACLE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (the HIP path behind
 * include/rt_amd.h) never calls into oracle/ and has no CPU fallback.
 * =============================================================================
 *
 * It restates, literally and recursively, the algorithm of
 * foriequal0/homework-18-graphics-raytracer (Rust) — same control flow, same
 * floating-point operation order — so that it can act as the known-answer
 * generator for the iterative GPU kernel.  Each function cites the reference
 * file:line it follows.  Third-party arithmetic that is NOT in /root/reference
 * (un-vendored crates, Cargo.toml:7-15, no lockfile) is restated from those
 * crates' published algorithms:
 *   cgmath 0.16  — Vector3 dot/cross/magnitude/normalize, Matrix3*Vector3,
 *                  Quaternion::from_arc, Quaternion*Vector3, Deg->Rad
 *   palette 0.4  — LinSrgb arithmetic, Mix, into_luma, sRGB encode, u8 format
 *   approx 0.1   — ulps_eq! (used inside cgmath's from_arc)
 *   rand 0.5     — IsaacRng::new_from_u64, Uniform<f32>, Normal (ziggurat)
 *
 * PARITY PIN: the reference has no tests and cannot be built here (no Rust
 * toolchain).  It is pinned by the three full-frame images
 * the reference holds (tests/test_oracle_reference_png.py, all compared after
 * post_process + sRGB/u8 encode, bar: max |diff| 1 and >= 99.99 % of the
 * 3 686 400 channels identical):
 *   report/out_single_epoch.png  the Whitted pass (main.rs:1087-1115)
 *   report/out.png               Whitted + 7 depth-of-field epochs, blur 0.04
 *   report/out_small_blur.png    the same with blur 0.02 (main.rs:1117-1173)
 * (copies under tests/golden/), and by rand 0.5's own published vector for
 * IsaacRng::new_from_u64(0) (tests/test_oracle_rng.py).  The stochastic pass
 * is therefore pinned per pixel: ISAAC seeding and output order,
 * Uniform<f32>, the ziggurat Normal and its tables, shoot_focus,
 * weighted_select, scatter_hit, the is_normal filter and the in-place
 * renormalisation all reproduce two reference-held images to the last u8.
 *
 * Transcendentals: by default the deterministic binary64-evaluated functions of
 * rt_detmath.h (shared with the device so CPU == GPU bit-for-bit).  Build with
 * -DORC_USE_LIBM to call glibc's libm instead — that is what the Rust binary
 * would call — to measure how far the two are apart (tests do both).
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -ffast-math).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"
#include "../homework-18-graphics-raytracer_amd/csrc/rt_detmath.h"
#include "../homework-18-graphics-raytracer_amd/csrc/rt_ziggurat_tables.h"
#include "rt_oracle.h"

namespace orc {

/* ------------------------------------------------------------------------- */
/* libm seam                                                                  */
/* ------------------------------------------------------------------------- */
#ifdef ORC_USE_LIBM
static inline float m_sin(float x) { return ::sinf(x); }
static inline float m_cos(float x) { return ::cosf(x); }
static inline float m_tan(float x) { return ::tanf(x); }
static inline float m_acos(float x) { return ::acosf(x); }
static inline float m_atan2(float y, float x) { return ::atan2f(y, x); }
static inline float m_pow(float x, float y) { return ::powf(x, y); }
#else
static inline float m_sin(float x) { return rtdm::sinf(x); }
static inline float m_cos(float x) { return rtdm::cosf(x); }
static inline float m_tan(float x) { return rtdm::tanf(x); }
static inline float m_acos(float x) { return rtdm::acosf(x); }
static inline float m_atan2(float y, float x) { return rtdm::atan2f(y, x); }
static inline float m_pow(float x, float y) { return rtdm::powf(x, y); }
#endif
static inline float m_sqrt(float x) { return __builtin_sqrtf(x); }

static const float F_PI = 3.14159265358979323846f;       /* std::f32::consts::PI */
static const float F_EPSILON = 1.1920928955078125e-7f;   /* std::f32::EPSILON */

/* ------------------------------------------------------------------------- */
/* cgmath 0.16 Vector3 / Point3 (f32)                                         */
/* ------------------------------------------------------------------------- */
struct V3 {
    float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
/* InnerSpace::dot = mul_element_wise(..).sum() = (x + y) + z */
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float magnitude2(V3 a) { return dot(a, a); }
static inline float magnitude(V3 a) { return m_sqrt(magnitude2(a)); }
/* normalize = normalize_to(1) = self * (1 / magnitude) */
static inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
/* MetricSpace for Point3: distance(self, other) = (other - self).magnitude() */
static inline float distance(V3 self, V3 other) { return magnitude(other - self); }

struct V2 {
    float x, y;
};
static inline V2 operator*(V2 a, float s) { return V2{a.x * s, a.y * s}; }
static inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }

/* palette 0.4 LinSrgb<f32>: all operators are component-wise */
struct Rgb {
    float r, g, b;
};
static inline Rgb rgb(float r, float g, float b) { return Rgb{r, g, b}; }
static inline Rgb black() { return rgb(0.0f, 0.0f, 0.0f); } /* consts.rs:5 */
static inline Rgb operator+(Rgb a, Rgb b) { return rgb(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline Rgb operator-(Rgb a, Rgb b) { return rgb(a.r - b.r, a.g - b.g, a.b - b.b); }
static inline Rgb operator*(Rgb a, float s) { return rgb(a.r * s, a.g * s, a.b * s); }
static inline Rgb operator*(Rgb a, Rgb b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline Rgb operator/(Rgb a, float s) { return rgb(a.r / s, a.g / s, a.b / s); }
/* palette Mix::mix: factor clamped to [0,1]; self + (other - self) * factor */
static inline Rgb mix(Rgb self, Rgb other, float factor) {
    float f = factor < 0.0f ? 0.0f : (factor > 1.0f ? 1.0f : factor);
    return self + (other - self) * f;
}

/* ------------------------------------------------------------------------- */
/* approx 0.1 ulps_eq! for f32 (default epsilon = f32::EPSILON, max_ulps = 4)  */
/* ------------------------------------------------------------------------- */
static inline float signum(float x) {
    if (x != x) return x;
    return rtdm::sign_bit(x) ? -1.0f : 1.0f; /* Rust f32::signum: +-1.0 incl. zeros, NaN for NaN */
}
static inline bool ulps_eq(float a, float b) {
    float diff = a - b;
    if (rtdm::f_abs(diff) <= F_EPSILON) return true;
    if (signum(a) != signum(b)) return false;
    uint32_t ia = rtdm::f32_bits(a), ib = rtdm::f32_bits(b);
    uint32_t d = ia <= ib ? ib - ia : ia - ib;
    return d <= 4u;
}

/* ------------------------------------------------------------------------- */
/* cgmath 0.16 Quaternion                                                      */
/* ------------------------------------------------------------------------- */
struct Quat {
    float s;
    V3 v;
};
/* Quaternion::from_arc(src, dst, None) — quaternion.rs (cgmath 0.16) */
static Quat from_arc(V3 src, V3 dst) {
    float mag_avg = m_sqrt(magnitude2(src) * magnitude2(dst));
    float d = dot(src, dst);
    if (ulps_eq(d, mag_avg)) {
        return Quat{1.0f, v3(0.0f, 0.0f, 0.0f)};
    } else if (ulps_eq(d, -mag_avg)) {
        V3 axis = cross(v3(1.0f, 0.0f, 0.0f), src);
        if (ulps_eq(axis.x, 0.0f) && ulps_eq(axis.y, 0.0f) && ulps_eq(axis.z, 0.0f)) {
            axis = cross(v3(0.0f, 1.0f, 0.0f), src);
        }
        axis = normalize(axis);
        /* Quaternion::from_axis_angle(axis, Rad::turn_div_2()):
         *   (s, c) = sin_cos(angle * 0.5); from_sv(c, axis * s) */
        float half = F_PI * 0.5f;
        float s = m_sin(half), c = m_cos(half);
        return Quat{c, axis * s};
    } else {
        Quat q{mag_avg + d, cross(src, dst)};
        /* Quaternion::normalize: self * (1 / magnitude), magnitude2 = s*s + v.v */
        float m = m_sqrt(q.s * q.s + dot(q.v, q.v));
        float inv = 1.0f / m;
        return Quat{q.s * inv, q.v * inv};
    }
}
/* Quaternion * Vector3: tmp = q.v x vec + vec*q.s ; (q.v x tmp) * 2 + vec */
static inline V3 rotate(Quat q, V3 vec) {
    V3 tmp = cross(q.v, vec) + (vec * q.s);
    return (cross(q.v, tmp) * 2.0f) + vec;
}

/* ------------------------------------------------------------------------- */
/* scene views                                                                 */
/* ------------------------------------------------------------------------- */
enum Face { FRONT = 0, BACK = 1, BOTH = 2 }; /* main.rs:52-57 */
static inline Face invert(Face f) {          /* main.rs:60-66 */
    return f == FRONT ? BACK : (f == BACK ? FRONT : BOTH);
}
enum PrimKind { PRIM_SPHERE = 0, PRIM_TRIANGLE = 1 }; /* primitives.rs:31-34 */

struct Exclusion { /* main.rs:77-81 */
    bool some;
    PrimKind kind;
    uint32_t index;
    Face face;
};
struct Ray { /* main.rs:69-75 */
    V3 origin, direction;
    Face face;
    Exclusion exclude;
};
struct At { /* geometric.rs:42-47 */
    V3 position, normal;
    V2 uv;
};
struct Hit { /* main.rs:139-147 */
    uint32_t object_index;
    Ray ray;
    PrimKind kind;
    uint32_t index;
    At at;
    Face face;
    float distance;
};

/* materials.rs:21-31 */
struct ColorMaterial {
    V3 normal;
    Rgb diffuse_color;
    float shiness;
    Rgb specular_color;
    float smoothness, transparency, refraction_index, opaque_decay;
};

struct World {
    const rt_scene_desc *d;
    mutable uint64_t casts; /* per-thread copy */
};

/* primitives.rs:36-42 Triangle::face_normal */
static inline V3 face_normal(const rt_triangle &t) {
    V3 a = v3(t.vertices[1].position) - v3(t.vertices[0].position);
    V3 b = v3(t.vertices[2].position) - v3(t.vertices[1].position);
    return normalize(cross(a, b));
}
/* primitives.rs:44-46 */
static inline bool backface(const rt_triangle &t, V3 dir) { return dot(face_normal(t), dir) > 0.0f; }

/* main.rs:180-326 World::cast */
static bool cast(const World &w, const Ray &ray, Hit *out) {
    w.casts += 1;
    const rt_scene_desc &s = *w.d;
    bool have = false;
    float nearest = 0.0f;
    Hit best;
    memset(&best, 0, sizeof best);
    for (uint32_t i = 0; i < s.n_triangles; ++i) { /* main.rs:183-262 */
        const rt_triangle &tri = s.triangles[i];
        bool bf = backface(tri, ray.direction);
        if ((bf && ray.face == FRONT) || (!bf && ray.face == BACK)) continue;
        if (ray.exclude.some) {
            bool same_face = ray.exclude.kind == PRIM_TRIANGLE && ray.exclude.index == i;
            bool criteria = ray.exclude.face == FRONT ? !bf : (ray.exclude.face == BACK ? bf : true);
            if (same_face && criteria) continue;
        }
        V3 n = face_normal(tri);
        V3 v0 = v3(tri.vertices[0].position), v1 = v3(tri.vertices[1].position), v2 = v3(tri.vertices[2].position);
        float d = dot(n, v0);
        float t = (d - dot(n, ray.origin)) / dot(n, ray.direction);
        if (t <= 0.0f) continue;
        V3 p = ray.origin + ray.direction * t;
        float area[3] = {
            dot(cross(v2 - v1, p - v1), n),
            dot(cross(v0 - v2, p - v2), n),
            dot(cross(v1 - v0, p - v0), n),
        };
        if (area[0] < 0.0f || area[1] < 0.0f || area[2] < 0.0f) continue;
        if (have && nearest < t) continue;
        float area_of_triangle = dot(cross(v1 - v0, v2 - v0), n);
        V3 bary = v3(area[0], area[1], area[2]) / area_of_triangle;
        /* Matrix3::from_cols(n0,n1,n2) * bary = rows dotted with bary */
        V3 n0 = v3(tri.vertices[0].normal), n1 = v3(tri.vertices[1].normal), n2 = v3(tri.vertices[2].normal);
        V3 tmp = v3(dot(v3(n0.x, n1.x, n2.x), bary), dot(v3(n0.y, n1.y, n2.y), bary), dot(v3(n0.z, n1.z, n2.z), bary));
        V3 normal = bf ? -tmp : tmp;
        V2 uv0{tri.vertices[0].uv[0], tri.vertices[0].uv[1]};
        V2 uv1{tri.vertices[1].uv[0], tri.vertices[1].uv[1]};
        V2 uv2{tri.vertices[2].uv[0], tri.vertices[2].uv[1]};
        V2 uv = uv0 * bary.x + uv1 * bary.y + uv2 * bary.z;
        have = true;
        nearest = t;
        best.object_index = tri.object_index;
        best.ray = ray;
        best.kind = PRIM_TRIANGLE;
        best.index = i;
        best.at = At{p, normal, uv};
        best.distance = t;
        best.face = bf ? BACK : FRONT;
    }
    for (uint32_t i = 0; i < s.n_spheres; ++i) { /* main.rs:264-324 */
        const rt_sphere &sp = s.spheres[i];
        V3 c = v3(sp.center);
        float lsd = magnitude(cross(c - ray.origin, ray.direction));
        if (lsd > sp.radius) continue;
        V3 disp = c - ray.origin;
        float tc = dot(ray.direction, disp);
        float k = m_sqrt(sp.radius * sp.radius - lsd * lsd);
        float t;
        bool bf;
        if (ray.face == FRONT) { t = tc - k; bf = false; }
        else if (ray.face == BACK) { t = tc + k; bf = true; }
        else if (tc < k) { t = tc + k; bf = true; }
        else { t = tc - k; bf = false; }
        if (t <= 0.0f) continue;
        if (ray.exclude.some) {
            bool same_face = ray.exclude.kind == PRIM_SPHERE && ray.exclude.index == i;
            bool criteria = ray.exclude.face == FRONT ? !bf : (ray.exclude.face == BACK ? bf : true);
            if (same_face && criteria) continue;
        }
        if (have && nearest < t) continue;
        V3 p = ray.origin + ray.direction * t;
        V3 tmp = normalize(p - c);
        V3 normal = bf ? -tmp : tmp;
        V2 uv{m_acos(normal.y) / F_PI, m_atan2(normal.z, normal.x) / (F_PI * 2.0f) + 0.5f};
        have = true;
        nearest = t;
        best.object_index = sp.object_index;
        best.ray = ray;
        best.kind = PRIM_SPHERE;
        best.index = i;
        best.at = At{p, normal, uv};
        best.distance = t;
        best.face = bf ? BACK : FRONT;
    }
    if (have) *out = best;
    return have;
}

/* materials.rs:33-37 (ColorMaterial) and 85-103 (GenerativeMaterial) with the
 * closure bodies of main.rs:848-863, 1019-1026 */
static ColorMaterial approx(const rt_material &m, const At &at) {
    ColorMaterial c;
    c.shiness = m.shiness;
    c.specular_color = rgb(m.specular_color[0], m.specular_color[1], m.specular_color[2]);
    c.smoothness = m.smoothness;
    c.transparency = m.transparency;
    c.refraction_index = m.refraction_index;
    c.opaque_decay = m.opaque_decay;
    switch (m.diffuse_fn) {
        case RT_DIFFUSE_STRIPE_V: {
            int32_t cell = rtdm::f32_as_i32(at.uv.y * m.tex_frequency);
            c.diffuse_color = (cell % 2 == 0) ? rgb(m.tex_color_a[0], m.tex_color_a[1], m.tex_color_a[2])
                                              : rgb(m.tex_color_b[0], m.tex_color_b[1], m.tex_color_b[2]);
            break;
        }
        case RT_DIFFUSE_STRIPE_SUM: {
            int32_t cell = rtdm::f32_as_i32((at.uv.x + at.uv.y) * m.tex_frequency);
            c.diffuse_color = (cell % 2 == 0) ? rgb(m.tex_color_a[0], m.tex_color_a[1], m.tex_color_a[2])
                                              : rgb(m.tex_color_b[0], m.tex_color_b[1], m.tex_color_b[2]);
            break;
        }
        default:
            c.diffuse_color = rgb(m.diffuse_color[0], m.diffuse_color[1], m.diffuse_color[2]);
    }
    if (m.normal_fn == RT_NORMAL_WAVE_U) {
        float angle = at.uv.x * m.normal_frequency * 2.0f * F_PI;
        V3 v = v3(m_sin(angle), 0.0f, m_cos(angle));
        c.normal = (dot(v, v3(0.0f, 0.0f, 1.0f)) <= 0.0f) ? -v : v;
    } else {
        c.normal = v3(m.normal);
    }
    return c;
}

/* materials.rs:40-44 */
static inline V3 adjust_normal(const ColorMaterial &m, V3 normal) {
    Quat q = from_arc(v3(0.0f, 0.0f, 1.0f), normal);
    return rotate(q, m.normal);
}

struct Probe { /* materials.rs:9-14 (only the fields that are read) */
    V3 normal, view_direction, light_direction;
};
/* materials.rs:46-53 */
static inline Rgb get_diffuse(const ColorMaterial &m, const Probe &p) {
    float cosine = dot(p.light_direction, p.normal);
    return cosine > 0.0f ? m.diffuse_color * cosine : black();
}
/* materials.rs:55-66 */
static inline Rgb get_specular(const ColorMaterial &m, const Probe &p) {
    float cosine = dot(p.light_direction, p.normal);
    if (cosine <= 0.0f) return black();
    V3 reflected = 2.0f * cosine * p.normal - p.light_direction;
    float specular = 1.0f / (m.smoothness + F_EPSILON);
    float energy_conserving = (specular + 8.0f) / (8.0f * F_PI);
    float rv = dot(reflected, p.view_direction);
    float clamped = (rv != rv) ? 0.0f : (rv > 0.0f ? rv : 0.0f); /* f32::max(0.0): NaN -> 0.0 */
    float amount = m_pow(clamped, specular) * energy_conserving;
    return m.specular_color * amount;
}

struct Directional { /* lights.rs:6-11 */
    bool has_origin;
    V3 origin, direction;
    Rgb color;
};
/* lights.rs:48-93 */
static bool approximate_into_directional(const rt_light &l, V3 position, Directional *out) {
    Rgb color = rgb(l.color[0], l.color[1], l.color[2]);
    switch (l.kind) {
        case RT_LIGHT_DIRECTIONAL:
            out->has_origin = l.has_origin != 0;
            out->origin = v3(l.origin);
            out->direction = v3(l.direction);
            out->color = color;
            return true;
        case RT_LIGHT_SPOT: {
            V3 origin = v3(l.origin), direction = v3(l.direction);
            V3 offset = position - origin;
            /* InnerSpace::angle = acos(dot / (|a| * |b|)) */
            float angle = rtdm::f_abs(m_acos(dot(direction, offset) / (magnitude(direction) * magnitude(offset))));
            float spread = l.angle;
            if (angle > spread) return false;
            float angular = m_pow(1.0f - angle / spread, l.softness + F_EPSILON);
            float dist_att = 1.0f / (magnitude(offset) + F_EPSILON);
            out->has_origin = true;
            out->origin = origin;
            out->direction = normalize(position - origin);
            out->color = color * angular * dist_att;
            return true;
        }
        default: { /* Point */
            V3 origin = v3(l.origin);
            V3 offset = position - origin;
            float dist_att = 1.0f / (magnitude(offset) + F_EPSILON);
            out->has_origin = true;
            out->origin = origin;
            out->direction = normalize(offset);
            out->color = color * dist_att;
            return true;
        }
    }
}

/* main.rs:328-341 */
static Ray get_reflect(const Hit &hit) {
    V3 n = hit.at.normal, l = hit.ray.direction;
    V3 reflected = l - 2.0f * dot(l, n) * n;
    Ray r;
    r.origin = hit.at.position;
    r.direction = normalize(reflected);
    r.face = hit.ray.face;
    r.exclude = Exclusion{true, hit.kind, hit.index, invert(hit.face)};
    return r;
}

enum RefractionKind { ESCAPED, INFINITE, TRAPPED }; /* main.rs:149-158 */
struct Refraction {
    RefractionKind kind;
    float travel_distance;
    Ray escape_ray;
};
/* closure at main.rs:344-352 */
static bool refract(V3 n, V3 l, float k, V3 *out) {
    float c = -dot(l, n);
    if (k * k >= 1.0f - c * c) {
        V3 r = (l + n * c) / k - n * m_sqrt(1.0f - (1.0f - c * c) / (k * k));
        *out = normalize(r);
        return true;
    }
    return false;
}
/* main.rs:343-405 */
static Refraction get_refract(const World &w, const Hit &hit, float max_distance) {
    Refraction res;
    memset(&res, 0, sizeof res);
    float k = approx(w.d->materials[hit.object_index], hit.at).refraction_index;
    V3 refract_in;
    if (!refract(hit.at.normal, hit.ray.direction, k, &refract_in)) {
        res.kind = TRAPPED;
        return res;
    }
    Ray ray_inside;
    ray_inside.origin = hit.at.position;
    ray_inside.direction = normalize(refract_in);
    ray_inside.face = BACK;
    ray_inside.exclude = Exclusion{true, hit.kind, hit.index, FRONT};
    Hit hit_inside;
    if (!cast(w, ray_inside, &hit_inside)) {
        res.kind = INFINITE;
        return res;
    }
    float travel = distance(hit_inside.at.position, hit.at.position);
    V3 out_dir;
    bool have_out = refract(hit_inside.at.normal, hit_inside.ray.direction, 1.0f / k, &out_dir);
    int retry = 0;
    while (!have_out && travel <= max_distance && retry < 10) {
        V3 prev = hit_inside.at.position;
        Ray total_reflect = get_reflect(hit_inside);
        if (!cast(w, total_reflect, &hit_inside)) {
            res.kind = INFINITE;
            return res;
        }
        travel += distance(prev, hit_inside.at.position);
        have_out = refract(hit_inside.at.normal, hit_inside.ray.direction, 1.0f / k, &out_dir);
        retry += 1;
    }
    if (!have_out) {
        res.kind = TRAPPED;
        return res;
    }
    res.kind = ESCAPED;
    res.travel_distance = travel;
    res.escape_ray.origin = hit_inside.at.position;
    res.escape_ray.direction = normalize(out_dir);
    res.escape_ray.face = FRONT;
    res.escape_ray.exclude = Exclusion{true, hit_inside.kind, hit_inside.index, BACK};
    return res;
}

/* main.rs:407-464 */
static Rgb get_shade(const World &w, const Hit &hit) {
    ColorMaterial material = approx(w.d->materials[hit.object_index], hit.at);
    Ray ray = hit.ray;
    V3 normal = adjust_normal(material, hit.at.normal);
    Rgb sum = black();
    for (uint32_t li = 0; li < w.d->n_lights; ++li) {
        Directional light;
        if (!approximate_into_directional(w.d->lights[li], hit.at.position, &light)) continue;
        float cosine = -dot(light.direction, normal);
        if (cosine <= 0.0f) continue;
        Ray shadow;
        shadow.origin = hit.at.position;
        shadow.direction = -light.direction;
        shadow.face = BACK;
        shadow.exclude = Exclusion{true, hit.kind, hit.index, BACK};
        Hit occlusion;
        if (cast(w, shadow, &occlusion)) {
            if (light.has_origin) {
                float occlusion_distance = distance(hit.at.position, occlusion.at.position);
                float light_distance = distance(hit.at.position, light.origin);
                if (occlusion_distance < light_distance) continue;
            } else {
                continue;
            }
        }
        Probe probe{normal, -ray.direction, -light.direction};
        float shiness = material.shiness;
        Rgb diffuse = get_diffuse(material, probe) * light.color;
        Rgb specular = get_specular(material, probe) * light.color;
        sum = sum + diffuse * (1.0f - shiness) + specular * shiness;
    }
    return sum;
}

/* main.rs:466-519 World::ray_trace; TraceState::nested main.rs:673-680 */
static Rgb ray_trace(const World &w, int32_t depth, float contribution, const Ray &ray) {
    const float THRESHOLD = 0.001f;
    if (contribution < THRESHOLD) return black();
    Hit hit;
    if (!cast(w, ray, &hit)) return black();
    ColorMaterial material = approx(w.d->materials[hit.object_index], hit.at);
    float shade_contribution = (1.0f - material.shiness) * (1.0f - material.transparency);
    Rgb shade = (contribution * shade_contribution >= THRESHOLD) ? get_shade(w, hit) : black();
    if (depth <= 0) return shade;
    float reflection_contribution = material.shiness * (1.0f - material.transparency);
    Rgb reflection = black();
    if (contribution * reflection_contribution >= THRESHOLD) {
        Ray reflected = get_reflect(hit);
        reflection = ray_trace(w, depth - 1, contribution * reflection_contribution, reflected);
    }
    float refraction_contribution = material.transparency;
    Rgb refraction = black();
    if (contribution * refraction_contribution > THRESHOLD) {
        Refraction r = get_refract(w, hit, 100.0f);
        if (r.kind == ESCAPED) {
            Rgb s = ray_trace(w, depth - 1, contribution * refraction_contribution, r.escape_ray);
            refraction = s * m_pow(material.opaque_decay, r.travel_distance);
        }
    }
    return shade * shade_contribution + reflection * reflection_contribution + refraction * refraction_contribution;
}

/* main.rs:84-99 Camera::shoot */
static Ray shoot(const rt_camera &cam, float clip_x, float clip_y) {
    V3 toward = normalize(v3(cam.toward));
    V3 right = normalize(cross(toward, v3(cam.up)));
    V3 up = normalize(cross(right, toward));
    V3 x = m_tan(cam.fovy / 2.0f) * right;
    V3 y = m_tan(cam.fovy / 2.0f) * up;
    V3 direction = normalize(clip_x * x + clip_y * y + toward);
    V3 origin = v3(cam.center) + toward * cam.near;
    Ray r;
    r.origin = origin;
    r.direction = direction;
    r.face = FRONT;
    r.exclude = Exclusion{false, PRIM_SPHERE, 0, FRONT};
    return r;
}

/* main.rs:1093-1095 */
static inline void clip_of(uint32_t width, uint32_t height, uint32_t x, uint32_t y, float *cx, float *cy) {
    *cy = ((float)height / 2.0f - (float)y) / (float)height;
    *cx = ((float)x - (float)width / 2.0f) / (float)height;
}

/* ------------------------------------------------------------------------- */
/* post_process (main.rs:748-762) and Image::convert_from (image.rs:55-66)     */
/* ------------------------------------------------------------------------- */

/* palette 0.4: Rgb -> Xyz uses a matrix derived at run time (in T = f32) from
 * the sRGB primaries and the D65 white point; luma is Xyz.y.  Restated from
 * palette's matrix.rs / rgb primaries; crate semantics unverified here. */
static void palette_luma_row(float row[3]) {
    struct X3 { float x, y, z; };
    auto yxy_to_xyz = [](float x, float y, float luma) {
        X3 r{0.0f, luma, 0.0f};
        if (rtdm::is_normal(y)) {
            r.x = luma * x / y;
            r.z = luma * (1.0f - x - y) / y;
        }
        return r;
    };
    X3 r = yxy_to_xyz(0.6400f, 0.3300f, 0.212656f);
    X3 g = yxy_to_xyz(0.3000f, 0.6000f, 0.715158f);
    X3 b = yxy_to_xyz(0.1500f, 0.0600f, 0.072186f);
    float a[9] = {r.x, g.x, b.x, r.y, g.y, b.y, r.z, g.z, b.z};
    /* matrix_inverse */
    float d0 = a[4] * a[8] - a[5] * a[7];
    float d1 = a[3] * a[8] - a[5] * a[6];
    float d2 = a[3] * a[7] - a[4] * a[6];
    float det = a[0] * d0 - a[1] * d1 + a[2] * d2;
    float d3 = a[1] * a[8] - a[2] * a[7];
    float d4 = a[0] * a[8] - a[2] * a[6];
    float d5 = a[0] * a[7] - a[1] * a[6];
    float d6 = a[1] * a[5] - a[2] * a[4];
    float d7 = a[0] * a[5] - a[2] * a[3];
    float d8 = a[0] * a[4] - a[1] * a[3];
    float inv[9] = {d0 / det, -d3 / det, d6 / det, -d1 / det, d4 / det, -d7 / det, d2 / det, -d5 / det, d8 / det};
    /* S = inv * white (D65 = 0.95047, 1.0, 1.08883) */
    float wx = 0.95047f, wy = 1.0f, wz = 1.08883f;
    float sr = (inv[0] * wx) + (inv[1] * wy) + (inv[2] * wz);
    float sg = (inv[3] * wx) + (inv[4] * wy) + (inv[5] * wz);
    float sb = (inv[6] * wx) + (inv[7] * wy) + (inv[8] * wz);
    row[0] = a[3] * sr;
    row[1] = a[4] * sg;
    row[2] = a[5] * sb;
}

static inline float luma_of(const float row[3], float r, float g, float b) {
    return (row[0] * r) + (row[1] * g) + (row[2] * b);
}

/* ------------------------------------------------------------------------- */
/* rand 0.5: IsaacRng (ISAAC-32), Uniform<f32>, Normal (ziggurat), Open01       */
/* Restated from the crate's published algorithms (the crate is not in this    */
/* image); pinned by rand's new_from_u64(0) vector and by the two 7-epoch      */
/* reference images (header of this file).                                     */
/* ------------------------------------------------------------------------- */
struct Isaac {
    uint32_t mem[256];
    uint32_t a, b, c;
    uint32_t results[256];
    uint32_t index;
};

/* IsaacCore::init(mem, rounds = 1) as called by IsaacRng::new_from_u64 */
static void isaac_seed_u64(Isaac *r, uint64_t seed) {
    for (int i = 0; i < 256; ++i) r->mem[i] = 0u;
    r->mem[0] = (uint32_t)seed;
    r->mem[1] = (uint32_t)(seed >> 32);
    /* the golden ratio 0x9e3779b9 passed through mix() four times */
    uint32_t a = 0x1367df5au, b = 0x95d90059u, c = 0xc3163e4bu, d = 0x0f421ad8u;
    uint32_t e = 0xd92a4a78u, f = 0xa51a3c49u, g = 0xc4efea1bu, h = 0x30609119u;
    for (int i = 0; i < 256; i += 8) {
        a += r->mem[i]; b += r->mem[i + 1]; c += r->mem[i + 2]; d += r->mem[i + 3];
        e += r->mem[i + 4]; f += r->mem[i + 5]; g += r->mem[i + 6]; h += r->mem[i + 7];
        a ^= b << 11; d += a; b += c;
        b ^= c >> 2;  e += b; c += d;
        c ^= d << 8;  f += c; d += e;
        d ^= e >> 16; g += d; e += f;
        e ^= f << 10; h += e; f += g;
        f ^= g >> 4;  a += f; g += h;
        g ^= h << 8;  b += g; h += a;
        h ^= a >> 9;  c += h; a += b;
        r->mem[i] = a; r->mem[i + 1] = b; r->mem[i + 2] = c; r->mem[i + 3] = d;
        r->mem[i + 4] = e; r->mem[i + 5] = f; r->mem[i + 6] = g; r->mem[i + 7] = h;
    }
    r->a = r->b = r->c = 0u;
    r->index = 256u; /* BlockRng starts exhausted */
}

/* IsaacCore::generate: the classic ISAAC round; results are stored backwards so that reading them
 * forwards yields the reference implementation's order */
static void isaac_generate(Isaac *r) {
    r->c += 1u;
    uint32_t a = r->a, b = r->b + r->c;
    for (uint32_t i = 0; i < 256u; ++i) {
        uint32_t x = r->mem[i];
        uint32_t mixv;
        switch (i & 3u) {
            case 0: mixv = a ^ (a << 13); break;
            case 1: mixv = a ^ (a >> 6); break;
            case 2: mixv = a ^ (a << 2); break;
            default: mixv = a ^ (a >> 16); break;
        }
        a = mixv + r->mem[(i + 128u) & 255u];
        uint32_t y = a + b + r->mem[(x >> 2) & 255u];
        r->mem[i] = y;
        b = x + r->mem[(y >> 10) & 255u];
        r->results[255u - i] = b;
    }
    r->a = a;
    r->b = b;
}

static uint32_t next_u32(Isaac *r) {
    if (r->index >= 256u) { isaac_generate(r); r->index = 0u; }
    return r->results[r->index++];
}
/* BlockRng::next_u64: low word first; straddles a refill when only one word is left */
static uint64_t next_u64(Isaac *r) {
    if (r->index < 255u) {
        uint64_t x = r->results[r->index], y = r->results[r->index + 1];
        r->index += 2u;
        return (y << 32) | x;
    } else if (r->index >= 256u) {
        isaac_generate(r);
        r->index = 2u;
        return ((uint64_t)r->results[1] << 32) | r->results[0];
    } else {
        uint64_t x = r->results[255];
        isaac_generate(r);
        r->index = 1u;
        return ((uint64_t)r->results[0] << 32) | x;
    }
}
/* UniformFloat<f32>::sample_single(low, high): [1,2) float from 23 random bits, * scale + offset */
static float gen_range_f32(Isaac *r, float low, float high) {
    float scale = high - low;
    float offset = low - scale;
    float value1_2 = rtdm::f32_from_bits((next_u32(r) >> 9) | 0x3f800000u);
    return value1_2 * scale + offset;
}
/* Open01 for f64 */
static double open01_f64(Isaac *r) {
    uint64_t fraction = next_u64(r) >> 12;
    double v = rtdm::f64_from_bits(fraction | 0x3ff0000000000000ull);
    return v - (1.0 - 2.220446049250313e-16 / 2.0);
}
/* Standard for f64: 53 random bits scaled into [0,1) */
static double standard_f64(Isaac *r) {
    uint64_t value = next_u64(r) >> 11;
    return (1.0 / 9007199254740992.0) * (double)value;
}
static const double ZIG_X[257] = RT_ZIG_NORM_X;
static const double ZIG_F[257] = RT_ZIG_NORM_F;
/* f64 exp/ln: the deterministic binary64 kernels of rt_detmath.h (the reference calls libm's) */
static inline double d_exp(double z) { return z < -700.0 ? 0.0 : rtdm::exp_mid(z); }
static inline double d_ln(double x) { return rtdm::log_pos(x); }
/* StandardNormal via ziggurat(symmetric = true) */
static double standard_normal(Isaac *r) {
    for (;;) {
        uint64_t bits = next_u64(r);
        uint32_t i = (uint32_t)(bits & 0xffu);
        double u = rtdm::f64_from_bits((bits >> 12) | 0x4000000000000000ull) - 3.0; /* [2,4) - 3 */
        double x = u * ZIG_X[i];
        double test_x = x < 0.0 ? -x : x;
        if (test_x < ZIG_X[i + 1]) return x;
        if (i == 0u) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                double x_ = open01_f64(r);
                double y_ = open01_f64(r);
                xx = d_ln(x_) / RT_ZIG_NORM_R;
                yy = d_ln(y_);
            }
            return u < 0.0 ? xx - RT_ZIG_NORM_R : RT_ZIG_NORM_R - xx;
        }
        if (ZIG_F[i + 1] + (ZIG_F[i] - ZIG_F[i + 1]) * standard_f64(r) < d_exp(-x * x / 2.0)) return x;
    }
}
/* Normal::new(mean, std_dev).sample */
static double normal_sample(Isaac *r, double mean, double std_dev) { return mean + std_dev * standard_normal(r); }

/* main.rs:101-127 Camera::shoot_focus */
static Ray shoot_focus(const rt_camera &cam, float clip_x, float clip_y, Isaac *rng, float focus, float blur) {
    V3 toward = normalize(v3(cam.toward));
    V3 right = normalize(cross(toward, v3(cam.up)));
    V3 up = normalize(cross(right, toward));
    V3 x = m_tan(cam.fovy / 2.0f) * right;
    V3 y = m_tan(cam.fovy / 2.0f) * up;
    V3 direction = normalize(clip_x * x + clip_y * y + toward);
    float xoffset = (float)normal_sample(rng, 0.0, (double)blur);
    float yoffset = (float)normal_sample(rng, 0.0, (double)blur);
    V3 direction_offset = normalize(direction * focus + x * xoffset + y * yoffset);
    V3 origin = v3(cam.center) + normalize(toward) * cam.near - (x * xoffset + y * yoffset);
    Ray r;
    r.origin = origin;
    r.direction = direction_offset;
    r.face = FRONT;
    r.exclude = Exclusion{false, PRIM_SPHERE, 0, FRONT};
    return r;
}

/* main.rs:652-666 weighted_select over (Diffuse, Reflection, Refraction) */
static int weighted_select(Isaac *rng, const float w[3]) {
    float sum = 0.0f;
    for (int i = 0; i < 3; ++i) sum = sum + w[i];
    float r = gen_range_f32(rng, 0.0f, sum);
    float accum = 0.0f;
    for (int i = 0; i < 3; ++i) {
        accum += w[i];
        if (r < accum) return i;
    }
    return 2;
}

/* nested fn scatter_hit, main.rs:539-554 */
static Hit scatter_hit(Isaac *rng, const Hit &hit, V3 direction, float exponent) {
    float phi = m_acos(m_pow(1.0f - gen_range_f32(rng, 0.0f, 1.0f), exponent));
    float theta = gen_range_f32(rng, -F_PI, F_PI);
    Quat from_z = from_arc(v3(0.0f, 0.0f, 1.0f), normalize(direction));
    V3 new_dir = rotate(from_z, v3(m_sin(phi) * m_cos(theta), m_sin(phi) * m_sin(theta), m_cos(phi)));
    Hit out = hit;
    out.ray.direction = new_dir;
    return out;
}

/* main.rs:521-614.  get_shade(&hit) at line 524 is pure and only used at depth <= 0, so it is evaluated
 * there only (the cast counter counts what is evaluated; the kernel follows the same plan). */
static Rgb distributed_ray_trace(const World &w, int32_t depth, Isaac *rng, const Hit &hit) {
    if (depth <= 0) return get_shade(w, hit);
    ColorMaterial material = approx(w.d->materials[hit.object_index], hit.at);
    const float weights[3] = {(1.0f - material.shiness) * (1.0f - material.transparency),
                              material.shiness * (1.0f - material.transparency), material.transparency};
    int selected = weighted_select(rng, weights);
    if (selected == 0 || selected == 1) {
        Hit scattered = selected == 0 ? scatter_hit(rng, hit, -hit.at.normal, 1.0f)
                                      : scatter_hit(rng, hit, hit.ray.direction, material.smoothness);
        float cosine = -dot(hit.at.normal, scattered.ray.direction);
        if (cosine <= 0.0f) return black();
        Ray reflected = get_reflect(scattered);
        Hit reflected_hit;
        if (cast(w, reflected, &reflected_hit)) {
            Rgb x = distributed_ray_trace(w, depth - 1, rng, reflected_hit);
            Probe probe{scattered.at.normal, -hit.ray.direction, reflected.direction};
            Rgb s = x * (selected == 0 ? get_diffuse(material, probe) : get_specular(material, probe));
            return mix(get_shade(w, reflected_hit), s, 0.5f);
        }
        return get_shade(w, scattered);
    }
    Hit scattered = scatter_hit(rng, hit, hit.ray.direction, material.smoothness);
    float cosine = -dot(hit.at.normal, scattered.ray.direction);
    if (cosine <= 0.0f) return black();
    Refraction r = get_refract(w, scattered, 100.0f);
    if (r.kind != ESCAPED) return black();
    Hit refracted_hit;
    if (!cast(w, r.escape_ray, &refracted_hit)) return black();
    Rgb x = distributed_ray_trace(w, depth - 1, rng, refracted_hit);
    return (x + get_shade(w, refracted_hit)) * m_pow(material.opaque_decay, r.travel_distance);
}

} /* namespace orc */

using namespace orc;

/* ========================================================================= */
/* exported C entry points (see rt_oracle.h)                                  */
/* ========================================================================= */
extern "C" {

int orc_uses_libm(void) {
#ifdef ORC_USE_LIBM
    return 1;
#else
    return 0;
#endif
}

void orc_math(int op, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
            case ORC_MATH_SIN: out[i] = m_sin(x[i]); break;
            case ORC_MATH_COS: out[i] = m_cos(x[i]); break;
            case ORC_MATH_TAN: out[i] = m_tan(x[i]); break;
            case ORC_MATH_ACOS: out[i] = m_acos(x[i]); break;
            case ORC_MATH_ATAN2: out[i] = m_atan2(x[i], y[i]); break;
            case ORC_MATH_POW: out[i] = m_pow(x[i], y[i]); break;
            default: out[i] = 0.0f;
        }
    }
}

void orc_clip(uint32_t width, uint32_t height, uint32_t x, uint32_t y, float *clip_xy) {
    clip_of(width, height, x, y, &clip_xy[0], &clip_xy[1]);
}

static void ray_to_c(const Ray &r, orc_ray *o) {
    o->origin[0] = r.origin.x; o->origin[1] = r.origin.y; o->origin[2] = r.origin.z;
    o->direction[0] = r.direction.x; o->direction[1] = r.direction.y; o->direction[2] = r.direction.z;
    o->face_direction = (uint32_t)r.face;
    o->has_exclude = r.exclude.some ? 1u : 0u;
    o->exclude_kind = (uint32_t)r.exclude.kind;
    o->exclude_index = r.exclude.index;
    o->exclude_face = (uint32_t)r.exclude.face;
}
static Ray ray_from_c(const orc_ray *o) {
    Ray r;
    r.origin = v3(o->origin);
    r.direction = v3(o->direction);
    r.face = (Face)o->face_direction;
    r.exclude = Exclusion{o->has_exclude != 0, (PrimKind)o->exclude_kind, o->exclude_index, (Face)o->exclude_face};
    return r;
}
static void hit_to_c(const Hit &h, orc_hit *o) {
    o->kind = (uint32_t)h.kind;
    o->index = h.index;
    o->object_index = h.object_index;
    o->position[0] = h.at.position.x; o->position[1] = h.at.position.y; o->position[2] = h.at.position.z;
    o->normal[0] = h.at.normal.x; o->normal[1] = h.at.normal.y; o->normal[2] = h.at.normal.z;
    o->uv[0] = h.at.uv.x; o->uv[1] = h.at.uv.y;
    o->face_direction = (uint32_t)h.face;
    o->distance = h.distance;
}
static Hit hit_from_c(const orc_hit *o, const orc_ray *ray) {
    Hit h;
    memset(&h, 0, sizeof h);
    h.kind = (PrimKind)o->kind;
    h.index = o->index;
    h.object_index = o->object_index;
    h.at.position = v3(o->position);
    h.at.normal = v3(o->normal);
    h.at.uv = V2{o->uv[0], o->uv[1]};
    h.face = (Face)o->face_direction;
    h.distance = o->distance;
    h.ray = ray_from_c(ray);
    return h;
}

void orc_shoot(const rt_camera *cam, const float *clip_xy, orc_ray *out) {
    ray_to_c(shoot(*cam, clip_xy[0], clip_xy[1]), out);
}

int orc_cast(const rt_scene_desc *scene, const orc_ray *ray, orc_hit *out) {
    World w{scene, 0};
    Hit h;
    if (!cast(w, ray_from_c(ray), &h)) return 0;
    hit_to_c(h, out);
    return 1;
}

int orc_refract_dir(const float *n, const float *l, float k, float *out) {
    V3 o;
    if (!refract(v3(n), v3(l), k, &o)) return 0;
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
    return 1;
}

void orc_reflect(const orc_hit *hit, const orc_ray *incoming, orc_ray *out) {
    ray_to_c(get_reflect(hit_from_c(hit, incoming)), out);
}

int orc_get_refract(const rt_scene_desc *scene, const orc_hit *hit, const orc_ray *incoming, float max_distance,
                    float *travel, orc_ray *escape) {
    World w{scene, 0};
    Refraction r = get_refract(w, hit_from_c(hit, incoming), max_distance);
    if (r.kind == ESCAPED) {
        *travel = r.travel_distance;
        ray_to_c(r.escape_ray, escape);
    }
    return (int)r.kind;
}

int orc_light_directional(const rt_light *light, const float *position, float *direction, float *color,
                          float *origin, int *has_origin) {
    Directional d;
    if (!approximate_into_directional(*light, v3(position), &d)) return 0;
    direction[0] = d.direction.x; direction[1] = d.direction.y; direction[2] = d.direction.z;
    color[0] = d.color.r; color[1] = d.color.g; color[2] = d.color.b;
    origin[0] = d.origin.x; origin[1] = d.origin.y; origin[2] = d.origin.z;
    *has_origin = d.has_origin ? 1 : 0;
    return 1;
}

void orc_material_approx(const rt_material *m, const float *uv, float *out14) {
    At at{v3(0, 0, 0), v3(0, 0, 0), V2{uv[0], uv[1]}};
    ColorMaterial c = approx(*m, at);
    float v[14] = {c.normal.x, c.normal.y, c.normal.z, c.diffuse_color.r, c.diffuse_color.g, c.diffuse_color.b,
                   c.shiness, c.specular_color.r, c.specular_color.g, c.specular_color.b, c.smoothness,
                   c.transparency, c.refraction_index, c.opaque_decay};
    memcpy(out14, v, sizeof v);
}

void orc_adjust_normal(const float *material_normal, const float *normal, float *out) {
    ColorMaterial c;
    memset(&c, 0, sizeof c);
    c.normal = v3(material_normal);
    V3 r = adjust_normal(c, v3(normal));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void orc_diffuse_specular(const rt_material *m, const float *uv, const float *normal, const float *view,
                          const float *light_dir, float *diffuse, float *specular) {
    At at{v3(0, 0, 0), v3(0, 0, 0), V2{uv[0], uv[1]}};
    ColorMaterial c = approx(*m, at);
    Probe p{v3(normal), v3(view), v3(light_dir)};
    Rgb d = get_diffuse(c, p), s = get_specular(c, p);
    diffuse[0] = d.r; diffuse[1] = d.g; diffuse[2] = d.b;
    specular[0] = s.r; specular[1] = s.g; specular[2] = s.b;
}

void orc_get_shade(const rt_scene_desc *scene, const orc_hit *hit, const orc_ray *incoming, float *rgb3,
                   uint64_t *casts) {
    World w{scene, 0};
    Rgb s = get_shade(w, hit_from_c(hit, incoming));
    rgb3[0] = s.r; rgb3[1] = s.g; rgb3[2] = s.b;
    if (casts) *casts = w.casts;
}

void orc_ray_trace(const rt_scene_desc *scene, const orc_ray *ray, int32_t depth, float contribution, float *rgb3,
                   uint64_t *casts) {
    World w{scene, 0};
    Rgb s = ray_trace(w, depth, contribution, ray_from_c(ray));
    rgb3[0] = s.r; rgb3[1] = s.g; rgb3[2] = s.b;
    if (casts) *casts = w.casts;
}

/* The Whitted driver, main.rs:1087-1109, over an rt_frame tile.  n_threads <= 0
 * means hardware_concurrency (mirrors rayon's global pool). */
static void render_whitted_impl(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float *out_rgb,
                                uint64_t *out_casts, int n_threads, uint32_t *per_pixel_casts) {
    const uint32_t step = frame->y_step ? frame->y_step : 1;
    const uint32_t rows = frame->y1 > frame->y0 ? (frame->y1 - frame->y0 + step - 1) / step : 0;
    const uint32_t cols = frame->x1 > frame->x0 ? frame->x1 - frame->x0 : 0;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    std::atomic<uint32_t> next_row(0);
    std::atomic<uint64_t> total_casts(0);
    auto worker = [&]() {
        World w{scene, 0};
        for (;;) {
            uint32_t r = next_row.fetch_add(1);
            if (r >= rows) break;
            uint32_t y = frame->y0 + r * step;
            for (uint32_t c = 0; c < cols; ++c) {
                uint32_t x = frame->x0 + c;
                float cx, cy;
                clip_of(frame->width, frame->height, x, y, &cx, &cy);
                Ray ray = shoot(*camera, cx, cy);
                const uint64_t before = w.casts;
                Rgb photon = ray_trace(w, frame->max_depth, 1.0f, ray);
                if (per_pixel_casts) per_pixel_casts[(size_t)r * cols + c] = (uint32_t)(w.casts - before);
                float *px = out_rgb + ((size_t)r * cols + c) * 3;
                /* img[at] = img[at] + photon into a zeroed image (main.rs:1107) */
                px[0] = 0.0f + photon.r;
                px[1] = 0.0f + photon.g;
                px[2] = 0.0f + photon.b;
            }
        }
        total_casts.fetch_add(w.casts);
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (out_casts) *out_casts = total_casts.load();
}

void orc_render_whitted(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float *out_rgb,
                        uint64_t *out_casts, int n_threads) {
    render_whitted_impl(scene, camera, frame, out_rgb, out_casts, n_threads, nullptr);
}

/* same, also reporting the number of casts each pixel took (workload analysis) */
void orc_render_whitted_counts(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float *out_rgb,
                               uint32_t *per_pixel_casts, int n_threads) {
    render_whitted_impl(scene, camera, frame, out_rgb, nullptr, n_threads, per_pixel_casts);
}

/* main.rs:748-762.  luma_mode 0: palette matrix-derived row; 1: literal
 * Rec.709 constants.  Returns the p99 luma used (0 if the image has no normal
 * luma — the reference would panic there, main.rs:754). */
float orc_post_process(float *rgb, size_t n_pixels, int luma_mode) {
    float row[3];
    if (luma_mode == 0) palette_luma_row(row);
    else { row[0] = 0.2126f; row[1] = 0.7152f; row[2] = 0.0722f; }
    std::vector<float> lum;
    lum.reserve(n_pixels);
    for (size_t i = 0; i < n_pixels; ++i) {
        float l = luma_of(row, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        if (rtdm::is_normal(l)) lum.push_back(l);
    }
    if (lum.empty()) return 0.0f;
    std::sort(lum.begin(), lum.end());
    size_t idx = (size_t)((float)lum.size() * 0.99f);
    if (idx >= lum.size()) idx = lum.size() - 1;
    float p98 = lum[idx];
    if (p98 > F_EPSILON) {
        for (size_t i = 0; i < n_pixels * 3; ++i) rgb[i] = rgb[i] / p98;
    }
    return p98;
}

void orc_luma_row(int luma_mode, float *row3) {
    if (luma_mode == 0) palette_luma_row(row3);
    else { row3[0] = 0.2126f; row3[1] = 0.7152f; row3[2] = 0.0722f; }
}

/* image.rs:55-66: into_rgb().into_encoding::<Srgb>().into_format::<u8>() */
/* photon.rs:25-28 PhotonAccumulator::accumulate, once per surviving sample (the filter of main.rs:1157-1160 decides
 * which survive), in epoch order */
void orc_accumulate(const float *samples, const uint8_t *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight) {
    for (uint32_t e = 0; e < n_epochs; ++e)
        for (size_t i = 0; i < n_pixels; ++i) {
            if (!valid[(size_t)e * n_pixels + i]) continue;
            const float *ph = samples + ((size_t)e * n_pixels + i) * 3;
            sum[3 * i] = sum[3 * i] + ph[0]; /* self.sum = self.sum + photon */
            sum[3 * i + 1] = sum[3 * i + 1] + ph[1];
            sum[3 * i + 2] = sum[3 * i + 2] + ph[2];
            weight[i] += 1.0f;               /* self.weight_sum += 1.0 */
        }
}

/* photon.rs:15-23 into_rgb_internal: black while weight_sum < f32::EPSILON, else sum / weight_sum */
void orc_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb) {
    for (size_t i = 0; i < n_pixels; ++i) {
        const bool empty = weight[i] < F_EPSILON;
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = empty ? 0.0f : sum[3 * i + c] / weight[i];
    }
}

void orc_encode_srgb8(const float *rgb, size_t n_values, uint8_t *out) {
    for (size_t i = 0; i < n_values; ++i) {
        float x = rgb[i];
        float e = (x <= 0.0031308f) ? 12.92f * x : 1.055f * m_pow(x, 1.0f / 2.4f) - 0.055f;
        float scaled = e * 255.0f;
        /* palette 0.4 Component::convert (f32 -> u8): scale by 255, clamp, then a plain `as u8`
         * cast, i.e. TRUNCATION, not rounding — pinned by tests/golden/ref_out_single_epoch.png,
         * which a rounding conversion misses by +1 on 44 % of the channels */
        float r = scaled;
        if (!(r > 0.0f)) r = 0.0f; /* clamp; NaN -> 0 like a saturating `as u8` */
        if (r > 255.0f) r = 255.0f;
        out[i] = (uint8_t)r;
    }
}


/* ---- distributed pass (main.rs:1117-1161) ---------------------------------- */

size_t orc_rng_state_words(void) { return sizeof(Isaac) / sizeof(uint32_t); }

/* IsaacRng::new_from_u64(y * 2^33 + x) for every pixel of the tile (main.rs:1117-1127); layout:
 * tile-compact pixel index * orc_rng_state_words() */
void orc_rng_init(const rt_frame *frame, uint32_t *states) {
    const uint32_t step = frame->y_step ? frame->y_step : 1;
    const uint32_t rows = frame->y1 > frame->y0 ? (frame->y1 - frame->y0 + step - 1) / step : 0;
    const uint32_t cols = frame->x1 > frame->x0 ? frame->x1 - frame->x0 : 0;
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t c = 0; c < cols; ++c) {
            uint64_t y = frame->y0 + (uint64_t)r * step, x = frame->x0 + c;
            isaac_seed_u64(reinterpret_cast<Isaac *>(states) + ((size_t)r * cols + c), y * (2ull << 32) + x);
        }
}

void orc_rng_draw_u32(uint32_t *state, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = next_u32(reinterpret_cast<Isaac *>(state));
}
void orc_rng_draw_normal(uint32_t *state, double mean, double std_dev, double *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = normal_sample(reinterpret_cast<Isaac *>(state), mean, std_dev);
}
void orc_rng_draw_range_f32(uint32_t *state, float low, float high, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = gen_range_f32(reinterpret_cast<Isaac *>(state), low, high);
}

/* n_epochs passes of the closure at main.rs:1131-1161 over a tile.  For every epoch and pixel the sample
 * is written to samples[(e * pixels + p) * 3 ..] (black when the primary ray misses) and valid[e*pixels+p]
 * says whether it survives the filter at main.rs:1157-1160 (all three channels is_normal).  The RNG states
 * advance in place (the stream continues across epochs). */
void orc_render_distributed(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float focus,
                            float blur, uint32_t *rng_states, uint32_t n_epochs, float *samples, uint8_t *valid,
                            uint64_t *out_casts, int n_threads) {
    const uint32_t step = frame->y_step ? frame->y_step : 1;
    const uint32_t rows = frame->y1 > frame->y0 ? (frame->y1 - frame->y0 + step - 1) / step : 0;
    const uint32_t cols = frame->x1 > frame->x0 ? frame->x1 - frame->x0 : 0;
    const size_t pixels = (size_t)rows * cols;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    std::atomic<uint32_t> next_row(0);
    std::atomic<uint64_t> total_casts(0);
    auto worker = [&]() {
        World w{scene, 0};
        for (;;) {
            uint32_t r = next_row.fetch_add(1);
            if (r >= rows) break;
            uint32_t y = frame->y0 + r * step;
            for (uint32_t c = 0; c < cols; ++c) {
                uint32_t x = frame->x0 + c;
                size_t p = (size_t)r * cols + c;
                Isaac *rng = reinterpret_cast<Isaac *>(rng_states) + p;
                float cx, cy;
                clip_of(frame->width, frame->height, x, y, &cx, &cy);
                for (uint32_t e = 0; e < n_epochs; ++e) { /* epochs of one pixel are sequential in its RNG stream */
                    Ray ray = shoot_focus(*camera, cx, cy, rng, focus, blur);
                    Hit hit;
                    Rgb photon = black();
                    if (cast(w, ray, &hit)) photon = distributed_ray_trace(w, frame->max_depth, rng, hit);
                    float *o = samples + ((size_t)e * pixels + p) * 3;
                    o[0] = photon.r; o[1] = photon.g; o[2] = photon.b;
                    valid[(size_t)e * pixels + p] =
                        (rtdm::is_normal(photon.r) && rtdm::is_normal(photon.g) && rtdm::is_normal(photon.b)) ? 1 : 0;
                }
            }
        }
        total_casts.fetch_add(w.casts);
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (out_casts) *out_casts = total_casts.load();
}

} /* extern "C" */
