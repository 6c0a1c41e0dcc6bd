/*
 * rt_cast.h — device-side World::cast (main.rs:180-326) and the hit record, shared by the Whitted kernel
 * (rt_kernels.hip) and the distributed-pass kernel (rt_distributed.hip).  Device code only.
 */
#ifndef RT_CAST_H
#define RT_CAST_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_shade.h"
#include "rt_kernels.h"

namespace rt {

/* Scene records read with a wave-uniform index (a node of the triangle array, a sphere, a light): through the CONSTANT address
 * space, so that the compiler fetches them with scalar loads.  Left as ordinary global pointers it cannot — the kernels store to
 * global memory, and a load that a store might clobber has to be a vector load (global_load + v_readfirstlane, a full memory
 * round trip each and usually one after the other: the node walk of the chain kernel spent 40 % of the cast there).  The scene is
 * immutable while a kernel runs (rt_scene_create uploads it once), which is what the address space promises. */
#define RT_UNIFORM __attribute__((address_space(4)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); /* a built-in vector: HIP's uint4 class cannot be copied out of another address space */
template <class T> __device__ __forceinline__ const RT_UNIFORM T &uniform_ref(const T *p) {
    return *reinterpret_cast<const RT_UNIFORM T *>(reinterpret_cast<uintptr_t>(p));
}

/* ---- ray / hit records ------------------------------------------------------ */

enum : uint32_t { FACE_FRONT = 0u, FACE_BACK = 1u, FACE_BOTH = 2u }; /* main.rs:52-57 */

/* Packed exclusion (main.rs:77-81): 0 = None, else
 *   bit 31 = Some, bits 29..30 = face direction, bits 0..28 = primitive id.
 * Primitive ids: triangle i -> i, sphere i -> n_triangles + i (PrimitiveIndex, primitives.rs:31-34). */
__device__ __forceinline__ uint32_t pack_excl(uint32_t prim, uint32_t face) { return 0x80000000u | (face << 29) | prim; }

struct Ray {
    V3 o, d;
    uint32_t mode; /* FaceDirection of the ray (culling mode) */
    uint32_t excl; /* packed exclusion */
};

struct CastResult {
    float t;       /* travel distance of the nearest hit */
    int32_t prim;  /* -1 = miss */
    uint32_t bf;   /* backface flag of the hit */
    float a0, a1, a2; /* triangle hits: the three signed areas (main.rs:218-222) */
};

/* World::cast, main.rs:180-326.  Convergent: `i` is wave-uniform. */
template <bool USE_LDS, class Scene>
__device__ __forceinline__ CastResult cast(const Scene &sc, const DevTri *__restrict__ lds_tris, const Ray &ray) {
    CastResult best;
    best.prim = -1;
    best.t = 0.0f;
    best.bf = 0u;
    best.a0 = best.a1 = best.a2 = 0.0f;
    bool have = false;

    const bool cull_back = ray.mode == FACE_FRONT;  /* skip backfaces  (main.rs:185) */
    const bool cull_front = ray.mode == FACE_BACK;  /* skip frontfaces (main.rs:186) */
    const bool ex_some = (ray.excl >> 31) != 0u;
    const uint32_t ex_prim = ray.excl & 0x1fffffffu;
    const uint32_t ex_face = (ray.excl >> 29) & 3u;

    const uint32_t nt = sc.n_triangles;
    const DevTri *__restrict__ tris = USE_LDS ? lds_tris : sc.tris;
    for (uint32_t i = 0; i < nt; ++i) {
        const DevTri &T = tris[i];
        const V3 n = v3(T.n[0], T.n[1], T.n[2]);
        const float nd = dot(n, ray.d);
        const bool bf = nd > 0.0f; /* Triangle::backface, primitives.rs:44-46 */
        if (bf ? cull_back : cull_front) continue;
        if (ex_some && ex_prim == i) { /* main.rs:190-200 */
            const bool criteria = ex_face == FACE_FRONT ? !bf : (ex_face == FACE_BACK ? bf : true);
            if (criteria) continue;
        }
        const float t = (T.d - dot(n, ray.o)) / nd; /* main.rs:203-204 */
        if (t <= 0.0f) continue;                     /* NaN passes, as in the reference */
        const V3 p = ray.o + ray.d * t;
        const V3 v0 = v3(T.v0[0], T.v0[1], T.v0[2]);
        const V3 v1 = v3(T.v1[0], T.v1[1], T.v1[2]);
        const V3 v2 = v3(T.v2[0], T.v2[1], T.v2[2]);
        const float a0 = dot(cross(v3(T.e0[0], T.e0[1], T.e0[2]), p - v1), n);
        const float a1 = dot(cross(v3(T.e1[0], T.e1[1], T.e1[2]), p - v2), n);
        const float a2 = dot(cross(v3(T.e2[0], T.e2[1], T.e2[2]), p - v0), n);
        if (a0 < 0.0f || a1 < 0.0f || a2 < 0.0f) continue; /* NaN areas pass (main.rs:224) */
        if (have && best.t < t) continue;                  /* ties: the later primitive wins */
        have = true;
        best.t = t;
        best.prim = (int32_t)i;
        best.bf = bf ? 1u : 0u;
        best.a0 = a0; best.a1 = a1; best.a2 = a2;
    }

    const uint32_t ns = sc.n_spheres;
    for (uint32_t i = 0; i < ns; ++i) { /* main.rs:264-324 */
        const DevSphere &S = sc.spheres[i];
        const V3 c = v3(S.c[0], S.c[1], S.c[2]);
        const V3 disp = c - ray.o;
        const float lsd = magnitude(cross(disp, ray.d));
        if (lsd > S.radius) continue;
        const float tc = dot(ray.d, disp);
        const float k = rtdm::f_sqrt(S.r2 - lsd * lsd);
        float t;
        bool bf;
        if (ray.mode == FACE_FRONT) { t = tc - k; bf = false; }
        else if (ray.mode == FACE_BACK) { t = tc + k; bf = true; }
        else if (tc < k) { t = tc + k; bf = true; }
        else { t = tc - k; bf = false; }
        if (t <= 0.0f) continue;
        if (ex_some && ex_prim == nt + i) {
            const bool criteria = ex_face == FACE_FRONT ? !bf : (ex_face == FACE_BACK ? bf : true);
            if (criteria) continue;
        }
        if (have && best.t < t) continue;
        have = true;
        best.t = t;
        best.prim = (int32_t)(nt + i);
        best.bf = bf ? 1u : 0u;
    }
    return best;
}

/* ---- World::cast with the hand-scheduled triangle loop (rt_cast_asm.h, tools/gen_cast_asm.py) ----------
 * The asm block does main.rs:183-233 for all triangles and returns, per lane, the nearest accepted travel
 * distance and triangle index (best_t = NaN / best_prim = -1 while None: `nearest_t < t` is false for NaN,
 * which is exactly the reference's Option::None case), together with the winner's n.d (whose sign is the backface
 * flag) and three signed areas as they stood when it was accepted.  The sphere loop (main.rs:264-324) stays in C++. */
#ifdef RT_DIAG_STAGES /* diagnostic build: python tools/gen_cast_asm.py --count-stages > csrc/rt_cast_asm_diag.h */
#include "rt_cast_asm_diag.h"
#define RT_STAGE_OPERANDS , "+v"(stage_counts[0]), "+v"(stage_counts[1]), "+v"(stage_counts[2]), "+v"(stage_counts[3]), "+v"(stage_counts[4]), "+v"(stage_counts[5]), "+v"(stage_counts[6]), "+v"(stage_counts[7])
/* [0] planes evaluated (cull), [1] divides, [2] plane points, [3..5] signed areas, [6] accepts, [7] triangles that got past
 * "exclusion + nearest" (leaders and followers), [8] calls of the loop (per wave) */
static __device__ unsigned long long g_stage_totals[9]; /* per translation unit; read with RT_DIAG_STAGE_READER(name) */
#define RT_DIAG_STAGE_READER(name)                                                                              \
    extern "C" int name(unsigned long long *out9, int reset) {                                                  \
        if (hipMemcpyFromSymbol(out9, HIP_SYMBOL(rt::g_stage_totals), 9 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[9] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_stage_totals), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }
#else
#include "rt_cast_asm.h"
#define RT_STAGE_OPERANDS
#endif

/* Lane predicates of one cast as wave-wide masks (SGPR pairs), shared by all its segments */
struct CastMasks {
    unsigned long long keep_back, keep_front, ex_if_back, ex_if_front, filter_ok;
};
__device__ __forceinline__ CastMasks cast_masks(const Ray &ray, float filter_origin2) {
    const bool ex_some = (ray.excl >> 31) != 0u;
    const uint32_t ex_face = (ray.excl >> 29) & 3u;
    CastMasks m;
    m.keep_back = __builtin_amdgcn_ballot_w64(ray.mode != FACE_FRONT); /* backfaces survive culling */
    m.keep_front = __builtin_amdgcn_ballot_w64(ray.mode != FACE_BACK);
    m.ex_if_back = __builtin_amdgcn_ballot_w64(ex_some && ex_face != FACE_FRONT);  /* Back or Both */
    m.ex_if_front = __builtin_amdgcn_ballot_w64(ex_some && ex_face != FACE_BACK);  /* Front or Both */
    /* lanes whose origin is inside the scene's neighbourhood may use the bounding-sphere rejections (rt_device_scene.h) */
    m.filter_ok = __builtin_amdgcn_ballot_w64(dot(ray.o, ray.o) <= filter_origin2);
    return m;
}

/* The nearest accepted triangle hit of a cast so far: best_t = NaN / prim = -1 while None; nd_areas = the winner's n.d
 * (its sign is the backface flag) and its three signed areas as they stood when it was accepted. */
struct TriBest {
    float t;
    int32_t prim; /* global triangle index */
    float nd, a0, a1, a2;
};

/* The asm part over the triangle range [index_base, index_base + n): continues `best` under the reference's sequential
 * rule (ranges must be visited in index order).  The loop addresses the records of a call with a 32-bit byte offset, so
 * n * 128 must stay below 2^32: rt_scene_create refuses scenes of more than RT_MAX_TRIANGLES triangles.  The first triangle
 * of a call is treated as the leader of its plane whatever its record says, so a range may start anywhere. */
__device__ __forceinline__ void cast_asm_triangles(const DevTri *tris_range, uint32_t n, uint32_t index_base, const Ray &ray,
                                                        const CastMasks &m, TriBest *best) {
    const bool ex_some = (ray.excl >> 31) != 0u;
    const uint32_t ex_prim = ray.excl & 0x1fffffffu;
    const uint32_t exid = (ex_some && ex_prim >= index_base && ex_prim - index_base < n) ? ex_prim - index_base : 0xffffffffu;
    float best_t = best->t;
    int32_t local_prim = -1; /* index within the range of a hit accepted in this call */
    float best_nd = best->nd, best_a0 = best->a0, best_a1 = best->a1, best_a2 = best->a2;
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13;
    /* the range is wave-uniform by construction; say so, so that it is passed in SGPRs */
    const unsigned long long ptr_v = (unsigned long long)(uintptr_t)tris_range;
    const unsigned long long ptr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ptr_v >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ptr_v);
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
#ifdef RT_DIAG_STAGES
    uint32_t stage_counts[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#endif
    asm volatile(RT_CAST_ASM_TEXT
                 : "+v"(best_t), "+v"(local_prim), "+v"(best_nd), "+v"(best_a0), "+v"(best_a1), "+v"(best_a2), "=&v"(r0), "=&v"(r1),
                   "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9), "=&v"(r10), "=&v"(r11),
                   "=&v"(r12), "=&v"(r13) RT_STAGE_OPERANDS
                 : "v"(ray.o.x), "v"(ray.o.y), "v"(ray.o.z), "v"(ray.d.x), "v"(ray.d.y), "v"(ray.d.z), "v"(exid), "s"(m.keep_back),
                   "s"(m.keep_front), "s"(m.ex_if_back), "s"(m.ex_if_front), "s"(ptr), "s"(n), "s"(m.filter_ok)
                 : RT_CAST_ASM_CLOBBERS);
#ifdef RT_DIAG_STAGES
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) { /* first active lane: the counts are wave-uniform */
        for (int k = 0; k < 8; ++k) atomicAdd(&g_stage_totals[k], (unsigned long long)stage_counts[k]);
        atomicAdd(&g_stage_totals[8], 1ull); /* asm calls (per wave) */
    }
#endif
    best->t = best_t;
    if (local_prim >= 0) best->prim = local_prim + (int32_t)index_base;
    best->nd = best_nd;
    best->a0 = best_a0;
    best->a1 = best_a1;
    best->a2 = best_a2;
}

/* Can every lane skip a clustered segment (rt_device_scene.h "segments")?  Per lane: the ray's line misses the
 * cluster's bounding sphere by the margin folded into r2_hi, the ray starts inside the scene's neighbourhood, and it is
 * not (nearly) parallel to any of the cluster's face planes — so each of its triangles would compute a finite t, a
 * finite plane point outside its own bounding circle, and reject it on a negative signed area.  Comparisons are written
 * so that NaN anywhere means "cannot skip". */
#ifdef RT_DIAG_NEED /* diagnostic build: how many of the triangle tests a wave runs its lanes actually need (tools/diag_need.py) */
/* wave-casts | triangles visited | lane-tests run (active lanes x visited wave-uniformly + 64 x pair passes) | lane-tests needed |
 * [4] clustered leaves tested pair-wise | [5] pair passes | [6] of them with a pair that reached the signed areas | [7] such pairs |
 * [8] pairs accepted | [9] clustered leaves run wave-uniformly | [10] casts redone for a NaN distance | [11] casts with lanes missing |
 * [16 + n] clustered leaves that n lanes needed */
#define RT_DIAG_NEED_WORDS 96
static __device__ unsigned long long g_need_totals[RT_DIAG_NEED_WORDS];
#define RT_DIAG_NEED_READER(name)                                                                               \
    extern "C" int name(unsigned long long *out4, int reset) {                                                  \
        if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(rt::g_need_totals), RT_DIAG_NEED_WORDS * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[RT_DIAG_NEED_WORDS] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_need_totals), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }
#endif

/* cluster_skippable's condition for ONE lane: this lane's ray gets no accepted hit from the node */
template <class Segment>
__device__ __forceinline__ bool cluster_skippable_lane(const Segment &g, const Ray &ray, const CastMasks &m) {
    if (g.n_normals == 0u) return false;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (((m.filter_ok >> lane) & 1ull) == 0ull) return false;
    const V3 disp = v3(g.c[0], g.c[1], g.c[2]) - ray.o;
    const V3 cr = cross(disp, ray.d);
    const float dd = dot(ray.d, ray.d);
    if (!(dot(cr, cr) > g.r2_hi * dd)) return false;
    bool steep = true;
    if (g.n_normals == RT_SEGMENT_CONE) {
        const float ad = dot(v3(g.normals[0][0], g.normals[0][1], g.normals[0][2]), ray.d);
        steep = ad * ad >= g.normals[0][3] * dd;
    } else {
        for (uint32_t k = 0; k < g.n_normals; ++k)
            steep = steep && rtdm::f_abs(dot(v3(g.normals[k][0], g.normals[k][1], g.normals[k][2]), ray.d)) >= 1.0e-3f;
    }
    return steep;
}

template <class Segment>
__device__ __forceinline__ bool cluster_skippable(const Segment &g, const Ray &ray, const CastMasks &m) {
    const V3 disp = v3(g.c[0], g.c[1], g.c[2]) - ray.o;
    const V3 cr = cross(disp, ray.d);
    const float dd = dot(ray.d, ray.d);
    const bool miss = dot(cr, cr) > g.r2_hi * dd;
    if (__builtin_amdgcn_ballot_w64(miss) != (__builtin_amdgcn_ballot_w64(true) & m.filter_ok) ||
        (__builtin_amdgcn_ballot_w64(true) & ~m.filter_ok) != 0ull)
        return false; /* some lane may hit the sphere, or starts far outside the scene */
    bool steep = true;
    if (g.n_normals == RT_SEGMENT_CONE) { /* a normal cone instead of a list (rt_device_scene.h) */
        const float ad = dot(v3(g.normals[0][0], g.normals[0][1], g.normals[0][2]), ray.d);
        steep = ad * ad >= g.normals[0][3] * dd;
    } else {
        for (uint32_t k = 0; k < g.n_normals; ++k)
            steep = steep && rtdm::f_abs(dot(v3(g.normals[k][0], g.normals[k][1], g.normals[k][2]), ray.d)) >= 1.0e-3f;
    }
    return __builtin_amdgcn_ballot_w64(steep) == __builtin_amdgcn_ballot_w64(true);
}

/* A node's record in ONE round trip: its head and all eight normal slots, scalar loads issued together.  (Read field by field
 * the compiler fetches each where it is first needed — the head, then one normal per trip of a loop — one round trip after the
 * other.) */
struct NodeRec {
    u32x4 h0, h1; /* first, count, n_normals, r2_hi | c[3], skip_to */
    float4 nrm[RT_SEGMENT_NORMALS];
};
__device__ __forceinline__ NodeRec load_node(const DevSegment *node) {
    const RT_UNIFORM u32x4 *gp = reinterpret_cast<const RT_UNIFORM u32x4 *>(reinterpret_cast<uintptr_t>(node));
    NodeRec r;
    r.h0 = gp[0];
    r.h1 = gp[1];
#pragma unroll
    for (uint32_t q = 0; q < RT_SEGMENT_NORMALS; ++q) {
        const u32x4 w = gp[2u + q];
        r.nrm[q] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    }
    return r;
}
/* cluster_skippable_lane in two halves: the ray's line misses the node's sphere (and the ray may use the rejections at all) ... */
__device__ __forceinline__ bool node_missed(const NodeRec &g, const Ray &ray, const CastMasks &m, uint32_t lane) {
    const V3 disp = v3(__uint_as_float(g.h1.x), __uint_as_float(g.h1.y), __uint_as_float(g.h1.z)) - ray.o;
    const V3 cr = cross(disp, ray.d);
    return ((m.filter_ok >> lane) & 1ull) != 0ull && dot(cr, cr) > __uint_as_float(g.h0.w) * dot(ray.d, ray.d);
}
/* ... and it is not (nearly) parallel to a plane below the node: without a branch per normal */
__device__ __forceinline__ bool node_steep(const NodeRec &g, const Ray &ray) {
    const uint32_t n_normals = g.h0.z;
    const bool cone = n_normals == RT_SEGMENT_CONE;
    const float ad = dot(v3(g.nrm[0].x, g.nrm[0].y, g.nrm[0].z), ray.d);
    bool steep = cone ? ad * ad >= g.nrm[0].w * dot(ray.d, ray.d) : rtdm::f_abs(ad) >= 1.0e-3f;
#pragma unroll
    for (uint32_t q = 1; q < 4u; ++q)
        steep = steep & ((rtdm::f_abs(dot(v3(g.nrm[q].x, g.nrm[q].y, g.nrm[q].z), ray.d)) >= 1.0e-3f) | (cone | (q >= n_normals)));
    if (!cone && n_normals > 4u) {
#pragma unroll
        for (uint32_t q = 4u; q < RT_SEGMENT_NORMALS; ++q)
            steep = steep & ((rtdm::f_abs(dot(v3(g.nrm[q].x, g.nrm[q].y, g.nrm[q].z), ray.d)) >= 1.0e-3f) | (q >= n_normals));
    }
    return steep;
}

/* Everything of World::cast after the triangle loop: the winner's backface flag and signed areas, then the
 * sphere loop (main.rs:264-324), starting from the triangles' nearest hit. */
template <class Scene> /* KernelScene, by value or in the kernel-argument segment (constant address space) */
__device__ __forceinline__ CastResult cast_finish(const Scene &sc, const Ray &ray, float best_t, int32_t best_prim,
                                                  const TriBest *kept = nullptr) {
    const uint32_t nt = sc.n_triangles;
    const bool ex_some = (ray.excl >> 31) != 0u;
    const uint32_t ex_prim = ray.excl & 0x1fffffffu;
    const uint32_t ex_face = (ray.excl >> 29) & 3u;
    CastResult best;
    best.prim = best_prim;
    best.t = best_t;
    best.bf = 0u;
    best.a0 = best.a1 = best.a2 = 0.0f;
    bool have = best_prim >= 0;
    if (have && kept != nullptr) { /* kept by the loop at the accept: n.d and the three signed areas of the winner */
        best.bf = kept->nd > 0.0f ? 1u : 0u;
        best.a0 = kept->a0;
        best.a1 = kept->a1;
        best.a2 = kept->a2;
    } else if (have) { /* the winner's backface flag and signed areas (main.rs:184, 218-222), same operations as in the loop */
        const DevTri &T = sc.tris[best_prim];
        const V3 n = v3(T.n[0], T.n[1], T.n[2]);
        best.bf = dot(n, ray.d) > 0.0f ? 1u : 0u;
        const V3 p = ray.o + ray.d * best_t;
        best.a0 = dot(cross(v3(T.e0[0], T.e0[1], T.e0[2]), p - v3(T.v1[0], T.v1[1], T.v1[2])), n);
        best.a1 = dot(cross(v3(T.e1[0], T.e1[1], T.e1[2]), p - v3(T.v2[0], T.v2[1], T.v2[2])), n);
        best.a2 = dot(cross(v3(T.e2[0], T.e2[1], T.e2[2]), p - v3(T.v0[0], T.v0[1], T.v0[2])), n);
    }
    const uint32_t ns = sc.n_spheres;
    for (uint32_t i = 0; i < ns; ++i) { /* main.rs:264-324 */
        const auto &S = uniform_ref(sc.spheres + i);
        const V3 c = v3(S.c[0], S.c[1], S.c[2]);
        const V3 disp = c - ray.o;
        const V3 cr = cross(disp, ray.d);
        const float q = dot(cr, cr);
#ifndef RT_NO_SPHERE_PRETEST /* A/B */
        /* every lane misses clearly: the correctly rounded root of a q above q_miss is above the radius (the root is monotone;
         * rt_device_scene.h), which is main.rs:265-268's `continue` — taken before the root.  NaN compares false: not clear. */
        if (__builtin_amdgcn_ballot_w64(!(q > S.q_miss)) == 0ull) continue;
#endif
        const float lsd = rtdm::f_sqrt(q); /* magnitude(cross(disp, ray.d)) */
        if (lsd > S.radius) continue;
        const float tc = dot(ray.d, disp);
        const float k = rtdm::f_sqrt(S.r2 - lsd * lsd);
        float t;
        bool bf;
        if (ray.mode == FACE_FRONT) { t = tc - k; bf = false; }
        else if (ray.mode == FACE_BACK) { t = tc + k; bf = true; }
        else if (tc < k) { t = tc + k; bf = true; }
        else { t = tc - k; bf = false; }
        if (t <= 0.0f) continue;
        if (ex_some && ex_prim == nt + i) {
            const bool criteria = ex_face == FACE_FRONT ? !bf : (ex_face == FACE_BACK ? bf : true);
            if (criteria) continue;
        }
        if (have && best.t < t) continue;
        have = true;
        best.t = t;
        best.prim = (int32_t)(nt + i);
        best.bf = bf ? 1u : 0u;
    }
    return best;
}

template <class Scene> /* KernelScene, by value or in the kernel-argument segment (constant address space) */
__device__ __forceinline__ CastResult cast_asm(const Scene &sc, const Ray &ray) {
    const CastMasks m = cast_masks(ray, sc.filter_origin2);
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    TriBest best;
    best.t = rtdm::quiet_nan();
    best.prim = -1;
    best.nd = best.a0 = best.a1 = best.a2 = 0.0f;
    /* The triangles in index order: a walk over the node array (rt_device_scene.h) — a node that no lane can hit is left
     * out together with everything below it, an inner node that somebody may hit is descended into, and the triangles of
     * neighbouring leaves that are visited go through the loop in one call (its set-up and first fetches are paid once; the loop
     * is instantiated once). */
    uint32_t run_first = 0u, run_count = 0u;
    const uint32_t n_nodes = sc.n_segments;
#ifdef RT_DIAG_NEED
    const unsigned long long diag_act = __builtin_amdgcn_ballot_w64(true);
    const bool diag_first = __builtin_amdgcn_mbcnt_hi((uint32_t)(diag_act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)diag_act, 0u)) == 0u;
    if (diag_first) atomicAdd(&g_need_totals[0], 1ull);
#endif
    for (uint32_t k = 0;;) {
        bool skip = false, leaf = false, more = k < n_nodes;
        uint32_t g_first = 0u, g_count = 0u, next_k = k;
        if (more) {
            const NodeRec g = load_node(sc.segments + k);
            g_first = g.h0.x;
            g_count = g.h0.y;
            if (g.h0.z != 0u) { /* the normals only if every lane's line misses the sphere */
                const bool missed = node_missed(g, ray, m, lane);
                if (__builtin_amdgcn_ballot_w64(!missed) == 0ull) skip = __builtin_amdgcn_ballot_w64(!node_steep(g, ray)) == 0ull;
            }
            leaf = !skip && g_count != 0u;
            next_k = skip ? g.h1.w : k + 1u;
#ifdef RT_DIAG_NEED
            if (leaf) {
                const unsigned long long need = __builtin_amdgcn_ballot_w64(!(g.h0.z != 0u && node_missed(g, ray, m, lane) && node_steep(g, ray)));
                if (diag_first) {
                    atomicAdd(&g_need_totals[1], (unsigned long long)g_count);
                    atomicAdd(&g_need_totals[2], (unsigned long long)g_count * (unsigned long long)__builtin_popcountll(diag_act));
                    atomicAdd(&g_need_totals[3], (unsigned long long)g_count * (unsigned long long)__builtin_popcountll(need));
                }
            }
#endif
        }
        /* the run gathered so far, when what comes next is not its continuation */
        if (run_count != 0u && (!more || skip || (leaf && run_first + run_count != g_first))) {
            cast_asm_triangles(sc.tris + run_first, run_count, run_first, ray, m, &best);
            run_count = 0u;
        }
        if (!more) break;
        k = next_k;
        if (leaf) {
            if (run_count == 0u) run_first = g_first;
            run_count += g_count;
        }
    }
    return cast_finish(sc, ray, best.t, best.prim, &best);
}

/* ---- World::cast with the sparsely needed clusters tested PAIR-WISE ------------------------------------------------------
 * cast_asm runs a clustered leaf for the whole wave as soon as ONE lane's ray may hit it: a wave of 64 unrelated rays (the
 * scattered rays of the stochastic pass, shadow rays of scattered hit points) visits 56 of the reference scene's 64 triangles
 * per cast while a lane's own ray needs 15 of them (profiles/r02_lane_tests_needed.txt).  Here the few rays that need a leaf
 * pay for it: the leaf's (ray, triangle) pairs are dealt to the 64 lanes — lane l holds triangle l % ck of the (l / ck)-th
 * (ray, chunk) sub-job of the pass (rt_device_scene.h RT_SEG_PAIR_*) — the ray comes over from its owner's lane
 * (ds_bpermute), the triangle's plane and bounding sphere from DevTriHead (two 16-byte loads per lane), and the lane runs the
 * reference's single-triangle test (main.rs:184-224) in the reference's operation order: culling, exclusion, t, `t <= 0`,
 * the conservative bounding-sphere rejection of rt_cast_asm.h, and — for the few pairs that get that far — the three signed
 * areas.  What is left of main.rs:229-233, the sequential nearest rule "replace unless nearest_t < t", picks among all the
 * candidates of a cast the one with the smallest t and, among equal t, the LAST index — as long as no candidate's t is NaN
 * (t <= 0 is gone, so every t is a positive float or +inf and orders like its bits).  So each accepted pair does one
 * ds_min_u64 of (bits(t) << 32 | ~index) on its ray's slot, the wave-uniform loop covers the leaves that many lanes need
 * (same rule, its own running minimum), and the owner takes the smaller key of the two.  A NaN t anywhere — the one case
 * where "sequential" and "minimum" part ways (main.rs:205 lets NaN through, and `nearest_t < NaN` is false) — sends the whole
 * wave through cast_asm instead: rare, and exact without an argument.  The winner's n.d and signed areas (finish_hit's
 * barycentrics) are re-evaluated by its owner: the same operations on the same values.
 *
 * All 64 lanes must be executing (the callers sit in wave-uniform control flow and pass `active`); if some are not, the
 * wave takes cast_asm.  One PairLds per wave. */
#ifndef RT_PAIR_CANDIDATES
#define RT_PAIR_CANDIDATES 128u /* >= 128: a pass may add 64 and the list is drained when it holds more than 64 */
#endif
struct PairLds {
    unsigned long long key[64]; /* per ray (owner lane): the smallest (bits(t) << 32 | ~triangle) accepted pair-wise */
    uint32_t rank_lane[64];     /* the current leaf's needing lanes, in lane order */
    uint2 cand[RT_PAIR_CANDIDATES]; /* pairs that got as far as the signed areas: bits(t), owner lane | triangle << 6 */
    float4 heads[64][2];        /* the current leaf's DevTriHead records (a leaf tested pair-wise has 64 triangles at most) */
    float4 kept[64];            /* per ray: n.d and the three signed areas of the pair that holds its key (what finish_hit needs of the winner) */
};
/* the same without the leaf's plane records: the passes then fetch them from global memory (the L1 holds them), for kernels whose
 * LDS budget decides their occupancy */
struct PairLdsSlim {
    unsigned long long key[64];
    uint32_t rank_lane[64];
    uint2 cand[RT_PAIR_CANDIDATES];
    float4 kept[64];
};
template <class P> struct pair_lds_has_heads { static constexpr bool value = false; };
template <> struct pair_lds_has_heads<PairLds> { static constexpr bool value = true; };
#ifndef RT_PAIR_MAX_NEED
#define RT_PAIR_MAX_NEED 24u /* a clustered leaf that more lanes than this need is run wave-uniformly */
#endif

__device__ __forceinline__ float pair_fetch(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ void pair_sync() { /* LDS written by some lanes of the wave, read by others */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* The signed areas (main.rs:218-224) for the first `count` entries of the candidate list, one per lane; a pair that passes
 * puts its key on its ray's slot.  Returns whether a NaN distance was accepted. */
template <class Scene, class PairScratch>
__device__ __forceinline__ bool pair_areas(const Scene &sc, const Ray &ray, PairScratch *pl, const uint32_t lane, const uint32_t count) {
    const bool mine = lane < count;
    const uint2 c = pl->cand[lane];
    const uint32_t owner = c.y & 63u, tri = mine ? c.y >> 6 : 0u;
    const int from = (int)(owner << 2);
    const V3 o = v3(pair_fetch(from, ray.o.x), pair_fetch(from, ray.o.y), pair_fetch(from, ray.o.z));
    const V3 d = v3(pair_fetch(from, ray.d.x), pair_fetch(from, ray.d.y), pair_fetch(from, ray.d.z));
    bool nan_seen = false, accepted = false;
    unsigned long long key = ~0ull;
    float4 mine_kept = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (mine) {
        const DevTri &T = sc.tris[tri];
        const float t = __uint_as_float(c.x);
        const V3 n = v3(T.n[0], T.n[1], T.n[2]);
        const V3 p = o + d * t; /* main.rs:210 */
        const float a0 = dot(cross(v3(T.e0[0], T.e0[1], T.e0[2]), p - v3(T.v1[0], T.v1[1], T.v1[2])), n);
        const float a1 = dot(cross(v3(T.e1[0], T.e1[1], T.e1[2]), p - v3(T.v2[0], T.v2[1], T.v2[2])), n);
        const float a2 = dot(cross(v3(T.e2[0], T.e2[1], T.e2[2]), p - v3(T.v0[0], T.v0[1], T.v0[2])), n);
        if (!(a0 < 0.0f || a1 < 0.0f || a2 < 0.0f)) { /* NaN areas pass (main.rs:224) */
            nan_seen = t != t;
            accepted = true;
            key = ((unsigned long long)c.x << 32) | (unsigned long long)(0xffffffffu - tri);
            mine_kept = make_float4(dot(n, d), a0, a1, a2);
            __hip_atomic_fetch_min(&pl->key[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifdef RT_DIAG_NEED
            atomicAdd(&g_need_totals[8], 1ull);
#endif
        }
    }
    /* whoever holds its ray's key now leaves the winner's n.d and signed areas beside it (keys are unique per ray: one per triangle; a
     * later, smaller key overwrites) */
    pair_sync();
    if (accepted && pl->key[owner] == key) pl->kept[owner] = mine_kept;
    return nan_seen;
}

#ifdef RT_DIAG_PAIR_TIME
static __device__ unsigned long long g_pair_time[16];
static __device__ unsigned long long g_chain_critical[4]; /* over the waves of the launches since the last read: [0] most steps of one wave, [1] longest wave (ticks), [2] waves, [3] - */
#define RT_DIAG_PAIR_TIME_READER(name)                                                                          \
    extern "C" int name(unsigned long long *out16, int reset) {                                                 \
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rt::g_pair_time), 16 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_pair_time), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }                                                                                                           \
    extern "C" int name##_critical(unsigned long long *out4, int reset) {                                       \
        if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(rt::g_chain_critical), 4 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_chain_critical), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }
#define RT_PAIR_TIME_ARG , unsigned long long *dt
#else
#define RT_PAIR_TIME_ARG
#endif
template <class Scene, class PairScratch>
__device__ __forceinline__ CastResult cast_pairs(const Scene &sc, const Ray &ray, const bool active, PairScratch *pl RT_PAIR_TIME_ARG) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const CastMasks m = cast_masks(ray, sc.filter_origin2);
    /* the ray's lane predicates as bits, to travel with it: excluded triangle (or none) | keep_back keep_front ex_if_back ex_if_front filter_ok */
    const bool ex_some = (ray.excl >> 31) != 0u;
    const uint32_t ex_face = (ray.excl >> 29) & 3u;
    const uint32_t ex_prim = ray.excl & 0x1fffffffu;
    const uint32_t ray_bits = ((ex_some && ex_prim < sc.n_triangles) ? ex_prim : 0x07ffffffu) | (ray.mode != FACE_FRONT ? 1u << 27 : 0u) |
                              (ray.mode != FACE_BACK ? 1u << 28 : 0u) | ((ex_some && ex_face != FACE_FRONT) ? 1u << 29 : 0u) |
                              ((ex_some && ex_face != FACE_BACK) ? 1u << 30 : 0u) | ((uint32_t)((m.filter_ok >> lane) & 1ull) << 31);
    const uint32_t n_nodes = sc.n_segments;
    /* every leaf wave-uniformly — what cast_asm does — when some lanes of the wave are not here to help, and, the second time
     * round, when the first found a NaN distance */
    bool dense_only = __builtin_amdgcn_ballot_w64(true) != ~0ull;
#ifdef RT_DIAG_NEED
    if (dense_only && lane == 0u) atomicAdd(&g_need_totals[11], 1ull);
#endif
#ifdef RT_DIAG_PAIR_TIME
    /* wave time by part: dt[1] classify nodes, [2] wave-uniform runs, [3] leaf set-up, [4] passes, [5] signed areas, [6] finish; [7] the cast */
    unsigned long long tick = __builtin_readcyclecounter();
    const unsigned long long tick0 = tick;
#define RT_PAIR_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); dt[k] += now_ - tick; tick = now_; }
#else
#define RT_PAIR_TICK(k)
#endif
    TriBest best;
    for (;;) {
        best.t = rtdm::quiet_nan();
        best.prim = -1;
        best.nd = best.a0 = best.a1 = best.a2 = 0.0f;
        if (!dense_only) pl->key[lane] = ~0ull;
        bool nan_seen = false;
        uint32_t n_cand = 0u; /* wave-uniform */
        uint32_t run_first = 0u, run_count = 0u;
#ifdef RT_DIAG_NEED
        if (lane == 0u) atomicAdd(&g_need_totals[0], 1ull);
#endif
        for (uint32_t k = 0;;) {
            /* what to do with node k: 0 skip it and its subtree, 1 descend, 2 its triangles wave-uniformly, 3 pair-wise, 4 no node left */
            uint32_t what = 4u, g_first = 0u, g_count = 0u, n_need = 0u, pair_word = 0u, pair_mck = 0u, pair_mk = 0u, next_k = k;
            bool need = false;
            unsigned long long needing = 0ull;
            if (k < n_nodes) {
                const NodeRec g = load_node(sc.segments + k);
                g_first = g.h0.x;
                g_count = g.h0.y;
                const uint32_t n_normals = g.h0.z;
                pair_word = __float_as_uint(g.nrm[1].w);
                pair_mck = __float_as_uint(g.nrm[2].w);
                pair_mk = __float_as_uint(g.nrm[3].w);
                const u32x4 h1 = g.h1;
                need = active;
                if (n_normals != 0u) { /* cluster_skippable_lane */
                    const bool missed = node_missed(g, ray, m, lane), steep = node_steep(g, ray);
                    need = active && !(missed && steep);
                }
                needing = __builtin_amdgcn_ballot_w64(need);
                n_need = (uint32_t)__builtin_popcountll(needing);
                if (needing == 0ull) { what = 0u; next_k = h1.w; }
                else if (g_count == 0u) { what = 1u; next_k = k + 1u; }
                else {
                    what = (dense_only || n_normals == 0u || pair_word == 0u || n_need > RT_PAIR_MAX_NEED) ? 2u : 3u;
                    next_k = k + 1u;
#ifdef RT_DIAG_NEED
                    const unsigned long long diag_active = __builtin_amdgcn_ballot_w64(active);
                    if (lane == 0u) {
                        atomicAdd(&g_need_totals[1], (unsigned long long)g_count);
                        atomicAdd(&g_need_totals[3], (unsigned long long)g_count * n_need);
                        if (n_normals != 0u) atomicAdd(&g_need_totals[16u + n_need], 1ull);
                        if (what == 2u) {
                            atomicAdd(&g_need_totals[2], (unsigned long long)g_count * (unsigned long long)__builtin_popcountll(diag_active));
                            if (n_normals != 0u) atomicAdd(&g_need_totals[9], 1ull);
                        }
                    }
#endif
                }
            }
            RT_PAIR_TICK(1)
            /* the one instance of the wave-uniform loop: the run of leaves gathered so far, when the next thing is not its
             * continuation */
            if (run_count != 0u && (what == 0u || what >= 3u || (what == 2u && run_first + run_count != g_first))) {
                if (active) cast_asm_triangles(sc.tris + run_first, run_count, run_first, ray, m, &best);
                run_count = 0u;
                RT_PAIR_TICK(2)
            }
            if (what == 4u) break;
            k = next_k;
            if (what == 2u) {
                if (run_count == 0u) run_first = g_first;
                run_count += g_count;
            }
            if (what != 3u) continue;

            /* pair-wise: the leaf's plane records into LDS (one per lane), the needing lanes in lane order, then the passes */
            const uint32_t ck = pair_word & 0xffu, K = (pair_word >> 8) & 0xffu, R = pair_word >> 16;
            const uint32_t mck = pair_mck, mk = pair_mk;
            if (need) pl->rank_lane[__builtin_amdgcn_mbcnt_hi((uint32_t)(needing >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needing, 0u))] = lane;
            if constexpr (pair_lds_has_heads<PairScratch>::value) {
                const float4 *h = reinterpret_cast<const float4 *>(sc.heads + g_first + (lane < g_count ? lane : 0u));
                const float4 ha = h[0], hb = h[1];
                pl->heads[lane][0] = ha;
                pl->heads[lane][1] = hb;
            }
            pair_sync();
            const uint32_t slot = (lane * mck) >> 16, in_chunk = lane - slot * ck;
            const uint32_t n_sub = n_need * K;
#ifdef RT_DIAG_NEED
            if (lane == 0u) {
                atomicAdd(&g_need_totals[2], (unsigned long long)((n_sub + R - 1u) / R) * 64ull);
                atomicAdd(&g_need_totals[4], 1ull);
                atomicAdd(&g_need_totals[5], (unsigned long long)((n_sub + R - 1u) / R));
            }
#endif
            RT_PAIR_TICK(3)
            for (uint32_t sub0 = 0u; sub0 < n_sub; sub0 += R) {
                const uint32_t sub = sub0 + slot;
                const uint32_t rank = (sub * mk) >> 16, chunk = sub - rank * K;
                const uint32_t local = chunk * ck + in_chunk;
                const bool pair = slot < R && sub < n_sub && local < g_count;
                const uint32_t owner = pl->rank_lane[rank & 63u] & 63u;
                float4 ha, hb;
                if constexpr (pair_lds_has_heads<PairScratch>::value) {
                    ha = pl->heads[local & 63u][0];
                    hb = pl->heads[local & 63u][1];
                } else {
                    const float4 *h = reinterpret_cast<const float4 *>(sc.heads + g_first + (pair ? local : 0u));
                    ha = h[0];
                    hb = h[1];
                }
                const uint32_t tri = g_first + local;
                const int from = (int)(owner << 2);
                const V3 o = v3(pair_fetch(from, ray.o.x), pair_fetch(from, ray.o.y), pair_fetch(from, ray.o.z));
                const V3 d = v3(pair_fetch(from, ray.d.x), pair_fetch(from, ray.d.y), pair_fetch(from, ray.d.z));
                const uint32_t bits = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)ray_bits);
                const V3 n = v3(ha.x, ha.y, ha.z);
                const float nd = dot(n, d);
                const bool bf = nd > 0.0f; /* Triangle::backface, primitives.rs:44-46 */
                /* culling (main.rs:185-188) and exclusion (main.rs:190-200): the ray's flags for this facing — keep (bit 27 / 28), excluded if it
                 * is the excluded triangle (bit 29 / 30) */
                const uint32_t facing = bits & (bf ? 0x28000000u : 0x50000000u);
                const bool keep = (facing & 0x18000000u) != 0u, ex_facing = (facing & 0x60000000u) != 0u, ex_tri = (bits & 0x07ffffffu) == tri;
                const float t = (ha.w - dot(n, o)) / nd; /* main.rs:203-204 */
                /* `t <= 0` rejects, NaN passes (main.rs:205).  (main.rs:229-233, "nearest_t < t", is the minimum taken at the end.) */
                bool go = pair & keep & !(ex_tri & ex_facing) & !(t <= 0.0f);
                const V3 p = o + d * t;                                     /* main.rs:210 */
                {   /* the conservative bounding-sphere rejection of the wave-uniform loop (rt_cast_asm.h), same operations */
                    const V3 w = p - v3(hb.x, hb.y, hb.z);
                    const float q = (w.x * w.x + w.y * w.y) + w.z * w.z;
                    if (hb.w < q && 1.0e30f > q && (bits >> 31) != 0u) go = false;
                }
                const unsigned long long going = __builtin_amdgcn_ballot_w64(go);
                if (going == 0ull) continue;
                /* on to the signed areas — later, together with the other pairs that get this far */
#ifdef RT_DIAG_NEED
                if (lane == 0u) {
                    atomicAdd(&g_need_totals[6], 1ull);
                    atomicAdd(&g_need_totals[7], (unsigned long long)__builtin_popcountll(going));
                }
#endif
                if (go) pl->cand[n_cand + __builtin_amdgcn_mbcnt_hi((uint32_t)(going >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)going, 0u))] =
                    make_uint2(__float_as_uint(t), owner | (tri << 6));
                n_cand += (uint32_t)__builtin_popcountll(going);
                if (n_cand > RT_PAIR_CANDIDATES - 64u) { /* room for another pass's worth is gone: the first 64 now */
                    RT_PAIR_TICK(4)
                    pair_sync();
                    nan_seen = pair_areas(sc, ray, pl, lane, 64u) || nan_seen;
                    const uint2 moved = pl->cand[64u + lane];
                    pair_sync();
                    pl->cand[lane] = moved;
                    n_cand -= 64u;
                    RT_PAIR_TICK(5)
                }
            }
            RT_PAIR_TICK(4)
        }
        if (dense_only) break;
        pair_sync();
        RT_PAIR_TICK(1)
        if (n_cand != 0u) {
            nan_seen = pair_areas(sc, ray, pl, lane, n_cand) || nan_seen;
            pair_sync();
        }
        RT_PAIR_TICK(5)
        const unsigned long long key = pl->key[lane];
        const bool odd = nan_seen || (active && best.prim >= 0 && best.t != best.t);
        if (__builtin_amdgcn_ballot_w64(odd) != 0ull) { /* a NaN distance somewhere: the reference's sequential rule, literally */
#ifdef RT_DIAG_NEED
            if (lane == 0u) atomicAdd(&g_need_totals[10], 1ull);
#endif
            dense_only = true;
            continue;
        }
        if (active && key != ~0ull) {
            const float kt = __uint_as_float((uint32_t)(key >> 32));
            const int32_t kp = (int32_t)(0xffffffffu - (uint32_t)key);
            if (best.prim < 0 || kt < best.t || (kt == best.t && kp > best.prim)) {
                const float4 w = pl->kept[lane];
                best.t = kt;
                best.prim = kp;
                best.nd = w.x;
                best.a0 = w.y;
                best.a1 = w.z;
                best.a2 = w.w;
            }
        }
        break;
    }
    CastResult cr;
    cr.prim = -1;
    cr.t = 0.0f;
    cr.bf = 0u;
    cr.a0 = cr.a1 = cr.a2 = 0.0f;
    if (active) cr = cast_finish(sc, ray, best.t, best.prim, &best);
#ifdef RT_DIAG_PAIR_TIME
    RT_PAIR_TICK(6)
    dt[7] += tick - tick0;
#endif
    return cr;
}

} /* namespace rt */

#include "rt_cast_bfs.h" /* cast_bfs: the breadth-first walk for scenes beyond the caches */

namespace rt {

/* What the state machine keeps of a Hit (main.rs:139-147). */
struct HitGeom {
    V3 pos, normal;
    float u, v;
    uint32_t prim, bf, obj;
};

/* The tail of the accept branches of World::cast (main.rs:235-252, 304-313),
 * evaluated once for the winning primitive instead of on every improvement. */
template <class Scene> /* KernelScene, by value or in the kernel-argument segment (constant address space) */
__device__ __forceinline__ HitGeom finish_hit(const Scene &sc, const Ray &ray, const CastResult &r, bool want_sphere_uv_always) {
    HitGeom h;
    h.prim = (uint32_t)r.prim;
    h.bf = r.bf;
    h.pos = ray.o + ray.d * r.t;
    if ((uint32_t)r.prim < sc.n_triangles) {
        const DevTri &T = sc.tris[r.prim];
        const DevTriAttr &A = sc.attrs[r.prim];
        h.obj = T.obj & RT_TRI_OBJ_MASK; /* bit 31 is the loop's "same plane as the previous triangle" flag */
        const V3 bary = v3(r.a0, r.a1, r.a2) / T.area;
        /* Matrix3::from_cols(n0,n1,n2) * bary: rows dotted with bary */
        const V3 tmp = v3(dot(v3(A.n0[0], A.n1[0], A.n2[0]), bary),
                          dot(v3(A.n0[1], A.n1[1], A.n2[1]), bary),
                          dot(v3(A.n0[2], A.n1[2], A.n2[2]), bary));
        h.normal = r.bf ? -tmp : tmp;
        h.u = (A.uv0x * bary.x + A.uv1x * bary.y) + A.uv2x * bary.z;
        h.v = (A.uv0y * bary.x + A.uv1y * bary.y) + A.uv2y * bary.z;
    } else {
        const DevSphere &S = sc.spheres[(uint32_t)r.prim - sc.n_triangles];
        h.obj = S.obj;
        const V3 tmp = normalize(h.pos - v3(S.c[0], S.c[1], S.c[2]));
        h.normal = r.bf ? -tmp : tmp;
        h.u = 0.0f;
        h.v = 0.0f;
        /* uv (main.rs:310-313) costs an acos and an atan2 and is only read by
         * generative materials: evaluate it only for those (pure, so identical) */
        if (want_sphere_uv_always || material_reads_uv(sc.materials[h.obj])) {
            h.u = rtdm::acosf(h.normal.y) / RT_F_PI;
            h.v = rtdm::atan2f(h.normal.z, h.normal.x) / (RT_F_PI * 2.0f) + 0.5f;
        }
    }
    return h;
}

} /* namespace rt */

#endif /* RT_CAST_H */
