/*
 * rt_device_scene.h — device-resident scene layout.
 *
 * rt_scene_create() converts the ABI arrays (include/rt_amd.h) into these
 * records once.  Everything precomputed here is a pure function of one
 * primitive's vertices, evaluated with the reference's own operation order, so
 * it is bit-identical to what the reference recomputes per ray:
 *
 *   n   = normalize((v1-v0) x (v2-v1))       primitives.rs:36-42, used at main.rs:184,202
 *   d   = n . v0                             main.rs:203
 *   e0  = v2-v1, e1 = v0-v2, e2 = v1-v0      main.rs:219-221
 *   area = ((v1-v0) x (v2-v0)) . n           main.rs:235
 *   r2  = radius * radius                    main.rs:272 (powi(2))
 *
 * Bounding sphere (bc, bq): centre and 1.05 x squared radius of the triangle's minimal enclosing circle.  The
 * intersection loop uses it for a CONSERVATIVE rejection only — a plane hit point farther than that from the
 * centre is outside the triangle by a margin far above the rounding error of the reference's signed areas, so
 * the areas need not be evaluated to know that one of them is negative (rt_cast_asm.h).  It never decides an
 * accept, so results stay bit-identical.  rt_scene_create switches it off (bq = +inf) for triangles where that
 * argument would be thin: slivers (an angle under ~1.15 degrees), non-finite or huge coordinates, triangles tiny
 * against the scene's extent; rays whose origin is far outside the scene do not use it either (filter_origin2).
 *
 * DevTri is 128 bytes; the intersection loop indexes it with a wave-uniform index
 * and fetches its first 112 bytes with scalar loads (s_load_dwordx16 + x8 + x4)
 * into SGPRs (rt_cast_asm.h), or the kernel stages it in LDS (variant bit 0) —
 * either way one fetch serves all 64 lanes.
 */
#ifndef RT_DEVICE_SCENE_H
#define RT_DEVICE_SCENE_H

#include <stdint.h>

namespace rt {

/* DevTri::obj: the object (material) index, and in bit 31 FOLLOWS — the triangle's plane (n and d) is the previous
 * triangle's and both belong to the same segment, e.g. the second half of a square() (main.rs:741-746): the intersection
 * loop then reuses n.d, t, the plane point and the culling / t > 0 mask of the previous triangle instead of recomputing
 * identical values (rt_cast_asm.h).  Bit 30 (WEAK, with bit 31): the two planes are equal only up to the signs of zero
 * components — (-0, 1, 0) after (0, 1, 0), what triangle() (main.rs:730-739) gives the two halves of most axis-aligned
 * squares — and the loop shares the plane only while no lane has n.d == 0, the one case in which a zero's sign reaches a
 * result (rt_cast_asm.h / tools/gen_cast_asm.py).  Purely a saving: the values are the same either way. */
#define RT_MAX_TRIANGLES (1u << 24) /* 2 GiB of DevTri records; the loop's byte offset is 32 bits wide */
#define RT_TRI_FOLLOWS 0x80000000u
#define RT_TRI_FOLLOWS_WEAK 0x40000000u
#define RT_TRI_OBJ_MASK 0x3fffffffu

struct alignas(32) DevTri {
    float n[3];  float d;          /* plane */
    float v0[3]; uint32_t obj;     /* object index | RT_TRI_FOLLOWS */
    float v1[3]; float area;       /* area of the whole triangle (barycentric denominator) */
    float v2[3]; float bq;         /* bounding sphere: 1.05 R^2, +inf = filter off (see below) */
    float e0[3]; float bcx;        /* v2 - v1            | bounding-sphere centre x */
    float e1[3]; float bcy;        /* v0 - v2            |                        y */
    float e2[3]; float bcz;        /* v1 - v0            |                        z */
    float pad4[4];
};
static_assert(sizeof(DevTri) == 128, "DevTri must be 128 bytes");

/* The plane and the bounding sphere of a triangle once more, 32 bytes together: what the PAIR-WISE part of the intersection
 * loop (rt_cast.h cast_pairs: one (ray, triangle) pair per lane, records fetched per lane) needs to decide that a pair can
 * go no further — two 16-byte loads instead of five out of DevTri.  Same values, bit for bit. */
struct alignas(16) DevTriHead {
    float n[3]; float d;
    float bc[3]; float bq;
};
static_assert(sizeof(DevTriHead) == 32, "DevTriHead must be 32 bytes");

/* per-vertex attributes, only read for the winning primitive of a cast */
struct alignas(16) DevTriAttr {
    float n0[3]; float uv0x;
    float n1[3]; float uv0y;
    float n2[3]; float uv1x;
    float uv1y, uv2x, uv2y, pad;
};
static_assert(sizeof(DevTriAttr) == 64, "DevTriAttr must be 64 bytes");

/* The triangles as a pre-order array of NODES.  A LEAF (count != 0) is a run of consecutive triangles; an INNER node
 * (count == 0) covers the leaves that follow it up to skip_to.  The intersection loop walks the array in order (so the
 * reference's sequential nearest-hit rule is kept: triangles are still visited in index order, only ones that nobody can hit
 * are left out): a node with n_normals != 0 is SKIPPED — the walk continues at skip_to — when, for every lane of the wave,
 * the ray's line misses the node's bounding sphere inflated by 5 % (folded into r2_hi), the origin is in the scene's
 * neighbourhood, and the ray is not (nearly) parallel to any face plane below: |n . d| >= 1e-3 for each representative
 * face normal (n_normals = 1..8: one per plane direction, antipodal ones merged), or, for a node with more plane directions
 * than that, (a . d)^2 >= K^2 (d . d) for its normal CONE (n_normals = RT_SEGMENT_CONE: axis a in normals[0][0..2], K^2 in
 * normals[0][3]; rt_api.hip derives K from the cone's half-angle so that the condition implies the per-normal one).  Then
 * every triangle below computes a finite t and a finite plane point that lies outside the triangle's own bounding circle by
 * the margin of the per-triangle rejection above, i.e. the reference would reject it on a negative signed area.
 * rt_scene_create only builds such nodes over triangles all of which qualify for that per-triangle rejection; an inner node
 * is only emitted if it can be skipped at all; a leaf with n_normals == 0 is always visited. */
#define RT_SEGMENT_NORMALS 8
#define RT_SEGMENT_CONE 0xffffffffu
#ifndef RT_LEAF_TRIANGLES
#define RT_LEAF_TRIANGLES 16u
#endif
/* How a clustered leaf's triangles are dealt to the 64 lanes when only a few rays of a wave need it (cast_pairs): the leaf is
 * cut into K chunks of ck triangles, a pass tests R = 64 / ck (ray, chunk) sub-jobs at once, lane l holding triangle l % ck of
 * sub-job l / ck.  rt_scene_create picks K for the best fill and stores the numbers where the record has room: the fourth
 * components of normals[1..3] (a cone uses normals[0] only, explicit normals use three components each).
 *   normals[1][3]  ck | K << 8 | R << 16   (0: the leaf is never tested pair-wise)
 *   normals[2][3]  ceil(65536 / ck): (l * it) >> 16 == l / ck for l < 64
 *   normals[3][3]  ceil(65536 / K) likewise for sub-job numbers below 4096 */
#define RT_SEG_PAIR_WORD(g) __float_as_uint((g).normals[1][3])
#define RT_SEG_PAIR_MCK(g) __float_as_uint((g).normals[2][3])
#define RT_SEG_PAIR_MK(g) __float_as_uint((g).normals[3][3])
struct alignas(16) DevSegment {
    uint32_t first, count; /* count == 0: inner node */
    uint32_t n_normals;    /* 0: plain leaf, always visited */
    float r2_hi;           /* (1.05 R)^2, rounded up */
    float c[3];
    uint32_t skip_to;      /* index of the first node after this one's subtree (leaf: its own index + 1) */
    float normals[RT_SEGMENT_NORMALS][4];
};
static_assert(sizeof(DevSegment) == 160, "DevSegment layout");

struct alignas(32) DevSphere {
    float c[3]; float radius;
    float r2;   uint32_t obj;
    float q_miss; /* a ray whose squared distance-times-|d| from the centre is above this misses for sure: (radius (1 + 2^-22))^2 rounded
                   * up — its square root, correctly rounded, is above `radius` — or +inf when the radius is not a positive finite number
                   * (rt_cast.h cast_finish: a wave all of whose lanes miss that clearly leaves the sphere before the square root) */
    float pad;
};
static_assert(sizeof(DevSphere) == 32, "DevSphere must be 32 bytes");

} /* namespace rt */

#endif
