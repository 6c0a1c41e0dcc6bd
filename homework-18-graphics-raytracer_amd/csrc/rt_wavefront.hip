/*
 * rt_wavefront.hip — the Whitted render path as a level-synchronous wavefront (RT_VARIANT_WAVEFRONT).
 *
 * Same arithmetic as the per-pixel kernel of rt_kernels.hip (and therefore as the reference's recursion,
 * src/main.rs:466-519), organised the other way round: instead of one lane carrying a pixel through its whole ray
 * tree — ~50 dependent casts for the deepest pixels, which is what bounds that kernel's frame time — every
 * ray_trace activation ("node") is a work item, and the frame is a short sequence of wide kernels:
 *
 *   for level L = 0 .. max_depth:
 *     wf_node(L)    one lane per node of the level: the node's own cast (main.rs:473), the hit, the material's
 *                   contributions (478-505); emits the reflection child (get_reflect, 328-341) into level L+1,
 *                   a shade task if get_shade is due, a refraction task if get_refract is due
 *     wf_refr(L)    one lane per refraction task: get_refract's inside cast and total-internal-reflection
 *                   bounces (343-405); emits the escape ray as a node of level L+1
 *   wf_shade        one lane per shade task of ANY level: get_shade's light loop with its shadow casts
 *                   (407-464) — half of all casts of a frame, off the levels' critical path, perfectly convergent
 *                   (the light index is wave-uniform)
 *   for level L = max_depth-1 .. 0:
 *     wf_combine(L) value = (shade*sc + reflection*rc) + (refraction*decay)*fc  (main.rs:516-518) from the
 *                   children's finished values; level 0 writes the pixels
 *
 * A node's subtrees are pure functions of their rays, so evaluating them in this order changes nothing; the
 * association of the combine and every operation inside the helpers (rt_shade.h, rt_cast.h) are the per-pixel
 * kernel's, so the two paths agree bit for bit (tests/test_gpu_wavefront.py).
 *
 * Memory: nodes live in one array in level order (level L+1 is appended while level L runs), 32 B of input
 * (ray + contribution) and 32 B of record (shade term, contributions, decay, child ids) per node; shade tasks
 * 64 B, refraction tasks 48 B.  ~3.4 nodes per pixel on the reference scene at depth 8.  The arrays have a fixed
 * capacity; if a frame overflows it, an overflow flag makes every later wavefront kernel a no-op and the
 * launcher's trailing per-pixel kernel (which is a no-op otherwise) renders the frame instead.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_shade.h"
#include "rt_kernels.h"
#include "rt_cast.h"

namespace rt {

#define WF_NO_CHILD 0xffffffffu /* child slot of a record: no such child (it counts as black) */
#define WF_FINAL 0xfffffffeu    /* record.cr: the record's value is final as stored (miss, or shade at depth 0) */
#define WF_MODE_SHIFT 27u       /* node input: the ray's face mode rides in bits 27-28 of the exclusion word */

#ifndef WF_MIN_WAVES
#define WF_MIN_WAVES 4
#endif

__device__ __forceinline__ uint32_t fu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float uf(uint32_t x) { return __uint_as_float(x); }

/* wave-aggregated append: lanes with `want` get consecutive indices of the list counted by *counter.
 * Returns false (for the whole wave) when the list would outgrow `limit`; the overflow flag is then raised. */
__device__ __forceinline__ bool wave_append(uint32_t *counter, bool want, uint32_t limit, uint32_t *overflow, uint32_t *index) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    *index = 0u;
    if (mask == 0ull) return true;
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    uint32_t base = 0u;
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(counter, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(mask));
    if (base > limit || n > limit - base) {
        if ((threadIdx.x & 63u) == 0u) atomicExch(overflow, 1u);
        return false;
    }
    *index = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return true;
}

__device__ __forceinline__ uint32_t next_chunk(uint32_t *counter) {
    uint32_t c = 0u;
    if ((threadIdx.x & 63u) == 0u) c = atomicAdd(counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
}

__device__ __forceinline__ void add_casts(uint32_t *counters, uint32_t casts) {
    for (int off = 32; off > 0; off >>= 1) casts += __shfl_down(casts, off, 64);
    if ((threadIdx.x & 63u) == 0u && casts != 0u)
        atomicAdd(reinterpret_cast<unsigned long long *>(counters + WF_C_CASTS), (unsigned long long)casts);
}

/* first node id of `level`: the levels before it are complete when a kernel of `level` runs */
__device__ __forceinline__ uint32_t level_base(const uint32_t *counters, uint32_t level) {
    uint32_t base = 0u;
    for (uint32_t j = 0; j < level; ++j) base += counters[WF_C_LEVEL + j];
    return base;
}

/* slot -> pixel of the tile: 8-row bands, column-major inside a band (64 consecutive slots = an 8x8 block) */
__device__ __forceinline__ void slot_to_pixel(const KernelFrame &fr, uint32_t slot, uint32_t *row, uint32_t *col) {
    const uint32_t band_slots = fr.cols << 3;
    const uint32_t band = slot / band_slots;
    const uint32_t r = slot - band * band_slots;
    const uint32_t rows_left = fr.rows - (band << 3);
    const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
    *col = r / band_rows;
    *row = (band << 3) + (r - *col * band_rows);
}

__global__ void wf_init_kernel(uint32_t *counters, uint32_t n_level0) {
    for (uint32_t i = threadIdx.x; i < WF_COUNTER_WORDS; i += blockDim.x) counters[i] = (i == WF_C_LEVEL) ? n_level0 : 0u;
}

/* ---- wf_node: ray_trace's own cast and everything that does not need another cast (main.rs:466-505) ------ */

template <bool FIRST>
__global__ __launch_bounds__(64, WF_MIN_WAVES) void wf_node_kernel(const KernelScene sc, const KernelFrame fr, const WfBuffers wb,
                                                                    const uint32_t level) {
    uint32_t *C = wb.counters;
    if (C[WF_C_OVERFLOW] != 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base = FIRST ? 0u : level_base(C, level);
    const uint32_t count = C[WF_C_LEVEL + level];
    const uint32_t next_base = base + count;                 /* level+1 starts where this level ends */
    const uint32_t next_limit = wb.capacity - next_base;     /* room left for level+1 (count <= capacity - base: no overflow so far) */
    const int32_t depth = fr.max_depth - (int32_t)level;
    const float THRESHOLD = 0.001f; /* main.rs:467 */
    uint32_t casts = 0u;

    for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {
        if ((unsigned long long)chunk * 64ull >= (unsigned long long)count) break;
        const uint32_t idx = chunk * 64u + lane;
        const bool active = idx < count;
        const uint32_t g = base + idx;

        Ray req;
        req.o = v3(0.0f, 0.0f, 0.0f);
        req.d = v3(0.0f, 0.0f, 1.0f);
        req.mode = FACE_FRONT;
        req.excl = 0u;
        float contribution = 1.0f;
        if (active) {
            if (FIRST) {
                /* main.rs:1093-1096 + Camera::shoot (main.rs:84-99), per-frame basis hoisted to the host;
                 * TraceState { depth: max_depth, contribution: 1.0 } (main.rs:1097-1100) */
                uint32_t row, col;
                slot_to_pixel(fr, idx, &row, &col);
                const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                const float clip_y = (fr.half_height - (float)y) / fr.height_f;
                const float clip_x = ((float)x - fr.half_width) / fr.height_f;
                const V3 cx = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
                const V3 cy = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
                const V3 ct = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
                req.o = v3(fr.cam_origin[0], fr.cam_origin[1], fr.cam_origin[2]);
                req.d = normalize(clip_x * cx + clip_y * cy + ct);
            } else {
                const uint4 a = wb.node_in[(size_t)g * 2u], b = wb.node_in[(size_t)g * 2u + 1u];
                req.o = v3(uf(a.x), uf(a.y), uf(a.z));
                req.d = v3(uf(a.w), uf(b.x), uf(b.y));
                req.mode = (b.z >> WF_MODE_SHIFT) & 3u;
                req.excl = b.z & ~(3u << WF_MODE_SHIFT);
                contribution = uf(b.w);
            }
        }

        CastResult cr;
        cr.prim = -1;
        cr.t = 0.0f;
        cr.bf = 0u;
        cr.a0 = cr.a1 = cr.a2 = 0.0f;
        if (active) {
            cr = cast_asm(sc, req);
            casts += 1u;
        }

        /* the node's record; a miss is black and final (main.rs:475) */
        V3 acc = v3(0.0f, 0.0f, 0.0f);
        float rc = 0.0f, fc = 0.0f;
        uint32_t rec_cr = WF_FINAL, rec_cf = WF_NO_CHILD;
        bool want_shade = false, want_refl = false, want_refr = false;
        HitGeom nh;
        nh.pos = nh.normal = v3(0.0f, 0.0f, 0.0f);
        nh.u = nh.v = 0.0f;
        nh.prim = nh.bf = nh.obj = 0u;
        V3 inside_d = v3(0.0f, 0.0f, 0.0f);
        if (active && cr.prim >= 0) {
            nh = finish_hit(sc, req, cr, false);
            const rt_material &rm = sc.materials[nh.obj];
            const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
            want_shade = contribution * shade_contribution >= THRESHOLD; /* main.rs:480-483 */
            if (depth > 0) {
                /* shade * shade_contribution with shade = black; wf_shade overwrites it when get_shade is due */
                acc = v3(0.0f, 0.0f, 0.0f) * shade_contribution;
                rc = rm.shiness * (1.0f - rm.transparency); /* main.rs:493 */
                fc = rm.transparency;                       /* main.rs:502 */
                rec_cr = WF_NO_CHILD;
                want_refl = contribution * rc >= THRESHOLD; /* main.rs:494-495 */
                if (contribution * fc > THRESHOLD) {        /* main.rs:502-505, strict */
                    V3 refract_in;
                    if (refract_dir(nh.normal, req.d, rm.refraction_index, &refract_in)) { /* else Trapped */
                        inside_d = normalize(refract_in); /* second normalize, main.rs:362 */
                        want_refr = true;
                    }
                }
            }
            /* depth <= 0 (main.rs:488-490): the value is the unscaled shade, black unless wf_shade fills it in */
        }

        /* reflection child -> level+1 (get_reflect, main.rs:328-341) */
        uint32_t k_refl, k_shade, k_refr;
        if (wave_append(&C[WF_C_LEVEL + level + 1u], want_refl, next_limit, &C[WF_C_OVERFLOW], &k_refl) && want_refl) {
            const uint32_t child = next_base + k_refl;
            const V3 d = reflect_dir(nh.normal, req.d);
            const uint32_t excl = pack_excl(nh.prim, nh.bf ? FACE_FRONT : FACE_BACK) | (req.mode << WF_MODE_SHIFT);
            wb.node_in[(size_t)child * 2u] = make_uint4(fu(nh.pos.x), fu(nh.pos.y), fu(nh.pos.z), fu(d.x));
            wb.node_in[(size_t)child * 2u + 1u] = make_uint4(fu(d.y), fu(d.z), excl, fu(contribution * rc));
            rec_cr = child;
        }
        /* shade task (any level) */
        if (wave_append(&C[WF_C_SHADE_COUNT], want_shade, wb.capacity, &C[WF_C_OVERFLOW], &k_shade) && want_shade) {
            uint4 *t = wb.shade + (size_t)k_shade * 4u;
            t[0] = make_uint4(g, nh.prim, nh.obj | (depth > 0 ? 0u : 0x80000000u), fu(nh.u));
            t[1] = make_uint4(fu(nh.v), fu(nh.pos.x), fu(nh.pos.y), fu(nh.pos.z));
            t[2] = make_uint4(fu(nh.normal.x), fu(nh.normal.y), fu(nh.normal.z), fu(req.d.x));
            t[3] = make_uint4(fu(req.d.y), fu(req.d.z), 0u, 0u);
        }
        /* refraction task of this level: the ray into the glass (main.rs:358-366) */
        if (wave_append(&C[WF_C_REFR_COUNT + level], want_refr, wb.capacity, &C[WF_C_OVERFLOW], &k_refr) && want_refr) {
            uint4 *t = wb.refr + (size_t)k_refr * 3u;
            t[0] = make_uint4(fu(nh.pos.x), fu(nh.pos.y), fu(nh.pos.z), fu(inside_d.x));
            t[1] = make_uint4(fu(inside_d.y), fu(inside_d.z), pack_excl(nh.prim, FACE_FRONT), g);
            t[2] = make_uint4(nh.obj, fu(contribution * fc), 0u, 0u);
        }
        if (active) {
            wb.nodes[(size_t)g * 2u] = make_uint4(fu(acc.x), fu(acc.y), fu(acc.z), fu(rc));
            wb.nodes[(size_t)g * 2u + 1u] = make_uint4(fu(fc), 0u, rec_cr, rec_cf);
        }
    }
    add_casts(C, casts);
}

/* ---- wf_refr: get_refract from the inside cast on (main.rs:366-405) ------------------------------------- */

__global__ __launch_bounds__(64, WF_MIN_WAVES) void wf_refr_kernel(const KernelScene sc, const KernelFrame fr, const WfBuffers wb,
                                                                    const uint32_t level) {
    uint32_t *C = wb.counters;
    if (C[WF_C_OVERFLOW] != 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t count = C[WF_C_REFR_COUNT + level];
    if (count == 0u) return;
    const uint32_t next_base = level_base(C, level + 1u);
    const uint32_t next_limit = wb.capacity - next_base;
    uint32_t casts = 0u;

    for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {
        if ((unsigned long long)chunk * 64ull >= (unsigned long long)count) break;
        const uint32_t idx = chunk * 64u + lane;
        bool live = idx < count;

        Ray req;
        req.o = v3(0.0f, 0.0f, 0.0f);
        req.d = v3(0.0f, 0.0f, 1.0f);
        req.mode = FACE_BACK;
        req.excl = 0u;
        uint32_t parent = 0u, obj = 0u;
        float child_contribution = 0.0f;
        if (live) {
            const uint4 *t = wb.refr + (size_t)idx * 3u;
            const uint4 a = t[0], b = t[1], c = t[2];
            req.o = v3(uf(a.x), uf(a.y), uf(a.z));
            req.d = v3(uf(a.w), uf(b.x), uf(b.y));
            req.excl = b.z;
            parent = b.w;
            obj = c.x;
            child_contribution = uf(c.y);
        }
        const V3 node_pos = req.o; /* hit.at.position of the node */
        float travel = 0.0f;
        int32_t retry = -1; /* -1: the pending cast is the first inside cast (main.rs:371) */
        bool has_escape = false;
        V3 esc_o = v3(0.0f, 0.0f, 0.0f), esc_d = esc_o;
        uint32_t esc_excl = 0u;
        float decay = 0.0f;

        while (__builtin_amdgcn_ballot_w64(live) != 0ull) {
            if (live) {
                const CastResult cr = cast_asm(sc, req);
                casts += 1u;
                if (cr.prim < 0) {
                    live = false; /* Refraction::Infinite (main.rs:373, 383) */
                } else {
                    const HitGeom ih = finish_hit(sc, req, cr, false);
                    if (retry < 0) {
                        travel = distance(ih.pos, node_pos); /* main.rs:375 */
                        retry = 0;
                    } else {
                        travel += distance(req.o, ih.pos); /* main.rs:385; req.o is the previous inside hit */
                        retry += 1;
                    }
                    const rt_material &rm = sc.materials[obj];
                    V3 out_dir;
                    const bool have_out = refract_dir(ih.normal, req.d, 1.0f / rm.refraction_index, &out_dir);
                    if (!have_out && travel <= 100.0f && retry < 10) { /* main.rs:378 */
                        /* get_reflect(&hit_inside), main.rs:328-341; the bounce keeps the ray's face mode */
                        const V3 d = reflect_dir(ih.normal, req.d);
                        req.o = ih.pos;
                        req.d = d;
                        req.excl = pack_excl(ih.prim, ih.bf ? FACE_FRONT : FACE_BACK);
                    } else {
                        live = false;
                        if (have_out) { /* Escaped, main.rs:392-403; else Trapped */
                            has_escape = true;
                            esc_o = ih.pos;
                            esc_d = normalize(out_dir);
                            esc_excl = pack_excl(ih.prim, FACE_BACK);
                            decay = rtdm::powf(rm.opaque_decay, travel); /* main.rs:508 */
                        }
                    }
                }
            }
        }

        uint32_t k;
        if (wave_append(&C[WF_C_LEVEL + level + 1u], has_escape, next_limit, &C[WF_C_OVERFLOW], &k) && has_escape) {
            const uint32_t child = next_base + k;
            wb.node_in[(size_t)child * 2u] = make_uint4(fu(esc_o.x), fu(esc_o.y), fu(esc_o.z), fu(esc_d.x));
            wb.node_in[(size_t)child * 2u + 1u] = make_uint4(fu(esc_d.y), fu(esc_d.z), esc_excl | (FACE_FRONT << WF_MODE_SHIFT), fu(child_contribution));
            uint32_t *rec = reinterpret_cast<uint32_t *>(wb.nodes + (size_t)parent * 2u);
            rec[5] = fu(decay);
            rec[7] = child;
        }
    }
    add_casts(C, casts);
}

/* ---- wf_shade: get_shade (main.rs:407-464) for every node that needs it ----------------------------------- */

__global__ __launch_bounds__(64, WF_MIN_WAVES) void wf_shade_kernel(const KernelScene sc, const KernelFrame fr, const WfBuffers wb) {
    uint32_t *C = wb.counters;
    if (C[WF_C_OVERFLOW] != 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t count = C[WF_C_SHADE_COUNT];
    uint32_t casts = 0u;

    for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {
        if ((unsigned long long)chunk * 64ull >= (unsigned long long)count) break;
        const uint32_t idx = chunk * 64u + lane;
        const bool active = idx < count;

        uint32_t node = 0u, prim = 0u, objf = 0u;
        float u = 0.0f, v = 0.0f;
        V3 pos = v3(0.0f, 0.0f, 0.0f), normal = v3(0.0f, 0.0f, 1.0f), in_dir = v3(0.0f, 0.0f, 1.0f);
        if (active) {
            const uint4 *t = wb.shade + (size_t)idx * 4u;
            const uint4 a = t[0], b = t[1], c = t[2], d = t[3];
            node = a.x; prim = a.y; objf = a.z; u = uf(a.w);
            v = uf(b.x);
            pos = v3(uf(b.y), uf(b.z), uf(b.w));
            normal = v3(uf(c.x), uf(c.y), uf(c.z));
            in_dir = v3(uf(c.w), uf(d.x), uf(d.y));
        }
        const rt_material &rm = sc.materials[objf & 0x7fffffffu];
        const Mat m = material_approx(rm, u, v);
        const V3 adj_n = adjust_normal(m.normal, normal); /* main.rs:410 */
        V3 sum = v3(0.0f, 0.0f, 0.0f);

        for (uint32_t light_i = 0; light_i < sc.n_lights; ++light_i) { /* wave-uniform */
            const rt_light &L = sc.lights[light_i];
            DirLight dl;
            dl.direction = dl.color = v3(0.0f, 0.0f, 0.0f);
            bool need = false;
            if (active && approximate_into_directional(L, pos, &dl)) { /* main.rs:413-433 */
                const float cosine = -dot(dl.direction, adj_n);
                need = !(cosine <= 0.0f);
            }
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
            if (need) {
                Ray req;
                req.o = pos;
                req.d = -dl.direction;
                req.mode = FACE_BACK;
                req.excl = pack_excl(prim, FACE_BACK);
                const CastResult cr = cast_asm(sc, req);
                casts += 1u;
                /* main.rs:435-448 */
                bool lit = true;
                if (cr.prim >= 0) {
                    const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                    if (has_origin) {
                        const V3 occ = req.o + req.d * cr.t;
                        const float occlusion_distance = distance(pos, occ);
                        const float light_distance = distance(pos, v3(L.origin[0], L.origin[1], L.origin[2]));
                        if (occlusion_distance < light_distance) lit = false;
                    } else {
                        lit = false;
                    }
                }
                if (lit) { /* main.rs:450-461 */
                    const V3 light_direction = req.d; /* = -light.direction */
                    const V3 view_direction = -in_dir;
                    const V3 diffuse = get_diffuse(m, adj_n, light_direction) * dl.color;
                    const V3 specular = get_specular(m, adj_n, view_direction, light_direction) * dl.color;
                    sum = sum + diffuse * (1.0f - m.shiness) + specular * m.shiness;
                }
            }
        }
        if (active) {
            V3 acc = sum; /* depth <= 0: the unscaled shade (main.rs:488-490) */
            if ((objf & 0x80000000u) == 0u) {
                const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                acc = sum * shade_contribution;
            }
            float *rec = reinterpret_cast<float *>(wb.nodes + (size_t)node * 2u);
            rec[0] = acc.x;
            rec[1] = acc.y;
            rec[2] = acc.z;
        }
    }
    add_casts(C, casts);
}

/* ---- wf_combine: main.rs:516-518, children first ------------------------------------------------------------ */

template <bool ROOT>
__global__ __launch_bounds__(256) void wf_combine_kernel(const KernelFrame fr, const WfBuffers wb, const uint32_t level,
                                                         float *__restrict__ out) {
    const uint32_t *C = wb.counters;
    if (C[WF_C_OVERFLOW] != 0u) return;
    const uint32_t base = ROOT ? 0u : level_base(C, level);
    const uint32_t count = C[WF_C_LEVEL + level];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        const uint32_t g = base + idx;
        const uint4 a = wb.nodes[(size_t)g * 2u], b = wb.nodes[(size_t)g * 2u + 1u];
        V3 value = v3(uf(a.x), uf(a.y), uf(a.z));
        if (b.z != WF_FINAL) {
            const float rc = uf(a.w), fc = uf(b.x), decay = uf(b.y);
            V3 reflection = v3(0.0f, 0.0f, 0.0f), refraction = v3(0.0f, 0.0f, 0.0f);
            if (b.z != WF_NO_CHILD) {
                const uint4 c = wb.nodes[(size_t)b.z * 2u];
                reflection = v3(uf(c.x), uf(c.y), uf(c.z));
            }
            if (b.w != WF_NO_CHILD) {
                const uint4 c = wb.nodes[(size_t)b.w * 2u];
                refraction = v3(uf(c.x), uf(c.y), uf(c.z)) * decay; /* main.rs:508 */
            }
            value = (value + reflection * rc) + refraction * fc;
            if (!ROOT) {
                float *rec = reinterpret_cast<float *>(wb.nodes + (size_t)g * 2u);
                rec[0] = value.x;
                rec[1] = value.y;
                rec[2] = value.z;
            }
        }
        if (ROOT) {
            /* img[at] = img[at] + photon on a zeroed image (main.rs:1107) */
            uint32_t row, col;
            slot_to_pixel(fr, idx, &row, &col);
            float *px = out + ((size_t)row * fr.cols + col) * 3u;
            px[0] = 0.0f + value.x;
            px[1] = 0.0f + value.y;
            px[2] = 0.0f + value.z;
        }
    }
}

__global__ void wf_finish_kernel(const uint32_t *counters, unsigned long long *ray_count) {
    if (counters[WF_C_OVERFLOW] == 0u && ray_count != nullptr)
        *ray_count += *reinterpret_cast<const unsigned long long *>(counters + WF_C_CASTS);
}

/* ---- launcher ------------------------------------------------------------------------------------------ */

static hipEvent_t g_wf_ev[WF_STAGES][2];
static bool g_wf_ev_on = false;
void set_wavefront_events(hipEvent_t (*events)[2]) {
    g_wf_ev_on = events != nullptr;
    if (events)
        for (int i = 0; i < WF_STAGES; ++i) { g_wf_ev[i][0] = events[i][0]; g_wf_ev[i][1] = events[i][1]; }
}

hipError_t launch_wavefront(const KernelScene &sc, KernelFrame fr, float *out, unsigned long long *ray_count, const WfBuffers &wb,
                            uint32_t waves, hipStream_t stream) {
    const uint32_t total = fr.cols * fr.rows;
    fr.n_chunks = (total + 63u) / 64u;
    if (total == 0u) return hipSuccess;
    const uint32_t levels = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0) + 1u;
    auto grid = [&](uint32_t upper_items) {
        /* persistent waves pulling 64-item chunks; never more waves than chunks that could exist */
        const uint32_t chunks = (upper_items + 63u) / 64u;
        return dim3(chunks < waves ? (chunks ? chunks : 1u) : waves);
    };
    auto mark = [&](int stage, int which) {
        if (g_wf_ev_on) (void)hipEventRecord(g_wf_ev[stage][which], stream);
    };
    hipLaunchKernelGGL(wf_init_kernel, dim3(1), dim3(256), 0, stream, wb.counters, total);
    mark(WF_STAGE_TRACE, 0);
    for (uint32_t level = 0; level < levels; ++level) {
        if (level == 0u)
            hipLaunchKernelGGL((wf_node_kernel<true>), grid(total), dim3(64), 0, stream, sc, fr, wb, level);
        else
            hipLaunchKernelGGL((wf_node_kernel<false>), grid(wb.capacity), dim3(64), 0, stream, sc, fr, wb, level);
        if (level + 1u < levels) hipLaunchKernelGGL(wf_refr_kernel, grid(wb.capacity), dim3(64), 0, stream, sc, fr, wb, level);
    }
    mark(WF_STAGE_TRACE, 1);
    mark(WF_STAGE_SHADE, 0);
    hipLaunchKernelGGL(wf_shade_kernel, grid(wb.capacity), dim3(64), 0, stream, sc, fr, wb);
    mark(WF_STAGE_SHADE, 1);
    mark(WF_STAGE_COMBINE, 0);
    for (uint32_t level = levels - 1u; level-- > 1u;)
        hipLaunchKernelGGL((wf_combine_kernel<false>), dim3(2048), dim3(256), 0, stream, fr, wb, level, out);
    hipLaunchKernelGGL((wf_combine_kernel<true>), dim3(2048), dim3(256), 0, stream, fr, wb, 0u, out);
    hipLaunchKernelGGL(wf_finish_kernel, dim3(1), dim3(1), 0, stream, wb.counters, ray_count);
    mark(WF_STAGE_COMBINE, 1);
    return hipGetLastError();
}

} /* namespace rt */

#ifdef RT_DIAG_STAGES
RT_DIAG_STAGE_READER(rt_diag_read_stages_wavefront)
#endif
