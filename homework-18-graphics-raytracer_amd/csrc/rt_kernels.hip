/*
 * rt_kernels.hip — the render path as hand-written HIP for gfx950 (MI355X).
 *
 * Replaces the rayon closure at src/main.rs:1090-1104 (camera.shoot ->
 * world.ray_trace) with one kernel:
 *
 *   - one work-item per primary ray, a wave = an 8x8 pixel tile;
 *   - World::cast (main.rs:180-326) is a wave-convergent brute-force loop: the
 *     primitive index is wave-uniform, so each triangle record is fetched ONCE
 *     per wave (scalar loads into SGPRs, or an LDS broadcast read in the LDS
 *     variant) and tested by all 64 lanes; only the per-primitive accept
 *     predicate diverges;
 *   - the recursion of World::ray_trace (main.rs:466-519), get_shade's light
 *     loop (407-464) and get_refract's bounce loop (343-405) are unrolled into
 *     a per-lane state machine whose only expensive step is "cast one ray":
 *     every trip of the outer loop, every live lane casts whatever ray its own
 *     state needs next (primary, shadow, reflection, inside-glass bounce,
 *     escape) through the SAME convergent intersection loop, then advances its
 *     state with cheap divergent code.  A wave ballot ends the loop;
 *   - the post-order combine `shade*sc + reflection*rc + refraction*fc`
 *     (main.rs:516-518) keeps its association through an explicit per-lane
 *     frame stack (one frame per node that has children), so results are
 *     bit-identical to the recursive form.  Subtrees are pure, so the kernel
 *     is free to evaluate get_refract's casts before descending into the
 *     reflection child; that lets a frame hold the ready-made escape ray
 *     instead of the whole hit.
 *
 * Floating point: IEEE binary32 with no contraction (-ffp-contract=off),
 * correctly rounded divide/sqrt (hipcc default), denormals kept; the
 * transcendentals are rt_detmath.h.  See DESIGN.md "Numerics".
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_shade.h"
#include "rt_kernels.h"
#include "rt_cast.h"

namespace rt {

/* ---- the per-lane state machine ---------------------------------------------- */

enum : uint32_t {
    PH_DONE = 0u,
    PH_NODE = 1u,        /* the pending cast is ray_trace's own cast          (main.rs:473) */
    PH_SHADOW = 2u,      /* ... a shadow ray of get_shade's light loop        (main.rs:435) */
    PH_REFR_INSIDE = 3u, /* ... get_refract's first inside cast               (main.rs:371) */
    PH_REFR_BOUNCE = 4u  /* ... a total-internal-reflection bounce            (main.rs:381) */
};

/* One frame per ray_trace activation that has at least one child. */
struct Frame {
    V3 acc;            /* shade*sc, then (shade*sc + reflection*rc) */
    float rc, fc;      /* reflection_contribution, refraction_contribution */
    float decay;       /* opaque_decay.powf(travel_distance) */
    float child_contribution; /* contribution of the refraction child */
    V3 esc_o, esc_d;   /* escape ray (main.rs:393-401), valid when has_escape */
    uint32_t esc_excl;
    uint32_t flags;    /* bit0: reflection child running (else refraction child); bit1: has_escape */
};

/* The kernel renders one 64-slot chunk (an 8x8 pixel tile) per wave, in image order.  (Round 1 also tried, behind variant
 * bits, lanes that refill pixel by pixel from a queue, a two-phase scheme that parks the last few lanes of a tile and packs
 * them in a second pass, a cost probe with most-expensive-first dispatch, and cooperative workgroups whose four waves split
 * every tile's triangle loop; all bit-identical, all slower or equal — numbers in profiles/README.md — and removed in round 2
 * when the persistent wavefront kernel of rt_pwf.hip had long been the default and this kernel its fallback.) */
template <int MAXD, bool USE_LDS>
__device__ __forceinline__ void whitted_body(const KernelScene &sc, const KernelFrame &fr, float *__restrict__ out,
                                             unsigned long long *__restrict__ ray_count, const KernelQueues &qs, const DevTri *lds_tris,
                                             const uint32_t wave) {

    /* Work assignment.  The tile image (cols x rows) is enumerated as "slots": 8-row bands, column-major
     * inside a band, so 64 consecutive slots are an 8x8 pixel block (8 x fewer rows in a ragged last band).
     * Wave w owns slots [64 w, 64 w + 64). */
    const uint32_t lane = threadIdx.x & 63u;
#ifdef RT_DIAG_TIMELINE /* diagnostic build only: wave start/end on the 100 MHz constant clock, iterations, HW id */
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long diag_cast_cycles = 0ull;
#endif
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t band_slots = fr.cols << 3;
    uint32_t q_next = 0u, q_end = 0u;
    {
        const uint32_t chunk = wave;
        q_next = chunk * 64u < total_slots ? chunk * 64u : total_slots;
        q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
    }
    uint32_t iteration = 0u;
    uint32_t out_index = 0u; /* row * cols + col of the pixel this lane is working on */

    uint32_t phase = PH_DONE;
    Ray req;
    req.o = v3(0.0f, 0.0f, 0.0f);
    req.d = v3(0.0f, 0.0f, 1.0f);
    req.mode = FACE_FRONT;
    req.excl = 0u;
    uint32_t casts = 0u;

    /* node context */
    HitGeom nh;
    nh.pos = nh.normal = v3(0.0f, 0.0f, 0.0f);
    nh.u = nh.v = 0.0f;
    nh.prim = nh.bf = nh.obj = 0u;
    V3 n_in_dir = v3(0.0f, 0.0f, 0.0f); /* hit.ray.direction of the node hit */
    uint32_t n_in_mode = FACE_FRONT;    /* hit.ray.face_direction */
    float contribution = 1.0f;
    int32_t sp = 0;                     /* depth = max_depth - sp */
    /* shading context */
    V3 sum = v3(0.0f, 0.0f, 0.0f), adj_n = v3(0.0f, 0.0f, 0.0f), l_color = v3(0.0f, 0.0f, 0.0f);
    uint32_t light_i = 0u;
    /* refraction context */
    float travel = 0.0f;
    int32_t retry = 0;
    V3 node_acc = v3(0.0f, 0.0f, 0.0f); /* shade * shade_contribution of the current node */

    Frame stack[MAXD];

    const float THRESHOLD = 0.001f; /* main.rs:467 */

    for (;;) {
        /* ---- refill idle lanes ---- */
        unsigned long long need = __builtin_amdgcn_ballot_w64(phase == PH_DONE);
        while (need != 0ull && q_next != q_end) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t avail = q_end - q_next;
            if (phase == PH_DONE && rank < avail) {
                const uint32_t slot = q_next + rank;
                const uint32_t band = slot / band_slots;
                const uint32_t r = slot - band * band_slots;
                const uint32_t rows_left = fr.rows - (band << 3);
                const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
                const uint32_t col = r / band_rows;
                const uint32_t row = (band << 3) + (r - col * band_rows);
                out_index = row * fr.cols + col;
                /* main.rs:1093-1096 + Camera::shoot (main.rs:84-99) with the per-frame basis hoisted to the host */
                const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                const float clip_y = (fr.half_height - (float)y) / fr.height_f;
                const float clip_x = ((float)x - fr.half_width) / fr.height_f;
                const V3 cx = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
                const V3 cy = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
                const V3 ct = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
                req.o = v3(fr.cam_origin[0], fr.cam_origin[1], fr.cam_origin[2]);
                req.d = normalize(clip_x * cx + clip_y * cy + ct);
                req.mode = FACE_FRONT;
                req.excl = 0u;
                /* TraceState { depth: max_depth, contribution: 1.0 } (main.rs:1097-1100); the entry check
                 * of ray_trace (main.rs:469) always passes at the root */
                contribution = 1.0f;
                sp = 0;
                phase = PH_NODE;
            }
            const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
            q_next += n_need < avail ? n_need : avail;
            need = __builtin_amdgcn_ballot_w64(phase == PH_DONE);
        }
        const unsigned long long active = __builtin_amdgcn_ballot_w64(phase != PH_DONE);
        if (active == 0ull) break;
        iteration += 1u; /* read by the RT_DIAG_TIMELINE build only */
        (void)iteration;
        CastResult cr;
        cr.prim = -1;
        cr.t = 0.0f;
        cr.bf = 0u;
        cr.a0 = cr.a1 = cr.a2 = 0.0f;
#ifdef RT_DIAG_TIMELINE
        const unsigned long long diag_ca = __builtin_amdgcn_s_memtime();
#endif
        if (phase != PH_DONE) {
#ifdef RT_CAST_COMPILER /* the compiler-generated loop of the first builds, for A/B */
            cr = cast<USE_LDS>(sc, lds_tris, req);
#else
            cr = USE_LDS ? cast<USE_LDS>(sc, lds_tris, req) : cast_asm(sc, req);
#endif
            casts += 1u;
        }
#ifdef RT_DIAG_TIMELINE
        diag_cast_cycles += __builtin_amdgcn_s_memtime() - diag_ca;
#endif

        /* ---- advance this lane until it needs another cast or finishes ---- */
        if (phase != PH_DONE) {
            /* `value` carries a finished subtree result up the frame stack */
            V3 value = v3(0.0f, 0.0f, 0.0f);
            enum { GO_NONE, GO_NEXT_LIGHT, GO_AFTER_SHADE, GO_TRY_EXIT, GO_CHILDREN, GO_RETURN } go = GO_NONE;
            bool has_escape = false;
            V3 esc_o = v3(0.0f, 0.0f, 0.0f), esc_d = esc_o;
            uint32_t esc_excl = 0u;
            float decay = 0.0f;
            /* inside hit of get_refract, live only within this advance step */
            HitGeom ih = nh;
            V3 i_in_dir = req.d;
            uint32_t i_in_mode = req.mode;

            if (phase == PH_NODE) {
                if (cr.prim < 0) {
                    value = v3(0.0f, 0.0f, 0.0f); /* main.rs:475 */
                    go = GO_RETURN;
                } else {
                    nh = finish_hit(sc, req, cr, false);
                    n_in_dir = req.d;
                    n_in_mode = req.mode;
                    const rt_material &rm = sc.materials[nh.obj];
                    const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                    if (contribution * shade_contribution >= THRESHOLD) { /* main.rs:480-483 */
                        const Mat m = material_approx(rm, nh.u, nh.v);
                        adj_n = adjust_normal(m.normal, nh.normal); /* main.rs:410 */
                        sum = v3(0.0f, 0.0f, 0.0f);
                        light_i = 0u;
                        go = GO_NEXT_LIGHT;
                    } else {
                        sum = v3(0.0f, 0.0f, 0.0f);
                        go = GO_AFTER_SHADE;
                    }
                }
            } else if (phase == PH_SHADOW) {
                /* main.rs:435-448 */
                const rt_light &L = sc.lights[light_i];
                bool lit = true;
                if (cr.prim >= 0) {
                    const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                    if (has_origin) {
                        const V3 occ = req.o + req.d * cr.t;
                        const float occlusion_distance = distance(nh.pos, occ);
                        const float light_distance = distance(nh.pos, v3(L.origin[0], L.origin[1], L.origin[2]));
                        if (occlusion_distance < light_distance) lit = false;
                    } else {
                        lit = false;
                    }
                }
                if (lit) { /* main.rs:450-461 */
                    const rt_material &rm = sc.materials[nh.obj];
                    const Mat m = material_approx(rm, nh.u, nh.v);
                    const V3 light_direction = req.d; /* = -light.direction */
                    const V3 view_direction = -n_in_dir;
                    const V3 diffuse = get_diffuse(m, adj_n, light_direction) * l_color;
                    const V3 specular = get_specular(m, adj_n, view_direction, light_direction) * l_color;
                    sum = sum + diffuse * (1.0f - m.shiness) + specular * m.shiness;
                }
                light_i += 1u;
                go = GO_NEXT_LIGHT;
            } else { /* PH_REFR_INSIDE / PH_REFR_BOUNCE */
                if (cr.prim < 0) {
                    has_escape = false; /* Refraction::Infinite (main.rs:373, 383) */
                    go = GO_CHILDREN;
                } else {
                    ih = finish_hit(sc, req, cr, false);
                    i_in_dir = req.d;
                    i_in_mode = req.mode;
                    if (phase == PH_REFR_INSIDE) {
                        travel = distance(ih.pos, nh.pos); /* main.rs:375 */
                        retry = 0;
                    } else {
                        travel += distance(req.o, ih.pos); /* main.rs:385; req.o is the previous inside hit */
                        retry += 1;
                    }
                    go = GO_TRY_EXIT;
                }
            }

            /* small per-lane control loop; every path ends in a new cast request or PH_DONE */
            for (;;) {
                if (go == GO_NEXT_LIGHT) {
                    /* the `for light in &self.lights` loop of get_shade up to the shadow cast (main.rs:413-433) */
                    bool issued = false;
                    while (light_i < sc.n_lights) {
                        DirLight dl;
                        if (approximate_into_directional(sc.lights[light_i], nh.pos, &dl)) {
                            const float cosine = -dot(dl.direction, adj_n);
                            if (!(cosine <= 0.0f)) {
                                req.o = nh.pos;
                                req.d = -dl.direction;
                                req.mode = FACE_BACK;
                                req.excl = pack_excl(nh.prim, FACE_BACK);
                                l_color = dl.color;
                                phase = PH_SHADOW;
                                issued = true;
                                break;
                            }
                        }
                        light_i += 1u;
                    }
                    if (issued) break;
                    go = GO_AFTER_SHADE;
                } else if (go == GO_AFTER_SHADE) {
                    /* `sum` is get_shade's result, or black when the shade branch was skipped */
                    const int32_t depth = fr.max_depth - sp;
                    if (depth <= 0) { /* main.rs:488-490: unscaled shade */
                        value = sum;
                        go = GO_RETURN;
                        continue;
                    }
                    const rt_material &rm = sc.materials[nh.obj];
                    const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                    node_acc = sum * shade_contribution;
                    const float refraction_contribution = rm.transparency;
                    if (contribution * refraction_contribution > THRESHOLD) { /* main.rs:502-505, strict */
                        /* get_refract (main.rs:343-405) */
                        V3 refract_in;
                        if (refract_dir(nh.normal, n_in_dir, rm.refraction_index, &refract_in)) {
                            req.o = nh.pos;
                            req.d = normalize(refract_in); /* second normalize, main.rs:362 */
                            req.mode = FACE_BACK;
                            req.excl = pack_excl(nh.prim, FACE_FRONT);
                            phase = PH_REFR_INSIDE;
                            break;
                        }
                        /* Trapped */
                    }
                    has_escape = false;
                    go = GO_CHILDREN;
                } else if (go == GO_TRY_EXIT) {
                    const rt_material &rm = sc.materials[nh.obj];
                    const float k = rm.refraction_index;
                    V3 out_dir;
                    const bool have_out = refract_dir(ih.normal, i_in_dir, 1.0f / k, &out_dir);
                    if (!have_out && travel <= 100.0f && retry < 10) { /* main.rs:378 */
                        /* get_reflect(&hit_inside), main.rs:328-341 */
                        req.o = ih.pos;
                        req.d = reflect_dir(ih.normal, i_in_dir);
                        req.mode = i_in_mode;
                        req.excl = pack_excl(ih.prim, ih.bf ? FACE_FRONT : FACE_BACK); /* invert(hit.face_direction) */
                        phase = PH_REFR_BOUNCE;
                        break;
                    }
                    if (have_out) { /* Escaped, main.rs:392-403 */
                        has_escape = true;
                        esc_o = ih.pos;
                        esc_d = normalize(out_dir);
                        esc_excl = pack_excl(ih.prim, FACE_BACK);
                        decay = rtdm::powf(rm.opaque_decay, travel); /* main.rs:508 */
                    } else {
                        has_escape = false; /* Trapped */
                    }
                    go = GO_CHILDREN;
                } else if (go == GO_CHILDREN) {
                    const rt_material &rm = sc.materials[nh.obj];
                    const float rc = rm.shiness * (1.0f - rm.transparency); /* main.rs:493 */
                    const float fc = rm.transparency;                       /* main.rs:502 */
                    const bool want_refl = contribution * rc >= THRESHOLD;  /* main.rs:494-495 */
                    const V3 black = v3(0.0f, 0.0f, 0.0f);
                    if (!want_refl && !has_escape) {
                        value = node_acc + black * rc + black * fc; /* main.rs:516-518 with both children black */
                        go = GO_RETURN;
                        continue;
                    }
                    /* write only what this activation will read back: the escape ray (9 of the 15 dwords) exists
                     * only when the refraction child does — most frames are reflection-only */
                    Frame &f = stack[sp];
                    f.rc = rc;
                    f.fc = fc;
                    if (has_escape) {
                        f.decay = decay;
                        f.child_contribution = contribution * fc;
                        f.esc_o = esc_o;
                        f.esc_d = esc_d;
                        f.esc_excl = esc_excl;
                    }
                    if (want_refl) {
                        f.acc = node_acc;
                        f.flags = 1u | (has_escape ? 2u : 0u);
                        /* get_reflect(&hit), main.rs:328-341 */
                        req.o = nh.pos;
                        req.d = reflect_dir(nh.normal, n_in_dir);
                        req.mode = n_in_mode;
                        req.excl = pack_excl(nh.prim, nh.bf ? FACE_FRONT : FACE_BACK);
                        contribution = contribution * rc;
                    } else {
                        f.acc = node_acc + black * rc;
                        f.flags = 2u;
                        req.o = esc_o;
                        req.d = esc_d;
                        req.mode = FACE_FRONT;
                        req.excl = esc_excl;
                        contribution = contribution * fc;
                    }
                    sp += 1;
                    phase = PH_NODE;
                    break;
                } else { /* GO_RETURN: unwind finished activations */
                    if (sp == 0) {
                        /* img[at] = img[at] + photon on a zeroed image (main.rs:1107) */
                        float *px = out + (size_t)out_index * 3u;
                        px[0] = 0.0f + value.x;
                        px[1] = 0.0f + value.y;
                        px[2] = 0.0f + value.z;
                        phase = PH_DONE;
                        break;
                    }
                    Frame &f = stack[sp - 1];
                    if (f.flags & 1u) { /* the reflection child just returned */
                        f.acc = f.acc + value * f.rc;
                        if (f.flags & 2u) {
                            f.flags = 2u;
                            req.o = f.esc_o;
                            req.d = f.esc_d;
                            req.mode = FACE_FRONT;
                            req.excl = f.esc_excl;
                            contribution = f.child_contribution;
                            phase = PH_NODE;
                            break;
                        }
                        value = f.acc + v3(0.0f, 0.0f, 0.0f) * f.fc;
                    } else { /* the refraction child returned: shade * decay, then * fc */
                        value = f.acc + (value * f.decay) * f.fc;
                    }
                    sp -= 1;
                    /* restore the parent's contribution is unnecessary: a parent only needs it
                     * before its children start, and every start writes `contribution` afresh */
                }
            }
        }
    }

#ifdef RT_DIAG_TIMELINE
    if (qs.timeline != nullptr && lane == 0u) {
        const unsigned long long diag_t1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *rec = qs.timeline + (size_t)wave * 4u;
        rec[0] = diag_t0;
        rec[1] = diag_t1;
        rec[2] = iteration;
        rec[2] = (unsigned long long)iteration | ((__builtin_amdgcn_s_memtime() - diag_c0) << 16); /* iterations | wave cycles */
        rec[3] = diag_cast_cycles; /* shader cycles spent inside cast() */
    }
#endif
    if (ray_count != nullptr) {
        uint32_t c = casts;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0u && c != 0u) atomicAdd(ray_count, (unsigned long long)c);
    }
}

template <int MAXD, bool USE_LDS>
__global__ RT_LAUNCH_BOUNDS void whitted_kernel(const KernelScene sc, const KernelFrame fr, float *__restrict__ out,
                                                unsigned long long *__restrict__ ray_count, const KernelQueues qs) {
    if (qs.run_if != nullptr && *qs.run_if == 0u) return; /* fallback launch that is not needed */
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    const DevTri *lds_tris = nullptr;
    if (USE_LDS) {
        /* stage the triangle records once per workgroup: coalesced 16-byte loads, then broadcast reads */
        const uint32_t n16 = sc.n_triangles * (uint32_t)(sizeof(DevTri) / 16);
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.tris);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        lds_tris = reinterpret_cast<const DevTri *>(lds_raw);
    }
    /* one tile per wave when the grid covers the frame (the per-pixel render paths); the wavefront path's fallback launch is
     * a small grid — it usually has nothing to do, and 32 K workgroups that leave at once cost 7 us — whose waves take
     * tiles a grid apart */
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < fr.n_chunks; wave += n_waves)
        whitted_body<MAXD, USE_LDS>(sc, fr, out, ray_count, qs, lds_tris, wave);
}

} /* namespace rt */

/* ---- launchers ------------------------------------------------------------------ */

namespace rt {

template <int MAXD>
static hipError_t launch_tiles(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                               const KernelQueues &qs, uint32_t waves, hipStream_t stream, bool use_lds) {
    if (waves == 0u) return hipSuccess;
    const uint32_t waves_per_block = RT_BLOCK_THREADS / 64;
    const uint32_t blocks = (waves + waves_per_block - 1) / waves_per_block;
    if (use_lds) {
        const size_t lds = (size_t)sc.n_triangles * sizeof(DevTri);
        hipLaunchKernelGGL((whitted_kernel<MAXD, true>), dim3(blocks), dim3(RT_BLOCK_THREADS), lds, stream, sc, fr, out, ray_count, qs);
    } else {
        hipLaunchKernelGGL((whitted_kernel<MAXD, false>), dim3(blocks), dim3(RT_BLOCK_THREADS), 0, stream, sc, fr, out, ray_count, qs);
    }
    return hipGetLastError();
}

/* optional HIP events recorded on the launch stream right around the dominant (render) kernel of a call */
/* thread-local: set by a render call on its own host thread and read by the launchers it calls on that thread */
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local bool g_ev_muted = false; /* set around launches that are not "the" render kernel (the wavefront path's fallback) */
void set_main_kernel_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }
void mute_main_kernel_events(bool muted) { g_ev_muted = muted; }
void record_main_kernel_event(int which, hipStream_t stream) {
    hipEvent_t ev = which == 0 ? g_ev_start : g_ev_stop;
    if (ev && !g_ev_muted) (void)hipEventRecord(ev, stream);
}

template <int MAXD>
static hipError_t launch_maxd(const KernelScene &sc, KernelFrame fr, float *out, unsigned long long *ray_count, const KernelQueues &qs,
                              hipStream_t stream, int variant) {
    const bool use_lds = (variant & RT_VARIANT_LDS) != 0 && (size_t)sc.n_triangles * sizeof(DevTri) <= RT_LDS_SCENE_LIMIT;
    const uint32_t total = fr.cols * fr.rows;
    fr.n_chunks = (total + 63u) / 64u;
    record_main_kernel_event(0, stream);
    uint32_t waves = fr.n_chunks;
    if (qs.run_if != nullptr && waves > RT_FALLBACK_WAVES) waves = RT_FALLBACK_WAVES;
    const hipError_t e = launch_tiles<MAXD>(sc, fr, out, ray_count, qs, waves, stream, use_lds);
    record_main_kernel_event(1, stream);
    return e;
}

hipError_t launch_whitted(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                          const KernelQueues &qs, hipStream_t stream, int variant) {
    if (fr.max_depth <= 8) return launch_maxd<8>(sc, fr, out, ray_count, qs, stream, variant);
    return launch_maxd<RT_MAX_DEPTH>(sc, fr, out, ray_count, qs, stream, variant);
}

} /* namespace rt */

/* ---- diagnostics: rt_detmath on the device ----------------------------------------- */

namespace rt {

/* host + device so rt_math_eval_host runs the very same source */
__host__ __device__ float math_eval_one(int op, float x, float y) {
    switch (op) {
        case RT_MATH_SIN: return rtdm::sinf(x);
        case RT_MATH_COS: return rtdm::cosf(x);
        case RT_MATH_TAN: return rtdm::tanf(x);
        case RT_MATH_ACOS: return rtdm::acosf(x);
        case RT_MATH_ATAN2: return rtdm::atan2f(x, y);
        case RT_MATH_POW: return rtdm::powf(x, y);
        case RT_MATH_F32_DIV: return x / y;
        case RT_MATH_F32_SQRT: return rtdm::f_sqrt(x);
        case RT_MATH_F64_SQRT_HI: return rtdm::f32_from_bits((uint32_t)(rtdm::f64_bits(rtdm::d_sqrt((double)x * (double)y)) >> 32));
        case RT_MATH_F64_SQRT_LO: return rtdm::f32_from_bits((uint32_t)(rtdm::f64_bits(rtdm::d_sqrt((double)x * (double)y))));
        case RT_MATH_F64_DIV_HI: return rtdm::f32_from_bits((uint32_t)(rtdm::f64_bits((double)x / (double)y) >> 32));
        case RT_MATH_F64_DIV_LO: return rtdm::f32_from_bits((uint32_t)(rtdm::f64_bits((double)x / (double)y)));
        case RT_MATH_ROUND: return rtdm::f_round(x);
        case RT_MATH_SINCOS_SIN:
        case RT_MATH_SINCOS_COS: {
            float s, c;
            rtdm::sincosf(x, &s, &c);
            return op == RT_MATH_SINCOS_SIN ? s : c;
        }
        default: return 0.0f;
    }
}

__global__ void math_eval_kernel(int op, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = math_eval_one(op, x[i], y[i]);
}

hipError_t launch_math_eval(int op, const float *d_x, const float *d_y, float *d_out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(math_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, op, d_x, d_y, d_out, n);
    return hipGetLastError();
}

void math_eval_host(int op, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = math_eval_one(op, x[i], y ? y[i] : 0.0f);
}

} /* namespace rt */

#ifdef RT_DIAG_STAGES
RT_DIAG_STAGE_READER(rt_diag_read_stages_kernels)
#endif
