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

/* Serialise / restore a lane's state.  Macros because they touch two dozen of the kernel's locals. */
#define CONT_F(i, x) rec[i] = __float_as_uint(x)
#define STORE_CONT(rec)                                                                                     \
    do {                                                                                                    \
        rec[0] = phase; rec[1] = (uint32_t)sp; rec[2] = out_index; rec[3] = req.mode; rec[4] = req.excl;    \
        rec[5] = nh.prim; rec[6] = nh.bf; rec[7] = nh.obj; rec[8] = n_in_mode; rec[9] = light_i;            \
        rec[10] = (uint32_t)retry;                                                                          \
        CONT_F(11, req.o.x); CONT_F(12, req.o.y); CONT_F(13, req.o.z);                                      \
        CONT_F(14, req.d.x); CONT_F(15, req.d.y); CONT_F(16, req.d.z);                                      \
        CONT_F(17, nh.pos.x); CONT_F(18, nh.pos.y); CONT_F(19, nh.pos.z);                                   \
        CONT_F(20, nh.normal.x); CONT_F(21, nh.normal.y); CONT_F(22, nh.normal.z);                          \
        CONT_F(23, nh.u); CONT_F(24, nh.v);                                                                 \
        CONT_F(25, n_in_dir.x); CONT_F(26, n_in_dir.y); CONT_F(27, n_in_dir.z);                             \
        CONT_F(28, contribution);                                                                           \
        CONT_F(29, sum.x); CONT_F(30, sum.y); CONT_F(31, sum.z);                                            \
        CONT_F(32, adj_n.x); CONT_F(33, adj_n.y); CONT_F(34, adj_n.z);                                      \
        CONT_F(35, l_color.x); CONT_F(36, l_color.y); CONT_F(37, l_color.z);                                \
        CONT_F(38, travel);                                                                                 \
        CONT_F(39, node_acc.x); CONT_F(40, node_acc.y); CONT_F(41, node_acc.z);                             \
        for (int32_t fi = 0; fi < sp; ++fi) {                                                               \
            const Frame &f = stack[fi];                                                                     \
            uint32_t *fr_ = rec + CONT_FIXED + FRAME_DWORDS * (uint32_t)fi;                                 \
            fr_[0] = __float_as_uint(f.acc.x); fr_[1] = __float_as_uint(f.acc.y); fr_[2] = __float_as_uint(f.acc.z); \
            fr_[3] = __float_as_uint(f.rc); fr_[4] = __float_as_uint(f.fc); fr_[5] = __float_as_uint(f.decay); \
            fr_[6] = __float_as_uint(f.child_contribution);                                                 \
            fr_[7] = __float_as_uint(f.esc_o.x); fr_[8] = __float_as_uint(f.esc_o.y); fr_[9] = __float_as_uint(f.esc_o.z); \
            fr_[10] = __float_as_uint(f.esc_d.x); fr_[11] = __float_as_uint(f.esc_d.y); fr_[12] = __float_as_uint(f.esc_d.z); \
            fr_[13] = f.esc_excl; fr_[14] = f.flags;                                                        \
        }                                                                                                   \
    } while (0)
#define CONT_G(i) __uint_as_float(rec[i])
#define LOAD_CONT(rec)                                                                                      \
    do {                                                                                                    \
        phase = rec[0]; sp = (int32_t)rec[1]; out_index = rec[2]; req.mode = rec[3]; req.excl = rec[4];     \
        nh.prim = rec[5]; nh.bf = rec[6]; nh.obj = rec[7]; n_in_mode = rec[8]; light_i = rec[9];            \
        retry = (int32_t)rec[10];                                                                           \
        req.o = v3(CONT_G(11), CONT_G(12), CONT_G(13)); req.d = v3(CONT_G(14), CONT_G(15), CONT_G(16));     \
        nh.pos = v3(CONT_G(17), CONT_G(18), CONT_G(19)); nh.normal = v3(CONT_G(20), CONT_G(21), CONT_G(22)); \
        nh.u = CONT_G(23); nh.v = CONT_G(24);                                                               \
        n_in_dir = v3(CONT_G(25), CONT_G(26), CONT_G(27)); contribution = CONT_G(28);                       \
        sum = v3(CONT_G(29), CONT_G(30), CONT_G(31)); adj_n = v3(CONT_G(32), CONT_G(33), CONT_G(34));       \
        l_color = v3(CONT_G(35), CONT_G(36), CONT_G(37)); travel = CONT_G(38);                              \
        node_acc = v3(CONT_G(39), CONT_G(40), CONT_G(41));                                                  \
        for (int32_t fi = 0; fi < sp; ++fi) {                                                               \
            const uint32_t *fr_ = rec + CONT_FIXED + FRAME_DWORDS * (uint32_t)fi;                           \
            Frame f;                                                                                        \
            f.acc = v3(__uint_as_float(fr_[0]), __uint_as_float(fr_[1]), __uint_as_float(fr_[2]));          \
            f.rc = __uint_as_float(fr_[3]); f.fc = __uint_as_float(fr_[4]); f.decay = __uint_as_float(fr_[5]); \
            f.child_contribution = __uint_as_float(fr_[6]);                                                 \
            f.esc_o = v3(__uint_as_float(fr_[7]), __uint_as_float(fr_[8]), __uint_as_float(fr_[9]));        \
            f.esc_d = v3(__uint_as_float(fr_[10]), __uint_as_float(fr_[11]), __uint_as_float(fr_[12]));     \
            f.esc_excl = fr_[13]; f.flags = fr_[14];                                                        \
            stack[fi] = f;                                                                                  \
        }                                                                                                   \
    } while (0)

/* cooperative cast (MODE_COOP): LDS layout in dwords */
#define COOP_WAVES 4u
#define COOP_RAYS_OFFSET 0u                                  /* [tile][8][64] */
#define COOP_RES_OFFSET (COOP_WAVES * 8u * 64u)              /* [tile][chunk][2][64] */
#define COOP_MASK_OFFSET (COOP_RES_OFFSET + COOP_WAVES * COOP_WAVES * 2u * 64u) /* [tile][2] */
#define COOP_LDS_BYTES ((COOP_MASK_OFFSET + COOP_WAVES * 2u) * 4u)

#ifndef RT_PROBE_ITERS
#define RT_PROBE_ITERS 6u /* casts the cost probe follows a pixel for (profiles/README.md) */
#endif

#ifndef RT_PRIO_STEP1
#define RT_PRIO_STEP1 9u
#define RT_PRIO_STEP2 14u
#define RT_PRIO_STEP3 20u
#endif

/* Work-source modes of the kernel (see "Work assignment" below). */
enum : int {
    MODE_STATIC = 0,     /* one 64-slot chunk per wave, run to completion                                   */
    MODE_PERSISTENT = 1, /* lanes refill pixel by pixel from a global chunk counter                          */
    MODE_PHASE1 = 2,     /* one chunk per wave; when few lanes are left their state is evicted to a queue     */
    MODE_PHASE2 = 3,     /* lanes load evicted states (continuations) from that queue, refilling as they end  */
    MODE_COOP = 5,       /* 256-thread workgroups of four tiles; the waves split every tile's triangle loop four ways (below) */
    MODE_COST = 4        /* probe: lane k traces the middle pixel of chunk k for a few casts and records how far it got */
};

/* A continuation = everything a lane carries between two casts: CONT_FIXED dwords + its frame stack. */
#define CONT_FIXED 48u
#define FRAME_DWORDS 15u

template <int MAXD, bool USE_LDS, int MODE>
__device__ __forceinline__ void whitted_body(const KernelScene &sc, const KernelFrame &fr, float *__restrict__ out,
                                             unsigned long long *__restrict__ ray_count, const KernelQueues &qs) {
    uint32_t *__restrict__ work_queue = qs.work_queue;
    if (MODE == MODE_STATIC && qs.run_if != nullptr && *qs.run_if == 0u) return; /* fallback launch that is not needed */
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

    /* Work assignment.  The tile image (cols x rows) is enumerated as "slots": 8-row bands, column-major
     * inside a band, so 64 consecutive slots are an 8x8 pixel block (8 x fewer rows in a ragged last band).
     * A wave owns a local run of slots [q_next, q_end).  Lanes whose pixel is finished take the next slots
     * of the run (ballot + prefix count, no atomics); when the run is empty the wave pulls the next
     * 64-slot chunk from a global counter (PERSISTENT) — one atomic per 64 pixels — so every lane stays
     * busy until the frame runs dry and no wave outlives the others by more than one pixel's work.
     * Without PERSISTENT each wave gets exactly one chunk (the round-1 v1 scheme, kept for A/B). */
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
#ifdef RT_DIAG_TIMELINE /* diagnostic build only: wave start/end on the 100 MHz constant clock, iterations, HW id */
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long diag_cast_cycles = 0ull;
#endif
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t band_slots = fr.cols << 3;
    uint32_t q_next = 0u, q_end = 0u;
    bool exhausted = false;
    /* MODE_COOP: a workgroup of COOP_WAVES waves owns COOP_WAVES tiles taken from far-apart parts of the image
     * (tile = group + k * n_groups), so that cheap and expensive tiles share a group; see "cooperative cast". */
    const uint32_t wslot = (threadIdx.x >> 6);
    uint32_t *coop_lds = reinterpret_cast<uint32_t *>(lds_raw);
    if (MODE == MODE_COOP) {
        const uint32_t chunk = blockIdx.x + wslot * gridDim.x;
        q_next = chunk < fr.n_chunks ? chunk * 64u : total_slots;
        q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
        exhausted = true;
    }
    if (MODE == MODE_STATIC || MODE == MODE_PHASE1) {
        /* which chunk this wave renders: dispatch position `wave`, or — when a cost-sorted order exists —
         * the wave-th most expensive chunk, so that the long tiles start first and the cheap ones fill the end */
        uint32_t chunk = wave;
        if (MODE == MODE_STATIC && qs.tile_order != nullptr && wave < fr.n_chunks) {
            /* the probe left one list of chunk ids per cost class; dispatch position `wave` walks them from the
             * most expensive class down (a sort without a sort kernel) */
            uint32_t pos = wave;
            for (int32_t cls = (int32_t)RT_PROBE_ITERS + 1; cls >= 0; --cls) {
                const uint32_t n_cls = qs.class_count[cls];
                if (pos < n_cls) { chunk = qs.tile_order[(uint32_t)cls * fr.n_chunks + pos]; break; }
                pos -= n_cls;
            }
        }
        q_next = chunk * 64u < total_slots ? chunk * 64u : total_slots;
        q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
        exhausted = true;
    }
    if (MODE == MODE_COST) {
        /* one probe pixel per lane: the middle slot of chunk (wave*64 + lane), handed out by the refill code
         * below on its first trip (a pretend run of 64 slots that is consumed at once) */
        q_next = 0u;
        q_end = 64u;
        exhausted = true;
    }
    /* MODE_PHASE2: slots are continuation records; their number was left in qs.cont_count by phase 1 */
    const uint32_t n_cont = MODE == MODE_PHASE2 ? *qs.cont_count : 0u;
    const uint32_t cont_stride = CONT_FIXED + FRAME_DWORDS * (uint32_t)MAXD;
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

    bool cost_started = false;
    for (;;) {
        /* ---- refill idle lanes ---- */
        unsigned long long need = __builtin_amdgcn_ballot_w64(phase == PH_DONE);
        if (MODE == MODE_COST) need = cost_started ? 0ull : need;
        while (need != 0ull) {
            if (q_next == q_end) { /* wave-uniform */
                if (exhausted) break;
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(work_queue, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                const uint32_t limit = MODE == MODE_PHASE2 ? n_cont : total_slots;
                if (c * 64u >= limit) { exhausted = true; break; }
                q_next = c * 64u;
                q_end = q_next + 64u < limit ? q_next + 64u : limit;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            uint32_t avail = q_end - q_next;
            uint32_t probe_slot = 0u;
            if (MODE == MODE_COST) {
                const uint32_t chunk = wave * 64u + lane;
                const uint32_t first = chunk * 64u;
                const uint32_t last = first + 64u < total_slots ? first + 64u : total_slots;
                probe_slot = first + ((last > first ? last - first : 0u) >> 1) + 4u < last ? first + ((last - first) >> 1) + 4u : first;
                avail = chunk < fr.n_chunks ? 64u : rank; /* lanes beyond the last chunk get nothing */
                cost_started = true;
            }
            if (phase == PH_DONE && rank < avail) {
                const uint32_t slot = MODE == MODE_COST ? probe_slot : q_next + rank;
                if (MODE == MODE_PHASE2) {
                    const uint32_t *rec = qs.cont_buf + (size_t)slot * cont_stride;
                    LOAD_CONT(rec);
                } else {
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
            }
            if (MODE == MODE_COST) {
                q_next = q_end;
                need = 0ull;
            } else {
                const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
                q_next += n_need < avail ? n_need : avail;
                need = __builtin_amdgcn_ballot_w64(phase == PH_DONE);
            }
        }
        const unsigned long long active = __builtin_amdgcn_ballot_w64(phase != PH_DONE);
        if (MODE == MODE_COOP) {
            /* publish this tile's 64 ray requests and its live-lane mask, then decide together whether anyone is left */
            uint32_t *my = coop_lds + COOP_RAYS_OFFSET + (wslot * 8u) * 64u + lane;
            my[0 * 64] = __float_as_uint(req.o.x); my[1 * 64] = __float_as_uint(req.o.y); my[2 * 64] = __float_as_uint(req.o.z);
            my[3 * 64] = __float_as_uint(req.d.x); my[4 * 64] = __float_as_uint(req.d.y); my[5 * 64] = __float_as_uint(req.d.z);
            my[6 * 64] = req.mode; my[7 * 64] = req.excl;
            if (lane == 0u) {
                coop_lds[COOP_MASK_OFFSET + wslot * 2u] = (uint32_t)active;
                coop_lds[COOP_MASK_OFFSET + wslot * 2u + 1u] = (uint32_t)(active >> 32);
            }
            __syncthreads();
            uint32_t any = 0u;
            for (uint32_t j = 0; j < COOP_WAVES * 2u; ++j) any |= coop_lds[COOP_MASK_OFFSET + j];
            if (__builtin_amdgcn_readfirstlane((int)any) == 0) break;
        } else if (active == 0ull) {
            break;
        }
        if (MODE == MODE_COST && iteration >= RT_PROBE_ITERS) {
            /* the probe only has to tell long tiles from short ones: pixels still going after RT_PROBE_ITERS casts
             * are graded "long" and the probe stops (its own latency is on the frame's critical path) */
            if (phase != PH_DONE) {
                const uint32_t cls = RT_PROBE_ITERS + 1u;
                qs.tile_order[cls * fr.n_chunks + atomicAdd(&qs.class_count[cls], 1u)] = wave * 64u + lane;
            }
            break;
        }
        if (MODE == MODE_PHASE1 && iteration >= qs.evict_min_iterations &&
            (uint32_t)__builtin_popcountll(active) <= qs.evict_threshold) {
            /* Few lanes left: a wave that keeps going runs the full intersection loop for a handful of
             * rays.  Park their state in the continuation queue (one wave-aggregated atomic) and leave;
             * phase 2 packs 64 of these per wave. */
            const uint32_t n_act = (uint32_t)__builtin_popcountll(active);
            uint32_t base = 0u;
            if (lane == 0u) base = atomicAdd(qs.cont_count, n_act);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (phase != PH_DONE) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u));
                uint32_t *rec = qs.cont_buf + (size_t)(base + rank) * cont_stride;
                STORE_CONT(rec);
            }
            break;
        }
        iteration += 1u;
#ifdef RT_WAVE_PRIORITY /* measured slower (profiles/README.md): any extra code here tips the compiler's allocation of the loop */
        /* The frame's critical path is its deepest pixel: ~50 dependent casts in one wave, each of which
         * takes 4x longer while three other waves share the SIMD.  Waves that turn out to be long raise their
         * issue priority step by step, so they run at close to solo speed and the short waves fill the gaps. */
        {
            /* s_setprio is a scalar instruction: it must sit under a branch the compiler knows to be
             * wave-uniform (readfirstlane), or it runs every trip regardless of the condition */
            const uint32_t it_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)iteration);
            if (it_u == RT_PRIO_STEP1) __builtin_amdgcn_s_setprio(1);
            if (it_u == RT_PRIO_STEP2) __builtin_amdgcn_s_setprio(2);
            if (it_u == RT_PRIO_STEP3) __builtin_amdgcn_s_setprio(3);
        }
#endif

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
            if (MODE != MODE_COOP) cr = USE_LDS ? cast<USE_LDS>(sc, lds_tris, req) : cast_asm(sc, req);
#endif
            casts += 1u;
        }
        if (MODE == MODE_COOP) {
            /* ---- cooperative cast ----
             * The frame's critical path is its deepest tile: ~50 dependent casts, each a 64-triangle loop that one
             * wave cannot run faster than ~12 us.  Here the four waves of a group split EVERY tile's loop: wave w
             * tests triangles [w*nt/4, (w+1)*nt/4) for the rays of each tile that is still alive, so a tile whose
             * three neighbours have finished gets its casts done four times sooner, and a group with four live
             * tiles does the same total work as before.  Rays and partial results go through LDS (16 KB).
             * Merging the four partial (t, prim) in triangle order with the reference's own rule
             * (`nearest_t < t -> skip`, ties to the later primitive) equals the sequential scan — except when a
             * candidate distance is NaN (degenerate geometry), where the scan's history matters: then the owning
             * wave simply redoes its cast sequentially. */
            const uint32_t nt = sc.n_triangles;
            const uint32_t per = (nt + COOP_WAVES - 1u) / COOP_WAVES;
            const uint32_t t_base = wslot * per < nt ? wslot * per : nt;
            const uint32_t t_cnt = t_base + per < nt ? per : nt - t_base;
            for (uint32_t j = 0; j < COOP_WAVES; ++j) {
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)coop_lds[COOP_MASK_OFFSET + j * 2u]);
                const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)coop_lds[COOP_MASK_OFFSET + j * 2u + 1u]);
                const unsigned long long mj = ((unsigned long long)mhi << 32) | mlo;
                if (mj == 0ull) continue;
                TriBest part;
                part.t = rtdm::quiet_nan();
                part.prim = -1;
                part.nd = part.a0 = part.a1 = part.a2 = 0.0f;
                if ((mj >> lane) & 1ull) {
                    const uint32_t *rj = coop_lds + COOP_RAYS_OFFSET + (j * 8u) * 64u + lane;
                    Ray r;
                    r.o = v3(__uint_as_float(rj[0 * 64]), __uint_as_float(rj[1 * 64]), __uint_as_float(rj[2 * 64]));
                    r.d = v3(__uint_as_float(rj[3 * 64]), __uint_as_float(rj[4 * 64]), __uint_as_float(rj[5 * 64]));
                    r.mode = rj[6 * 64];
                    r.excl = rj[7 * 64];
                    cast_asm_triangles(sc.tris + t_base, t_cnt, t_base, r, cast_masks(r, sc.filter_origin2), &part);
                }
                const float bt = part.t;
                const int32_t bp = part.prim;
                uint32_t *res = coop_lds + COOP_RES_OFFSET + ((j * COOP_WAVES + wslot) * 2u) * 64u + lane;
                res[0] = __float_as_uint(bt);
                res[64] = (uint32_t)bp;
            }
            __syncthreads();
            if (phase != PH_DONE) {
                float best_t = rtdm::quiet_nan();
                int32_t best_prim = -1;
                bool saw_nan = false;
                for (uint32_t c = 0; c < COOP_WAVES; ++c) {
                    const uint32_t *res = coop_lds + COOP_RES_OFFSET + ((wslot * COOP_WAVES + c) * 2u) * 64u + lane;
                    const float t = __uint_as_float(res[0]);
                    const int32_t pr = (int32_t)res[64];
                    if (pr >= 0) {
                        saw_nan = saw_nan || (t != t);
                        if (!(best_prim >= 0 && best_t < t)) { best_t = t; best_prim = pr; }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(saw_nan) != 0ull) {
                    TriBest whole;
                    whole.t = rtdm::quiet_nan();
                    whole.prim = -1;
                    whole.nd = whole.a0 = whole.a1 = whole.a2 = 0.0f;
                    cast_asm_triangles(sc.tris, nt, 0u, req, cast_masks(req, sc.filter_origin2), &whole);
                    best_t = whole.t;
                    best_prim = whole.prim;
                }
                cr = cast_finish(sc, req, best_t, best_prim);
            }
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
                        if (MODE == MODE_COST) {
                            /* probe finished early: its class is its exact cast count; the radiance is discarded */
                            const uint32_t cls = casts < RT_PROBE_ITERS ? casts : RT_PROBE_ITERS;
                            qs.tile_order[cls * fr.n_chunks + atomicAdd(&qs.class_count[cls], 1u)] = wave * 64u + lane;
                        } else {
                            /* img[at] = img[at] + photon on a zeroed image (main.rs:1107) */
                            float *px = out + (size_t)out_index * 3u;
                            px[0] = 0.0f + value.x;
                            px[1] = 0.0f + value.y;
                            px[2] = 0.0f + value.z;
                        }
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
    if (qs.timeline != nullptr && MODE == MODE_STATIC && lane == 0u) {
        const unsigned long long diag_t1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *rec = qs.timeline + (size_t)wave * 4u;
        rec[0] = diag_t0;
        rec[1] = diag_t1;
        rec[2] = iteration;
        rec[2] = (unsigned long long)iteration | ((__builtin_amdgcn_s_memtime() - diag_c0) << 16); /* iterations | wave cycles */
        rec[3] = diag_cast_cycles; /* shader cycles spent inside cast() */
    }
#endif
    if (ray_count != nullptr && MODE != MODE_COST) {
        uint32_t c = casts;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0u && c != 0u) atomicAdd(ray_count, (unsigned long long)c);
    }
}

template <int MAXD, bool USE_LDS, int MODE>
__global__ RT_LAUNCH_BOUNDS void whitted_kernel(const KernelScene sc, const KernelFrame fr, float *__restrict__ out,
                                                unsigned long long *__restrict__ ray_count, const KernelQueues qs) {
    whitted_body<MAXD, USE_LDS, MODE>(sc, fr, out, ray_count, qs);
}

/* the cooperative variant: 256-thread workgroups (four waves, four tiles), same body */
template <int MAXD>
__global__ __launch_bounds__(COOP_WAVES * 64, RT_MIN_WAVES) void whitted_coop_kernel(const KernelScene sc, const KernelFrame fr,
                                                                                      float *__restrict__ out,
                                                                                      unsigned long long *__restrict__ ray_count,
                                                                                      const KernelQueues qs) {
    whitted_body<MAXD, false, MODE_COOP>(sc, fr, out, ray_count, qs);
}

} /* namespace rt */

/* ---- launchers ------------------------------------------------------------------ */

namespace rt {

template <int MAXD, int MODE>
static hipError_t launch_mode(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                              const KernelQueues &qs, uint32_t waves, hipStream_t stream, bool use_lds) {
    if (waves == 0u) return hipSuccess;
    const uint32_t waves_per_block = RT_BLOCK_THREADS / 64;
    const uint32_t blocks = (waves + waves_per_block - 1) / waves_per_block;
    if (use_lds) {
        const size_t lds = (size_t)sc.n_triangles * sizeof(DevTri);
        hipLaunchKernelGGL((whitted_kernel<MAXD, true, MODE>), dim3(blocks), dim3(RT_BLOCK_THREADS), lds, stream, sc, fr, out, ray_count, qs);
    } else {
        hipLaunchKernelGGL((whitted_kernel<MAXD, false, MODE>), dim3(blocks), dim3(RT_BLOCK_THREADS), 0, stream, sc, fr, out, ray_count, qs);
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

template <int MAXD, int MODE>
static hipError_t launch_main(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                              const KernelQueues &qs, uint32_t waves, hipStream_t stream, bool use_lds) {
    record_main_kernel_event(0, stream);
    const hipError_t e = launch_mode<MAXD, MODE>(sc, fr, out, ray_count, qs, waves, stream, use_lds);
    record_main_kernel_event(1, stream);
    return e;
}

template <int MAXD>
static hipError_t launch_maxd(const KernelScene &sc, KernelFrame fr, float *out, unsigned long long *ray_count, const KernelQueues &qs,
                              uint32_t resident_waves, hipStream_t stream, int variant) {
    const bool use_lds = (variant & RT_VARIANT_LDS) != 0 && (size_t)sc.n_triangles * sizeof(DevTri) <= RT_LDS_SCENE_LIMIT;
    const uint32_t total = fr.cols * fr.rows;
    fr.n_chunks = (total + 63u) / 64u;
    const int scheme = variant & RT_VARIANT_SCHEME_MASK;
    if (scheme == RT_VARIANT_STATIC) {
        KernelQueues q2 = qs;
        q2.tile_order = nullptr;
        return launch_main<MAXD, MODE_STATIC>(sc, fr, out, ray_count, q2, fr.n_chunks, stream, use_lds);
    }
    if (scheme == RT_VARIANT_SORTED) {
        /* 1. probe: follow the middle pixel of every chunk for a few casts and file the chunk under its cost class,
         * 2. render, walking the class lists from the most expensive class down */
        if (fr.n_chunks <= resident_waves || qs.tile_order == nullptr) { /* everything is resident at once: order is moot */
            KernelQueues q2 = qs;
            q2.tile_order = nullptr;
            return launch_main<MAXD, MODE_STATIC>(sc, fr, out, ray_count, q2, fr.n_chunks, stream, use_lds);
        }
        hipError_t e = launch_mode<MAXD, MODE_COST>(sc, fr, out, ray_count, qs, (fr.n_chunks + 63u) / 64u, stream, use_lds);
        if (e != hipSuccess) return e;
        return launch_main<MAXD, MODE_STATIC>(sc, fr, out, ray_count, qs, fr.n_chunks, stream, use_lds);
    }
    if ((variant & RT_VARIANT_COOP) != 0 && !use_lds) {
        const uint32_t groups = (fr.n_chunks + COOP_WAVES - 1u) / COOP_WAVES;
        if (groups == 0u) return hipSuccess;
        record_main_kernel_event(0, stream);
        hipLaunchKernelGGL((whitted_coop_kernel<MAXD>), dim3(groups), dim3(COOP_WAVES * 64), COOP_LDS_BYTES, stream, sc, fr, out, ray_count, qs);
        record_main_kernel_event(1, stream);
        return hipGetLastError();
    }
    if (scheme == RT_VARIANT_PERSISTENT) {
        const uint32_t waves = fr.n_chunks < resident_waves ? fr.n_chunks : resident_waves;
        return launch_main<MAXD, MODE_PERSISTENT>(sc, fr, out, ray_count, qs, waves, stream, use_lds);
    }
    /* two-phase (default): coherent tiles that evict their stragglers, then the stragglers packed 64 per wave */
    hipError_t e = launch_mode<MAXD, MODE_PHASE1>(sc, fr, out, ray_count, qs, fr.n_chunks, stream, use_lds);
    if (e != hipSuccess) return e;
    const uint32_t max_cont_waves = (qs.cont_capacity + 63u) / 64u;
    const uint32_t waves = max_cont_waves < resident_waves ? max_cont_waves : resident_waves;
    return launch_mode<MAXD, MODE_PHASE2>(sc, fr, out, ray_count, qs, waves, stream, use_lds);
}

uint32_t cont_record_dwords(int32_t max_depth) { return CONT_FIXED + FRAME_DWORDS * (uint32_t)(max_depth <= 8 ? 8 : RT_MAX_DEPTH); }

/* qs.work_queue and qs.cont_count must be zero; resident_waves: CUs * 4 * RT_MIN_WAVES */
hipError_t launch_whitted(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                          const KernelQueues &qs, uint32_t resident_waves, hipStream_t stream, int variant) {
    if (fr.max_depth <= 8) return launch_maxd<8>(sc, fr, out, ray_count, qs, resident_waves, stream, variant);
    return launch_maxd<RT_MAX_DEPTH>(sc, fr, out, ray_count, qs, resident_waves, stream, variant);
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
