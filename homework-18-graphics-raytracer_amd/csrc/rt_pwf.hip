/*
 * rt_pwf.hip — the Whitted render path as ONE persistent kernel of workgroup-local wavefronts (RT_VARIANT_PWF, the default).
 *
 * The per-pixel kernel (rt_kernels.hip) keeps a lane on one pixel for its whole ray tree: ~50 dependent casts for the
 * deepest pixels bound the frame, a wave's lanes sit in different phases (so the code between two casts runs once per
 * phase present), and its live state allows three waves per SIMD.  Here every cast is a work item of its own:
 *
 *   NODE   a ray_trace activation (main.rs:466-519): its own cast, the hit, the material's contributions; emits the
 *          reflection child (get_reflect, 328-341), a REFR item if get_refract is due, a SHADE item if get_shade is
 *   REFR   one cast of get_refract (343-405): the inside cast, then one item per total-internal-reflection bounce;
 *          emits the escape ray as a NODE
 *   SHADE  one shadow cast of get_shade's light loop (407-464); the item carries the running sum and goes round once
 *          per light that needs a cast, so the lights are still added in order
 *
 * A workgroup (eight waves) owns an arena in HBM with its node records and three queues, and runs a loop of
 * barrier-separated iterations; in each, every wave takes one 64-item chunk — NODE and REFR chunks first (they are the
 * critical path and make new work), a batch of fresh 8x8 tiles from the frame-wide counter when those run short, SHADE
 * chunks in whatever slots are left (about half of all casts of a frame, needed only at the end: the filler that keeps
 * the lanes busy while the deep chains of the last tiles play out).  All 64 lanes of a chunk are in the same phase, so
 * the code between casts runs once; an item is a few dozen bytes, so nothing but the cast's own temporaries is live
 * across the intersection loop.  When a workgroup's queues are dry it folds its records bottom-up,
 * value = (shade*sc + reflection*rc) + (refraction*decay)*fc (main.rs:516-518), and writes its pixels.
 *
 * Subtrees are pure functions of their rays and every helper (rt_shade.h, rt_cast.h) and the association of the fold
 * are the per-pixel kernel's, so the two paths agree bit for bit with each other and with the oracle
 * (tests/test_gpu_wavefront.py).  No inter-workgroup communication except the tile counter (one atomic per eight tiles)
 * and the final cast count; queue bookkeeping is LDS atomics.  The two rings never overflow: a node has at most one
 * SHADE and one REFR item alive (plus the successor a wave is writing while the item is still being read), and the
 * rings hold node_cap + 1024 items.  Arenas have a fixed capacity: a workgroup stops taking
 * tiles when its arena fills up, and if a frame cannot be finished that way an overflow flag makes the launcher's
 * trailing per-pixel kernel (a no-op otherwise) render the frame instead.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_shade.h"
#include "rt_kernels.h"
#include "rt_cast.h"
#include "rt_pwf_common.h"

namespace rt {

#ifndef PW_WAVES
#define PW_WAVES 8u
#endif
#define PW_THREADS (PW_WAVES * 64u)
#define PW_TILE_BATCH 8u      /* tiles fetched per global atomic and turned into root nodes by one wave */
#ifndef PW_ROUNDS
#define PW_ROUNDS 4u /* chunks a wave may take per iteration when that many are queued: fewer barriers per cast while there is
                      * plenty of work; one chunk per wave (the shortest step for the dependent chains) when there is not */
#endif
#ifndef PW_PENDING_CHUNKS
#define PW_PENDING_CHUNKS 12u /* a workgroup takes new tiles only while fewer chunks than this are queued: tiles must last
                                           * to the end of the frame, or the workgroups that met the expensive ones finish long after the rest */
#endif
#ifndef PW_MIN_WAVES
#define PW_MIN_WAVES 6
#endif

enum : uint32_t { PW_T_NONE = 0u, PW_T_NODE = 1u, PW_T_REFR = 2u, PW_T_TILES = 3u, PW_T_SHADE = 4u };

struct PwShared {
    uint32_t n_alloc, n_taken;         /* nodes: ids are queue positions (every node is queued once, when it is created) */
    uint32_t f_alloc, f_taken;         /* refraction ring (positions; slot = position & ring_mask) */
    uint32_t s_alloc, s_taken;         /* shade ring */
    uint32_t tile_buf[PW_TILE_BATCH];  /* tiles fetched from the frame-wide counter, not started yet */
    uint32_t tile_count;               /* valid entries of tile_buf */
    uint32_t tiles_exhausted;          /* the frame-wide counter ran out, or this arena has no room for more tiles */
    uint32_t tile_list_count;          /* tiles this workgroup started */
    uint32_t abort;                    /* arena overflow: stop; the frame falls back to the per-pixel kernel */
    uint32_t next_chunk;               /* chunks of the running iteration handed out so far */
    uint32_t tiles_seen;               /* the frame-wide tile counter as of this workgroup's last fetch */
};

/* the frame description travels through memory (pp.frame, written by pwf_init_kernel): it is read twice per tile, and as
 * a by-value argument its 25 dwords would sit in SGPRs across the intersection loop, which needs those itself */
__global__ __launch_bounds__(PW_THREADS, PW_MIN_WAVES) void pwf_kernel(const KernelScene sc, const PwParams pp, float *__restrict__ out) {
    __shared__ PwShared S;
    const KernelFrame &fr = *pp.frame;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t n_tiles = fr.n_chunks;
    const float THRESHOLD = 0.001f; /* main.rs:467 */

    /* this workgroup's arena */
    unsigned char *arena = pp.arena + (size_t)blockIdx.x * pp.arena_stride;
    uint4 *node_in = reinterpret_cast<uint4 *>(arena);                 /* node_cap x 2: ray, exclusion|mode|depth, contribution */
    uint4 *nodes = node_in + (size_t)pp.node_cap * 2u;                   /* node_cap x 2: shade term, rc | fc, decay, children   */
    uint4 *shade_q = nodes + (size_t)pp.node_cap * 2u;                   /* ring_cap x 5 */
    uint4 *refr_q = shade_q + (size_t)pp.ring_cap * 5u;                  /* ring_cap x 3 */
    uint32_t *tile_list = reinterpret_cast<uint32_t *>(refr_q + (size_t)pp.ring_cap * 3u); /* (tile, first node) pairs */
    const uint32_t ring_mask = pp.ring_cap - 1u;
    const uint32_t tile_cap = pp.node_cap / 64u;

    if (threadIdx.x == 0u) {
        S.n_alloc = S.n_taken = 0u;
        S.f_alloc = S.f_taken = 0u;
        S.s_alloc = S.s_taken = 0u;
        S.tile_count = 0u;
        S.tiles_exhausted = 0u;
        S.tile_list_count = 0u;
        S.abort = 0u;
        S.tiles_seen = 0u;
    }
    uint32_t casts = 0u;
#ifdef PW_STATS /* diagnostic build: iteration and slot-use totals into the global words */
    uint32_t st_iter = 0u, st_n = 0u, st_f = 0u, st_t = 0u, st_s = 0u, st_partial = 0u, st_work = 0u, st_exh_t = 0u, st_exh_iter = 0u, st_exh_pend = 0u;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    for (;;) {
        __syncthreads(); /* (A) last iteration's appends are complete and visible */
        if (S.abort != 0u) break;
        const uint32_t n_alloc = S.n_alloc, n_taken = S.n_taken;
        const uint32_t f_alloc = S.f_alloc, f_taken = S.f_taken;
        const uint32_t s_alloc = S.s_alloc, s_taken = S.s_taken;
        const uint32_t tile_count = S.tile_count, exhausted = S.tiles_exhausted;

        /* ---- the iteration's schedule: the same arithmetic in every wave ---- */
        const uint32_t availN = n_alloc - n_taken, availF = f_alloc - f_taken, availS = s_alloc - s_taken;
        const uint32_t fullN = availN >> 6, fullF = availF >> 6, fullS = availS >> 6;
#ifdef PW_EXP_ENDGAME /* experiment: short iterations once no more tiles can be had */
        uint32_t slots = (exhausted != 0u && tile_count == 0u) ? PW_WAVES * PW_EXP_ENDGAME : PW_WAVES * PW_ROUNDS;
#else
        uint32_t slots = PW_WAVES * PW_ROUNDS;
#endif
        /* fresh tiles while the critical queues are short: one wave turns the fetched batch into root nodes */
        const uint32_t tT = (tile_count != 0u && fullN + fullF + fullS < PW_PENDING_CHUNKS) ? 1u : 0u; slots -= tT;
        const uint32_t tN = fullN < slots ? fullN : slots; slots -= tN;
        const uint32_t tF = fullF < slots ? fullF : slots; slots -= tF;
        const uint32_t tS = fullS < slots ? fullS : slots; slots -= tS;
        /* partial chunks as soon as a queue's full chunks are all taken: waiting for a chunk to fill up would delay the
         * dependent chains behind its items, which costs more than the idle lanes (measured) */
        const uint32_t pN = (slots != 0u && tN == fullN && (availN & 63u) != 0u) ? 1u : 0u; slots -= pN;
        const uint32_t pF = (slots != 0u && tF == fullF && (availF & 63u) != 0u) ? 1u : 0u; slots -= pF;
        const uint32_t pS = (slots != 0u && tS == fullS && (availS & 63u) != 0u) ? 1u : 0u; slots -= pS;
        const uint32_t takeN = tN * 64u + (pN ? (availN & 63u) : 0u);
        const uint32_t takeF = tF * 64u + (pF ? (availF & 63u) : 0u);
        const uint32_t takeS = tS * 64u + (pS ? (availS & 63u) : 0u);
        const uint32_t busy = tN + pN + tF + pF + tT + tS + pS;
        if (busy == 0u && tile_count == 0u && exhausted != 0u) break; /* nothing queued, nothing left to fetch */
#ifdef PW_STATS
        if (st_exh_t == 0u && exhausted != 0u && tile_count == 0u) {
            st_exh_t = (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_t0); st_exh_iter = st_iter;
            st_exh_pend = fullN | (fullF << 10) | (fullS << 20);
        }
        st_iter += 1u; st_n += tN + pN; st_f += tF + pF; st_t += tT; st_s += tS + pS; st_partial += pN + pF + pS;
#endif

        const uint32_t my_tile = lane < PW_TILE_BATCH ? S.tile_buf[lane] : 0u; /* thread 0 may refill the buffer after (B) */
        if (threadIdx.x == 0u) S.next_chunk = PW_WAVES; /* the first PW_WAVES chunks go to the waves by number, the rest to whoever is free */
        __syncthreads(); /* (B) everyone has read the queue state */
        /* Fetch tiles one iteration before they can be needed, and only as many as there are waves that would otherwise
         * have nothing queued next time: a tile's work cannot be predicted (64 primary rays can grow into thousands of
         * casts), so tiles must last to the end of the frame and be handed out in small portions, or the workgroups that
         * met the expensive ones finish long after the rest.  The atomic is issued here and its result used after this
         * iteration's chunks, so its latency is not on the iteration's path. */
        uint32_t fetch_need = 0u, fetch_first = 0u;
        if (threadIdx.x == 0u) {
            S.n_taken = n_taken + takeN;
            S.f_taken = f_taken + takeF;
            S.s_taken = s_taken + takeS;
            if (tT != 0u) S.tile_count = 0u;
            const uint32_t left_over = (fullN - tN) + (fullF - tF) + (fullS - tS);
            uint32_t need = left_over < PW_WAVES ? PW_WAVES - left_over : 0u;
            /* ... and never more than a fair share of what is left (guided self-scheduling): the last tiles of a frame, or
             * all tiles of a small one, are spread over the workgroups one at a time */
            const uint32_t seen = S.tiles_seen;
            const uint32_t share = (n_tiles - (seen < n_tiles ? seen : n_tiles)) / (2u * gridDim.x);
            if (need > share) need = share != 0u ? share : (need != 0u ? 1u : 0u);
            if ((tT != 0u || tile_count == 0u) && exhausted == 0u && need != 0u) {
                /* room for the batch's rays with a margin for their trees: stop taking tiles when the arena is nearly full */
                const uint32_t room = pp.node_cap - (n_alloc < pp.node_cap ? n_alloc : pp.node_cap);
                if (room < need * 64u * pp.tile_reserve || S.tile_list_count + need > tile_cap) {
                    S.tiles_exhausted = 1u;
                } else {
                    fetch_need = need;
                    fetch_first = atomicAdd(pp.global + PW_G_TILE, need);
                }
            }
        }
        /* the chunks of this iteration, dealt round-robin: tiles first (their rays are due next time), then NODE, REFR, SHADE */
        const uint32_t cN = tN + pN, cF = tF + pF;
#ifdef PW_STATS
        const unsigned long long st_w0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (uint32_t v = wave; v < busy;) {
            uint32_t type, start = 0u, count = 0u;
            {
                uint32_t w = v;
                /* this wave's next chunk: whichever is next when it gets here (waves that drew short chunks take more of them) */
                uint32_t nx = 0u;
                if (lane == 0u) nx = atomicAdd(&S.next_chunk, 1u);
                v = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
                if (w < tT) { type = PW_T_TILES; }
                else if ((w -= tT) < cN) { type = PW_T_NODE; start = n_taken + w * 64u; count = takeN - w * 64u; }
                else if ((w -= cN) < cF) { type = PW_T_REFR; start = f_taken + w * 64u; count = takeF - w * 64u; }
                else { w -= cF; type = PW_T_SHADE; start = s_taken + w * 64u; count = takeS - w * 64u; }
                if (count > 64u) count = 64u;
            }

            if (type == PW_T_TILES) {
                /* main.rs:1093-1100: the primary rays of up to PW_TILE_BATCH tiles become root nodes (depth max_depth,
                 * contribution 1.0); they are cast from the next iteration on like any other node */
                for (uint32_t k = 0; k < tile_count; ++k) {
                    const uint32_t tile = (uint32_t)__builtin_amdgcn_readlane((int)my_tile, (int)k);
                    const uint32_t first_slot = tile * 64u;
                    const uint32_t nv = total_slots - first_slot < 64u ? total_slots - first_slot : 64u;
                    uint32_t base = 0u, entry = 0u;
                    if (lane == 0u) {
                        base = atomicAdd(&S.n_alloc, nv);
                        entry = atomicAdd(&S.tile_list_count, 1u);
                    }
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry);
                    if (base + nv > pp.node_cap || entry >= tile_cap) { /* cannot happen with the reserve above; be safe */
                        if (lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
                        break;
                    }
                    if (lane == 0u) { tile_list[entry * 2u] = tile; tile_list[entry * 2u + 1u] = base; }
                    if (lane < nv) {
                        uint32_t row, col;
                        pw_slot_to_pixel(fr, first_slot + lane, &row, &col);
                        /* Camera::shoot (main.rs:84-99), per-frame basis hoisted to the host */
                        const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                        const float clip_y = (fr.half_height - (float)y) / fr.height_f;
                        const float clip_x = ((float)x - fr.half_width) / fr.height_f;
                        const V3 cx = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
                        const V3 cy = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
                        const V3 ct = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
                        const V3 d = normalize(clip_x * cx + clip_y * cy + ct);
                        const uint32_t depth = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0);
                        const uint32_t id = base + lane;
                        node_in[(size_t)id * 2u] = make_uint4(pfu(fr.cam_origin[0]), pfu(fr.cam_origin[1]), pfu(fr.cam_origin[2]), pfu(d.x));
                        node_in[(size_t)id * 2u + 1u] = make_uint4(pfu(d.y), pfu(d.z), (depth << PW_DEPTH_SHIFT) | (FACE_FRONT << PW_MODE_SHIFT), pfu(1.0f));
                    }
                }
                continue;
            }

            /* ---- load the chunk's items and set up their rays ---- */
            const bool active = lane < count;
            Ray req;
            req.o = v3(0.0f, 0.0f, 0.0f);
            req.d = v3(0.0f, 0.0f, 1.0f);
            req.mode = FACE_FRONT;
            req.excl = 0u;
            bool do_cast = active;
            uint32_t id = 0u;            /* NODE: this node; REFR: the parent node; SHADE: the shaded node */
            uint32_t depth = 0u;         /* NODE: depth left for this node; REFR: depth left for the escape child */
            float contribution = 1.0f;   /* NODE: contribution; REFR: the child's contribution */
            uint32_t obj = 0u;           /* REFR, SHADE */
            float travel = 0.0f;         /* REFR */
            int32_t retry = -1;          /* REFR: -1 = the pending cast is the first inside cast (main.rs:371) */
            uint32_t prim = 0u, light_i = 0u, sflags = 0u; /* SHADE */
            V3 spos = v3(0.0f, 0.0f, 0.0f), adj_n = v3(0.0f, 0.0f, 1.0f), in_dir = v3(0.0f, 0.0f, 1.0f), sdiffuse = spos, sum = spos;
            DirLight dl;
            dl.direction = dl.color = v3(0.0f, 0.0f, 0.0f);

            if (type == PW_T_NODE) {
                id = start + lane;
                if (active) {
                    const uint4 a = node_in[(size_t)id * 2u], b = node_in[(size_t)id * 2u + 1u];
                    req.o = v3(puf(a.x), puf(a.y), puf(a.z));
                    req.d = v3(puf(a.w), puf(b.x), puf(b.y));
                    req.mode = (b.z >> PW_MODE_SHIFT) & 3u;
                    depth = (b.z >> PW_DEPTH_SHIFT) & 63u;
                    req.excl = b.z & PW_EXCL_MASK;
                    contribution = puf(b.w);
                }
            } else if (type == PW_T_REFR) {
                if (active) {
                    const uint4 *t = refr_q + (size_t)((start + lane) & ring_mask) * 3u;
                    const uint4 a = t[0], b = t[1], c = t[2];
                    req.o = v3(puf(a.x), puf(a.y), puf(a.z));
                    req.d = v3(puf(a.w), puf(b.x), puf(b.y));
                    req.mode = (b.z >> PW_MODE_SHIFT) & 3u;
                    depth = (b.z >> PW_DEPTH_SHIFT) & 63u;
                    req.excl = b.z & PW_EXCL_MASK;
                    id = b.w;
                    obj = c.x;
                    contribution = puf(c.y);
                    travel = puf(c.z);
                    retry = (int32_t)c.w;
                }
            } else { /* PW_T_SHADE */
                if (active) {
                    const uint4 *t = shade_q + (size_t)((start + lane) & ring_mask) * 5u;
                    const uint4 a = t[0], b = t[1], c = t[2], d = t[3], e = t[4];
                    id = a.x; prim = a.y; obj = a.z & 0xffffu; light_i = (a.z >> 16) & 0x7fffu; sflags = a.z >> 31;
                    spos = v3(puf(b.x), puf(b.y), puf(b.z)); sum.x = puf(b.w);
                    adj_n = v3(puf(c.x), puf(c.y), puf(c.z)); sum.y = puf(c.w);
                    in_dir = v3(puf(d.x), puf(d.y), puf(d.z)); sum.z = puf(d.w);
                    sdiffuse = v3(puf(e.x), puf(e.y), puf(e.z));
                    do_cast = next_shadow_ray(sc, &light_i, spos, adj_n, &dl); /* always true for a queued item */
                    req.o = spos;
                    req.d = -dl.direction;
                    req.mode = FACE_BACK;
                    req.excl = pack_excl(prim, FACE_BACK);
                }
            }

            /* ---- the cast: the one place the intersection loop is instantiated ---- */
            CastResult cr;
            cr.prim = -1;
            cr.t = 0.0f;
            cr.bf = 0u;
            cr.a0 = cr.a1 = cr.a2 = 0.0f;
            if (do_cast) {
                cr = cast_asm(sc, req);
                casts += 1u;
            }

            if (type == PW_T_NODE) {
                /* ---- ray_trace after its cast (main.rs:475-505) ---- */
                V3 acc = v3(0.0f, 0.0f, 0.0f);
                float rc = 0.0f, fc = 0.0f;
                uint32_t rec_cr = PW_FINAL, rec_cf = PW_NO_CHILD; /* a miss is black and final (main.rs:475) */
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
                    if (depth > 0u) {
                        acc = v3(0.0f, 0.0f, 0.0f) * shade_contribution; /* black * shade_contribution unless a SHADE item fills it in */
                        rc = rm.shiness * (1.0f - rm.transparency);      /* main.rs:493 */
                        fc = rm.transparency;                            /* main.rs:502 */
                        rec_cr = PW_NO_CHILD;
                        want_refl = contribution * rc >= THRESHOLD;      /* main.rs:494-495 */
                        if (contribution * fc > THRESHOLD) {             /* main.rs:502-505, strict */
                            V3 refract_in;
                            if (refract_dir(nh.normal, req.d, rm.refraction_index, &refract_in)) { /* else Trapped */
                                inside_d = normalize(refract_in);        /* second normalize, main.rs:362 */
                                want_refr = true;
                            }
                        }
                    }
                    /* depth == 0 (main.rs:488-490): the value is the unscaled shade */
                    if (want_shade) {
                        const Mat m = material_approx(rm, nh.u, nh.v);
                        adj_n = adjust_normal(m.normal, nh.normal); /* main.rs:410 */
                        sdiffuse = m.diffuse;
                        light_i = 0u;
                        want_shade = next_shadow_ray(sc, &light_i, nh.pos, adj_n, &dl); /* no light needs a cast: get_shade = black */
                    }
                }
                /* reflection child (get_reflect, main.rs:328-341) */
                const uint32_t k_refl = lds_append(&S.n_alloc, want_refl);
                bool overflow = want_refl && k_refl >= pp.node_cap;
                if (want_refl && !overflow) {
                    const V3 d = reflect_dir(nh.normal, req.d);
                    const uint32_t word = pack_excl(nh.prim, nh.bf ? FACE_FRONT : FACE_BACK) | (req.mode << PW_MODE_SHIFT) | ((depth - 1u) << PW_DEPTH_SHIFT);
                    node_in[(size_t)k_refl * 2u] = make_uint4(pfu(nh.pos.x), pfu(nh.pos.y), pfu(nh.pos.z), pfu(d.x));
                    node_in[(size_t)k_refl * 2u + 1u] = make_uint4(pfu(d.y), pfu(d.z), word, pfu(contribution * rc));
                    rec_cr = k_refl;
                }
                /* the first shadow cast of get_shade; the ring cannot overflow: at most one SHADE item per node is queued */
                const uint32_t k_shade = lds_append(&S.s_alloc, want_shade);
                if (want_shade) {
                    uint4 *t = shade_q + (size_t)(k_shade & ring_mask) * 5u;
                    t[0] = make_uint4(id, nh.prim, nh.obj | (light_i << 16) | (depth > 0u ? 0u : 0x80000000u), 0u);
                    t[1] = make_uint4(pfu(nh.pos.x), pfu(nh.pos.y), pfu(nh.pos.z), pfu(0.0f));
                    t[2] = make_uint4(pfu(adj_n.x), pfu(adj_n.y), pfu(adj_n.z), pfu(0.0f));
                    t[3] = make_uint4(pfu(req.d.x), pfu(req.d.y), pfu(req.d.z), pfu(0.0f));
                    t[4] = make_uint4(pfu(sdiffuse.x), pfu(sdiffuse.y), pfu(sdiffuse.z), 0u);
                }
                /* the ray into the glass (main.rs:358-366) */
                const uint32_t k_refr = lds_append(&S.f_alloc, want_refr);
                if (want_refr) {
                    uint4 *t = refr_q + (size_t)(k_refr & ring_mask) * 3u;
                    const uint32_t word = pack_excl(nh.prim, FACE_FRONT) | (FACE_BACK << PW_MODE_SHIFT) | ((depth - 1u) << PW_DEPTH_SHIFT);
                    t[0] = make_uint4(pfu(nh.pos.x), pfu(nh.pos.y), pfu(nh.pos.z), pfu(inside_d.x));
                    t[1] = make_uint4(pfu(inside_d.y), pfu(inside_d.z), word, id);
                    t[2] = make_uint4(nh.obj, pfu(contribution * fc), pfu(0.0f), 0xffffffffu);
                }
                if (active) {
                    nodes[(size_t)id * 2u] = make_uint4(pfu(acc.x), pfu(acc.y), pfu(acc.z), pfu(rc));
                    nodes[(size_t)id * 2u + 1u] = make_uint4(pfu(fc), 0u, rec_cr, rec_cf);
                }
                if (__builtin_amdgcn_ballot_w64(overflow) != 0ull && lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
            } else if (type == PW_T_REFR) {
                /* ---- one step of get_refract (main.rs:371-403) ---- */
                bool requeue = false, escape = false;
                V3 esc_o = v3(0.0f, 0.0f, 0.0f), esc_d = esc_o;
                uint32_t esc_excl = 0u;
                float decay = 0.0f;
                if (active && cr.prim >= 0) { /* a miss is Refraction::Infinite (main.rs:373, 383): no child */
                    const HitGeom ih = finish_hit(sc, req, cr, false);
                    if (retry < 0) {
                        travel = distance(ih.pos, req.o); /* main.rs:375; req.o is the node's hit position */
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
                        requeue = true;
                    } else if (have_out) { /* Escaped, main.rs:392-403; else Trapped */
                        escape = true;
                        esc_o = ih.pos;
                        esc_d = normalize(out_dir);
                        esc_excl = pack_excl(ih.prim, FACE_BACK);
                        decay = rtdm::powf(rm.opaque_decay, travel); /* main.rs:508 */
                    }
                }
                const uint32_t k_again = lds_append(&S.f_alloc, requeue);
                if (requeue) {
                    /* at most one REFR item per node is queued, so the ring has room; the slot may not be one this iteration reads */
                    uint4 *t = refr_q + (size_t)(k_again & ring_mask) * 3u;
                    const uint32_t word = req.excl | (req.mode << PW_MODE_SHIFT) | (depth << PW_DEPTH_SHIFT);
                    t[0] = make_uint4(pfu(req.o.x), pfu(req.o.y), pfu(req.o.z), pfu(req.d.x));
                    t[1] = make_uint4(pfu(req.d.y), pfu(req.d.z), word, id);
                    t[2] = make_uint4(obj, pfu(contribution), pfu(travel), (uint32_t)retry);
                }
                const uint32_t k_child = lds_append(&S.n_alloc, escape);
                const bool overflow = escape && k_child >= pp.node_cap;
                if (escape && !overflow) {
                    const uint32_t word = esc_excl | (FACE_FRONT << PW_MODE_SHIFT) | (depth << PW_DEPTH_SHIFT);
                    node_in[(size_t)k_child * 2u] = make_uint4(pfu(esc_o.x), pfu(esc_o.y), pfu(esc_o.z), pfu(esc_d.x));
                    node_in[(size_t)k_child * 2u + 1u] = make_uint4(pfu(esc_d.y), pfu(esc_d.z), word, pfu(contribution));
                    uint32_t *rec = reinterpret_cast<uint32_t *>(nodes + (size_t)id * 2u);
                    rec[5] = pfu(decay);
                    rec[7] = k_child;
                }
                if (__builtin_amdgcn_ballot_w64(overflow) != 0ull && lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
            } else {
                /* ---- one light of get_shade (main.rs:435-461) ---- */
                bool again = false;
                if (active) {
                    const rt_material &rm = sc.materials[obj];
                    if (do_cast) {
                        const rt_light &L = sc.lights[light_i];
                        bool lit = true;
                        if (cr.prim >= 0) {
                            const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                            if (has_origin) {
                                const V3 occ = req.o + req.d * cr.t;
                                const float occlusion_distance = distance(spos, occ);
                                const float light_distance = distance(spos, v3(L.origin[0], L.origin[1], L.origin[2]));
                                if (occlusion_distance < light_distance) lit = false;
                            } else {
                                lit = false;
                            }
                        }
                        if (lit) { /* main.rs:450-461 */
                            Mat m;
                            m.normal = v3(0.0f, 0.0f, 0.0f); /* already folded into adj_n */
                            m.diffuse = sdiffuse;
                            m.specular = v3(rm.specular_color[0], rm.specular_color[1], rm.specular_color[2]);
                            m.shiness = rm.shiness;
                            m.smoothness = rm.smoothness;
                            m.transparency = rm.transparency;
                            m.refraction_index = rm.refraction_index;
                            m.opaque_decay = rm.opaque_decay;
                            const V3 light_direction = req.d; /* = -light.direction */
                            const V3 view_direction = -in_dir;
                            const V3 diffuse = get_diffuse(m, adj_n, light_direction) * dl.color;
                            const V3 specular = get_specular(m, adj_n, view_direction, light_direction) * dl.color;
                            sum = sum + diffuse * (1.0f - m.shiness) + specular * m.shiness;
                        }
                        light_i += 1u;
                        again = next_shadow_ray(sc, &light_i, spos, adj_n, &dl);
                    }
                    if (!again) {
                        V3 acc = sum; /* depth 0: the unscaled shade (main.rs:488-490) */
                        if (sflags == 0u) {
                            const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                            acc = sum * shade_contribution;
                        }
                        float *rec = reinterpret_cast<float *>(nodes + (size_t)id * 2u);
                        rec[0] = acc.x;
                        rec[1] = acc.y;
                        rec[2] = acc.z;
                    }
                }
                const uint32_t k_again = lds_append(&S.s_alloc, again);
                if (again) {
                    uint4 *t = shade_q + (size_t)(k_again & ring_mask) * 5u;
                    t[0] = make_uint4(id, prim, obj | (light_i << 16) | (sflags << 31), 0u);
                    t[1] = make_uint4(pfu(spos.x), pfu(spos.y), pfu(spos.z), pfu(sum.x));
                    t[2] = make_uint4(pfu(adj_n.x), pfu(adj_n.y), pfu(adj_n.z), pfu(sum.y));
                    t[3] = make_uint4(pfu(in_dir.x), pfu(in_dir.y), pfu(in_dir.z), pfu(sum.z));
                    t[4] = make_uint4(pfu(sdiffuse.x), pfu(sdiffuse.y), pfu(sdiffuse.z), 0u);
                }
            }
    
        }
#ifdef PW_STATS
        st_work += (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_w0);
#endif
        if (wave == 0u) { /* the tiles fetched above: lane k files the k-th one (read after the next barrier) */
            fetch_need = (uint32_t)__builtin_amdgcn_readfirstlane((int)fetch_need);
            fetch_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)fetch_first);
            if (fetch_need != 0u) {
                const uint32_t got = fetch_first >= n_tiles ? 0u : (n_tiles - fetch_first < fetch_need ? n_tiles - fetch_first : fetch_need);
                /* consecutive fetches land far apart in the image (tile_stride is coprime to n_tiles) */
                if (lane < got)
                    S.tile_buf[lane] = pp.tile_order != nullptr ? pp.tile_order[fetch_first + lane]
                                                                : (uint32_t)(((unsigned long long)(fetch_first + lane) * pp.tile_stride) % n_tiles);
                if (lane == 0u) {
                    S.tiles_seen = fetch_first + fetch_need;
                    S.tile_count = got;
                    if (fetch_first + fetch_need >= n_tiles) S.tiles_exhausted = 1u;
                }
            }
        }
    }

#ifdef PW_STATS
    if (lane == 0u) atomicAdd(pp.global + 19, st_work); /* 100 MHz ticks the waves spent on their chunks */
    if (threadIdx.x == 0u) {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(pp.global + 8, st_iter); atomicAdd(pp.global + 9, st_n); atomicAdd(pp.global + 10, st_f);
        atomicAdd(pp.global + 11, st_t); atomicAdd(pp.global + 12, st_s); atomicAdd(pp.global + 13, st_partial);
        atomicAdd(pp.global + 14, (uint32_t)(st_t1 - st_t0)); atomicMax(pp.global + 15, (uint32_t)(st_t1 - st_t0));
        atomicMax(pp.global + 16, st_iter); atomicAdd(pp.global + 17, 1u); atomicMax(pp.global + 18, S.n_alloc);
    }
#endif
    /* ---- fold the records bottom-up (main.rs:516-518) and write the pixels ---- */
    __syncthreads();
    const bool aborted = S.abort != 0u;
    const uint32_t n_nodes = S.n_alloc < pp.node_cap ? S.n_alloc : pp.node_cap;
    const uint32_t max_depth = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0);
    if (!aborted) {
        /* nodes with `left` levels below them, children (left - 1) first; roots (left == max_depth) go last, by tile */
        for (uint32_t left = 1u; left < max_depth; ++left) {
            for (uint32_t id = threadIdx.x; id < n_nodes; id += PW_THREADS) {
                const uint32_t word = reinterpret_cast<const uint32_t *>(node_in + (size_t)id * 2u + 1u)[2];
                if (((word >> PW_DEPTH_SHIFT) & 63u) != left) continue;
                const uint4 a = nodes[(size_t)id * 2u], b = nodes[(size_t)id * 2u + 1u];
                if (b.z == PW_FINAL) continue;
                const float rc = puf(a.w), fc = puf(b.x), decay = puf(b.y);
                V3 reflection = v3(0.0f, 0.0f, 0.0f), refraction = v3(0.0f, 0.0f, 0.0f);
                if (b.z != PW_NO_CHILD) {
                    const uint4 c = nodes[(size_t)b.z * 2u];
                    reflection = v3(puf(c.x), puf(c.y), puf(c.z));
                }
                if (b.w != PW_NO_CHILD) {
                    const uint4 c = nodes[(size_t)b.w * 2u];
                    refraction = v3(puf(c.x), puf(c.y), puf(c.z)) * decay; /* main.rs:508 */
                }
                const V3 value = (v3(puf(a.x), puf(a.y), puf(a.z)) + reflection * rc) + refraction * fc;
                float *rec = reinterpret_cast<float *>(nodes + (size_t)id * 2u);
                rec[0] = value.x;
                rec[1] = value.y;
                rec[2] = value.z;
            }
            __syncthreads();
        }
        const uint32_t n_started = S.tile_list_count;
        for (uint32_t e = wave; e < n_started; e += PW_WAVES) {
            const uint32_t tile = tile_list[e * 2u], base = tile_list[e * 2u + 1u];
            const uint32_t first_slot = tile * 64u;
            const uint32_t nv = total_slots - first_slot < 64u ? total_slots - first_slot : 64u;
            if (lane >= nv) continue;
            const uint32_t id = base + lane;
            const uint4 a = nodes[(size_t)id * 2u], b = nodes[(size_t)id * 2u + 1u];
            V3 value = v3(puf(a.x), puf(a.y), puf(a.z));
            if (b.z != PW_FINAL) {
                const float rc = puf(a.w), fc = puf(b.x), decay = puf(b.y);
                V3 reflection = v3(0.0f, 0.0f, 0.0f), refraction = v3(0.0f, 0.0f, 0.0f);
                if (b.z != PW_NO_CHILD) {
                    const uint4 c = nodes[(size_t)b.z * 2u];
                    reflection = v3(puf(c.x), puf(c.y), puf(c.z));
                }
                if (b.w != PW_NO_CHILD) {
                    const uint4 c = nodes[(size_t)b.w * 2u];
                    refraction = v3(puf(c.x), puf(c.y), puf(c.z)) * decay;
                }
                value = (value + reflection * rc) + refraction * fc;
            }
            uint32_t row, col;
            pw_slot_to_pixel(fr, first_slot + lane, &row, &col);
            /* img[at] = img[at] + photon on a zeroed image (main.rs:1107) */
            float *px = out + ((size_t)row * fr.cols + col) * 3u;
            px[0] = 0.0f + value.x;
            px[1] = 0.0f + value.y;
            px[2] = 0.0f + value.z;
        }
        if (threadIdx.x == 0u && n_started != 0u) atomicAdd(pp.global + PW_G_TILES_DONE, n_started);
    }
#ifdef PW_STATS
    __syncthreads();
    if (threadIdx.x == 0u) { /* per-workgroup record over the (dead) start of the arena */
        uint32_t *rec = reinterpret_cast<uint32_t *>(arena);
        rec[0] = st_iter; rec[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_t0); rec[2] = S.n_alloc; rec[3] = S.tile_list_count;
        rec[4] = (uint32_t)(st_t0 & 0xffffffffu); rec[5] = st_exh_t; rec[6] = st_exh_iter; rec[7] = st_exh_pend;
    }
#endif
    for (int off = 32; off > 0; off >>= 1) casts += __shfl_down(casts, off, 64);
    if (lane == 0u && casts != 0u) atomicAdd(reinterpret_cast<unsigned long long *>(pp.global + PW_G_CASTS), (unsigned long long)casts);
}

__global__ void pwf_init_kernel(uint32_t *global, KernelFrame *frame, const KernelFrame fr) {
    if (threadIdx.x < 32u) global[threadIdx.x] = 0u; /* PW_G_WORDS, plus the diagnostic words of PW_STATS builds */
    if (threadIdx.x == 0u) *frame = fr;
}

/* every tile rendered and no arena overflow: publish the cast count; else raise the flag the fallback launch looks at */
__global__ void pwf_finish_kernel(uint32_t *global, uint32_t n_tiles, unsigned long long *ray_count) {
    if (global[PW_G_OVERFLOW] == 0u && global[PW_G_TILES_DONE] != n_tiles) global[PW_G_OVERFLOW] = 1u;
    if (global[PW_G_OVERFLOW] == 0u && ray_count != nullptr) *ray_count += *reinterpret_cast<const unsigned long long *>(global + PW_G_CASTS);
}

uint32_t pwf_threads() { return PW_THREADS; }
int pwf_workgroups_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwf_kernel, (int)PW_THREADS, 0) != hipSuccess || n < 1) n = 1;
    return n;
}
size_t pwf_arena_bytes(uint32_t node_cap, uint32_t ring_cap) {
    return ((size_t)node_cap * 4u + (size_t)ring_cap * 8u) * sizeof(uint4) + (size_t)(node_cap / 64u) * 2u * sizeof(uint32_t) + 256u;
}

#ifdef PW_STATS
static uint32_t *g_pw_last_global = nullptr;
static PwParams g_pw_last;
static uint32_t g_pw_last_groups = 0;
extern "C" int rt_diag_read_pwf_groups(uint32_t *out8, int max_groups) {
    const int n = (int)g_pw_last_groups < max_groups ? (int)g_pw_last_groups : max_groups;
    if (hipMemcpy2D(out8, 32, g_pw_last.arena, g_pw_last.arena_stride, 32, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
extern "C" int rt_diag_read_pwf(uint32_t *out32) {
    if (!g_pw_last_global) return -1;
    return hipMemcpy(out32, g_pw_last_global, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_pwf(const KernelScene &sc, KernelFrame fr, float *out, unsigned long long *ray_count, const PwParams &pp,
                      uint32_t workgroups, hipStream_t stream, bool async) {
    const uint32_t total = fr.cols * fr.rows;
    fr.n_chunks = (total + 63u) / 64u;
    if (total == 0u) return hipSuccess;
#ifdef PW_STATS
    g_pw_last_global = pp.global;
    g_pw_last = pp;
    g_pw_last_groups = workgroups;
#endif
    hipLaunchKernelGGL(pwf_init_kernel, dim3(1), dim3(64), 0, stream, pp.global, const_cast<KernelFrame *>(pp.frame), fr);
    record_main_kernel_event(0, stream);
    if (async)
        launch_pwf_async_main(sc, pp, out, workgroups, stream);
    else
        hipLaunchKernelGGL(pwf_kernel, dim3(workgroups), dim3(PW_THREADS), 0, stream, sc, pp, out);
    record_main_kernel_event(1, stream);
    hipLaunchKernelGGL(pwf_finish_kernel, dim3(1), dim3(1), 0, stream, pp.global, fr.n_chunks, ray_count);
    return hipGetLastError();
}

} /* namespace rt */
