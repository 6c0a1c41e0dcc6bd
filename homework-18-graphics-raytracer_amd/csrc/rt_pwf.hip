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
 * A workgroup (eight waves) owns an arena in HBM with its node records and three queues, and keeps the head of each queue
 * in LDS: a few pages of SHADE, REFR and NODE items (PA_LDS_*_PAGES; 44 KB per workgroup, three workgroups per CU) that
 * producers fill while there is room and consumers empty first — the arena's rings take only what does not fit, which on
 * the reference frame is little: 0.85 GB of HBM traffic per 1080p frame instead of 2.5 GB, 1.20 ms instead of 1.28
 * (profiles/README.md, round 2).  A page's slot is written again only after its consumer has the items in registers
 * (pa_release_page: pages are released in the order they were claimed).  All 64 lanes of a chunk are
 * in the same phase, so the code between casts runs once; an item is a few dozen bytes, so nothing but the cast's own
 * temporaries is live across the intersection loop (80 VGPRs, six waves per SIMD).  When a workgroup's queues are dry
 * it folds its records bottom-up, value = (shade*sc + reflection*rc) + (refraction*decay)*fc (main.rs:516-518) — the
 * nodes that have something below them, listed level by level in LDS first; the others were written complete — and
 * writes its pixels.  Subtrees are pure functions of their rays and every helper (rt_shade.h, rt_cast.h) and the
 * association of the fold are the per-pixel kernel's, so the two paths agree bit for bit with each other and with the
 * oracle (tests/test_gpu_wavefront.py).
 *
 * Scheduling.  There are no barriers in the main loop: every wave loops on its own —
 *
 *   claim a page (64 consecutive queue positions) of NODE, else REFR items, full or not (they are the dependent chains:
 *   a ray_trace activation, its refraction casts, its child, ...); else start a fresh 8x8 tile — one of the workgroup's
 *   own (half of its even share of the frame's tiles; all of it up to eight), then from the frame-wide counter — while
 *   little SHADE work is queued; else a full page of SHADE items (half of all casts of a frame, needed only at the end:
 *   the filler); else a tile; else a partly filled SHADE page; else sleep until somebody publishes.  (The counters of all
 *   eight queues are read together, in one LDS round trip, before any of this.)
 *
 * — so a chain advances as fast as single chunks take and nobody waits for anybody.  (A first version iterated between
 * two barriers, every wave taking up to four chunks per iteration: an item made in one iteration could be picked up in
 * the next at the earliest, ~45 us later, and the waves waited for the slowest chunk each time: 1.75 ms per frame
 * against 1.57 ms now; profiles/README.md.)  The bookkeeping that makes this safe, all in LDS:
 *
 *   - producers reserve positions with one wave-aggregated atomic on `alloc`, write their items, fence, and then add
 *     the number written to `ready[page]` (an item range may straddle two pages) and bump `gen`;
 *   - consumers claim pages in order with a compare-and-swap on `taken`; a page may be claimed when its ready count is
 *     64, or when it has been SEALED: a consumer that finds only a partly filled last page, all of whose reserved
 *     positions are written, moves `alloc` to the next page boundary (compare-and-swap, so no reservation can slip in)
 *     and marks the page sealed with the count it had;
 *   - a wave that finds nothing counts itself idle and sleeps until `gen` moves; the wave whose count makes all eight
 *     is the last one awake — nobody else can make items or fetch tiles — so if it still finds nothing the queues are
 *     final and the loop ends.
 *
 * No wave ever waits for a particular other wave, so there is nothing to deadlock on; an (unreachable) spin limit turns
 * a would-be hang into the overflow fallback.  Root nodes (primary rays) are made in registers and cast at once; their
 * ids come from the top of the arena so that they do not appear in the NODE queue, whose positions are node ids.  No
 * inter-workgroup communication except the tile counter (one atomic per tile that is not the workgroup's own; consecutive
 * fetches are spread over the image) and the final cast count (one add per workgroup).  The two rings never overflow: a node has at most one SHADE and one REFR item alive
 * (plus the successor a wave is writing while the item is still being read), and the rings hold node_cap + 1024 items.
 * Arenas have a fixed capacity: a workgroup stops taking tiles when its arena fills up, and if a frame cannot be
 * finished that way an overflow flag makes the launcher's trailing per-pixel kernel (a no-op otherwise) render it.
 * The last workgroup to leave (a count of workgroups done) decides that, publishes the cast count and zeroes the block of
 * global words the next launch on the workspace will use: a frame is one launch (rt_kernels.h PwParams, rt_api.hip).
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

#ifndef PA_WAVES
#define PA_WAVES 8u
#endif
#define PA_THREADS (PA_WAVES * 64u)
#ifndef PA_MIN_WAVES
#define PA_MIN_WAVES 6
#endif
#ifndef PA_SHADE_PAGES
#define PA_SHADE_PAGES 4u /* a wave starts a fresh tile rather than a SHADE page while fewer SHADE pages than this are queued */
#endif
#ifndef PA_MIN_PARTIAL
#define PA_MIN_PARTIAL 1u /* a partly filled page is taken (sealed) as soon as a wave has nothing fuller to do, whatever it holds: measured
                           * 1.61 ms per frame against 1.65 / 1.70 / 1.94 ms for a minimum of 4 / 16 / 64 items — the chains behind the
                           * items matter more than the idle lanes */
#endif
#ifndef PA_CHAIN_PARTIAL_MIN
#define PA_CHAIN_PARTIAL_MIN 1u /* partly filled NODE/REFR pages of at least this many items go before fresh tiles and SHADE pages (0: after
                                 * them): they are the dependent chains (1.63 -> 1.57 ms) */
#endif
#ifndef PA_PRIO_CHAIN
#define PA_PRIO_CHAIN 1
#define PA_PRIO_TILE 1
#endif
#ifndef PA_SLEEP
#define PA_SLEEP 16
#endif
/* A SHADE item is 80 bytes and its consumer evaluates its light again (for a spot light an acos and a powf in binary64).  Twice
 * the light was made to travel with the item instead — as direction + colour in a sixth field (96 B), and as the colour in place
 * of the diffuse colour, which the consumer then re-derived from the material and uv (80 B) — and twice the frame got SLOWER,
 * 1.51 and 1.31 ms against 1.28 (profiles/README.md, round 2): evaluating the colour where the item is made spills (95
 * registers).  A third attempt with the items in LDS — direction and colour evaluated apart, the colour after the other
 * fields were stored, 8 spills — was as fast as this and no faster (r02_ab15.txt). */
#define PA_SHADE_U4 5u /* uint4s per SHADE item */
#ifndef PA_LDS_PAGES
#define PA_LDS_PAGES 2u /* pages of SHADE items held in LDS PER QUEUE (a power of two; 5 KB each); what does not fit goes to the ring in the arena */
#endif
#ifndef PA_LQ
#define PA_LQ 3u /* SHADE queues in LDS, by the light the item asks next: 0, 1, ..., and PA_LQ - 1 or beyond.  A chunk from one of
                  * them has ONE light (unless it is the last queue of a scene with more lights): its record comes through scalar
                  * loads and the code for its kind — a spot light's acos and powf in binary64 — runs only in chunks that need it.
                  * Measured on the reference frame (three lights): PA_LQ 3 with 2 pages per queue and 2 + 2 pages of REFR and NODE
                  * items (the same 44 KB) executes 7 % fewer VALU instructions and was 1.5–2 % faster in round 2 (1.18 against 1.20 ms)
                  * at the price of more items through the arena, 1.22 GB of HBM traffic per frame against 1.00 GB
                  * (profiles/r02_ab17…20.txt).  Round 3, with the spill traffic gone and the kernel plainly bound by VALU issue, the
                  * same instructions are worth 4.3 %: 1.052 against 1.100 ms (profiles/r03_ab5.txt) — the default now (PA_LQ 1 with four
                  * pages per queue is the A/B). */
#endif
#define PA_LQ_SHIFT 4u /* type bits 4-5: which of them the claimed page belongs to */
#ifndef PA_LDS_F_PAGES
#define PA_LDS_F_PAGES 2u /* the same for REFR items (3 KB each) */
#endif
#ifndef PA_LDS_N_PAGES
#define PA_LDS_N_PAGES 2u /* and for NODE items (3 KB each: the ray, its word and contribution, the node id) */
#endif
#ifndef PA_DRAIN_ROOM
#define PA_DRAIN_ROOM 64u /* measured on the reference frame: SHADE items through the arena 2.68 M -> 2.01 M of 9.2 M, 1.048 -> 1.040 ms (0: off; 127:
                           * 1.96 M but 1.052 ms; profiles/r03_ab8.txt) */
#endif
#define PA_IN_LDS 8u /* type bit: the claimed page is one of an LDS queue */
#define PA_SPIN_LIMIT (1u << 22)

enum : uint32_t { PA_T_NONE = 0u, PA_T_NODE = 1u, PA_T_REFR = 2u, PA_T_TILE = 3u, PA_T_SHADE = 4u };

struct PaShared {
    PaQueue n, f, s;          /* NODE (positions are node ids), REFR ring, SHADE ring */
    PaQueue l[PA_LQ];         /* the SHADE items held in LDS, by light */
    uint32_t l_released[PA_LQ]; /* pages of `l` whose items have been read: their slots may be written again */
    uint32_t ready_l[PA_LQ][PA_LDS_PAGES];
    PaQueue lf;               /* the REFR items held in LDS */
    uint32_t lf_released;
    uint32_t ready_lf[PA_LDS_F_PAGES];
    PaQueue ln;               /* the NODE items held in LDS: their nodes' ids come from the top of the arena, like the roots' */
    uint32_t ln_released;
    uint32_t ready_ln[PA_LDS_N_PAGES];
    uint32_t root_alloc;      /* root nodes, handed out from the top of the arena downwards */
    uint32_t tiles_exhausted; /* the frame-wide counter ran out, or this arena has no room for another tile */
    uint32_t tile_list_count;
    uint32_t static_next;     /* tiles of this workgroup's own share taken so far */
    uint32_t idle;            /* waves asleep: they found nothing and wait for `gen` to move */
    uint32_t gen;             /* bumped whenever items are published */
    uint32_t done;            /* all waves idle at once: the queues are final */
    uint32_t abort;
};

/* the five fields of a SHADE item (t: field 0 of its entry, in the arena's ring or in LDS) */
template <class P>
__device__ __forceinline__ void pa_store_shade(P *t, uint32_t id, uint32_t prim, uint32_t word, V3 spos, V3 adj_n, V3 in_dir, V3 sdiffuse, V3 sum) {
    t[PA_F(0u)] = make_uint4(id, prim, word, 0u);
    t[PA_F(1u)] = make_uint4(pfu(spos.x), pfu(spos.y), pfu(spos.z), pfu(sum.x));
    t[PA_F(2u)] = make_uint4(pfu(adj_n.x), pfu(adj_n.y), pfu(adj_n.z), pfu(sum.y));
    t[PA_F(3u)] = make_uint4(pfu(in_dir.x), pfu(in_dir.y), pfu(in_dir.z), pfu(sum.z));
    t[PA_F(4u)] = make_uint4(pfu(sdiffuse.x), pfu(sdiffuse.y), pfu(sdiffuse.z), 0u);
}

/* a NODE item: two fields in the arena (its position is the node id), three in LDS */
template <class P>
__device__ __forceinline__ void pa_store_node(P *t, V3 o, V3 d, uint32_t word, float contribution, uint32_t id) {
    t[PA_F(0u)] = make_uint4(pfu(o.x), pfu(o.y), pfu(o.z), pfu(d.x));
    t[PA_F(1u)] = make_uint4(pfu(d.y), pfu(d.z), word, pfu(contribution));
    if (id != 0xffffffffu) t[PA_F(2u)] = make_uint4(id, 0u, 0u, 0u);
}

/* all lanes.  Ids for nodes that are not queued in the arena — the roots and the nodes whose items live in LDS — come from
 * its top, downwards; *overflow when they would meet the queued nodes */
__device__ __forceinline__ uint32_t pa_top_node(uint32_t *root_alloc, const uint32_t *n_alloc, uint32_t node_cap, bool want, bool *overflow) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    *overflow = false;
    if (mask == 0ull) return 0u;
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    const int leader = (int)__builtin_ctzll(mask);
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(root_alloc, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (base + n + lds_load(n_alloc) > node_cap) { *overflow = want; return 0u; }
    return node_cap - 1u - (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)));
}

/* the three fields of a REFR item */
template <class P>
__device__ __forceinline__ void pa_store_refr(P *t, V3 o, V3 d, uint32_t word, uint32_t parent, uint32_t obj, float contribution, float travel, uint32_t retry) {
    t[PA_F(0u)] = make_uint4(pfu(o.x), pfu(o.y), pfu(o.z), pfu(d.x));
    t[PA_F(1u)] = make_uint4(pfu(d.y), pfu(d.z), word, parent);
    t[PA_F(2u)] = make_uint4(obj, pfu(contribution), pfu(travel), retry);
}

/* all lanes, after loading a page of an LDS queue: its slot may be written again once the items are in registers.  Pages are
 * released in the order they were claimed: a wave may wait here for the one before it, which is between its claim and this
 * point too and has nothing but its own LDS reads to wait for */
__device__ __forceinline__ void pa_release_page(uint32_t *released, uint32_t page, uint32_t lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0u) {
        while (lds_load(released) != page) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(released, page + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

#ifdef PA_STATS
/* diagnostic build: wave time by phase (s_memtime ticks, summed over all waves) and chunk / lane counts by item type:
 * [0..3] find work, [4..7] load items, [8..11] the cast, [12..15] after the cast — each by type NODE(0) REFR(1) TILE(2)
 * SHADE(3); [16..19] chunks, [20..23] active lanes; [24] sleep/idle time, [25] fold */
__device__ unsigned long long pa_phase_stats[32];
#define PA_TICK() __builtin_readcyclecounter()
#endif

/* the frame description travels through memory (pp.frame, written by pwf_init_kernel): it is read once per tile, and as a
 * by-value argument its 25 dwords would sit in SGPRs across the intersection loop, which needs those itself */
/* PACKED: the arena queues' page counters two to a word (rt_pwf_common.h): frames of several megapixels.
 * BFS: the intersection loop as a breadth-first walk of the node tree, ray by ray (rt_cast_bfs.h cast_bfs): scenes beyond the caches
 * (KernelScene::bfs_walk); 20 KB more LDS per workgroup for the waves' ray tables */
template <bool PACKED, bool BFS = false>
__global__ __launch_bounds__(PA_THREADS, BFS ? 2 : PA_MIN_WAVES) void pwf_kernel(const KernelScene sc, const PwParams pp, float *__restrict__ out) {
    BfsLds *bfs_lds = nullptr;
    BfsScratch bfs_ws = {nullptr, nullptr, nullptr, 0u, 0u};
    if constexpr (BFS) { /* (nothing of this exists in the other instantiation) */
        __shared__ BfsLds bfs_lds_all[PA_WAVES];
        bfs_lds = &bfs_lds_all[threadIdx.x >> 6];
        uint2 *const mine = reinterpret_cast<uint2 *>(pp.bfs_scratch) + ((size_t)blockIdx.x * PA_WAVES + (threadIdx.x >> 6)) * (2u * (size_t)pp.bfs_items_cap + pp.bfs_jobs_cap);
        bfs_ws.items_a = mine;
        bfs_ws.items_b = mine + pp.bfs_items_cap;
        bfs_ws.jobs = mine + 2u * (size_t)pp.bfs_items_cap;
        bfs_ws.items_cap = pp.bfs_items_cap;
        bfs_ws.jobs_cap = pp.bfs_jobs_cap;
    }
    extern __shared__ uint32_t pa_ready[]; /* node pages | shade ring pages | refraction ring pages */
    __shared__ PaShared S;
    __shared__ uint4 lds_shade[PA_LQ * PA_LDS_PAGES * PA_SHADE_U4 * 64u];
    __shared__ uint4 lds_refr[PA_LDS_F_PAGES * 3u * 64u];
    __shared__ uint4 lds_node[PA_LDS_N_PAGES * 3u * 64u];
    const auto &fr = uniform_ref(pp.frame); /* written by the launch before this one, read-only here: scalar loads */
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t n_tiles = fr.n_chunks;
    const float THRESHOLD = 0.001f; /* main.rs:467 */

    /* this workgroup's arena */
    unsigned char *arena = pp.arena + (size_t)blockIdx.x * pp.arena_stride;
    uint4 *node_in = reinterpret_cast<uint4 *>(arena);                 /* node_cap x 2: ray, exclusion|mode|depth, contribution */
    uint4 *nodes = node_in + (size_t)pp.node_cap * 2u;                   /* node_cap x 2: shade term, rc | fc, decay, children   */
    uint4 *shade_q = nodes + (size_t)pp.node_cap * 2u;                   /* ring_cap x PA_SHADE_U4 */
    uint4 *refr_q = shade_q + (size_t)pp.ring_cap * PA_SHADE_U4;         /* ring_cap x 3 */
    uint32_t *tile_list = reinterpret_cast<uint32_t *>(refr_q + (size_t)pp.ring_cap * 3u); /* (tile, first root node) pairs */
    const uint32_t ring_mask = pp.ring_cap - 1u;
    const uint32_t tile_cap = pp.node_cap / 64u;
    /* one byte per queued node: the level it is folded at (its depth left) if it has a hit below the depth limit, else 0;
     * 16-byte aligned because tile_cap * 8 bytes is a multiple of 16 only for even tile_cap */
    unsigned char *fold_level = reinterpret_cast<unsigned char *>(((uintptr_t)(tile_list + (size_t)tile_cap * 2u) + 15u) & ~(uintptr_t)15u);
    const uint32_t node_pages = (pp.node_cap + 63u) / 64u, ring_pages = pp.ring_cap / 64u;
    uint32_t *ready_n = pa_ready, *ready_s = pa_ready + PA_READY_WORDS(node_pages, PACKED), *ready_f = ready_s + PA_READY_WORDS(ring_pages, PACKED);
    const uint32_t ring_page_mask = ring_pages - 1u;

    for (uint32_t i = threadIdx.x; i < PA_READY_WORDS(node_pages, PACKED) + 2u * PA_READY_WORDS(ring_pages, PACKED); i += PA_THREADS) pa_ready[i] = 0u;
    if (threadIdx.x == 0u) {
        S.n.alloc = S.n.taken = 0u;
        S.f.alloc = S.f.taken = 0u;
        S.s.alloc = S.s.taken = 0u;
        for (uint32_t q = 0; q < PA_LQ; ++q) {
            S.l[q].alloc = S.l[q].taken = 0u;
            S.l_released[q] = 0u;
            for (uint32_t k = 0; k < PA_LDS_PAGES; ++k) S.ready_l[q][k] = 0u;
        }
        S.lf.alloc = S.lf.taken = 0u;
        S.lf_released = 0u;
        for (uint32_t k = 0; k < PA_LDS_F_PAGES; ++k) S.ready_lf[k] = 0u;
        S.ln.alloc = S.ln.taken = 0u;
        S.ln_released = 0u;
        for (uint32_t k = 0; k < PA_LDS_N_PAGES; ++k) S.ready_ln[k] = 0u;
        S.root_alloc = 0u;
        S.tiles_exhausted = 0u;
        S.tile_list_count = 0u;
        S.static_next = 0u;
        S.idle = 0u;
        S.gen = 0u;
        S.done = 0u;
        S.abort = 0u;
    }
    __syncthreads();
    uint32_t casts = 0u;
    /* all lanes: queue a SHADE item for the lanes that `want` one — in the LDS queue of its light while there is room, else in
     * the arena's ring */
    auto queue_shade = [&](bool want, uint32_t id, uint32_t prim, uint32_t word, uint32_t light_i, V3 spos, V3 adj_n, V3 in_dir, V3 sdiffuse, V3 sum) {
        const uint32_t mine = light_i < PA_LQ - 1u ? light_i : PA_LQ - 1u;
#pragma unroll
        for (uint32_t q = 0; q < PA_LQ; ++q) {
            const bool w = want && mine == q;
            if (__builtin_amdgcn_ballot_w64(w) == 0ull) continue;
            bool fits; /* per lane: a queue takes what it has room for, the rest of the wave's items go on */
            const uint32_t k = pa_try_append_some(&S.l[q], &S.l_released[q], PA_LDS_PAGES * 64u, w, &fits);
            if (fits) pa_store_shade(lds_shade + q * (PA_LDS_PAGES * PA_SHADE_U4 * 64u) + pa_entry(k & (PA_LDS_PAGES * 64u - 1u), PA_SHADE_U4), id, prim, word, spos, adj_n, in_dir, sdiffuse, sum);
            pa_publish(S.ready_l[q], PA_LDS_PAGES - 1u, fits, k, &S.gen);
            want = want && !fits;
        }
#ifdef PA_STATS /* which light's queue had no room */
        for (uint32_t q = 0; q < 3u; ++q) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(want && (mine < 2u ? mine : 2u) == q);
            if (m != 0ull && lane == (uint32_t)__builtin_ctzll(m)) atomicAdd(pp.global + 28 + q, (uint32_t)__builtin_popcountll(m));
        }
#endif
        const uint32_t k = lds_append(&S.s.alloc, want);
        if (want) pa_store_shade(shade_q + pa_entry(k & ring_mask, PA_SHADE_U4), id, prim, word, spos, adj_n, in_dir, sdiffuse, sum);
        pa_publish<PACKED>(ready_s, ring_page_mask, want, k, &S.gen);
    };
#ifdef PA_STATS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t st_hist[5] = {0u, 0u, 0u, 0u, 0u}; /* chunks by item count: <= 8, <= 16, <= 32, < 64, 64 */
    uint32_t st_shadow[3] = {0u, 0u, 0u};
    unsigned long long ph[26];
    for (int k = 0; k < 26; ++k) ph[k] = 0ull;
#endif

    for (;;) {
        if (lds_load(&S.done) != 0u || lds_load(&S.abort) != 0u) break;

        /* ---- find work (lane 0 decides, the wave follows) ---- */
#ifdef PA_STATS
        const unsigned long long ph_t0 = PA_TICK();
#endif
        uint32_t type = PA_T_NONE, start = 0u, count = 0u;
        if (lane == 0u) {
            const uint32_t gen = lds_load(&S.gen); /* before looking: a publish during the look must not be slept through */
            /* which queues hold unclaimed positions at all — their counters read together, one LDS round trip for all eight:
             * towards the end of a frame most of them are empty most of the time, and looking at them one after the other (two
             * dependent reads each) was a microsecond of every step of the chains the frame then waits for.  A queue that gets
             * its first item after this look is noticed through `gen` (read above), as before. */
            auto look = [&]() -> uint32_t {
                const uint32_t a0 = lds_load(&S.ln.alloc), a1 = lds_load(&S.n.alloc), a2 = lds_load(&S.lf.alloc), a3 = lds_load(&S.f.alloc);
                const uint32_t t0 = lds_load(&S.ln.taken), t1 = lds_load(&S.n.taken), t2 = lds_load(&S.lf.taken), t3 = lds_load(&S.f.taken);
                uint32_t al[PA_LQ], tl[PA_LQ];
                for (uint32_t q = 0; q < PA_LQ; ++q) { al[q] = lds_load(&S.l[q].alloc); tl[q] = lds_load(&S.l[q].taken); }
                const uint32_t as = lds_load(&S.s.alloc), ts = lds_load(&S.s.taken);
                uint32_t m = (a0 > t0 * 64u ? 1u : 0u) | (a1 > t1 * 64u ? 2u : 0u) | (a2 > t2 * 64u ? 4u : 0u) | (a3 > t3 * 64u ? 8u : 0u);
                for (uint32_t q = 0; q < PA_LQ; ++q) m |= al[q] > tl[q] * 64u ? 16u << q : 0u;
                m |= as > ts * 64u ? 16u << PA_LQ : 0u;
                return m;
            };
            uint32_t have = look();
            /* the dependent chains: NODE, then REFR pages (those in LDS first), full ones, then whatever there is */
            auto claim_chain = [&](uint32_t min_partial) {
                if ((have & 1u) != 0u) {
                    count = pa_claim(&S.ln, S.ready_ln, PA_LDS_N_PAGES - 1u, min_partial, &start);
                    if (count != 0u) { type = PA_T_NODE | PA_IN_LDS; return; }
                }
                if ((have & 2u) != 0u) {
                    count = pa_claim<PACKED>(&S.n, ready_n, 0xffffffffu, min_partial, &start);
                    if (count != 0u) { type = PA_T_NODE; return; }
                }
                if ((have & 4u) != 0u) {
                    count = pa_claim(&S.lf, S.ready_lf, PA_LDS_F_PAGES - 1u, min_partial, &start);
                    if (count != 0u) { type = PA_T_REFR | PA_IN_LDS; return; }
                }
                if ((have & 8u) != 0u) {
                    count = pa_claim<PACKED>(&S.f, ready_f, ring_page_mask, min_partial, &start);
                    if (count != 0u) type = PA_T_REFR;
                }
            };
            auto claim_shade = [&](uint32_t min_partial) {
                for (uint32_t q = 0; q < PA_LQ; ++q) {
                    if ((have & (16u << q)) == 0u) continue;
                    count = pa_claim(&S.l[q], S.ready_l[q], PA_LDS_PAGES - 1u, min_partial, &start);
                    if (count != 0u) { type = PA_T_SHADE | PA_IN_LDS | (q << PA_LQ_SHIFT); return; }
                }
                if ((have & (16u << PA_LQ)) == 0u) return;
                count = pa_claim<PACKED>(&S.s, ready_s, ring_page_mask, min_partial, &start);
                if (count != 0u) type = PA_T_SHADE;
            };
            claim_chain(0u);
            /* a light's LDS queue with less than PA_DRAIN_ROOM positions left is served before partly filled chain pages: what
             * does not fit a queue goes through the arena's ring in HBM (80 bytes written and read back per item).  Only while
             * there are tiles to start: after that the chains are all that the frame waits for */
            if (PA_DRAIN_ROOM != 0u && type == PA_T_NONE && lds_load(&S.tiles_exhausted) == 0u) {
                for (uint32_t q = 0; q < PA_LQ; ++q) {
                    if (lds_load(&S.l[q].alloc) - (lds_load(&S.l_released[q]) << 6) + PA_DRAIN_ROOM > PA_LDS_PAGES * 64u) {
                        count = pa_claim(&S.l[q], S.ready_l[q], PA_LDS_PAGES - 1u, 0u, &start);
                        if (count != 0u) { type = PA_T_SHADE | PA_IN_LDS | (q << PA_LQ_SHIFT); break; }
                    }
                }
            }
            if (PA_CHAIN_PARTIAL_MIN != 0u && type == PA_T_NONE) claim_chain(PA_CHAIN_PARTIAL_MIN);
            bool tried_tile = false;
            for (int pass = 0; pass < 2 && type == PA_T_NONE; ++pass) {
                /* a fresh tile: before SHADE work while little of it is queued (pass 0), else after the full pages (pass 1) */
                if (!tried_tile && lds_load(&S.tiles_exhausted) == 0u) {
                    uint32_t shade_pages = (lds_load(&S.s.alloc) >> 6) - lds_load(&S.s.taken);
                    for (uint32_t q = 0; q < PA_LQ; ++q) shade_pages += (lds_load(&S.l[q].alloc) >> 6) - lds_load(&S.l[q].taken);
                    if (pass == 1 || (int32_t)shade_pages < (int32_t)PA_SHADE_PAGES) {
                        tried_tile = true;
                        const uint32_t used = lds_load(&S.n.alloc) + lds_load(&S.root_alloc);
                        const uint32_t room = pp.node_cap > used ? pp.node_cap - used : 0u;
                        if (room < 64u * pp.tile_reserve || lds_load(&S.tile_list_count) >= tile_cap) {
                            S.tiles_exhausted = 1u; /* this arena is nearly full: the other workgroups take the rest */
                        } else {
                            /* Half of a workgroup's even share of the tiles are its own — workgroup w takes tiles w, w + G, w + 2 G, ...
                             * off a counter in LDS — and the rest come from the frame-wide counter, which evens out what the tiles turn out
                             * to cost; a share of eight tiles or less is the workgroup's own entirely.  One counter word serves ~88 fetches
                             * per microsecond, and whoever comes first takes: with every tile from it, a 1/8 share's 4 050 tiles went to
                             * the eight waves each of the 500 workgroups launched first and a third of the chip got none; and in a full
                             * frame the counter sat on the path of every tile started (profiles/r03_ab11.txt: 1.003 -> 0.939 ms per frame,
                             * a 1/8 share 0.419 -> 0.336 ms). */
                            const uint32_t share = n_tiles / gridDim.x;
#ifdef PA_STATIC_EIGHTHS /* A/B: a fixed fraction */
                            const uint32_t own = (uint32_t)(((unsigned long long)share * PA_STATIC_EIGHTHS) >> 3);
#else
                            const uint32_t own = share / 2u > (share < 8u ? share : 8u) ? share / 2u : (share < 8u ? share : 8u);
#endif
                            uint32_t k;
                            const uint32_t mine = own != 0u && lds_load(&S.static_next) < own ? atomicAdd(&S.static_next, 1u) : own;
                            if (mine < own) k = blockIdx.x + gridDim.x * mine;
                            else k = gridDim.x * own + atomicAdd(pp.global + PW_G_TILE, 1u);
                            if (k >= n_tiles) {
                                S.tiles_exhausted = 1u;
                            } else {
                                /* consecutive fetches land far apart in the image (tile_stride is coprime to n_tiles) */
                                start = pp.tile_order != nullptr ? pp.tile_order[k] : (uint32_t)(((unsigned long long)k * pp.tile_stride) % n_tiles);
                                count = total_slots - start * 64u < 64u ? total_slots - start * 64u : 64u;
                                type = PA_T_TILE;
                            }
                        }
                    }
                }
                if (type == PA_T_NONE && pass == 0) claim_shade(0u);
            }
            /* Partly filled pages: waiting for them to fill would hold up the chains behind their items, but a page taken
             * with a handful of items costs a full intersection loop.  Half a page at least while other waves are awake
             * and may add to it; anything once this wave is the last one awake (then nobody will). */
            for (int last = 0; last < 2 && type == PA_T_NONE; ++last) {
                const uint32_t min_partial = last ? 1u : PA_MIN_PARTIAL;
                have = look(); /* afresh: the second time round this wave has counted itself the last one awake */
                claim_chain(min_partial);
                if (type != PA_T_NONE) break;
                claim_shade(min_partial);
                if (type != PA_T_NONE) break;
                if (last) {
                    /* the last wave awake found nothing whatsoever: the queues are final.  (Idle waves stay counted while
                     * they sleep, so the count is still PA_WAVES - 1 unless one has just been woken by new items.) */
                    if (lds_load(&S.idle) == PA_WAVES - 1u) S.done = 1u;
                    break;
                }
                /* count this wave idle; if that makes all of them, it is the last one awake: look once more, for anything */
                if (atomicAdd(&S.idle, 1u) + 1u == PA_WAVES) {
                    atomicSub(&S.idle, 1u);
                    continue;
                }
                /* sleep until somebody publishes items (or everything is over), then look again */
                uint32_t spins = 0u;
                while (lds_load(&S.gen) == gen && lds_load(&S.done) == 0u && lds_load(&S.abort) == 0u) {
                    __builtin_amdgcn_s_sleep(PA_SLEEP);
                    if (++spins > PA_SPIN_LIMIT) { /* cannot happen; a hang would cost a GPU, the fallback only a frame */
                        S.abort = 1u;
                        atomicExch(pp.global + PW_G_OVERFLOW, 1u);
                    }
                }
                atomicSub(&S.idle, 1u);
                break;
            }
        }
        type = (uint32_t)__builtin_amdgcn_readfirstlane((int)type);
        start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
        count = (uint32_t)__builtin_amdgcn_readfirstlane((int)count);
        const bool in_lds = (type & PA_IN_LDS) != 0u;
        const uint32_t lq = (type >> PA_LQ_SHIFT) & 3u; /* SHADE pages in LDS: the queue, i.e. the light */
        type &= 7u;
        const bool one_light = type == PA_T_SHADE && in_lds && (lq < PA_LQ - 1u || sc.n_lights <= PA_LQ);
#ifdef PA_STATS
        const unsigned long long ph_t1 = PA_TICK();
        if (type == PA_T_NONE) ph[24] += ph_t1 - ph_t0;
        const uint32_t ph_k = type == PA_T_NODE ? 0u : (type == PA_T_REFR ? 1u : (type == PA_T_TILE ? 2u : 3u));
#endif
        if (type == PA_T_NONE) continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* the page's items were written before they were counted */
        /* the dependent chains issue ahead of fresh tiles, and those ahead of the filler (1.41 -> 1.38 ms) */
        if (type == PA_T_SHADE) __builtin_amdgcn_s_setprio(0);
        else if (type == PA_T_TILE) __builtin_amdgcn_s_setprio(PA_PRIO_TILE);
        else __builtin_amdgcn_s_setprio(PA_PRIO_CHAIN);
#ifdef PA_STATS
        st_hist[count <= 8u ? 0 : (count <= 16u ? 1 : (count <= 32u ? 2 : (count < 64u ? 3 : 4)))] += 1u;
#endif

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
        bool from_tile = false;

        if (type == PA_T_TILE) {
            /* main.rs:1093-1100: the tile's primary rays are root nodes (depth max_depth, contribution 1.0), cast right away */
            const uint32_t tile = start;
            uint32_t base = 0u, entry = 0u;
            if (lane == 0u) {
                base = atomicAdd(&S.root_alloc, count);
                entry = atomicAdd(&S.tile_list_count, 1u);
            }
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry);
            if (base + count + lds_load(&S.n.alloc) > pp.node_cap || entry >= tile_cap) { /* the reserve makes this unreachable; be safe */
                if (lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
                break;
            }
            const uint32_t first_id = pp.node_cap - base - count; /* roots fill the arena from the top */
            if (lane == 0u) { tile_list[entry * 2u] = tile; tile_list[entry * 2u + 1u] = first_id; }
            id = first_id + lane;
            depth = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0);
            if (active) {
                uint32_t row, col;
                pw_slot_to_pixel(fr, tile * 64u + lane, &row, &col);
                /* Camera::shoot (main.rs:84-99), per-frame basis hoisted to the host */
                const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                const float clip_y = (fr.half_height - (float)y) / fr.height_f;
                const float clip_x = ((float)x - fr.half_width) / fr.height_f;
                const V3 cx = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
                const V3 cy = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
                const V3 ct = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
                req.o = v3(fr.cam_origin[0], fr.cam_origin[1], fr.cam_origin[2]);
                req.d = normalize(clip_x * cx + clip_y * cy + ct);
            }
            type = PA_T_NODE;
            from_tile = true;
        } else if (type == PA_T_NODE) {
            uint4 a, b;
            a = b = make_uint4(0u, 0u, 0u, 0u);
            if (in_lds) {
                if (active) {
                    const uint4 *t = lds_node + pa_entry((start + lane) & (PA_LDS_N_PAGES * 64u - 1u), 3u);
                    a = t[PA_F(0u)], b = t[PA_F(1u)];
                    id = t[PA_F(2u)].x;
                }
                pa_release_page(&S.ln_released, start >> 6, lane);
            } else {
                id = start + lane;
                /* a sealed page: the positions after its last item are node ids nobody owns; the fold walks all ids below
                 * n.alloc, so they must not look like nodes (level 0 is never folded) */
                if (!active && id < pp.node_cap) fold_level[id] = 0u;
                if (active) a = node_in[pa_entry(id, 2u)], b = node_in[pa_entry(id, 2u) + PA_F(1u)];
            }
            if (active) {
                req.o = v3(puf(a.x), puf(a.y), puf(a.z));
                req.d = v3(puf(a.w), puf(b.x), puf(b.y));
                req.mode = (b.z >> PW_MODE_SHIFT) & 3u;
                depth = (b.z >> PW_DEPTH_SHIFT) & 63u;
                req.excl = b.z & PW_EXCL_MASK;
                contribution = puf(b.w);
            }
        } else if (type == PA_T_REFR) {
            uint4 a, b, c;
            a = b = c = make_uint4(0u, 0u, 0u, 0u);
            if (in_lds) {
                if (active) {
                    const uint4 *t = lds_refr + pa_entry((start + lane) & (PA_LDS_F_PAGES * 64u - 1u), 3u);
                    a = t[PA_F(0u)], b = t[PA_F(1u)], c = t[PA_F(2u)];
                }
                pa_release_page(&S.lf_released, start >> 6, lane);
            } else if (active) {
                const uint4 *t = refr_q + pa_entry((start + lane) & ring_mask, 3u);
                a = t[PA_F(0u)], b = t[PA_F(1u)], c = t[PA_F(2u)];
            }
            if (active) {
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
        } else { /* PA_T_SHADE */
            uint4 a, b, c, d, e;
            a = b = c = d = e = make_uint4(0u, 0u, 0u, 0u);
            if (in_lds) {
                if (active) {
                    const uint4 *t = lds_shade + lq * (PA_LDS_PAGES * PA_SHADE_U4 * 64u) + pa_entry((start + lane) & (PA_LDS_PAGES * 64u - 1u), PA_SHADE_U4);
                    a = t[PA_F(0u)], b = t[PA_F(1u)], c = t[PA_F(2u)], d = t[PA_F(3u)], e = t[PA_F(4u)];
                }
                pa_release_page(&S.l_released[lq], start >> 6, lane);
            } else if (active) {
                const uint4 *t = shade_q + pa_entry((start + lane) & ring_mask, PA_SHADE_U4);
                a = t[PA_F(0u)], b = t[PA_F(1u)], c = t[PA_F(2u)], d = t[PA_F(3u)], e = t[PA_F(4u)];
            }
            if (active) {
                id = a.x; prim = a.y; obj = a.z & 0xffffu; light_i = (a.z >> 16) & 0x3fffu; sflags = a.z >> 30; /* 2: depth 0, 1: nothing below the node */
                spos = v3(puf(b.x), puf(b.y), puf(b.z)); sum.x = puf(b.w);
                adj_n = v3(puf(c.x), puf(c.y), puf(c.z)); sum.y = puf(c.w);
                in_dir = v3(puf(d.x), puf(d.y), puf(d.z)); sum.z = puf(d.w);
                sdiffuse = v3(puf(e.x), puf(e.y), puf(e.z));
                /* always true for a queued item: its light asks for a cast */
                if (one_light) do_cast = approximate_into_directional(uniform_ref(sc.lights + lq), spos, &dl);
                else do_cast = next_shadow_ray(sc, &light_i, spos, adj_n, &dl);
                req.o = spos;
                req.d = -dl.direction;
                req.mode = FACE_BACK;
                req.excl = pack_excl(prim, FACE_BACK);
            }
        }

        /* ---- the cast: the one place the intersection loop is instantiated ---- */
#ifdef PA_STATS
        const unsigned long long ph_t2 = PA_TICK();
#endif
        CastResult cr;
        cr.prim = -1;
        cr.t = 0.0f;
        cr.bf = 0u;
        cr.a0 = cr.a1 = cr.a2 = 0.0f;
        if constexpr (BFS) {
            cr = cast_bfs(sc, req, do_cast, bfs_lds, bfs_ws); /* all lanes: those without a ray help */
            if (do_cast) casts += 1u;
        } else if (do_cast) {
            cr = cast_asm(sc, req);
            casts += 1u;
        }
#ifdef PA_STATS
        const unsigned long long ph_t3 = PA_TICK();
#endif

        if (type == PA_T_NODE) {
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
            }
            /* The chain first: the record (REFR items refer to it, and its children are written into it below), the reflection
             * child and the ray into the glass are written and published BEFORE get_shade's preparation below (a normal-map
             * sincos, cgmath's from_arc, a spot light's acos and powf in binary64: about half of this block) — every
             * microsecond here is on the frame's critical path, the SHADE item is not. */
            bool refl_in_lds, overflow;
            const uint32_t q_refl = pa_try_append(&S.ln, &S.ln_released, PA_LDS_N_PAGES * 64u, want_refl, &refl_in_lds); /* queue position */
            const uint32_t k_refl = refl_in_lds ? pa_top_node(&S.root_alloc, &S.n.alloc, pp.node_cap, want_refl, &overflow) /* node id */
                                                : lds_append(&S.n.alloc, want_refl);
            if (!refl_in_lds) overflow = want_refl && k_refl + lds_load(&S.root_alloc) >= pp.node_cap;
            if (want_refl && !overflow) rec_cr = k_refl;
            /* A node with nothing below it — a miss, a hit at the depth limit, a hit whose material asks for no reflection and no
             * refraction — is complete once its shade term is: whoever writes that term writes the node's VALUE,
             * (term + black * rc) + black * fc for the third kind (main.rs:516-518 with both children black), and the node has no
             * second field and is never folded.  Half of all nodes: 16 bytes less written and read each, and a fold less. */
            const bool below = active && cr.prim >= 0 && depth > 0u && (want_refl || want_refr);
            const bool bare = active && cr.prim >= 0 && depth > 0u && !below;
            if (active) {
                if (below) nodes[pa_entry(id, 2u) + PA_F(1u)] = make_uint4(pfu(fc), 0u, rec_cr, rec_cf);
                /* the level the node is folded at; roots are folded by tile: 0xff marks one that is complete as it stands */
                fold_level[id] = (unsigned char)(below ? (from_tile ? 0u : depth) : (from_tile ? 0xffu : 0u));
            }
            /* reflection child (get_reflect, main.rs:328-341) */
            if (want_refl && !overflow) {
                const V3 d = reflect_dir(nh.normal, req.d);
                const uint32_t word = pack_excl(nh.prim, nh.bf ? FACE_FRONT : FACE_BACK) | (req.mode << PW_MODE_SHIFT) | ((depth - 1u) << PW_DEPTH_SHIFT);
                if (refl_in_lds) pa_store_node(lds_node + pa_entry(q_refl & (PA_LDS_N_PAGES * 64u - 1u), 3u), nh.pos, d, word, contribution * rc, k_refl);
                else pa_store_node(node_in + pa_entry(k_refl, 2u), nh.pos, d, word, contribution * rc, 0xffffffffu);
            }
            /* the ray into the glass (main.rs:358-366) */
            bool refr_in_lds;
            uint32_t k_refr = pa_try_append(&S.lf, &S.lf_released, PA_LDS_F_PAGES * 64u, want_refr, &refr_in_lds);
            if (!refr_in_lds) k_refr = lds_append(&S.f.alloc, want_refr);
            if (want_refr) {
                const uint32_t word = pack_excl(nh.prim, FACE_FRONT) | (FACE_BACK << PW_MODE_SHIFT) | ((depth - 1u) << PW_DEPTH_SHIFT);
                if (refr_in_lds) pa_store_refr(lds_refr + pa_entry(k_refr & (PA_LDS_F_PAGES * 64u - 1u), 3u), nh.pos, inside_d, word, id, nh.obj, contribution * fc, 0.0f, 0xffffffffu);
                else pa_store_refr(refr_q + pa_entry(k_refr & ring_mask, 3u), nh.pos, inside_d, word, id, nh.obj, contribution * fc, 0.0f, 0xffffffffu);
            }
            const bool any_overflow = __builtin_amdgcn_ballot_w64(overflow) != 0ull; /* then the frame is abandoned: nothing to count in */
            if (refl_in_lds) pa_publish(S.ready_ln, PA_LDS_N_PAGES - 1u, want_refl && !any_overflow, q_refl, &S.gen);
            else pa_publish<PACKED>(ready_n, 0xffffffffu, want_refl && !any_overflow, k_refl, &S.gen);
            if (refr_in_lds) pa_publish(S.ready_lf, PA_LDS_F_PAGES - 1u, want_refr, k_refr, &S.gen);
            else pa_publish<PACKED>(ready_f, ring_page_mask, want_refr, k_refr, &S.gen);
            if (any_overflow && lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
            /* get_shade up to its first shadow cast (main.rs:407-433) */
            if (want_shade) {
                const rt_material &rm = sc.materials[nh.obj];
                const Mat m = material_approx(rm, nh.u, nh.v);
                adj_n = adjust_normal(m.normal, nh.normal); /* main.rs:410 */
                sdiffuse = m.diffuse;
                light_i = 0u;
            }
            want_shade = next_shadow_ray_in_step(sc, 0u, want_shade, &light_i, nh.pos, adj_n); /* no light needs a cast: get_shade = black */
            /* the record's first field (shade term, rc): written here unless a SHADE item will, with the term filled in */
            if (active && !want_shade) {
                if (bare) { const V3 black = v3(0.0f, 0.0f, 0.0f); acc = (acc + black * rc) + black * fc; }
                nodes[pa_entry(id, 2u)] = make_uint4(pfu(acc.x), pfu(acc.y), pfu(acc.z), pfu(rc));
            }
            queue_shade(want_shade, id, nh.prim, nh.obj | (light_i << 16) | (depth > 0u ? (bare ? 0x40000000u : 0u) : 0x80000000u), light_i, nh.pos, adj_n, req.d,
                        sdiffuse, v3(0.0f, 0.0f, 0.0f));
        } else if (type == PA_T_REFR) {
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
                }
            }
            bool again_in_lds;
            uint32_t k_again = pa_try_append(&S.lf, &S.lf_released, PA_LDS_F_PAGES * 64u, requeue, &again_in_lds);
            if (!again_in_lds) k_again = lds_append(&S.f.alloc, requeue);
            if (requeue) {
                const uint32_t word = req.excl | (req.mode << PW_MODE_SHIFT) | (depth << PW_DEPTH_SHIFT);
                if (again_in_lds) pa_store_refr(lds_refr + pa_entry(k_again & (PA_LDS_F_PAGES * 64u - 1u), 3u), req.o, req.d, word, id, obj, contribution, travel, (uint32_t)retry);
                else pa_store_refr(refr_q + pa_entry(k_again & ring_mask, 3u), req.o, req.d, word, id, obj, contribution, travel, (uint32_t)retry);
            }
            bool child_in_lds, overflow;
            const uint32_t q_child = pa_try_append(&S.ln, &S.ln_released, PA_LDS_N_PAGES * 64u, escape, &child_in_lds);
            const uint32_t k_child = child_in_lds ? pa_top_node(&S.root_alloc, &S.n.alloc, pp.node_cap, escape, &overflow) : lds_append(&S.n.alloc, escape);
            if (!child_in_lds) overflow = escape && k_child + lds_load(&S.root_alloc) >= pp.node_cap;
            if (escape && !overflow) {
                const uint32_t word = esc_excl | (FACE_FRONT << PW_MODE_SHIFT) | (depth << PW_DEPTH_SHIFT);
                if (child_in_lds) pa_store_node(lds_node + pa_entry(q_child & (PA_LDS_N_PAGES * 64u - 1u), 3u), esc_o, esc_d, word, contribution, k_child);
                else pa_store_node(node_in + pa_entry(k_child, 2u), esc_o, esc_d, word, contribution, 0xffffffffu);
                reinterpret_cast<uint32_t *>(nodes + pa_entry(id, 2u) + PA_F(1u))[3] = k_child; /* record word 7: the refraction child */
            }
            const bool any_overflow = __builtin_amdgcn_ballot_w64(overflow) != 0ull;
            if (again_in_lds) pa_publish(S.ready_lf, PA_LDS_F_PAGES - 1u, requeue, k_again, &S.gen);
            else pa_publish<PACKED>(ready_f, ring_page_mask, requeue, k_again, &S.gen);
            if (child_in_lds) pa_publish(S.ready_ln, PA_LDS_N_PAGES - 1u, escape && !any_overflow, q_child, &S.gen);
            else pa_publish<PACKED>(ready_n, 0xffffffffu, escape && !any_overflow, k_child, &S.gen);
            if (any_overflow && lane == 0u) { S.abort = 1u; atomicExch(pp.global + PW_G_OVERFLOW, 1u); }
            /* the decay (a powf in binary64) is only read by the fold: after the child is on its way */
            if (escape && !overflow) {
                decay = rtdm::powf(sc.materials[obj].opaque_decay, travel); /* main.rs:508 */
                reinterpret_cast<uint32_t *>(nodes + pa_entry(id, 2u) + PA_F(1u))[1] = pfu(decay); /* record word 5 */
            }
        } else {
            /* ---- one light of get_shade (main.rs:435-461) ---- */
            bool again = false;
            if (active) {
                const rt_material &rm = sc.materials[obj];
                if (do_cast) {
                    bool lit = true;
                    if (cr.prim >= 0) {
                        auto occluded = [&](const auto &L) {
                            const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                            if (!has_origin) return true;
                            const V3 occ = req.o + req.d * cr.t;
                            const float occlusion_distance = distance(spos, occ);
                            const float light_distance = distance(spos, v3(L.origin[0], L.origin[1], L.origin[2]));
                            return occlusion_distance < light_distance;
                        };
                        lit = one_light ? !occluded(uniform_ref(sc.lights + lq)) : !occluded(sc.lights[light_i]);
                    }
#ifdef PA_STATS
                    st_shadow[0] += 1u; /* shadow casts: all, with a hit, occluded (per lane; added up when the wave leaves) */
                    if (cr.prim >= 0) st_shadow[1] += 1u;
                    if (!lit) st_shadow[2] += 1u;
#endif
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
                    if (!one_light) again = next_shadow_ray(sc, &light_i, spos, adj_n, &dl);
                }
            }
            if (one_light) again = next_shadow_ray_in_step(sc, lq + 1u, active && do_cast, &light_i, spos, adj_n);
            if (active) {
                const rt_material &rm = sc.materials[obj];
                if (!again) {
                    V3 acc = sum; /* depth 0: the unscaled shade (main.rs:488-490) */
                    float rc = 0.0f;
                    if ((sflags & 2u) == 0u) {
                        const float shade_contribution = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                        acc = sum * shade_contribution;
                        rc = rm.shiness * (1.0f - rm.transparency); /* main.rs:493, as the node's own step has it */
                        if ((sflags & 1u) != 0u) { /* nothing below the node: its value (main.rs:516-518 with both children black) */
                            const V3 black = v3(0.0f, 0.0f, 0.0f);
                            acc = (acc + black * rc) + black * rm.transparency;
                        }
                    }
                    nodes[pa_entry(id, 2u)] = make_uint4(pfu(acc.x), pfu(acc.y), pfu(acc.z), pfu(rc));
                }
            }
            queue_shade(again, id, prim, obj | (light_i << 16) | (sflags << 30), light_i, spos, adj_n, in_dir, sdiffuse, sum);
        }
#ifdef PA_STATS
        {
            const unsigned long long ph_t4 = PA_TICK();
            ph[0 + ph_k] += ph_t1 - ph_t0;
            ph[4 + ph_k] += ph_t2 - ph_t1;
            ph[8 + ph_k] += ph_t3 - ph_t2;
            ph[12 + ph_k] += ph_t4 - ph_t3;
            ph[16 + ph_k] += 1ull;
            ph[20 + ph_k] += count;
        }
#endif
    }

#ifdef PA_STATS /* diagnostic build: per-workgroup main-loop and fold times (100 MHz ticks) into the global words */
    const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- fold the records bottom-up (main.rs:516-518) and write the pixels ---- */
    __syncthreads();
#ifdef PA_STATS
    const unsigned long long st_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    const bool aborted = S.abort != 0u;
    const uint32_t n_nodes = S.n.alloc < pp.node_cap ? S.n.alloc : pp.node_cap; /* the queued nodes; roots sit at the top */
    const uint32_t max_depth = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0);
    /* the nodes at the top of the arena: the roots (level 0: folded by tile) and the nodes whose items were queued in LDS */
    const uint32_t n_top = S.root_alloc < pp.node_cap - n_nodes ? S.root_alloc : pp.node_cap - n_nodes;
    const uint32_t top_first = pp.node_cap - n_top;
    if (!aborted) {
        /* levels 1 .. max_depth-1 (children before parents; level max_depth are the roots, folded by tile below) */
        auto fold_node = [&](uint32_t id) {
            const uint4 a = nodes[pa_entry(id, 2u)], b = nodes[pa_entry(id, 2u) + PA_F(1u)];
            const float rc = puf(a.w), fc = puf(b.x), decay = puf(b.y);
            V3 reflection = v3(0.0f, 0.0f, 0.0f), refraction = v3(0.0f, 0.0f, 0.0f);
            if (b.z != PW_NO_CHILD) {
                const uint4 c = nodes[pa_entry(b.z, 2u)];
                reflection = v3(puf(c.x), puf(c.y), puf(c.z));
            }
            if (b.w != PW_NO_CHILD) {
                const uint4 c = nodes[pa_entry(b.w, 2u)];
                refraction = v3(puf(c.x), puf(c.y), puf(c.z)) * decay; /* main.rs:508 */
            }
            const V3 value = (v3(puf(a.x), puf(a.y), puf(a.z)) + reflection * rc) + refraction * fc;
            float *rec = reinterpret_cast<float *>(nodes + pa_entry(id, 2u));
            rec[0] = value.x;
            rec[1] = value.y;
            rec[2] = value.z;
        };
        /* The nodes to fold are listed level by level first — in the LDS that held the SHADE queues, which are empty now — by a
         * counting sort over the level bytes: two passes over them instead of one per level, and a level's nodes dealt evenly
         * to the 512 threads, one or two each, all their loads in flight together (the fold is the serial end of a workgroup's
         * frame: 73 -> ~40 us of a 1.04 ms frame, profiles/r03_ab9.txt).  More nodes than the list holds: the scan per level. */
        uint32_t *const f_list = reinterpret_cast<uint32_t *>(lds_shade);
        uint32_t *const f_cursor = reinterpret_cast<uint32_t *>(lds_refr); /* [64]: a level's count, then where its next node goes */
        uint32_t *const f_start = f_cursor + 64u;                           /* [65] */
        const uint32_t f_cap = PA_LQ * PA_LDS_PAGES * PA_SHADE_U4 * 64u * 4u;
        const uint32_t n_all = n_nodes + n_top;
        static_assert(PA_LDS_F_PAGES * 3u * 64u * 4u >= 129u, "the fold's counters live where the REFR queue was");
        if (threadIdx.x < 64u) f_cursor[threadIdx.x] = 0u;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            for (uint32_t v0 = 0; v0 < n_all; v0 += PA_THREADS) { /* every thread takes part in every trip: wave-wide ballots below */
                const uint32_t v = v0 + threadIdx.x;
                const uint32_t id = v < n_nodes ? v : top_first + (v - n_nodes);
                uint32_t lv = v < n_all ? (uint32_t)fold_level[id] : 0u;
                if (lv >= max_depth) lv = 0u; /* 0xff: a complete root */
                unsigned long long todo = __builtin_amdgcn_ballot_w64(lv != 0u);
                while (todo != 0ull) { /* one LDS atomic per level present in the wave */
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)__builtin_ctzll(todo));
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(lv == l);
                    uint32_t base = 0u;
                    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&f_cursor[l], (uint32_t)__builtin_popcountll(m));
                    if (pass == 1 && lv == l) {
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
                        const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (at < f_cap) f_list[at] = id;
                    }
                    todo &= ~m;
                }
            }
            __syncthreads();
            if (pass == 0) {
                if (threadIdx.x == 0u) { /* counts -> starts = cursors */
                    uint32_t sum = 0u; /* levels 1 .. max_depth - 1 have nodes (max_depth <= 63: six bits of a queued ray's word) */
                    for (uint32_t l = 0; l <= max_depth && l < 64u; ++l) { const uint32_t c = f_cursor[l]; f_start[l] = sum; f_cursor[l] = sum; sum += c; }
                    f_start[64] = sum;
                }
                __syncthreads();
                if (f_start[64] > f_cap) break; /* wave-uniform: LDS */
            }
        }
        const bool listed = f_start[64] <= f_cap;
#ifdef PA_DIAG_NO_LEVELS /* timing experiment: the wrong image */
        for (uint32_t left = max_depth; left < max_depth; ++left) {
#else
        for (uint32_t left = 1u; left < max_depth; ++left) {
#endif
            if (listed) {
                const uint32_t end = f_start[left + 1u];
                for (uint32_t i = f_start[left] + threadIdx.x; i < end; i += PA_THREADS) fold_node(f_list[i]);
            } else {
                for (uint32_t first = threadIdx.x; first < n_all; first += PA_THREADS * 16u) {
                    uint32_t lv[16];
#pragma unroll
                    for (uint32_t k = 0; k < 16u; ++k) {
                        const uint32_t v = first + k * PA_THREADS;
                        lv[k] = v < n_all ? (uint32_t)fold_level[v < n_nodes ? v : top_first + (v - n_nodes)] : 0u;
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 16u; ++k) {
                        if (lv[k] != left) continue;
                        const uint32_t v = first + k * PA_THREADS;
                        fold_node(v < n_nodes ? v : top_first + (v - n_nodes));
                    }
                }
            }
            __syncthreads();
        }
        /* the roots, by tile: a wave takes four tiles at a time so that their dependent loads (record, then children)
         * overlap */
        const uint32_t n_started = S.tile_list_count;
#ifdef PA_DIAG_NO_ROOTS /* timing experiment: no image */
        for (uint32_t e0 = n_started; e0 < n_started; e0 += PA_WAVES * 4u) {
#else
        for (uint32_t e0 = threadIdx.x >> 6; e0 < n_started; e0 += PA_WAVES * 4u) {
#endif
            uint32_t id[4], slot[4], tile_of[4];
            uint4 ra[4], rb[4];
            bool live[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t e = e0 + j * PA_WAVES;
                live[j] = false;
                id[j] = slot[j] = 0u;
                tile_of[j] = 0xffffffffu;
                if (e < n_started) {
                    const uint32_t tile = tile_list[e * 2u], base = tile_list[e * 2u + 1u];
                    tile_of[j] = tile;
                    const uint32_t first_slot = tile * 64u;
                    const uint32_t nv = total_slots - first_slot < 64u ? total_slots - first_slot : 64u;
                    live[j] = lane < nv;
                    id[j] = base + lane;
                    slot[j] = first_slot + lane;
                }
                if (live[j]) {
                    ra[j] = nodes[pa_entry(id[j], 2u)];
                    rb[j] = make_uint4(0u, 0u, PW_FINAL, PW_NO_CHILD);
                    if (fold_level[id[j]] != 0xffu) rb[j] = nodes[pa_entry(id[j], 2u) + PA_F(1u)]; /* else complete as it stands: no second field */
                }
            }
            if (pp.tile_cost != nullptr) { /* what the tile cost, roughly: how many of its pixels recursed */
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const unsigned long long deep = __builtin_amdgcn_ballot_w64(live[j] && rb[j].z != PW_FINAL);
                    if (lane == 0u && tile_of[j] != 0xffffffffu) pp.tile_cost[tile_of[j]] = (uint32_t)__builtin_popcountll(deep);
                }
            }
            uint4 cr4[4], cf4[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                cr4[j] = cf4[j] = make_uint4(0u, 0u, 0u, 0u);
                if (live[j] && rb[j].z != PW_FINAL) {
                    if (rb[j].z != PW_NO_CHILD) cr4[j] = nodes[pa_entry(rb[j].z, 2u)];
                    if (rb[j].w != PW_NO_CHILD) cf4[j] = nodes[pa_entry(rb[j].w, 2u)];
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (!live[j]) continue;
                const uint4 a = ra[j], b = rb[j];
                V3 value = v3(puf(a.x), puf(a.y), puf(a.z));
                if (b.z != PW_FINAL) {
                    const float rc = puf(a.w), fc = puf(b.x), decay = puf(b.y);
                    V3 reflection = v3(0.0f, 0.0f, 0.0f), refraction = v3(0.0f, 0.0f, 0.0f);
                    if (b.z != PW_NO_CHILD) reflection = v3(puf(cr4[j].x), puf(cr4[j].y), puf(cr4[j].z));
                    if (b.w != PW_NO_CHILD) refraction = v3(puf(cf4[j].x), puf(cf4[j].y), puf(cf4[j].z)) * decay; /* main.rs:508 */
                    value = (value + reflection * rc) + refraction * fc;
                }
                uint32_t row, col;
                pw_slot_to_pixel(fr, slot[j], &row, &col);
                /* img[at] = img[at] + photon on a zeroed image (main.rs:1107) */
                float *px = out + ((size_t)row * fr.cols + col) * 3u;
                px[0] = 0.0f + value.x;
                px[1] = 0.0f + value.y;
                px[2] = 0.0f + value.z;
            }
        }
        if (threadIdx.x == 0u && n_started != 0u) atomicAdd(pp.global + PW_G_TILES_DONE, n_started);
    }
#ifdef PA_STATS
    __syncthreads();
    {
        const unsigned long long st_t3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0u) { atomicAdd(pp.global + 8, (uint32_t)(st_t1 - st_t0)); atomicMax(pp.global + 9, (uint32_t)(st_t1 - st_t0)); } /* per wave: own loop */
        if (lane == 0u) for (int k = 0; k < 5; ++k) atomicAdd(pp.global + 20 + k, st_hist[k]);
        for (int k = 0; k < 3; ++k) {
            uint32_t v = st_shadow[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0u && v != 0u) atomicAdd(pp.global + (k == 0 ? 18 : (k == 1 ? 19 : 31)), v);
        }
        if (lane == 0u) {
            ph[25] = st_t3 - st_t2;
            for (int k = 0; k < 26; ++k) atomicAdd(&pa_phase_stats[k], ph[k]);
        }
        if (threadIdx.x == 0u) {
            atomicAdd(pp.global + 10, (uint32_t)(st_t2 - st_t0)); atomicMax(pp.global + 11, (uint32_t)(st_t2 - st_t0)); /* until the last wave left the loop */
            atomicAdd(pp.global + 12, (uint32_t)(st_t3 - st_t2)); atomicMax(pp.global + 13, (uint32_t)(st_t3 - st_t2)); /* fold */
            atomicAdd(pp.global + 14, 1u); atomicAdd(pp.global + 15, S.n.alloc);
            atomicAdd(pp.global + 25, S.s.alloc); atomicAdd(pp.global + 26, S.f.alloc); atomicAdd(pp.global + 27, S.root_alloc); /* items that went through the arena's rings; nodes at the top */
            atomicMin(pp.global + 16, (uint32_t)(st_t3 - st_t0)); atomicMax(pp.global + 17, (uint32_t)(st_t3 - st_t0));
        }
    }
#endif
    /* the workgroup's casts: summed in LDS (in `gen`, which nobody reads any more), one global add per workgroup — by the thread
     * whose fence and count of workgroups done follow it, so the workgroup that closes the frame reads a complete sum */
    if (threadIdx.x == 0u) S.gen = 0u;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) casts += __shfl_down(casts, off, 64);
    if (lane == 0u && casts != 0u) atomicAdd(&S.gen, casts);

    /* ---- the last workgroup to leave closes the frame (no launch of its own for that): every tile rendered and no arena
     * overflow -> publish the cast count; else raise the flag the trailing per-pixel launch looks at.  And it zeroes the
     * block of global words the NEXT launch on this workspace will use (this launch's block stays as it is until then: the
     * per-pixel launch reads the flag from it). ---- */
    __syncthreads();
    if (threadIdx.x == 0u) {
        if (S.gen != 0u) atomicAdd(reinterpret_cast<unsigned long long *>(pp.global + PW_G_CASTS), (unsigned long long)S.gen);
        __threadfence();
        if (atomicAdd(pp.global + PW_G_GROUPS_DONE, 1u) + 1u == gridDim.x) {
            __threadfence();
            uint32_t overflow = __hip_atomic_load(pp.global + PW_G_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tiles_done = __hip_atomic_load(pp.global + PW_G_TILES_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (overflow == 0u && tiles_done != n_tiles) {
                overflow = 1u;
                atomicExch(pp.global + PW_G_OVERFLOW, 1u);
            }
            if (overflow == 0u && pp.ray_count != nullptr)
                atomicAdd(pp.ray_count, /* atomic: a caller may hand one counter to renders on several streams (include/rt_amd.h: "added to") */
                          __hip_atomic_load(reinterpret_cast<unsigned long long *>(pp.global + PW_G_CASTS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            for (uint32_t k = 0; k < PW_G_BLOCK_WORDS; ++k) pp.global_next[k] = 0u;
        }
    }
}

static size_t pwf_dynamic_lds(uint32_t node_cap, uint32_t ring_cap) {
    const bool packed = pa_ready_packed(node_cap, ring_cap);
    return (size_t)(PA_READY_WORDS((node_cap + 63u) / 64u, packed) + 2u * PA_READY_WORDS(ring_cap / 64u, packed)) * sizeof(uint32_t);
}

int pwf_workgroups_per_cu(uint32_t node_cap, uint32_t ring_cap, bool bfs_walk) {
    int n = 0;
    const size_t lds = pwf_dynamic_lds(node_cap, ring_cap);
    const bool packed = pa_ready_packed(node_cap, ring_cap);
    const hipError_t e = bfs_walk ? (packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwf_kernel<true, true>, (int)PA_THREADS, lds)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwf_kernel<false, true>, (int)PA_THREADS, lds))
                                  : (packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwf_kernel<true, false>, (int)PA_THREADS, lds)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pwf_kernel<false, false>, (int)PA_THREADS, lds));
    if (e != hipSuccess || n < 1) n = 1;
    return n;
}

__global__ void pwf_init_kernel(uint32_t *global, KernelFrame *frame, const KernelFrame fr) {
    if (threadIdx.x < PW_G_BLOCK_WORDS) global[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) *frame = fr;
}

size_t pwf_arena_bytes(uint32_t node_cap, uint32_t ring_cap) {
    /* node inputs + records, the two rings, the tile list, one "folds at level" byte per node (barrier-free kernel) */
    return ((size_t)node_cap * 4u + (size_t)ring_cap * (3u + PA_SHADE_U4)) * sizeof(uint4) + (size_t)(node_cap / 64u) * 2u * sizeof(uint32_t) +
           (((size_t)node_cap + 15u) & ~(size_t)15u) + 256u;
}

#ifdef PA_STATS
static uint32_t *g_pw_last_global = nullptr;
extern "C" int rt_diag_read_pwf(uint32_t *out32) {
    if (!g_pw_last_global) return -1;
    return hipMemcpy(out32, g_pw_last_global, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
/* wave time by phase; reset != 0 clears the counters after reading */
extern "C" int rt_diag_read_pwf_phases(unsigned long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(pa_phase_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pa_phase_stats), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t launch_pwf(const KernelScene &sc, KernelFrame fr, float *out, const PwParams &pp, uint32_t workgroups, hipStream_t stream,
                      bool init, bool first_band, bool last_band) {
    const uint32_t total = fr.cols * fr.rows;
    fr.n_chunks = (total + 63u) / 64u;
    if (total == 0u) return hipSuccess;
#ifdef PA_STATS
    g_pw_last_global = pp.global;
#endif
    if (init) hipLaunchKernelGGL(pwf_init_kernel, dim3(1), dim3(64), 0, stream, pp.global, const_cast<KernelFrame *>(pp.frame), fr);
    if (first_band) record_main_kernel_event(0, stream); /* the pair brackets all bands of a call (one, up to ~8 Mpixel) */
    const size_t lds = pwf_dynamic_lds(pp.node_cap, pp.ring_cap);
    const bool packed = pa_ready_packed(pp.node_cap, pp.ring_cap);
    if (sc.bfs_walk != 0u && pp.bfs_scratch != nullptr) {
        if (packed) hipLaunchKernelGGL((pwf_kernel<true, true>), dim3(workgroups), dim3(PA_THREADS), lds, stream, sc, pp, out);
        else hipLaunchKernelGGL((pwf_kernel<false, true>), dim3(workgroups), dim3(PA_THREADS), lds, stream, sc, pp, out);
    } else if (packed) hipLaunchKernelGGL((pwf_kernel<true, false>), dim3(workgroups), dim3(PA_THREADS), lds, stream, sc, pp, out);
    else hipLaunchKernelGGL((pwf_kernel<false, false>), dim3(workgroups), dim3(PA_THREADS), lds, stream, sc, pp, out);
    if (last_band) record_main_kernel_event(1, stream);
    return hipGetLastError();
}

} /* namespace rt */

#ifdef RT_DIAG_STAGES
RT_DIAG_STAGE_READER(rt_diag_read_stages_pwf)
#endif
#ifdef RT_DIAG_NEED
RT_DIAG_NEED_READER(rt_diag_read_need_pwf)
#endif
#ifdef RT_DIAG_BFS
RT_DIAG_BFS_READER(rt_diag_read_bfs)
#endif
