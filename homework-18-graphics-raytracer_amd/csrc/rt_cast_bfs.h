/*
 * rt_cast_bfs.h — World::cast (main.rs:183-324) for scenes beyond the caches: the node tree walked breadth-first, ray by ray.
 * Included by rt_cast.h (it ends in cast_finish and falls back on cast_asm); the kernels that have this walk are
 * rt_pwf.hip pwf_kernel<.., BFS> and rt_distributed.hip distributed_kernel<.., BFS>.
 */
#ifndef RT_CAST_BFS_H
#define RT_CAST_BFS_H

namespace rt {

/* ---- World::cast for scenes beyond the caches: the node tree walked BREADTH-FIRST, ray by ray (round 4) --------------------------
 * cast_asm visits, wave-uniformly and one record after the other, the UNION of the nodes and triangles its 64 rays need: at 147 484
 * triangles that is tens of thousands of dependent scalar fetches per wave-cast (485 ms for a 480 x 270 frame), and letting every
 * lane walk the pre-order array on its own only moves the chain of dependent fetches into the lanes (cast_lanes, round 4: 2-2.5x
 * slower at every size).  Here nothing waits for anything but its own level, and no lane runs a loop over records.  Everything is
 * a list of RECORDS (ray | count << 8 | flag << 16, first): one ray and up to 16 consecutive nodes or triangles; a pass takes FOUR
 * records and gives every lane one (ray, node) or (ray, triangle) pair — lane l has member l & 15 of record l >> 4.  Passes emit
 * records in lane order, so the records that follow each other in a list are one ray's sibling nodes or neighbouring leaves, and what
 * the walk reads of a node or a triangle lies in arrays of 16-byte pieces (KernelScene::bfs_soa): a load of a pass is one contiguous
 * kilobyte more often than not.
 *
 *   levels  the nodes are numbered in LEVEL ORDER (KernelScene::bfs_nodes): an inner node's children are a range of the same array, so
 *           "descend into this node" is one record.  Level 0: every active ray x the top-level nodes.  Per pair: three 16-byte pieces of
 *           the node (bounding sphere, counts, the first plane direction or the cone; a node with more plane directions has the others
 *           fetched when a lane gets that far), the ray from the wave's LDS table, cluster_skippable_lane's test.  A
 *           node the ray may hit: an inner node becomes a record of the next level, a LEAF a JOB — flagged BAND when the ray's line
 *           misses the leaf's sphere and the leaf was only kept because the ray is nearly parallel to one of its planes (one of
 *           SEVERAL: when the leaf has one plane direction all its triangles are nearly parallel to the ray, and it is a plain job).
 *   jobs    per pair: the triangle's plane and bounding sphere (two 16-byte loads per lane, issued a group of passes ahead, the
 *           records two groups), then the reference's single-triangle test (main.rs:184-224) in the reference's operation order: culling,
 *           exclusion, t, `t <= 0`, the conservative bounding-sphere rejection of rt_cast_asm.h — and, for the few pairs that get that
 *           far (the CANDIDATES, collected in LDS and taken 64 at a time), the whole test once more with the three signed areas.  An
 *           accepted pair does ONE ds_min_u64 of (bits(t) << 32 | ~index) on its ray's slot: among all accepted candidates of a cast
 *           the reference ends with the smallest t and, among equal t, the LAST index (main.rs:229-233) — as long as no accepted t is NaN.
 *   band jobs  (kept in a region of their own): the node rejection's argument is made triangle by triangle — the ray's line misses a
 *           sphere that contains the triangle's own, and the ray is not nearly parallel to THAT triangle's plane (|n . d| >= 1e-3,
 *           n . d being the very value the test computes) — so a pair of a band job only goes on to the test if its own plane is the
 *           nearly parallel one: one 16-byte load and one dot product for the others, and the few that go on join the candidates.
 *   NaN distances (a ray in a triangle's plane: 0 / 0) are the one case where "sequential" and "minimum" part ways; they are handled
 *   exactly by a second pass over the jobs (see there).  A list that overflows sends the wave through cast_asm.  Visiting a node that
 *   could have been skipped is always allowed (the skips are conservative), so the test per (ray, node) may be any subset of
 *   cluster_skippable_lane's.
 *
 * All 64 lanes must be executing (`active` says which have a ray).  One BfsLds and one BfsScratch per wave. */
/* Passes are taken in GROUPS and the groups are double-buffered in registers: while a group is computed the loads of the next one are
 * in flight and the records of the one after that are on their way — a wave alone on its SIMD (the tail of a frame, the one heavy
 * wave-cast of a tile of grazing rays) is otherwise one memory round trip per pass.  The kernel that uses this walk therefore gives
 * itself 256 VGPRs and a whole CU's LDS (rt_pwf.hip: one workgroup per CU). */
#ifndef RT_BFS_LEVEL_GROUP
#define RT_BFS_LEVEL_GROUP 4u /* passes per group of the level loop: 16 records */
#endif
#ifndef RT_BFS_PAIR_GROUP
#define RT_BFS_PAIR_GROUP 4u  /* of the job loop */
#endif
#ifndef RT_BFS_BAND_GROUP
#define RT_BFS_BAND_GROUP 8u  /* of the band-job loop: 32 records */
#endif
#define RT_BFS_CANDIDATES (64u + 64u * (RT_BFS_BAND_GROUP > RT_BFS_PAIR_GROUP ? RT_BFS_BAND_GROUP : RT_BFS_PAIR_GROUP)) /* a group may add 64 per pass; drained between groups down to 64 at most */
struct BfsLds {
    uint32_t nan_last[64];      /* per ray: 1 + the LAST triangle accepted with a NaN distance (0: none) */
    unsigned long long key[64]; /* per ray: the smallest (bits(t) << 32 | ~triangle) accepted so far */
    float4 ro[64];              /* origin, exclusion word */
    float4 rd[64];              /* direction, flags (mode | filter_ok << 2) */
    uint32_t cand[RT_BFS_CANDIDATES]; /* pairs that got as far as the signed areas: ray | triangle << 6 */
};
struct BfsScratch {
    uint2 *items_a, *items_b; /* the records of a level and of the next: items_cap each */
    uint2 *jobs;              /* jobs_cap records: the jobs from the front, the band jobs from the back */
    uint32_t items_cap, jobs_cap;
};
#define RT_BFS_BAND 0x10000u
#ifdef RT_DIAG_BFS /* diagnostic build (tools/diag_bfs.py): [0] wave-casts, [1] sent to cast_asm because a list overflowed, [2] rays with a NaN distance (second pass over the jobs),
                    * [3] (ray, node) pairs tested, [4] jobs, [5] most records in one level, [6] levels, [7] (ray, triangle) pairs of jobs, [8] band jobs, [9] pairs of band jobs,
                    * [10] pairs that reached the signed areas, [11] the longest walk of one wave-cast (cycles) */
#define RT_DIAG_BFS_WORDS 12
static __device__ unsigned long long g_bfs_stats[RT_DIAG_BFS_WORDS];
static __device__ unsigned long long g_bfs_ticks[16]; /* wave cycles: [0] the levels, [1] the jobs, [2] the band jobs, [3] cast_finish; waiting for memory at the top of a group / the rest of it: [4] [5] levels, [6] [7] jobs, [8] [9] band jobs */
#define RT_DIAG_BFS_READER(name)                                                                                \
    extern "C" int name(unsigned long long *out12, int reset) {                                                 \
        if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(rt::g_bfs_stats), RT_DIAG_BFS_WORDS * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[RT_DIAG_BFS_WORDS] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_bfs_stats), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }                                                                                                           \
    extern "C" int name##_ticks(unsigned long long *out4, int reset) {                                          \
        if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(rt::g_bfs_ticks), 16 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_bfs_ticks), z, sizeof z) != hipSuccess) return -1; } \
        return 0;                                                                                               \
    }
#define RT_BFS_STAMP_WAIT(acc) { const unsigned long long s0_ = __builtin_readcyclecounter(); __builtin_amdgcn_s_waitcnt(0x0F70); (acc) += __builtin_readcyclecounter() - s0_; }
#else
#define RT_BFS_STAMP_WAIT(acc)
#endif
__device__ __forceinline__ uint32_t bfs_rank(unsigned long long mask) { /* how many set bits of a ballot lie below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
/* a record from a list in global memory — said so: as a generic pointer the compiler makes it a FLAT load, which counts as an LDS
 * operation too, and every wait for the wave's LDS tables would then wait for this prefetch */
typedef uint32_t bfs_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 bfs_load_record(const uint2 *p) {
    const bfs_u32x2 v = *reinterpret_cast<const __attribute__((address_space(1))) bfs_u32x2 *>(reinterpret_cast<uintptr_t>(p));
    return make_uint2(v.x, v.y);
}
/* a group's records sit one per lane (lanes 0 .. 4 G - 1); pass q takes records 4 q .. 4 q + 3, lane l the (l >> 4)-th of them */
__device__ __forceinline__ uint2 bfs_record_of(const uint2 group, const uint32_t q, const uint32_t rec_lane) {
    const int from = (int)((4u * q + rec_lane) << 2);
    return make_uint2((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)group.x), (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)group.y));
}
/* The single-triangle test from its start (main.rs:184-224: culling, exclusion, t, `t <= 0`, the signed areas) for `count` entries of
 * the candidate list from `from` on, one per lane: the pairs the passes could not reject from the plane and the bounding sphere alone.
 * A pair that passes puts its key on its ray's slot, or — accepted with a NaN distance — moves its ray's last-NaN mark. */
template <class Scene>
__device__ __forceinline__ void bfs_areas(const Scene &sc, BfsLds *bl, const uint32_t lane, const uint32_t from, const uint32_t count) {
    const bool mine = lane < count;
    const uint32_t c = bl->cand[from + (mine ? lane : 0u)];
    const uint32_t r = c & 63u, tri = mine ? c >> 6 : 0u;
    const float4 a = bl->ro[r], b = bl->rd[r];
    const V3 o = v3(a.x, a.y, a.z), d = v3(b.x, b.y, b.z);
    const uint32_t mode = __float_as_uint(b.w) & 3u, excl = __float_as_uint(a.w);
    const DevTri &T = sc.tris[tri];
    const V3 n = v3(T.n[0], T.n[1], T.n[2]);
    const float nd = dot(n, d);
    const bool bf = nd > 0.0f; /* Triangle::backface, primitives.rs:44-46 */
    const bool culled = bf ? mode == FACE_FRONT : mode == FACE_BACK; /* main.rs:185-188 */
    bool excluded = false; /* main.rs:190-200 */
    if ((excl >> 31) != 0u && (excl & 0x1fffffffu) == tri) {
        const uint32_t ex_face = (excl >> 29) & 3u;
        excluded = ex_face == FACE_FRONT ? !bf : (ex_face == FACE_BACK ? bf : true);
    }
    const float t = (T.d - dot(n, o)) / nd; /* main.rs:203-204 */
    const V3 p = o + d * t; /* main.rs:210 */
    const float a0 = dot(cross(v3(T.e0[0], T.e0[1], T.e0[2]), p - v3(T.v1[0], T.v1[1], T.v1[2])), n);
    const float a1 = dot(cross(v3(T.e1[0], T.e1[1], T.e1[2]), p - v3(T.v2[0], T.v2[1], T.v2[2])), n);
    const float a2 = dot(cross(v3(T.e2[0], T.e2[1], T.e2[2]), p - v3(T.v0[0], T.v0[1], T.v0[2])), n);
    /* `t <= 0` rejects, NaN passes (main.rs:205); NaN areas pass (main.rs:224) */
    if (mine && !culled && !excluded && !(t <= 0.0f) && !(a0 < 0.0f || a1 < 0.0f || a2 < 0.0f)) {
        if (t != t) atomicMax(&bl->nan_last[r], tri + 1u); /* accepted with a NaN distance: not a key; it moves L */
        else atomicMin(&bl->key[r], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(~tri)); /* ties: the later triangle wins */
    }
}
/* the candidate list down to `keep` entries at most, 64 at a time from its end */
template <class Scene>
__device__ __forceinline__ uint32_t bfs_drain(const Scene &sc, BfsLds *bl, const uint32_t lane, uint32_t n_cand, const uint32_t keep) {
    if (n_cand <= keep) return n_cand;
    pair_sync();
    while (n_cand > keep) {
        const uint32_t take = n_cand < 64u ? n_cand : 64u;
        n_cand -= take;
        bfs_areas(sc, bl, lane, n_cand, take);
    }
    pair_sync();
    return n_cand;
}
/* The pair passes over the job records [0, n_rec) of `recs` (BAND: all of them band jobs).  second: only triangles behind their
 * ray's last NaN, of rays that have one.  Returns the candidates still in the list (64 at most). */
template <bool BAND, class Scene>
__device__ __forceinline__ uint32_t bfs_pairs(const Scene &sc, BfsLds *bl, const uint2 *recs, const uint32_t n_rec, const bool second, uint32_t n_cand,
                                              const uint32_t lane, unsigned long long *diag_cands) {
    constexpr uint32_t G = BAND ? RT_BFS_BAND_GROUP : RT_BFS_PAIR_GROUP, RG = 4u * G;
    const uint32_t n_groups = (n_rec + RG - 1u) / RG;
    if (n_groups == 0u) return n_cand;
    const uint32_t rec_lane = lane >> 4, member = lane & 15u;
    const uint2 no_job = make_uint2(0u, 0u); /* count 0: no pair; triangle 0 exists (there are jobs) */
    const float4 *planes = sc.bfs_soa + 3u * (size_t)sc.n_segments; /* n_triangles planes, then n_triangles bounding spheres */
    uint2 grp0 = (lane < RG && lane < n_rec) ? bfs_load_record(recs + (lane)) : no_job;
    uint2 grp1 = (lane < RG && RG + lane < n_rec) ? bfs_load_record(recs + (RG + lane)) : no_job;
    float4 ha[G], hb[G];
#pragma unroll
    for (uint32_t q = 0; q < G; ++q) {
        const uint2 jw = bfs_record_of(grp0, q, rec_lane);
        const uint32_t tri = member < ((jw.x >> 8) & 31u) ? jw.y + member : 0u;
        ha[q] = planes[tri];
        hb[q] = BAND ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : planes[sc.n_triangles + tri];
    }
#ifdef RT_DIAG_BFS
    unsigned long long diag_wait = 0ull, diag_t0 = __builtin_readcyclecounter(), diag_issue = 0ull, diag_lds = 0ull;
#endif
    for (uint32_t g = 0u; g < n_groups; ++g) {
        RT_BFS_STAMP_WAIT(diag_wait)
#ifdef RT_DIAG_BFS
        const unsigned long long st0_ = __builtin_readcyclecounter();
#endif
        const uint32_t at2 = RG * (g + 2u) + lane;
        const uint2 grp2 = (lane < RG && at2 < n_rec) ? bfs_load_record(recs + (at2)) : no_job;
        float4 ha1[G], hb1[G];
        {   /* the next group's plane records: in flight while this group is computed */
            uint2 jn[G];
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) jn[q] = bfs_record_of(grp1, q, rec_lane);
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) {
                const uint32_t tri = member < ((jn[q].x >> 8) & 31u) ? jn[q].y + member : 0u;
                ha1[q] = planes[tri];
                hb1[q] = BAND ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : planes[sc.n_triangles + tri];
            }
        }
#ifdef RT_DIAG_BFS
        const unsigned long long st1 = __builtin_readcyclecounter();
#endif
        /* this group: the records and the rays of all its passes first (one LDS round trip for the group, not one per pass) */
        uint2 jw[G];
        float4 dir[G];
        bool pr[G];
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) jw[q] = bfs_record_of(grp0, q, rec_lane);
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) dir[q] = bl->rd[jw[q].x & 63u];
#ifdef RT_DIAG_BFS
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
        const unsigned long long st2 = __builtin_readcyclecounter();
        if (BAND) { diag_issue += st1 - st0_; diag_lds += st2 - st1; }
#endif
        bool any = false;
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) {
            pr[q] = member < ((jw[q].x >> 8) & 31u);
            if (second) {
                const uint32_t behind = bl->nan_last[jw[q].x & 63u];
                pr[q] = pr[q] && behind != 0u && jw[q].y + member >= behind;
            }
            if constexpr (BAND) /* the ray's line misses this triangle's sphere: only a plane the ray is nearly parallel to goes on */
                pr[q] = pr[q] && !(rtdm::f_abs(dot(v3(ha[q].x, ha[q].y, ha[q].z), v3(dir[q].x, dir[q].y, dir[q].z))) >= 1.0e-3f);
            any = any || pr[q];
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) {
                bool go = pr[q];
                if (__builtin_amdgcn_ballot_w64(go) == 0ull) continue;
                const uint32_t r = jw[q].x & 63u, tri = jw[q].y + member;
                if constexpr (!BAND) { /* culling, exclusion, t and the conservative bounding-sphere rejection of the wave-uniform loop (rt_cast_asm.h) */
                    const float4 a = bl->ro[r];
                    const V3 o = v3(a.x, a.y, a.z), d = v3(dir[q].x, dir[q].y, dir[q].z);
                    const uint32_t flags = __float_as_uint(dir[q].w), excl = __float_as_uint(a.w);
                    const V3 n = v3(ha[q].x, ha[q].y, ha[q].z);
                    const float nd = dot(n, d);
                    const uint32_t mode = flags & 3u;
                    const bool bf = nd > 0.0f; /* Triangle::backface, primitives.rs:44-46 */
                    const bool culled = bf ? mode == FACE_FRONT : mode == FACE_BACK; /* main.rs:185-188 */
                    bool excluded = false; /* main.rs:190-200 */
                    if ((excl >> 31) != 0u && (excl & 0x1fffffffu) == tri) {
                        const uint32_t ex_face = (excl >> 29) & 3u;
                        excluded = ex_face == FACE_FRONT ? !bf : (ex_face == FACE_BACK ? bf : true);
                    }
                    const float t = (ha[q].w - dot(n, o)) / nd; /* main.rs:203-204 */
                    go = go & !culled & !excluded & !(t <= 0.0f); /* NaN passes, as in the reference (main.rs:205) */
                    const V3 w = (o + d * t) - v3(hb[q].x, hb[q].y, hb[q].z);
                    const float qq = (w.x * w.x + w.y * w.y) + w.z * w.z;
                    if (hb[q].w < qq && 1.0e30f > qq && (flags & 4u) != 0u) go = false;
                }
                const unsigned long long going = __builtin_amdgcn_ballot_w64(go);
                if (going != 0ull) { /* on to the whole test — later, together with the other pairs that get this far */
                    if (go) bl->cand[n_cand + bfs_rank(going)] = r | (tri << 6);
                    n_cand += (uint32_t)__builtin_popcountll(going);
#ifdef RT_DIAG_BFS
                    *diag_cands += (unsigned long long)__builtin_popcountll(going);
#endif
                }
            }
        }
        n_cand = bfs_drain(sc, bl, lane, n_cand, 64u); /* room for another group's worth */
        grp0 = grp1;
        grp1 = grp2;
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) { ha[q] = ha1[q]; hb[q] = hb1[q]; }
    }
#ifdef RT_DIAG_BFS
    if (lane == 0u && BAND) { atomicAdd(&g_bfs_ticks[10], diag_issue); atomicAdd(&g_bfs_ticks[11], diag_lds); atomicAdd(&g_bfs_ticks[12], (unsigned long long)n_groups); }
    if (lane == 0u) { atomicAdd(&g_bfs_ticks[BAND ? 8 : 6], diag_wait); atomicAdd(&g_bfs_ticks[BAND ? 9 : 7], __builtin_readcyclecounter() - diag_t0 - diag_wait); }
#endif
    return n_cand;
}
template <class Scene>
__device__ __forceinline__ CastResult cast_bfs(const Scene &sc, const Ray &ray, bool active, BfsLds *bl, const BfsScratch &ws) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    CastResult none;
    none.prim = -1;
    none.t = 0.0f;
    none.bf = 0u;
    none.a0 = none.a1 = none.a2 = 0.0f;
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    if (act == 0ull) return none;
    {   /* this lane's ray where the others can read it */
        const uint32_t flags = (ray.mode & 3u) | (dot(ray.o, ray.o) <= sc.filter_origin2 ? 4u : 0u);
        bl->key[lane] = ~0ull;
        bl->nan_last[lane] = 0u;
        bl->ro[lane] = make_float4(ray.o.x, ray.o.y, ray.o.z, __uint_as_float(ray.excl));
        bl->rd[lane] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(flags));
    }
    pair_sync();
    const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
    const uint32_t my_rank = bfs_rank(act);
    const uint32_t rec_lane = lane >> 4, member = lane & 15u;
    uint2 *list = ws.items_a, *next = ws.items_b;
    const uint32_t top_chunks = (sc.bfs_top + 15u) >> 4;
    uint32_t n_items = top_chunks * n_act, n_jobs = 0u, n_band = 0u;
    bool bad = n_items > ws.items_cap; /* wave-uniform: a list overflowed */
    if (!bad)
        for (uint32_t c = 0; c < top_chunks; ++c) { /* ray-major, as the passes emit them: the records of a pass name consecutive nodes */
            const uint32_t left = sc.bfs_top - 16u * c;
            if (active) list[my_rank * top_chunks + c] = make_uint2(lane | ((left < 16u ? left : 16u) << 8), 16u * c);
        }
    unsigned long long diag_cands = 0ull;
#ifdef RT_DIAG_BFS
    unsigned long long diag_items = 0ull, diag_levels = 0ull;
    uint32_t diag_max = n_items;
    const unsigned long long diag_c0 = __builtin_readcyclecounter();
#endif
    while (n_items != 0u && !bad) {
        pair_sync(); /* the list was written by other lanes */
        constexpr uint32_t G = RT_BFS_LEVEL_GROUP, RG = 4u * G;
        uint32_t n_next = 0u;
        const uint32_t n_groups = (n_items + RG - 1u) / RG;
        const uint2 no_rec = make_uint2(0u, 0u); /* count 0: no pair; node 0 exists (there are records) */
        uint2 grp0 = (lane < RG && lane < n_items) ? bfs_load_record(list + (lane)) : no_rec;
        uint2 grp1 = (lane < RG && RG + lane < n_items) ? bfs_load_record(list + (RG + lane)) : no_rec;
        float4 h0[G], h1[G], n0[G];
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) {
            const uint2 rec = bfs_record_of(grp0, q, rec_lane);
            const uint32_t k = member < ((rec.x >> 8) & 31u) ? rec.y + member : 0u;
            h0[q] = sc.bfs_soa[k]; h1[q] = sc.bfs_soa[sc.n_segments + k]; n0[q] = sc.bfs_soa[2u * sc.n_segments + k];
        }
#ifdef RT_DIAG_BFS
        unsigned long long diag_wait = 0ull, diag_t0 = __builtin_readcyclecounter();
#endif
        for (uint32_t g = 0u; g < n_groups && !bad; ++g) {
            RT_BFS_STAMP_WAIT(diag_wait)
            const uint32_t at2 = RG * (g + 2u) + lane;
            const uint2 grp2 = (lane < RG && at2 < n_items) ? bfs_load_record(list + (at2)) : no_rec;
            float4 h0n[G], h1n[G], n0n[G];
            {   /* the next group's node records: in flight while this group is computed */
                uint2 rn[G];
#pragma unroll
                for (uint32_t q = 0; q < G; ++q) rn[q] = bfs_record_of(grp1, q, rec_lane);
#pragma unroll
                for (uint32_t q = 0; q < G; ++q) {
                    const uint32_t k = member < ((rn[q].x >> 8) & 31u) ? rn[q].y + member : 0u;
                    h0n[q] = sc.bfs_soa[k]; h1n[q] = sc.bfs_soa[sc.n_segments + k]; n0n[q] = sc.bfs_soa[2u * sc.n_segments + k];
                }
            }
            /* this group: the records and the rays of all its passes first (one LDS round trip for the group, not one per pass) */
            uint2 recs[G];
            float4 org[G], dir[G];
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) recs[q] = bfs_record_of(grp0, q, rec_lane);
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) { org[q] = bl->ro[recs[q].x & 63u]; dir[q] = bl->rd[recs[q].x & 63u]; }
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) {
                const uint2 rec = recs[q];
                const bool have = member < ((rec.x >> 8) & 31u);
                if (bad || __builtin_amdgcn_ballot_w64(have) == 0ull) continue;
                const uint32_t r = rec.x & 63u, k = have ? rec.y + member : 0u;
                const V3 o = v3(org[q].x, org[q].y, org[q].z), d = v3(dir[q].x, dir[q].y, dir[q].z);
                const uint32_t first = __float_as_uint(h0[q].x), count = __float_as_uint(h0[q].y), nn = __float_as_uint(h0[q].z);
                const bool cone = nn == RT_SEGMENT_CONE;
                /* cluster_skippable_lane for ray r: its line misses the node's sphere, and it is not (nearly) parallel to a plane below */
                const V3 disp = v3(h1[q].x, h1[q].y, h1[q].z) - o;
                const V3 cr = cross(disp, d);
                const float dd = dot(d, d);
                const bool missed = dot(cr, cr) > h0[q].w * dd && nn != 0u && (__float_as_uint(dir[q].w) & 4u) != 0u;
                const float ad = dot(v3(n0[q].x, n0[q].y, n0[q].z), d);
                bool steep = cone ? ad * ad >= n0[q].w * dd : rtdm::f_abs(ad) >= 1.0e-3f;
                const bool more = have && missed && steep && !cone && nn > 1u; /* two to eight plane directions: the others */
                if (__builtin_amdgcn_ballot_w64(more) != 0ull) {
                    const float4 *gp = reinterpret_cast<const float4 *>(sc.bfs_nodes + k);
                    const float4 m1 = gp[3], m2 = gp[4], m3 = gp[5]; /* (all three on their way before the first is used) */
                    steep = steep & ((rtdm::f_abs(dot(v3(m1.x, m1.y, m1.z), d)) >= 1.0e-3f) | !more | (1u >= nn));
                    steep = steep & ((rtdm::f_abs(dot(v3(m2.x, m2.y, m2.z), d)) >= 1.0e-3f) | !more | (2u >= nn));
                    steep = steep & ((rtdm::f_abs(dot(v3(m3.x, m3.y, m3.z), d)) >= 1.0e-3f) | !more | (3u >= nn));
                    if (__builtin_amdgcn_ballot_w64(more && steep && nn > 4u) != 0ull) {
                        const float4 m4 = gp[6], m5 = gp[7], m6 = gp[8], m7 = gp[9];
                        steep = steep & ((rtdm::f_abs(dot(v3(m4.x, m4.y, m4.z), d)) >= 1.0e-3f) | !more | (4u >= nn));
                        steep = steep & ((rtdm::f_abs(dot(v3(m5.x, m5.y, m5.z), d)) >= 1.0e-3f) | !more | (5u >= nn));
                        steep = steep & ((rtdm::f_abs(dot(v3(m6.x, m6.y, m6.z), d)) >= 1.0e-3f) | !more | (6u >= nn));
                        steep = steep & ((rtdm::f_abs(dot(v3(m7.x, m7.y, m7.z), d)) >= 1.0e-3f) | !more | (7u >= nn));
                    }
                }
                const bool hit = have && !(missed && steep);
#ifdef RT_DIAG_BFS
                diag_items += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
#endif
                /* a leaf the ray may hit: a job (the band jobs from the back of the array); an inner node: a record of the next level that
                 * names its children.  One record each — 16 triangles or children at most — in straight-line code with the counters in
                 * scalar registers; what is left of a longer leaf or child list follows in a loop hardly any pass enters. */
                const bool leaf = hit && count != 0u, inner = hit && count == 0u;
                const bool band = missed && nn != 1u; /* (one plane direction: all its triangles are nearly parallel to the ray) */
                const uint32_t n_child = inner ? __float_as_uint(h1[q].w) : 0u;
                const unsigned long long m_full = __builtin_amdgcn_ballot_w64(leaf && !band), m_band = __builtin_amdgcn_ballot_w64(leaf && band);
                const unsigned long long m_child = __builtin_amdgcn_ballot_w64(n_child != 0u);
                const uint32_t n_f = (uint32_t)__builtin_popcountll(m_full), n_b = (uint32_t)__builtin_popcountll(m_band), n_c = (uint32_t)__builtin_popcountll(m_child);
                if (n_jobs + n_f + n_band + n_b > ws.jobs_cap || n_next + n_c > ws.items_cap) { bad = true; continue; }
                if (leaf) {
                    const uint2 job = make_uint2(r | ((count < 16u ? count : 16u) << 8) | (band ? RT_BFS_BAND : 0u), first);
                    ws.jobs[band ? ws.jobs_cap - 1u - (n_band + bfs_rank(m_band)) : n_jobs + bfs_rank(m_full)] = job;
                }
                if (n_child != 0u) next[n_next + bfs_rank(m_child)] = make_uint2(r | ((n_child < 16u ? n_child : 16u) << 8), first);
                n_jobs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_jobs + n_f));
                n_band = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_band + n_b));
                n_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_next + n_c));
                if (__builtin_amdgcn_ballot_w64((leaf && count > 16u) || n_child > 16u) != 0ull) { /* the rest of long lists */
                    const uint32_t n_sub = leaf ? (count + 15u) >> 4 : 0u, n_sub_c = (n_child + 15u) >> 4;
                    for (uint32_t s = 1u; !bad; ++s) {
                        const bool e_band = s < n_sub && band, e_full = s < n_sub && !band, e_child = s < n_sub_c;
                        const unsigned long long mf = __builtin_amdgcn_ballot_w64(e_full), mb = __builtin_amdgcn_ballot_w64(e_band), mc = __builtin_amdgcn_ballot_w64(e_child);
                        if ((mf | mb | mc) == 0ull) break;
                        const uint32_t kf = (uint32_t)__builtin_popcountll(mf), kb = (uint32_t)__builtin_popcountll(mb), kc = (uint32_t)__builtin_popcountll(mc);
                        if (n_jobs + kf + n_band + kb > ws.jobs_cap || n_next + kc > ws.items_cap) { bad = true; break; }
                        const uint32_t left = count - 16u * s, left_c = n_child - 16u * s;
                        const uint2 job = make_uint2(r | ((left < 16u ? left : 16u) << 8) | (band ? RT_BFS_BAND : 0u), first + 16u * s);
                        if (e_full) ws.jobs[n_jobs + bfs_rank(mf)] = job;
                        if (e_band) ws.jobs[ws.jobs_cap - 1u - (n_band + bfs_rank(mb))] = job;
                        if (e_child) next[n_next + bfs_rank(mc)] = make_uint2(r | ((left_c < 16u ? left_c : 16u) << 8), first + 16u * s);
                        n_jobs += kf;
                        n_band += kb;
                        n_next += kc;
                    }
                }
            }
            grp0 = grp1;
            grp1 = grp2;
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) { h0[q] = h0n[q]; h1[q] = h1n[q]; n0[q] = n0n[q]; }
        }
#ifdef RT_DIAG_BFS
        if (lane == 0u) { atomicAdd(&g_bfs_ticks[4], diag_wait); atomicAdd(&g_bfs_ticks[5], __builtin_readcyclecounter() - diag_t0 - diag_wait); }
#endif
        uint2 *t = list; list = next; next = t;
        n_items = n_next;
#ifdef RT_DIAG_BFS
        diag_levels += 1ull; if (n_next > diag_max) diag_max = n_next;
#endif
    }
#ifdef RT_DIAG_BFS
    const unsigned long long diag_c1 = __builtin_readcyclecounter();
    unsigned long long diag_band_cycles = 0ull;
#endif
    /* The jobs, once for every ray; and — should a ray have accepted a NaN distance (it lies in a triangle's plane: 0 / 0) — once
     * more for those rays.  The reference's rule "replace unless nearest_t < t" is a minimum only while no accepted t is NaN: a NaN
     * replaces whatever was nearest, and whatever is accepted next replaces the NaN.  So with L the LAST triangle a ray accepted with
     * a NaN distance, the ray ends with the minimum over its candidates BEHIND L (index > L) under the usual rule, or, if there are
     * none, with L itself and its NaN.  Pass 0 finds the minimum over everything and L; pass 1, for the rays that have an L, the
     * minimum over the candidates behind it. */
    pair_sync(); /* the jobs were written by other lanes */
    for (uint32_t pass = 0; pass < 2u && !bad && n_jobs + n_band != 0u; ++pass) {
        if (pass == 1u) {
            const uint32_t mine = bl->nan_last[lane];
            if (__builtin_amdgcn_ballot_w64(mine != 0u) == 0ull) break; /* no ray met a NaN: the rule was a minimum */
            if (mine != 0u) bl->key[lane] = ~0ull;
            pair_sync();
        }
        uint32_t n_cand = bfs_pairs<false>(sc, bl, ws.jobs, n_jobs, pass == 1u, 0u, lane, &diag_cands);
#ifdef RT_DIAG_BFS
        const unsigned long long diag_b0 = __builtin_readcyclecounter();
#endif
        n_cand = bfs_pairs<true>(sc, bl, ws.jobs + (ws.jobs_cap - n_band), n_band, pass == 1u, n_cand, lane, &diag_cands);
#ifdef RT_DIAG_BFS
        diag_band_cycles += __builtin_readcyclecounter() - diag_b0;
#endif
        n_cand = bfs_drain(sc, bl, lane, n_cand, 0u);
        pair_sync();
    }
    const uint32_t my_nan = bad ? 0u : bl->nan_last[lane];
#ifdef RT_DIAG_BFS
    const unsigned long long diag_c2 = __builtin_readcyclecounter();
    {
        const unsigned long long nan_rays = __builtin_amdgcn_ballot_w64(active && my_nan != 0u);
        uint32_t pj = 0u, pb = 0u; /* pairs of this lane's share of the jobs */
        for (uint32_t i = lane; i < n_jobs; i += 64u) pj += (ws.jobs[i].x >> 8) & 31u;
        for (uint32_t i = lane; i < n_band; i += 64u) pb += (ws.jobs[ws.jobs_cap - 1u - i].x >> 8) & 31u;
        for (int off = 32; off > 0; off >>= 1) { pj += __shfl_down(pj, off, 64); pb += __shfl_down(pb, off, 64); }
        if (lane == 0u) {
            atomicAdd(&g_bfs_stats[0], 1ull);
            if (bad) atomicAdd(&g_bfs_stats[1], 1ull);
            else if (nan_rays != 0ull) atomicAdd(&g_bfs_stats[2], (unsigned long long)__builtin_popcountll(nan_rays));
            atomicAdd(&g_bfs_stats[3], diag_items);
            atomicAdd(&g_bfs_stats[4], (unsigned long long)n_jobs);
            atomicMax(&g_bfs_stats[5], (unsigned long long)diag_max);
            atomicAdd(&g_bfs_stats[6], diag_levels);
            atomicAdd(&g_bfs_stats[7], (unsigned long long)pj);
            atomicAdd(&g_bfs_stats[8], (unsigned long long)n_band);
            atomicAdd(&g_bfs_stats[9], (unsigned long long)pb);
            atomicAdd(&g_bfs_stats[10], diag_cands);
            atomicMax(&g_bfs_stats[11], diag_c2 - diag_c0); /* the longest walk of one wave-cast, cycles */
        }
    }
#endif
    if (bad) { /* a list overflowed: the wave-uniform walk, exact whatever happened */
        CastResult cr = none;
        if (active) cr = cast_asm(sc, ray);
        return cr;
    }
#ifdef RT_DIAG_BFS
    const unsigned long long diag_c3 = __builtin_readcyclecounter();
#endif
    CastResult cr = none;
    if (active) {
        const unsigned long long key = bl->key[lane];
        float t = rtdm::quiet_nan();
        int32_t prim = -1;
        if (key != ~0ull) { t = __uint_as_float((uint32_t)(key >> 32)); prim = (int32_t)(~(uint32_t)key); }
        else if (my_nan != 0u) prim = (int32_t)(my_nan - 1u); /* nothing behind the last NaN: the ray ends with it (t stays NaN) */
        cr = cast_finish(sc, ray, t, prim); /* the winner's flag and areas re-evaluated, then the spheres: the same operations */
    }
#ifdef RT_DIAG_BFS
    if (lane == 0u) {
        atomicAdd(&g_bfs_ticks[0], diag_c1 - diag_c0);
        atomicAdd(&g_bfs_ticks[1], diag_c2 - diag_c1 - diag_band_cycles);
        atomicAdd(&g_bfs_ticks[2], diag_band_cycles);
        atomicAdd(&g_bfs_ticks[3], __builtin_readcyclecounter() - diag_c3);
    }
#endif
    return cr;
}

} /* namespace rt */

#endif /* RT_CAST_BFS_H */
