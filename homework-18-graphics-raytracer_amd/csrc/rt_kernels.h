/*
 * rt_kernels.h — interface between the C-ABI layer (rt_api.hip) and the kernels
 * (rt_kernels.hip).  Internal; the public boundary is include/rt_amd.h.
 */
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_shade.h"

#ifndef RT_BLOCK_THREADS
#define RT_BLOCK_THREADS 64 /* one wave per workgroup: a finished wave frees its slot at once (256-thread groups idled 40 % of the slots) */
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 3 /* <= 168 VGPRs: as fast as 4 waves/SIMD (<= 128) and spills 2 registers instead of 53 (profiles/README.md) */
#endif
#ifdef RT_MIN_WAVES
#define RT_LAUNCH_BOUNDS __launch_bounds__(RT_BLOCK_THREADS, RT_MIN_WAVES) /* 2nd arg: waves per SIMD */
#else
#define RT_LAUNCH_BOUNDS __launch_bounds__(RT_BLOCK_THREADS)
#endif
#define RT_FALLBACK_WAVES 4096u /* grid of the wavefront path's fallback launch of the per-pixel kernel (rt_kernels.hip) */
#define RT_LDS_SCENE_LIMIT (96u * 1024u) /* triangle records staged in LDS up to this many bytes */

/* kernel variants (A/B-selectable through RT_AMD_VARIANT or rt_set_variant): 2, 3, 18 (default), 19 */
#define RT_VARIANT_SGPR 0 /* triangle records fetched with wave-uniform scalar loads */
#define RT_VARIANT_LDS 1          /* bit 0 (per-pixel kernel): triangle records staged in LDS once per workgroup */
#define RT_VARIANT_STATIC 2       /* bit 1, always set: the per-pixel kernel renders one 8x8 tile per wave in image order (the bits round 1
                                   * used for its other per-pixel schemes — values 0, 4, 6, 8 — are no longer accepted) */
#define RT_VARIANT_PWF 16         /* bit 4: one persistent kernel of workgroup-local wavefronts (rt_pwf.hip, the default); a frame that
                                   * does not fit its arenas is rendered by the per-pixel kernel */
#define RT_VARIANT_DEFAULT (RT_VARIANT_PWF | RT_VARIANT_SGPR | RT_VARIANT_STATIC)

namespace rt {

/* device pointers + counts, passed by value as kernel arguments (lands in SGPRs) */
struct KernelScene {
    const DevTri *tris;
    const DevTriAttr *attrs;
    const DevSphere *spheres;
    const rt_material *materials;
    const rt_light *lights;
    uint32_t n_triangles, n_spheres, n_materials, n_lights;
    float filter_origin2; /* rays with |origin|^2 above this skip the bounding-sphere rejection (rt_device_scene.h) */
    const DevSegment *segments; /* the triangles as runs, clusters among them (rt_device_scene.h) */
    uint32_t n_segments;
    const DevTriHead *heads;    /* plane + bounding sphere per triangle, for the pair-wise tests (rt_cast.h cast_pairs) */
    const LightAux *light_aux;  /* per light: the cosine of a spot light's spread with its margins (rt_shade.h light_asks) */
    /* the node tree once more, in level order: bfs_nodes[0 .. bfs_top) are the top-level nodes, and an INNER node's record names its
     * children as bfs_nodes[first .. first + skip_to) (a leaf's first / count are its triangles, as in `segments`) — what a
     * breadth-first walk needs (rt_cast_bfs.h cast_bfs).  bfs_walk != 0: the scene is large enough for the kernels that have that walk to
     * use it (rt_scene_create) */
    const DevSegment *bfs_nodes;
    /* the same once more as arrays of 16-byte pieces, one after the other: n_segments x (first, count, n_normals, r2_hi), n_segments x
     * (centre, child count), n_segments x (the first plane direction or the cone), n_triangles x plane, n_triangles x bounding sphere */
    const float4 *bfs_soa;
    uint32_t bfs_top, bfs_walk;
};

/* frame/tile + the per-frame camera basis of Camera::shoot (main.rs:85-92),
 * hoisted to the host: it is the same for every ray */
struct KernelFrame {
    uint32_t cols, rows;     /* tile size in pixels */
    uint32_t n_chunks;       /* ceil(cols*rows / 64), filled by the launcher */
    uint32_t x0, y0, y_step; /* tile origin and row stride in the image */
    int32_t max_depth;
    float half_height;       /* height as f32 / 2.0   (main.rs:1094) */
    float half_width;        /* width as f32 / 2.0    (main.rs:1095) */
    float height_f;          /* height as f32 */
    float cam_origin[3];     /* center + toward * near */
    float cam_x[3];          /* tan(fovy/2) * right */
    float cam_y[3];          /* tan(fovy/2) * up' */
    float cam_toward[3];     /* normalize(toward) */
    float cam_origin_focus[3]; /* center + normalize(normalize(toward)) * near: shoot_focus normalises twice (main.rs:119) */
};

/* what a launch of the per-pixel kernel is told besides the scene and the frame */
struct KernelQueues {
    unsigned long long *timeline;  /* diagnostic builds (RT_DIAG_TIMELINE): 4 u64 per wave, else unused */
    const uint32_t *run_if;        /* when set: the kernel is a no-op unless *run_if != 0 (the persistent-wavefront path's overflow fallback) */
};

/* Process-wide switches — A/B knobs and test hooks (include/rt_amd.h rt_set_option).  Each is an integer that starts from the
 * environment variable of its name, read ONCE per process at the first use of any of them, and can be changed at run time through
 * rt_set_option; option() returns `unset` while it has no value.  (Defined in rt_api.hip; OPT_NAMES there is in this order.) */
enum Option : int {
    OPT_RNG_LOOKAHEAD,     /* 0: every IsaacCore::generate is left to the render kernels */
    OPT_RNG_OVERLAP,       /* 0: the look-ahead runs in line, before each chain kernel */
    OPT_DIST_PIPELINE,     /* 0: one workspace, a batch's kernels in line */
    OPT_DIST_BY_COST,      /* the chain kernel's pixels grouped by cost: 0 never, 1 always (unset: by the share's size) */
    OPT_DIST_OWN_FIRST,    /* a wave's first chunk its own: 0 / 1 (unset: by the share's size) */
    OPT_DIST_PREP_FIRST,   /* 0: the shade kernel and the look-ahead start together */
    OPT_DIST_WS_MB,        /* cap of the split pass's workspace(s), MiB */
    OPT_DIAG_WS_REFUSE,    /* test hook: pretend the first n workspace allocations fail */
    OPT_DIST_STATIC,       /* 1: the one-kernel organisation with one 64-pixel chunk per wave */
    OPT_DIST_CHAIN_WAVES,  /* waves per SIMD of the chain kernel's grid */
    OPT_SHADE_TILE, OPT_SHADE_SORT, /* the per-request shade kernel: samples per workgroup; 0: no bucket sort */
    OPT_MULTI_FORCE_STAGE, /* test hook: rt_multi_* stage every band as if it lived on another device */
    OPT_DIST_SPLIT,        /* 0: the one-kernel organisation (rt_set_distributed_split has the last word) */
    OPT_BFS_WALK_TRIANGLES, /* scenes of at least this many triangles are walked breadth-first by the wavefront kernel (0: never); read by rt_scene_create */
    OPT_WF_SHARE,           /* n > 1: a launch of the persistent wavefront kernel takes 1/n of the workgroups the device holds — a caller with n frames in flight on n streams runs them side by side instead of one behind the other's tail */
    OPT_DIAG_BFS_CAP,       /* test hook: that walk's record lists hold this many records at most (default: RT_BFS_ITEMS_CAP / RT_BFS_JOBS_CAP): a wave-cast that needs more takes the wave-uniform walk */
    OPT_COUNT
};
long long option(Option id, long long unset);

void set_main_kernel_events(hipEvent_t start, hipEvent_t stop); /* profiling hook, see rt_profile_* */
void record_main_kernel_event(int which, hipStream_t stream);    /* 0: start, 1: stop; no-op when profiling is off */
void mute_main_kernel_events(bool muted);                        /* launches in between are not the render kernel */

hipError_t launch_whitted(const KernelScene &sc, const KernelFrame &fr, float *out, unsigned long long *ray_count,
                          const KernelQueues &qs, hipStream_t stream, int variant);

/* persistent workgroup-local wavefronts (rt_pwf.hip) */
#define PW_G_TILE 0u       /* next tile of the frame */
#define PW_G_OVERFLOW 1u   /* the frame could not be finished within the arenas: the per-pixel kernel renders it */
#define PW_G_TILES_DONE 2u /* tiles whose pixels were written */
#define PW_G_GROUPS_DONE 3u /* workgroups that have left the kernel: the last one closes the frame */
#define PW_G_CASTS 4u      /* u64 */
#define PW_G_WORDS 8u
#define PW_G_BLOCK_WORDS 32u /* a block of global words (PW_G_WORDS of them in use, the rest for diagnostic builds) */
struct PwParams {
    uint32_t *global;       /* this launch's block of PW_G_BLOCK_WORDS u32: zero when the kernel starts */
    uint32_t *global_next;  /* the next launch's block: the last workgroup of this one zeroes it */
    unsigned long long *ray_count; /* += the frame's casts when it was finished within the arenas; may be null */
    const KernelFrame *frame; /* the frame description, in device memory */
    unsigned char *arena;   /* one arena of arena_stride bytes per workgroup */
    size_t arena_stride;
    uint32_t node_cap;      /* nodes per arena */
    uint32_t ring_cap;      /* power of two >= node_cap + 1024: items of the shade and refraction rings */
    uint32_t tile_reserve;  /* nodes an arena must have free per primary ray before it takes more tiles */
    uint32_t tile_stride;   /* the k-th tile handed out is (k * tile_stride) mod n_tiles; coprime to n_tiles */
    const uint32_t *tile_order; /* or, when set, tile_order[k] (n_tiles entries, a permutation) */
    uint32_t *tile_cost;        /* when set: per tile, how many of its pixels recursed (a root with children), written when the frame is folded */
    /* the breadth-first walk's scratch (scenes with KernelScene::bfs_walk): per wave of the grid two level lists of bfs_items_cap
     * records (two words each) and a job list of bfs_jobs_cap records, one after the other */
    uint32_t *bfs_scratch;
    uint32_t bfs_items_cap, bfs_jobs_cap;
};
#define RT_BFS_ITEMS_CAP 32768u
#define RT_BFS_JOBS_CAP 131072u
inline size_t pwf_bfs_scratch_words_per_wave() { return 4u * (size_t)RT_BFS_ITEMS_CAP + 2u * (size_t)RT_BFS_JOBS_CAP; }
int pwf_workgroups_per_cu(uint32_t node_cap, uint32_t ring_cap, bool bfs_walk = false);
size_t pwf_arena_bytes(uint32_t node_cap, uint32_t ring_cap);
/* init: zero this launch's block of global words and (re)write the frame description first — needed for a workspace's first
 * launch and whenever the frame description differs from the previous launch's; otherwise the previous launch has left
 * both as this one needs them */
hipError_t launch_pwf(const KernelScene &sc, KernelFrame fr, float *out, const PwParams &pp, uint32_t workgroups, hipStream_t stream,
                      bool init, bool first_band, bool last_band);

/* distributed pass (rt_distributed.hip) */
struct DistParams {
    uint32_t *rng_states;          /* RT_RNG_DEVICE_WORDS u32 per tile pixel */
    uint32_t n_epochs;
    float focus, blur;             /* main.rs:1147-1148 */
    float *accum;                  /* pixels*3, += every surviving sample; may be null */
    float *samples;                /* n_epochs*pixels*3 raw samples; may be null */
    unsigned char *valid;          /* n_epochs*pixels filter flags; may be null */
    unsigned long long *ray_count; /* may be null */
    uint32_t *work_queue;          /* zeroed chunk counter: persistent lanes; null: one chunk per wave */
    /* split pass (launch_distributed_split): the chain kernel records, per sample of the batch, what the shade and
     * unwind kernels need; all arrays are slot-major ([slot][sample]) so that neighbouring lanes read neighbouring records */
    uint32_t epoch0;               /* first epoch of this batch within the call (indexes samples/valid) */
    uint32_t *sp_hdr;              /* [sample] frames | has_terminal << 8 */
    uint4 *sp_req;                 /* [slot][sample] x 4: get_shade requests (hit, view direction) */
    float4 *sp_shade;              /* [slot][sample]: get_shade results */
    float4 *sp_frame;              /* [level][sample]: factor.xyz, kind */
    uint32_t sp_slots;             /* max_depth + 1 */
    /* Which pixel a lane of the chain kernel takes next: position q of its queue -> pixel_order[q] (null: the tiled image order).
     * The order groups the pixels by what their samples cost in an earlier batch (pixel_cost, written by the unwind kernel; null:
     * not kept), dearest first: a wave keeps stepping until the last of its 64 lanes has finished its pixel, so lanes with pixels
     * of like cost finish together — what decides the kernel's time when a GPU's share has no more pixels than the chip has lanes.
     * Used (rt_api.hip) when a lane gets two pixels at most: a 1/8 share of the 1080p frame 0.41 -> 0.38 ms per epoch, 1/4 0.60 ->
     * 0.565; with more pixels per lane the lanes even out by themselves and neighbouring pixels' rays are worth more: the whole
     * frame 1.68 -> 1.75 ms (profiles/r03_ab12.txt) */
    const uint32_t *pixel_order;
    uint32_t *pixel_cost;
    /* != 0: a wave's first 64-pixel chunk is its own (chunk w for wave w) and only the later ones come from the work counter.  For a
     * share with few chunks per wave: every wave asks at once at a launch, one counter word serves ~88 of them per microsecond, and
     * the chain kernel of a 1/8 share runs for a millisecond (0.422 -> 0.410 ms per epoch).  On the whole frame the counter's
     * staggered answers are worth more than they cost — waves that start together stay in step and ask memory together: 1 165 ->
     * 1 126 Msamples/s in 8-epoch calls with every wave's first chunk its own (profiles/r03_ab12.txt) */
    uint32_t own_first_chunk;
    /* scenes with KernelScene::bfs_walk (the one-kernel organisation only): the breadth-first walk's lists, per wave of the grid two
     * level lists of bfs_items_cap records and a job list of bfs_jobs_cap records, as in PwParams; null: the wave-uniform walk */
    uint32_t *bfs_scratch;
    uint32_t bfs_items_cap, bfs_jobs_cap;
};
size_t distributed_split_bytes_per_sample(int32_t max_depth);
#define RT_RNG_STATE_WORDS 516u   /* the oracle's / reference's record: what rt_rng_download returns per pixel */
#define RT_RNG_DEVICE_WORDS 1032u /* the device record: two banks of it (rt_distributed.hip) */
hipError_t launch_rng_seed(uint32_t *states, const KernelFrame &fr, hipStream_t stream);
/* look-ahead: generate the next block of every pixel that has none prepared; list = n_pixels + 1 words of scratch */
hipError_t launch_rng_prepare(uint32_t *states, uint32_t n_pixels, uint32_t *list, uint32_t compute_units, hipStream_t stream);
hipError_t launch_rng_export(const uint32_t *states, uint32_t n_pixels, uint32_t *out, hipStream_t stream);
hipError_t launch_distributed(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, uint32_t resident_waves, hipStream_t stream);
uint32_t dist_bfs_waves(uint32_t compute_units); /* the grid of launch_distributed, at most, when dp.bfs_scratch is set: one list set per wave */
/* the split pass, one batch of dp.n_epochs epochs: the chain kernel (all random draws; dp.work_queue zeroed), and — once it has
 * finished — the shade and unwind kernels, which only read what it recorded and never touch the RNG records: the caller may start
 * the look-ahead for the next batch, and the next batch's chain kernel on another workspace, beside them */
uint32_t dist_chain_waves(uint32_t resident_waves); /* the chain kernel's grid, at most */
hipError_t launch_dist_chain(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, uint32_t resident_waves, hipStream_t stream);
hipError_t launch_dist_shade_unwind(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, hipStream_t stream,
                                    const hipEvent_t *ev = nullptr); /* ev: four events of the caller's, recorded around the shade kernel and around the unwind */
/* order[] <- the pixels 0 .. n-1 grouped by cost[] (clipped to 255), dearest first; scratch: 512 words */
hipError_t launch_dist_pixel_order(const uint32_t *cost, uint32_t *order, uint32_t n_pixels, uint32_t *scratch, hipStream_t stream);

/* post_process / sRGB encode on the device (rt_post.hip) */
hipError_t launch_post_process(float *rgb, size_t n_pixels, const float luma_row[3], uint32_t *keys, uint32_t *state,
                               float *divisor_out, hipStream_t stream);
/* its passes one by one (rt_post.hip): keys n_pixels u32, state RT_POST_STATE_WORDS u32, both the caller's */
hipError_t launch_post_keys(const float *rgb, size_t n_pixels, const float luma_row[3], uint32_t *keys, uint32_t *state, hipStream_t stream);
hipError_t launch_post_hist(const uint32_t *keys, size_t n_pixels, int pass, uint32_t *state, hipStream_t stream);
hipError_t launch_post_pick(int pass, uint32_t *state, hipStream_t stream);
hipError_t launch_post_scale(float *rgb, size_t n_pixels, const uint32_t *state, float *divisor_out, hipStream_t stream);
hipError_t launch_encode_srgb8(const float *rgb, size_t n_values, unsigned char *out, hipStream_t stream);
hipError_t launch_accumulate(const float *samples, const unsigned char *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight,
                             hipStream_t stream);
hipError_t launch_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb, hipStream_t stream);

/* diagnostics: evaluate rt_detmath on the device (op codes = rt_math_op) */
hipError_t launch_math_eval(int op, const float *d_x, const float *d_y, float *d_out, size_t n, hipStream_t stream);

} /* namespace rt */

#endif
