/*
 * rt_amd.h — C ABI of the MI355X-native render path.
 *
 * This is the drop-in boundary for the per-pixel render loop of
 * foriequal0/homework-18-graphics-raytracer.  The reference has no FFI: its
 * render loop is two rayon closures inside main() (src/main.rs:1090-1104 for
 * the Whitted pass, src/main.rs:1131-1156 for the distributed pass).  The seam
 * is created exactly there: everything those closures read (World, Camera,
 * width/height/depth literals) comes in as flat POD arrays, and everything they
 * produce (one LinSrgb per pixel) goes out as row-major f32 RGB.
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns
 * RT_OK (0) or a negative rt_status and never unwinds or aborts the host
 * (the reference's convention is panic!/unwrap — src/main.rs:767-775,785 —
 * which cannot cross a C ABI).  rt_last_error() returns a thread-local message
 * for the last failing call.
 *
 * Threading (the reference shares `&World` immutably between rayon threads and
 * gives each pixel exclusive `&mut` access to its RNG, main.rs:1096, 1131):
 * an rt_scene is immutable and may be rendered from several host threads at
 * once, each on its own HIP stream (per-(scene, stream) workspaces are created
 * under a lock; the rt_set_* settings are atomics; the profiling hooks keep the
 * event pair of a call in thread-local state).  Calls on ONE stream, and calls
 * on one rt_rng, must be serialised by the caller.
 *
 * State between calls: the per-(scene, stream) workspace remembers which frame
 * description it holds on the device, so that a call with the same frame as the
 * one before it is a single kernel launch.  A render call may be captured into a
 * HIP graph once its workspace exists (allocation cannot be captured: render the
 * frame once uncaptured first); from the first capture on a stream, every
 * launch on that stream prepares its own state, as the captured one does, since
 * replays come unannounced.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1, /* null pointer, empty frame, index out of range ... */
    RT_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime failure at init */
    RT_ERR_HIP = -3,              /* a HIP call failed; rt_last_error() has hipGetErrorString */
    RT_ERR_OUT_OF_MEMORY = -4,
    RT_ERR_UNSUPPORTED = -5       /* e.g. max_depth above RT_MAX_DEPTH */
} rt_status;

/* Largest max_depth accepted by the render entry points (reference uses 5,
 * src/main.rs:1098,1139; the benchmark uses 8). */
#define RT_MAX_DEPTH 32

/* ---- scene model: flat mirrors of the reference's types ------------------- */

/* geometric.rs:42-47  PositionNormalUV { position, normal, uv } */
typedef struct rt_vertex {
    float position[3];
    float normal[3];
    float uv[2];
} rt_vertex;

/* primitives.rs:26-29  Triangle<PositionNormalUV> { object_index, vertices } */
typedef struct rt_triangle {
    uint32_t object_index; /* usize narrowed to u32 */
    rt_vertex vertices[3];
} rt_triangle;

/* primitives.rs:15-24  Sphere { object_index, geometry: { center, radius } } */
typedef struct rt_sphere {
    uint32_t object_index;
    float center[3];
    float radius;
} rt_sphere;

/* materials.rs:70-83 stores two Rust closures in GenerativeMaterial; closures
 * cannot cross a C ABI, so the closure bodies that exist in the reference
 * (src/main.rs:848-863 and 1019-1026) are enumerated. */
typedef enum rt_diffuse_fn {
    RT_DIFFUSE_CONST = 0,      /* ColorMaterial: diffuse_color                    (materials.rs:33-37) */
    RT_DIFFUSE_STRIPE_V = 1,   /* ((uv.y * f) as i32 % 2 == 0) ? tex_a : tex_b    (main.rs:848-854)   */
    RT_DIFFUSE_STRIPE_SUM = 2  /* (((uv.x + uv.y) * f) as i32 % 2 == 0) ? a : b   (main.rs:1019-1025) */
} rt_diffuse_fn;

typedef enum rt_normal_fn {
    RT_NORMAL_CONST = 0,       /* ColorMaterial.normal / |uv| (0,0,1)             (main.rs:1026)      */
    RT_NORMAL_WAVE_U = 1       /* a = uv.x*nf*2*PI; v=(sin a,0,cos a); v.z<=0 ? -v : v (main.rs:855-863) */
} rt_normal_fn;

/* materials.rs:21-31 ColorMaterial (14 f32) + the enumerated closure parameters */
typedef struct rt_material {
    uint32_t diffuse_fn;       /* rt_diffuse_fn */
    uint32_t normal_fn;        /* rt_normal_fn  */
    float normal[3];           /* tangent-space bump normal, (0,0,1) for flat */
    float diffuse_color[3];
    float shiness;
    float specular_color[3];
    float smoothness;
    float transparency;
    float refraction_index;
    float opaque_decay;
    float tex_color_a[3];      /* generative diffuse: colour when the cell index is even */
    float tex_color_b[3];      /*                     colour otherwise                   */
    float tex_frequency;       /* 20.0 (main.rs:849) / 10.0 (main.rs:1020) */
    float normal_frequency;    /* 10.0 (main.rs:856) */
} rt_material;

/* lights.rs:6-30 */
typedef enum rt_light_kind {
    RT_LIGHT_DIRECTIONAL = 0,
    RT_LIGHT_SPOT = 1,
    RT_LIGHT_POINT = 2
} rt_light_kind;

typedef struct rt_light {
    uint32_t kind;        /* rt_light_kind */
    uint32_t has_origin;  /* Directional.origin is Option<Point3> (lights.rs:8); Spot/Point always 1 */
    float origin[3];
    float direction[3];   /* Directional, Spot */
    float angle;          /* Spot: radians */
    float softness;       /* Spot */
    float color[3];
} rt_light;

/* src/main.rs:130-137 World { objects, triangles, spheres, lights }.
 * Array ORDER is semantically significant: nearest-hit ties resolve to the
 * later primitive (main.rs:229-233, 298-302) and lights are summed in order. */
typedef struct rt_scene_desc {
    const rt_triangle *triangles; uint32_t n_triangles;
    const rt_sphere   *spheres;   uint32_t n_spheres;
    const rt_material *materials; uint32_t n_materials; /* one per Object (primitives.rs:8-10) */
    const rt_light    *lights;    uint32_t n_lights;
} rt_scene_desc;

/* src/main.rs:43-49 */
typedef struct rt_camera {
    float fovy;       /* radians */
    float center[3];
    float toward[3];
    float up[3];
    float near;
} rt_camera;

/* Frame + tile.  The reference renders the full frame (main.rs:1084-1089);
 * the tile fields are what image-tile sharding over several GPUs needs.
 * Rendered pixels: x in [x0,x1), y in {y0, y0+y_step, ...} < y1.
 * Output is compact: out[((row * (x1-x0)) + (x - x0)) * 3 + c], row = (y-y0)/y_step.
 * With x0=y0=0, x1=width, y1=height, y_step=1 that is the reference's
 * row-major [y*width + x] (image.rs:35-47).  A tile must hold fewer than 2^32
 * pixels (RT_ERR_UNSUPPORTED otherwise: render it as several tiles). */
typedef struct rt_frame {
    uint32_t width, height;
    int32_t  max_depth;      /* TraceState.depth at the root (main.rs:1098); an i32 tested with `depth <= 0`
                              * (main.rs:488, 669), so a negative value renders like 0, as in the reference */
    uint32_t x0, y0, x1, y1;
    uint32_t y_step;         /* >= 1 */
} rt_frame;

typedef struct rt_scene rt_scene; /* opaque: device-resident, immutable after create */

/* ---- entry points --------------------------------------------------------- */

/* ABI version of the loaded library (== RT_ABI_VERSION it was built with). */
int rt_abi_version(void);

/* Thread-local message for the last failing call on this thread ("" if none). */
const char *rt_last_error(void);

/* Number of HIP devices visible, or a negative rt_status. */
int rt_device_count(void);

/* Select the HIP device used by subsequent calls on this thread. */
int rt_set_device(int device);

/* Number of rows / pixels a frame's tile covers (host arithmetic only). */
uint32_t rt_frame_rows(const rt_frame *frame);
uint64_t rt_frame_pixels(const rt_frame *frame);

/* Upload a scene to the current device.  Replaces the construction of `World`
 * (src/main.rs:811-1075) as seen by the render loop: the library precomputes
 * the per-triangle face normal and plane constant (primitives.rs:36-47,
 * main.rs:202-203; pure functions of the vertices, so bit-identical to the
 * reference's per-ray recomputation) and owns the device copy until destroy. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out_scene);
int rt_scene_destroy(rt_scene *scene);

/* Whitted pass over one tile: replaces the par_iter closure at
 * src/main.rs:1090-1104 (shoot -> ray_trace(depth, contribution 1.0)).
 *   d_rgb        device pointer, rt_frame_pixels(frame)*3 floats, linear radiance
 *                BEFORE post_process (what ray_trace returns, main.rs:1101-1102).
 *   d_ray_count  device pointer to one u64 or NULL; the number of World::cast
 *                evaluations performed is ADDED to it (the reference only counts
 *                pixels, main.rs:1108).
 *   hip_stream   hipStream_t (as void*), NULL = default stream.  The launch is
 *                stream-ordered and asynchronous. */
int rt_render_whitted(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame,
                      float *d_rgb, unsigned long long *d_ray_count, void *hip_stream);

/* Same, with host buffers: allocates, launches, copies back and synchronises.
 * *h_ray_count is overwritten with the cast count of this call (may be NULL). */
int rt_render_whitted_host(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame,
                           float *h_rgb, unsigned long long *h_ray_count);

/* ---- distributed (stochastic / depth-of-field) pass ----------------------------

 * Replaces the par_iter_mut closure at src/main.rs:1131-1156 and the per-pixel RNG construction at
 * main.rs:1117-1127.  rt_rng is the device-resident array of per-pixel IsaacRng states of one tile
 * (rand 0.5 IsaacRng::new_from_u64(y * 2^33 + x); 516 u32 per pixel: mem[256], a, b, c, results[256],
 * index — that is what rt_rng_download returns; on the device a pixel has two such banks, the block in use and the
 * next one, which a look-ahead pass generates before the render kernels can need it: 4128 B per pixel).  It is mutable
 * and exclusive to one render call at a time (the reference hands each pixel `&mut` access, main.rs:1131); the random
 * stream continues from call to call. */
typedef struct rt_rng rt_rng;

int rt_rng_state_words(void);
int rt_rng_create(const rt_frame *frame, rt_rng **out_rng);
int rt_rng_destroy(rt_rng *rng);
/* Copy the states to the host (rt_frame_pixels * rt_rng_state_words u32) — for tests. */
int rt_rng_download(const rt_rng *rng, uint32_t *h_states);

/* n_epochs passes over the tile.  Per pixel and epoch: shoot_focus(focus, blur) (main.rs:1144-1149,
 * reference literals 3.0 / 0.04) -> cast -> distributed_ray_trace(depth = max_depth).
 *   d_accum    device, pixels*3 floats or NULL: every sample that passes the filter of main.rs:1157-1160
 *              (all three channels is_normal) is ADDED, in epoch order (img[at] = img[at] + photon,
 *              main.rs:1165).  Calling with n_epochs = 1 and running post_process in between reproduces
 *              the reference's per-epoch renormalisation (main.rs:1171).
 *   d_samples  device, n_epochs*pixels*3 floats or NULL: the raw sample of every (epoch, pixel).
 *   d_valid    device, n_epochs*pixels bytes or NULL: 1 where the sample passed the filter.
 *   d_ray_count as in rt_render_whitted.
 * At least one of d_accum / d_samples must be given. */
int rt_render_distributed(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                          rt_rng *rng, uint32_t n_epochs, float *d_accum, float *d_samples, unsigned char *d_valid,
                          unsigned long long *d_ray_count, void *hip_stream);

/* Same, with a host image: h_accum (rt_frame_pixels * 3 floats) is uploaded, n_epochs passes are ADDED to it as above
 * and it is copied back — `img[at] = img[at] + photon` of src/main.rs:1163-1167 for n_epochs epochs in one call, which is
 * what a host that keeps `img` in its own memory binds.  *h_ray_count is overwritten with the
 * World::cast count of this call (may be NULL).  Synchronises. */
int rt_render_distributed_host(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                               rt_rng *rng, uint32_t n_epochs, float *h_accum, unsigned long long *h_ray_count);

/* ---- several GPUs from one process (SURVEY §8e without Python or MPI) -------------------
 * Image rows are interleaved over the entries of `devices` exactly as homework-18-graphics-raytracer_amd/dist.py interleaves
 * them over ranks (entry r renders rows y0 + r*y_step, y0 + (r+n)*y_step, ...): the scene is replicated, the bands are
 * rendered concurrently, copied to pinned host memory and de-interleaved into the caller's image.  A device index may
 * repeat (several bands on one GPU).  This is the form a single-process host — the reference's main() — binds to use a
 * whole node; the one-process-per-GPU form over torch.distributed / RCCL is dist.py's.  Results are those of the
 * single-device entry points bit for bit: a pixel's value, and its random stream (seeded by IMAGE coordinates), do not
 * depend on which device renders it. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, rt_multi **out);
int rt_multi_destroy(rt_multi *m);
/* rt_render_whitted_host over the devices: h_rgb = rt_frame_pixels(frame) * 3 floats; *h_ray_count = casts of all devices. */
int rt_multi_render_whitted_host(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float *h_rgb, unsigned long long *h_ray_count);
/* rt_render_distributed_host over the devices.  The per-pixel generators live on the device that owns the pixel's row; they
 * are created on the first call for a frame and continue from call to call (main.rs:1131); a call with a different frame
 * starts new ones. */
int rt_multi_render_distributed_host(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float focus, float blur, uint32_t n_epochs,
                                     float *h_accum, unsigned long long *h_ray_count);
/* The same two, device-resident: the frame is assembled in DEVICE memory on devices[0] (d_rgb / d_accum: rt_frame_pixels * 3
 * floats there; d_ray_count, may be NULL, a u64 there that the casts of all devices are ADDED to), so that
 * rt_post_process_device and rt_encode_srgb8_device can follow on `hip_stream` (a stream of devices[0]) without a trip
 * through host memory.  A band of another device travels by hipMemcpyPeerAsync (xGMI) into a staging buffer on devices[0]
 * and is de-interleaved there; no RCCL, no host bounce.  Asynchronous like rt_render_whitted: on return everything is
 * enqueued; work on `hip_stream` after the call sees the finished frame.  rt_multi_render_distributed continues from the
 * sums already in d_accum (main.rs:1165), exactly as rt_render_distributed does.  Replaces main.rs:1105-1109 / 1162-1167. */
int rt_multi_render_whitted(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float *d_rgb, unsigned long long *d_ray_count, void *hip_stream);
int rt_multi_render_distributed(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float focus, float blur, uint32_t n_epochs,
                                float *d_accum, unsigned long long *d_ray_count, void *hip_stream);

/* ---- the step after the path, on the device --------------------

 * post_process (src/main.rs:748-762): divide the image in place by the 99th-percentile luma of its normal
 * lumas (exact radix select; no sort).  *d_divisor (device float, may be NULL) receives the divisor used, 0 when
 * the image was left untouched (no normal luma, or percentile <= f32::EPSILON).  Bit-identical to
 * rt_post_process in librt_host.so.  Stream-ordered. */
int rt_post_process_device(float *d_rgb, size_t n_pixels, float *d_divisor, void *hip_stream);
/* rt_post_process_device keeps a grow-only scratch buffer per (device, stream); this frees those of the current device
 * (synchronises it first). */
int rt_post_release(void);
/* The same post_process for a frame whose row bands live on SEVERAL ranks or devices (src/main.rs:748-762 needs the 99th
 * percentile of ALL lumas): the passes of rt_post_process_device one by one on caller-owned device memory — d_keys: n_pixels
 * u32, d_state: RT_POST_STATE_WORDS u32 — so that the caller can sum d_state over the bands between them (an all-reduce on the
 * same stream order) and every band ends with the same divisor without a host round trip:
 *     rt_post_keys_device    zeroes d_state; one order-preserving key per pixel of THIS band; d_state[0] = its count of normal lumas
 *       -> sum d_state[0] over the bands
 *     for pass = 0 .. 3:
 *         rt_post_hist_device   d_state[4 .. 259] += histogram of this band's keys that match the digits chosen so far
 *           -> sum d_state[4 .. 259] over the bands
 *         rt_post_pick_device   every band picks the same digit from the same sums (and clears the histogram)
 *     rt_post_scale_device   divides this band by the selected luma (*d_divisor as for rt_post_process_device)
 * With one band and no sums in between this IS rt_post_process_device.  The k-th smallest does not depend on the order, so
 * the value equals the one main.rs:754 indexes after its sort.  Stream-ordered; n_pixels may be 0 (a rank without rows). */
#define RT_POST_STATE_WORDS 260
int rt_post_keys_device(const float *d_rgb, size_t n_pixels, uint32_t *d_keys, uint32_t *d_state, void *hip_stream);
int rt_post_hist_device(const uint32_t *d_keys, size_t n_pixels, int pass, uint32_t *d_state, void *hip_stream);
int rt_post_pick_device(int pass, uint32_t *d_state, void *hip_stream);
int rt_post_scale_device(float *d_rgb, size_t n_pixels, const uint32_t *d_state, float *d_divisor, void *hip_stream);
/* Image::<Srgb<u8>>::convert_from (src/image.rs:55-66): linear f32 -> sRGB-encoded u8, n_values = 3*pixels. */
int rt_encode_srgb8_device(const float *d_rgb, size_t n_values, unsigned char *d_out, void *hip_stream);
/* PhotonAccumulator (src/photon.rs:9-34; unused by the reference's main()) on the device, bit-identical to
 * rt_accumulate / rt_accumulator_resolve in librt_host.so: d_sum (3 f32 per pixel) and d_weight (1 f32 per pixel) start
 * at zero; rt_accumulate_device applies accumulate() for every sample whose filter flag is set (d_samples, d_valid: the
 * n_epochs x n_pixels outputs of rt_render_distributed), in epoch order; rt_accumulator_resolve_device writes
 * sum / weight, black while weight < f32::EPSILON.  Stream-ordered. */
int rt_accumulate_device(const float *d_samples, const unsigned char *d_valid, uint32_t n_epochs, size_t n_pixels, float *d_sum,
                         float *d_weight, void *hip_stream);
int rt_accumulator_resolve_device(const float *d_sum, const float *d_weight, size_t n_pixels, float *d_rgb, void *hip_stream);

/* ---- diagnostics ------------------------------------------------------------ */

/* Which kernel renders the Whitted pass (process-wide; same results bit for bit):
 *   18 (default)  the persistent wavefront kernel (csrc/rt_pwf.hip): one kernel whose workgroups keep queues of single-cast
 *                 work items (a ray_trace activation's own cast, one cast of get_refract, one shadow cast of get_shade)
 *                 and fold the results bottom-up at the end; a frame that does not fit its arenas is rendered by the
 *                 per-pixel kernel (2) within the same call
 *   2             the per-pixel kernel (csrc/rt_kernels.hip): one work-item per primary ray, a wave = an 8x8 tile, the
 *                 recursion unrolled into a per-lane state machine; triangle records fetched with wave-uniform scalar loads
 *   3, 19         as 2 / 18 with the per-pixel kernel's triangle records staged in LDS once per workgroup (the north_star's
 *                 wording; slower than the scalar fetches: DESIGN.md, profiles/)
 * or the value of the RT_AMD_VARIANT environment variable at load.  Anything else is RT_ERR_INVALID_ARGUMENT. */
/* Process-wide switches: A/B knobs of the launch plumbing and test hooks, none of which changes a result.  Each is an integer
 * named like the environment variable that seeds it — RT_AMD_DIST_PIPELINE, RT_AMD_DIST_WS_MB, RT_AMD_RNG_LOOKAHEAD,
 * RT_AMD_RNG_OVERLAP, RT_AMD_DIST_BY_COST, RT_AMD_DIST_OWN_FIRST, RT_AMD_DIST_PREP_FIRST, RT_AMD_DIST_SPLIT, RT_AMD_DIST_STATIC,
 * RT_AMD_DIST_CHAIN_WAVES, RT_AMD_SHADE_TILE, RT_AMD_SHADE_SORT, RT_AMD_DIAG_WS_REFUSE,
 * RT_AMD_MULTI_FORCE_STAGE, RT_AMD_BFS_WALK_TRIANGLES (read by rt_scene_create), RT_AMD_WF_SHARE, RT_AMD_DIAG_BFS_CAP (INTEGRATION.md says what each does).  The environment is read ONCE per process, at the first use;
 * after that only this call changes a switch: value = decimal integer, NULL or "" = unset (the library's own choice).  Render
 * calls read the switches without locks: set them between calls, not during one. */
int rt_set_option(const char *name, const char *value);
int rt_set_variant(int variant);
int rt_get_variant(void);

/* Persistent-wavefront path (variant bit 4): size of its arenas, in ray_trace activations per tile pixel (default 6,
 * RT_AMD_WF_NODES_PER_PIXEL; the reference scene needs 3.4 at depth 8; about 200 B of device memory each, rounded up
 * to a power of two per workgroup).  A frame that needs more is detected on the device and rendered by the per-pixel
 * kernel within the same call, so the budget changes speed and memory only, never results.  Arenas are never smaller
 * than 1.5 MB per workgroup unless the budget is below 4. */
int rt_set_wavefront_budget(unsigned nodes_per_pixel);

/* rt_render_distributed has two organisations with bit-identical results (samples, flags, RNG states, cast counts):
 *   1 (default)  three kernels per batch of epochs — the scatter chain with all random draws, every get_shade it
 *                asked for, the unwind + filter + accumulation — over a per-stream workspace (852 B per sample at depth 8,
 *                at most RT_AMD_DIST_WS_MB MiB, default 32768 for a call of several batches — which uses two workspaces in turn,
 *                batch k's get_shade and unwind kernels running beside batch k+1's chain kernel on streams the rt_rng owns, all of
 *                them behind the caller's stream again when the call returns (RT_AMD_DIST_PIPELINE=0: one workspace, in line) — and
 *                16384 for one of a single batch; a batch is as many epochs as fit, 16 at most, and fewer if the device cannot
 *                provide the memory — down to organisation 0 when not even one epoch fits; a workspace that holds at least half
 *                the batch wanted is kept rather than replaced);
 *   2            round 2's queued chain kernel: measured slower than 1 twice and removed in round 3 — the value now selects 1;
 *   0            one kernel, a lane stays on its pixel through chain, shades and unwind (no workspace).
 * -1 restores the default / the RT_AMD_DIST_SPLIT environment variable. */
int rt_set_distributed_split(int on);

/* Timing of the dominant (render) kernel alone: while enabled, each rt_render_whitted call records a HIP
 * event pair on its stream right around that kernel (a call may also launch a small probe kernel);
 * rt_profile_read synchronises the device, returns the summed elapsed milliseconds and the number of
 * launches since the last read, and resets. */
int rt_profile_enable(int on);
int rt_profile_read(double *kernel_ms_sum, unsigned *n_launches);
/* The same for the depth-of-field pass: while profiling is enabled, rt_render_distributed (the chain / shade / unwind organisation)
 * brackets every kernel launch with an event pair on the stream the launch is put on; this synchronises the device and returns, per
 * kernel — [0] the generators' look-ahead (rng_scan + rng_prepare), [1] dist_chain_kernel, [2] the shade kernel, [3] dist_unwind_kernel —
 * the summed milliseconds and the launches since the last read, and resets.  Kernels of a pipelined call overlap: these are each
 * kernel's own durations under that overlap (what a kernel trace shows), not shares of the call's wall time. */
int rt_profile_read_distributed(double ms_sum[4], unsigned n_launches[4]);

/* The deterministic f32 math the path computes with (csrc/rt_detmath.h),
 * evaluated element-wise on the host or on the device, so tests can prove the
 * two return identical bits.  op is an rt_math_op; y is ignored by unary ops.
 * The *_HI/_LO ops return the two 32-bit halves of a binary64 result
 * reinterpreted as f32 bit patterns (for checking f64 sqrt/div rounding). */
typedef enum rt_math_op {
    RT_MATH_SIN = 0, RT_MATH_COS = 1, RT_MATH_TAN = 2, RT_MATH_ACOS = 3,
    RT_MATH_ATAN2 = 4,   /* atan2(x, y): x is the ordinate */
    RT_MATH_POW = 5,     /* pow(x, y) */
    RT_MATH_F32_DIV = 6, RT_MATH_F32_SQRT = 7,
    RT_MATH_F64_SQRT_HI = 8, RT_MATH_F64_SQRT_LO = 9,   /* sqrt((double)x * (double)y) */
    RT_MATH_F64_DIV_HI = 10, RT_MATH_F64_DIV_LO = 11,   /* (double)x / (double)y */
    RT_MATH_ROUND = 12,
    RT_MATH_SINCOS_SIN = 13, RT_MATH_SINCOS_COS = 14    /* the two results of the fused sincosf the kernels call */
} rt_math_op;
int rt_math_eval_host(int op, const float *x, const float *y, float *out, size_t n);
int rt_math_eval_device(int op, const float *h_x, const float *h_y, float *h_out, size_t n);

/* The node array rt_scene_create builds over the triangles for the intersection loop (csrc/rt_device_scene.h: a pre-order
 * array of leaves — runs of consecutive triangles — and inner nodes with skip pointers), computed on the host without
 * touching a device: six words per node — first triangle, count (0: inner node), n_normals (0: plain leaf, always
 * visited; 0xffffffff: a normal cone), skip_to, the pair-wise dealing word (chunk | chunks << 8 | sub-jobs per pass << 16; 0:
 * never pair-wise), 0.  Writes at most cap_nodes nodes, always reports the count.  For tests of the builder's invariants. */
int rt_scene_describe_nodes(const rt_scene_desc *desc, uint32_t *out_words, uint32_t cap_nodes, uint32_t *n_nodes);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
