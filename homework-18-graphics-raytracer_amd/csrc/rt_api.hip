/*
 * rt_api.hip — implementation of the C ABI in include/rt_amd.h.
 *
 * Host-side plumbing only: argument validation, the one-time scene upload
 * (with the per-primitive precompute described in rt_device_scene.h), the
 * per-frame camera basis (Camera::shoot's ray-independent part, main.rs:85-92)
 * and stream-ordered kernel launches.  There is NO CPU rendering fallback: if
 * the HIP runtime or a device is missing every render entry point fails with
 * RT_ERR_NO_DEVICE / RT_ERR_HIP.
 */
#include "rt_api_internal.h"

/* Process-wide settings (rt_set_*): read by render calls on any thread, so they are atomics.  A value < 0 means "not set
 * yet": the first reader resolves it from the environment (two threads doing that at once compute the same value). */
static std::atomic<int> g_wf_nodes_per_pixel{-1};
static std::atomic<const uint32_t *> g_diag_tile_order{nullptr};
extern "C" void rt_diag_set_tile_order(const void *device_ptr) { g_diag_tile_order.store(static_cast<const uint32_t *>(device_ptr)); }
static std::atomic<uint32_t *> g_diag_tile_cost{nullptr};
extern "C" void rt_diag_set_tile_cost(void *device_ptr) { g_diag_tile_cost.store(static_cast<uint32_t *>(device_ptr)); }
#ifdef RT_DIAG_TIMELINE
static unsigned long long *g_diag_timeline = nullptr;
extern "C" void rt_diag_set_timeline(void *device_ptr) { g_diag_timeline = static_cast<unsigned long long *>(device_ptr); }
#endif

#ifndef RT_BFS_WALK_TRIANGLES_DEFAULT
#define RT_BFS_WALK_TRIANGLES_DEFAULT 8192 /* measured (profiles/r04_scene_sweep_*_bfs.jsonl against *_wave_uniform.jsonl): the breadth-first walk is 1.3-1.7x
                                              * ahead at 9 244 triangles, 2.1-2.3x at 36 892, 8-17x at 147 484; at 2 332 it loses (1.2x on flat meshes, 2x
                                              * on spherized ones) */
#endif
/* the switches of rt_kernels.h `Option`: name (also the environment variable that seeds it), whether it has a value, the value */
static const char *const OPT_NAMES[rt::OPT_COUNT] = {
    "RT_AMD_RNG_LOOKAHEAD", "RT_AMD_RNG_OVERLAP", "RT_AMD_DIST_PIPELINE", "RT_AMD_DIST_BY_COST", "RT_AMD_DIST_OWN_FIRST", "RT_AMD_DIST_PREP_FIRST",
    "RT_AMD_DIST_WS_MB", "RT_AMD_DIAG_WS_REFUSE", "RT_AMD_DIST_STATIC", "RT_AMD_DIST_CHAIN_WAVES", "RT_AMD_SHADE_TILE", "RT_AMD_SHADE_SORT",
    "RT_AMD_MULTI_FORCE_STAGE", "RT_AMD_DIST_SPLIT", "RT_AMD_BFS_WALK_TRIANGLES", "RT_AMD_WF_SHARE", "RT_AMD_DIAG_BFS_CAP"};
static std::atomic<int> g_opt_set[rt::OPT_COUNT];
static std::atomic<long long> g_opt_val[rt::OPT_COUNT];
static std::once_flag g_opt_once;
static void options_from_environment() {
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < rt::OPT_COUNT; ++i) {
            const char *v = getenv(OPT_NAMES[i]);
            if (v && *v) {
                g_opt_val[i].store(strtoll(v, nullptr, 10));
                g_opt_set[i].store(1);
            }
        }
    });
}
namespace rt {
long long option(Option id, long long unset) {
    options_from_environment();
    return g_opt_set[id].load(std::memory_order_acquire) ? g_opt_val[id].load(std::memory_order_relaxed) : unset;
}
} /* namespace rt */

static thread_local std::string g_error;
static thread_local bool t_prof_off = false; /* rt_api_internal.h ProfilingOff */
std::string &last_error() { return g_error; }
bool &profiling_off_flag() { return t_prof_off; }
static std::atomic<int> g_variant{-1};

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
int fail_hip(const char *what, hipError_t e) {
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE
           : (e == hipErrorOutOfMemory)                           ? RT_ERR_OUT_OF_MEMORY
                                                                  : RT_ERR_HIP;
}


/* 2 / 3: the per-pixel kernel with scalar / LDS triangle fetches; 18 / 19: the persistent wavefront kernel (with that as its fallback) */
static bool variant_ok(int v) { return v == 2 || v == 3 || v == 18 || v == 19; }
static int current_variant() {
    int v = g_variant.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("RT_AMD_VARIANT");
        v = (e && *e) ? atoi(e) : RT_VARIANT_DEFAULT;
        if (!variant_ok(v)) v = RT_VARIANT_DEFAULT;
        g_variant.store(v, std::memory_order_relaxed);
    }
    return v;
}
static int current_wf_budget() {
    int v = g_wf_nodes_per_pixel.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("RT_AMD_WF_NODES_PER_PIXEL");
        v = (e && *e) ? atoi(e) : 6; /* the reference scene needs 3.4 at depth 8 */
        if (v < 1 || v > 4096) v = 6;
        g_wf_nodes_per_pixel.store(v, std::memory_order_relaxed);
    }
    return v;
}

bool frame_ok(const rt_frame *f) {
    return f && f->width > 0 && f->height > 0 && f->y_step >= 1 && f->x0 < f->x1 && f->y0 < f->y1 && f->x1 <= f->width &&
           f->y1 <= f->height;
}
/* the kernels and launchers index a tile's pixels with 32-bit arithmetic: a tile of 2^32 pixels or more is refused
 * rather than wrapped (65536 x 65536 would wrap to 0 and "render" nothing) */
bool frame_fits(const rt_frame *f) {
    const uint64_t rows = ((uint64_t)f->y1 - f->y0 + f->y_step - 1) / f->y_step;
    return rows * (uint64_t)(f->x1 - f->x0) < (1ull << 32) - 64u;
}

static std::atomic<bool> g_prof_on{false}; /* rt_profile_enable */
bool profiling_on() { return g_prof_on.load() && !t_prof_off; }

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip("hipGetDeviceCount", e);
    return n;
}

int rt_set_device(int device) {
    RT_HIP(hipSetDevice(device));
    return RT_OK;
}

int rt_set_option(const char *name, const char *value) {
    if (!name) return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_option: null name");
    options_from_environment(); /* first, so that a later first use does not overwrite what is set here */
    for (int i = 0; i < rt::OPT_COUNT; ++i) {
        if (strcmp(name, OPT_NAMES[i]) != 0) continue;
        if (value && *value) {
            char *end = nullptr;
            const long long v = strtoll(value, &end, 10);
            if (end == value || *end != '\0') return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_option: the value is not an integer");
            g_opt_val[i].store(v, std::memory_order_relaxed);
            g_opt_set[i].store(1, std::memory_order_release);
        } else {
            g_opt_set[i].store(0, std::memory_order_release);
        }
        return RT_OK;
    }
    return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_set_option: unknown option ") + name);
}

int rt_set_variant(int variant) {
    if (!variant_ok(variant)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_variant: 2, 3, 18 or 19 (include/rt_amd.h)");
    g_variant.store(variant);
    return RT_OK;
}
int rt_get_variant(void) { return current_variant(); }

int rt_set_wavefront_budget(unsigned nodes_per_pixel) {
    if (nodes_per_pixel < 1u || nodes_per_pixel > 4096u) return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_wavefront_budget: 1..4096 nodes per pixel");
    g_wf_nodes_per_pixel.store((int)nodes_per_pixel);
    return RT_OK;
}

/* ---- profiling of the dominant kernel ----
 * bench.py's roofline needs the duration of the render kernel alone (a call may also launch the small probe
 * kernel).  When enabled, every rt_render_whitted call records a HIP event pair on the launch stream right
 * around that kernel; rt_profile_read() synchronises and sums the elapsed times.
 * Thread safety: the event list is under a mutex; the pair a call is recording into travels in thread-local state of the
 * calling thread (rt_kernels.hip), so render calls on several host threads do not see each other's events. */
static std::mutex g_prof_mutex;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static size_t g_prof_used = 0;


int rt_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on.store(on != 0);
    g_prof_used = 0;
    dist_profile_reset();
    rt::set_main_kernel_events(nullptr, nullptr);
    return RT_OK;
}

int rt_profile_read(double *kernel_ms_sum, unsigned *n_launches) {
    if (!kernel_ms_sum || !n_launches) return fail(RT_ERR_INVALID_ARGUMENT, "rt_profile_read: null argument");
    RT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    double sum = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, g_prof_events[i].first, g_prof_events[i].second));
        sum += ms;
    }
    *kernel_ms_sum = sum;
    *n_launches = (unsigned)g_prof_used;
    g_prof_used = 0;
    return RT_OK;
}

uint32_t rt_frame_rows(const rt_frame *f) {
    if (!frame_ok(f)) return 0;
    return (f->y1 - f->y0 + f->y_step - 1) / f->y_step;
}
uint64_t rt_frame_pixels(const rt_frame *f) {
    if (!frame_ok(f)) return 0;
    return (uint64_t)rt_frame_rows(f) * (uint64_t)(f->x1 - f->x0);
}

int rt_scene_create(const rt_scene_desc *desc, rt_scene **out_scene) {
    if (!desc || !out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null argument");
    *out_scene = nullptr;
    SceneLayout layout;
    {
        const int rc = layout_scene(desc, layout);
        if (rc != RT_OK) return rc;
    }
    const std::vector<rt::DevTri> &tris = layout.tris;
    const std::vector<rt::DevTriAttr> &attrs = layout.attrs;
    const std::vector<rt::DevSegment> &segments = layout.segments;
    const std::vector<rt::DevTriHead> &heads = layout.heads;
    const std::vector<rt::DevSphere> &spheres = layout.spheres;
    const double scene_extent = layout.scene_extent;

    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t off_tris = 0;
    const size_t off_attrs = off_tris + up(tris.size() * sizeof(rt::DevTri));
    const size_t off_spheres = off_attrs + up(attrs.size() * sizeof(rt::DevTriAttr));
    const size_t off_mats = off_spheres + up(spheres.size() * sizeof(rt::DevSphere));
    const size_t off_lights = off_mats + up(desc->n_materials * sizeof(rt_material));
    const size_t off_segments = off_lights + up(desc->n_lights * sizeof(rt_light));
    const size_t off_heads = off_segments + up(segments.size() * sizeof(rt::DevSegment));
    const size_t off_light_aux = off_heads + up(heads.size() * sizeof(rt::DevTriHead));
    /* the node tree once more in LEVEL ORDER (rt_kernels.h KernelScene::bfs_nodes): the top-level nodes first, then every inner node's
     * children one after the other — an inner node's record then names its children as a range [first, first + skip_to) of the same
     * array and a breadth-first walk needs no child list beside it.  From the pre-order array: the children of inner node k are
     * k + 1, then each one's skip_to, up to k's own skip_to; the top-level nodes likewise from 0. */
    std::vector<rt::DevSegment> bfs_nodes;
    uint32_t bfs_top = 0u;
    {
        std::vector<uint32_t> order; /* pre-order index of the node at each level-order position */
        order.reserve(segments.size());
        for (uint32_t k = 0; k < (uint32_t)segments.size(); k = segments[k].skip_to) { order.push_back(k); bfs_top += 1u; }
        bfs_nodes.reserve(segments.size());
        for (size_t at = 0; at < order.size(); ++at) {
            const uint32_t k = order[at];
            rt::DevSegment g = segments[k];
            if (g.count == 0u) { /* inner: first = its first child's position, skip_to = how many */
                g.first = (uint32_t)order.size();
                g.skip_to = 0u;
                for (uint32_t j = k + 1u; j < segments[k].skip_to && j < (uint32_t)segments.size(); j = segments[j].skip_to) { order.push_back(j); g.skip_to += 1u; }
            }
            bfs_nodes.push_back(g);
        }
    }
    /* ... and what the walk reads of every node and every triangle as arrays of 16-byte pieces (KernelScene::bfs_soa): the lanes of a
     * pass hold consecutive nodes or triangles, so each of its loads is one contiguous kilobyte */
    std::vector<float> bfs_soa((3u * bfs_nodes.size() + 2u * heads.size()) * 4u, 0.0f);
    for (size_t k = 0; k < bfs_nodes.size(); ++k) {
        memcpy(&bfs_soa[4u * k], &bfs_nodes[k], 16);                                /* first, count, n_normals, r2_hi */
        memcpy(&bfs_soa[4u * (bfs_nodes.size() + k)], &bfs_nodes[k].c[0], 16);       /* centre, child count */
        memcpy(&bfs_soa[4u * (2u * bfs_nodes.size() + k)], &bfs_nodes[k].normals[0][0], 16); /* the first plane direction, or the cone */
    }
    for (size_t t = 0; t < heads.size(); ++t) {
        memcpy(&bfs_soa[4u * (3u * bfs_nodes.size() + t)], &heads[t].n[0], 16);                  /* plane */
        memcpy(&bfs_soa[4u * (3u * bfs_nodes.size() + heads.size() + t)], &heads[t].bc[0], 16);  /* bounding sphere */
    }
    const size_t off_bfs_nodes = off_light_aux + up(desc->n_lights * sizeof(rt::LightAux));
    const size_t off_bfs_soa = off_bfs_nodes + up(bfs_nodes.size() * sizeof(rt::DevSegment));
    const size_t total = off_bfs_soa + up(bfs_soa.size() * sizeof(float)) + 256;
    /* a spot light's cone edge as a cosine, with margins (rt_shade.h light_asks); anything unusual switches the shortcut off */
    std::vector<rt::LightAux> light_aux(desc->n_lights);
    for (uint32_t i = 0; i < desc->n_lights; ++i) {
        light_aux[i].cos_in = std::numeric_limits<float>::infinity();
        light_aux[i].cos_out = -std::numeric_limits<float>::infinity();
        const double a = (double)desc->lights[i].angle;
        if (desc->lights[i].kind == RT_LIGHT_SPOT && a > 1e-3 && a < 3.14) {
            light_aux[i].cos_in = std::nextafter((float)(cos(a) + 1e-4), std::numeric_limits<float>::infinity());
            light_aux[i].cos_out = std::nextafter((float)(cos(a) - 1e-4), -std::numeric_limits<float>::infinity());
        }
    }

    std::vector<unsigned char> blob(total, 0);
    if (!tris.empty()) memcpy(&blob[off_tris], tris.data(), tris.size() * sizeof(rt::DevTri));
    if (!attrs.empty()) memcpy(&blob[off_attrs], attrs.data(), attrs.size() * sizeof(rt::DevTriAttr));
    if (!spheres.empty()) memcpy(&blob[off_spheres], spheres.data(), spheres.size() * sizeof(rt::DevSphere));
    if (desc->n_materials) memcpy(&blob[off_mats], desc->materials, desc->n_materials * sizeof(rt_material));
    if (desc->n_lights) memcpy(&blob[off_lights], desc->lights, desc->n_lights * sizeof(rt_light));
    if (!segments.empty()) memcpy(&blob[off_segments], segments.data(), segments.size() * sizeof(rt::DevSegment));
    if (!heads.empty()) memcpy(&blob[off_heads], heads.data(), heads.size() * sizeof(rt::DevTriHead));
    if (!light_aux.empty()) memcpy(&blob[off_light_aux], light_aux.data(), light_aux.size() * sizeof(rt::LightAux));
    if (!bfs_nodes.empty()) memcpy(&blob[off_bfs_nodes], bfs_nodes.data(), bfs_nodes.size() * sizeof(rt::DevSegment));
    if (!bfs_soa.empty()) memcpy(&blob[off_bfs_soa], bfs_soa.data(), bfs_soa.size() * sizeof(float));

    rt_scene *sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_OUT_OF_MEMORY, "rt_scene_create: host allocation failed");
    sc->d_blob = nullptr;
    hipError_t e = hipGetDevice(&sc->device);
    if (e == hipSuccess) e = hipMalloc(&sc->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(sc->d_blob, blob.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (sc->d_blob) (void)hipFree(sc->d_blob);
        delete sc;
        return fail_hip("rt_scene_create: device upload", e);
    }
    unsigned char *base = static_cast<unsigned char *>(sc->d_blob);
    sc->ks.tris = reinterpret_cast<const rt::DevTri *>(base + off_tris);
    sc->ks.attrs = reinterpret_cast<const rt::DevTriAttr *>(base + off_attrs);
    sc->ks.spheres = reinterpret_cast<const rt::DevSphere *>(base + off_spheres);
    sc->ks.materials = reinterpret_cast<const rt_material *>(base + off_mats);
    sc->ks.lights = reinterpret_cast<const rt_light *>(base + off_lights);
    sc->ks.n_triangles = desc->n_triangles;
    sc->ks.n_spheres = desc->n_spheres;
    sc->ks.n_materials = desc->n_materials;
    sc->ks.n_lights = desc->n_lights;
    sc->ks.segments = reinterpret_cast<const rt::DevSegment *>(base + off_segments);
    sc->ks.n_segments = (uint32_t)segments.size();
    sc->ks.heads = reinterpret_cast<const rt::DevTriHead *>(base + off_heads);
    sc->ks.light_aux = reinterpret_cast<const rt::LightAux *>(base + off_light_aux);
    sc->ks.filter_origin2 = (float)(16.0 * scene_extent * scene_extent); /* |origin| <= 4 x extent */
    sc->ks.bfs_nodes = reinterpret_cast<const rt::DevSegment *>(base + off_bfs_nodes);
    sc->ks.bfs_soa = reinterpret_cast<const float4 *>(base + off_bfs_soa);
    sc->ks.bfs_top = bfs_top;
    {   /* a scene this large is walked breadth-first by the wavefront kernel (rt_cast_bfs.h cast_bfs): node ids must fit 26 bits */
        const long long at = rt::option(rt::OPT_BFS_WALK_TRIANGLES, RT_BFS_WALK_TRIANGLES_DEFAULT);
        sc->ks.bfs_walk = at > 0 && (long long)desc->n_triangles >= at && segments.size() < (1u << 26) ? 1u : 0u;
    }
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device);
    if (e != hipSuccess || cus <= 0) cus = 256;
    sc->resident_waves = (uint32_t)cus * 4u * (uint32_t)RT_MIN_WAVES;
    sc->pwf_workgroups = (uint32_t)cus * (uint32_t)rt::pwf_workgroups_per_cu(7168u, 8192u, sc->ks.bfs_walk != 0u);
    *out_scene = sc;
    return RT_OK;
}

int rt_scene_destroy(rt_scene *scene) {
    if (!scene) return RT_OK;
    hipError_t e = hipSuccess;
    for (auto &kv : scene->workspaces) {
        if (kv.second.d_counters) (void)hipFree(kv.second.d_counters);
        if (kv.second.d_pwf) (void)hipFree(kv.second.d_pwf);
        if (kv.second.d_bfs) (void)hipFree(kv.second.d_bfs);
        if (kv.second.d_split) (void)hipFree(kv.second.d_split);
    }
    if (scene->d_blob) e = hipFree(scene->d_blob);
    delete scene;
    if (e != hipSuccess) return fail_hip("rt_scene_destroy: hipFree", e);
    return RT_OK;
}

} /* extern "C" */
int make_kernel_frame(const rt_camera *camera, const rt_frame *frame, rt::KernelFrame *kf) {
    if (!camera) return fail(RT_ERR_INVALID_ARGUMENT, "render: null camera");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "render: bad frame (need 0 <= x0 < x1 <= width, 0 <= y0 < y1 <= height, y_step >= 1)");
    if (!frame_fits(frame)) return fail(RT_ERR_UNSUPPORTED, "render: tile of 2^32 pixels or more (render it as several tiles)");
    if (frame->max_depth > RT_MAX_DEPTH) return fail(RT_ERR_UNSUPPORTED, "render: max_depth above RT_MAX_DEPTH");
    /* max_depth < 0 is valid and renders as 0: TraceState.depth is an i32 tested with `depth <= 0` (main.rs:488, 669) */
    using rt::V3;
    /* Camera::shoot, main.rs:85-92: the ray-independent part */
    const V3 toward = rt::normalize(rt::v3p(camera->toward));
    const V3 right = rt::normalize(rt::cross(toward, rt::v3p(camera->up)));
    const V3 up = rt::normalize(rt::cross(right, toward));
    const float th = rtdm::tanf(camera->fovy / 2.0f);
    const V3 x = th * right;
    const V3 y = th * up;
    const V3 origin = rt::v3p(camera->center) + toward * camera->near;
    kf->cols = frame->x1 - frame->x0;
    kf->rows = rt_frame_rows(frame);
    kf->x0 = frame->x0;
    kf->y0 = frame->y0;
    kf->y_step = frame->y_step;
    kf->max_depth = frame->max_depth;
    kf->half_height = (float)frame->height / 2.0f;
    kf->half_width = (float)frame->width / 2.0f;
    kf->height_f = (float)frame->height;
    kf->cam_origin[0] = origin.x; kf->cam_origin[1] = origin.y; kf->cam_origin[2] = origin.z;
    kf->cam_x[0] = x.x; kf->cam_x[1] = x.y; kf->cam_x[2] = x.z;
    kf->cam_y[0] = y.x; kf->cam_y[1] = y.y; kf->cam_y[2] = y.z;
    kf->cam_toward[0] = toward.x; kf->cam_toward[1] = toward.y; kf->cam_toward[2] = toward.z;
    const V3 origin_focus = rt::v3p(camera->center) + rt::normalize(toward) * camera->near; /* main.rs:118-119 */
    kf->cam_origin_focus[0] = origin_focus.x; kf->cam_origin_focus[1] = origin_focus.y; kf->cam_origin_focus[2] = origin_focus.z;
    return RT_OK;
}

extern "C" {
int rt_render_whitted(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float *d_rgb,
                      unsigned long long *d_ray_count, void *hip_stream) {
    if (!scene || !d_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_whitted: null argument");
    rt::KernelFrame kf;
    int rc = make_kernel_frame(camera, frame, &kf);
    if (rc != RT_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int variant = current_variant();
    /* the persistent-wavefront path packs the ray's face mode and the depth left next to a 21-bit primitive id */
    if ((variant & RT_VARIANT_PWF) && (uint64_t)scene->ks.n_triangles + scene->ks.n_spheres >= (1ull << 21)) variant &= ~RT_VARIANT_PWF;
    if ((variant & RT_VARIANT_PWF) && (scene->ks.n_lights >= (1u << 14) || scene->ks.n_materials >= (1u << 16))) variant &= ~RT_VARIANT_PWF; /* a SHADE item's word: 16 + 14 bits */
    const int wf_budget = current_wf_budget();
    rt::PwParams pw;
    memset(&pw, 0, sizeof pw);
    uint32_t pw_groups = 0, pw_band_rows = 0, pw_parity = 0;
    bool pw_init = true;
    rt::KernelQueues qs;
    memset(&qs, 0, sizeof qs);
    {
        rt_scene *mut = const_cast<rt_scene *>(scene); /* workspaces are the only mutable part of a scene */
        std::lock_guard<std::mutex> lock(mut->ws_mutex);
        Workspace &ws = mut->workspaces[stream];
        RT_HIP(ensure_counters(ws));
        if (variant & RT_VARIANT_PWF) {
            /* The kernel keeps one LDS word per 64 ring slots and per 64 nodes, so an arena holds 64 K ring slots at most.
             * A tile too large for that at the budget asked for (beyond ~8 Mpixel at 6 nodes per pixel) is rendered as
             * several bands of rows, one launch each, through the same workspace. */
            const uint64_t ring_max = 1ull << 16;
            const uint64_t pixels = (uint64_t)kf.cols * kf.rows;
            uint64_t groups = (pixels + 63u) / 64u; /* a workgroup fetches one to eight tiles at a time */
            if (groups > scene->pwf_workgroups) groups = scene->pwf_workgroups;
            {   /* frames in flight on several streams: each launch takes a part of the device, so that they run side by side */
                const long long share = rt::option(rt::OPT_WF_SHARE, 1);
                const uint64_t cap = share > 1 ? ((uint64_t)scene->pwf_workgroups + (uint64_t)share - 1u) / (uint64_t)share : (uint64_t)scene->pwf_workgroups;
                if (groups > cap) groups = cap;
            }
            if (groups < 1) groups = 1;
            const uint64_t max_pixels = (ring_max - 1024u) * groups / (uint64_t)wf_budget;
            pw_band_rows = kf.rows;
            if (pixels > max_pixels) {
                const uint64_t n_bands = (pixels + max_pixels - 1) / max_pixels;
                uint64_t rows = (kf.rows + n_bands - 1) / n_bands;
                rows = (rows + 7u) & ~7ull; /* whole 8-row tile bands */
                pw_band_rows = (uint32_t)(rows < kf.rows ? rows : kf.rows);
            }
            const uint64_t band_pixels = (uint64_t)kf.cols * pw_band_rows;
            const uint64_t want = (band_pixels * (uint64_t)wf_budget + groups - 1) / groups; /* nodes per arena */
            /* tiles are handed out dynamically, so a workgroup may end up with several times the average: arenas have a
             * floor of 8192 ring slots (1.5 MB) however small the frame (budgets below 4 waive it: tests of the fallback) */
            uint64_t ring = wf_budget >= 4 ? 8192 : 2048;
            while (ring < want + 1024u && ring < ring_max) ring <<= 1;
            pw.ring_cap = (uint32_t)ring;
            /* (the budget sizes the rings; an arena may use all of its ring's worth of nodes: tiles are handed out
             * dynamically, and a workgroup that met expensive ones needs more than the average) */
            pw.node_cap = (uint32_t)(ring - 1024u);
            pw.tile_reserve = 10;
            pw.arena_stride = (rt::pwf_arena_bytes(pw.node_cap, pw.ring_cap) + 255u) & ~(size_t)255u;
            const size_t need = 512 + (size_t)groups * pw.arena_stride;
            if (need > ws.pwf_bytes) {
                if (ws.d_pwf) (void)hipFree(ws.d_pwf);
                ws.d_pwf = nullptr;
                ws.pwf_bytes = 0;
                ws.pw_ready = false;
                /* OPT_DIAG_WS_REFUSE: test hook, pretend the allocation fails */
                if (rt::option(rt::OPT_DIAG_WS_REFUSE, 0) > 0 || hipMalloc(&ws.d_pwf, need) != hipSuccess) {
                    (void)hipGetLastError();
                    ws.d_pwf = nullptr;
                } else {
                    ws.pwf_bytes = need;
                }
            }
            /* the grid (groups) and the arenas (groups x arena_stride) come from the same numbers, under the same lock; should they ever
             * disagree with the allocation the launch must not happen (round 1's memory fault was a kernel writing one page past a workspace) */
            if (ws.d_pwf != nullptr && 512 + (size_t)groups * pw.arena_stride > ws.pwf_bytes) {
                (void)hipFree(ws.d_pwf);
                ws.d_pwf = nullptr;
                ws.pwf_bytes = 0;
                ws.pw_ready = false;
            }
            if (ws.d_pwf != nullptr && scene->ks.bfs_walk != 0u) { /* the breadth-first walk's item and job lists: per wave of the grid */
                const size_t words = (size_t)groups * 8u * rt::pwf_bfs_scratch_words_per_wave(); /* PA_WAVES = 8 */
                if (ws.bfs_words < words) {
                    if (ws.d_bfs) (void)hipFree(ws.d_bfs);
                    ws.d_bfs = nullptr;
                    ws.bfs_words = 0;
                    if (hipMalloc(reinterpret_cast<void **>(&ws.d_bfs), words * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); ws.d_bfs = nullptr; }
                    else ws.bfs_words = words;
                }
                if (ws.d_bfs == nullptr) { (void)hipFree(ws.d_pwf); ws.d_pwf = nullptr; ws.pwf_bytes = 0; ws.pw_ready = false; } /* no room: the per-pixel kernel */
                pw.bfs_scratch = ws.d_bfs;
                pw.bfs_items_cap = RT_BFS_ITEMS_CAP;
                pw.bfs_jobs_cap = RT_BFS_JOBS_CAP;
                const long long cap = rt::option(rt::OPT_DIAG_BFS_CAP, 0); /* test hook: shorter lists (the memory is the same) */
                if (cap > 0) {
                    pw.bfs_items_cap = (uint32_t)std::min<long long>(cap, RT_BFS_ITEMS_CAP);
                    pw.bfs_jobs_cap = (uint32_t)std::min<long long>(cap, RT_BFS_JOBS_CAP);
                }
            }
            if (ws.d_pwf == nullptr) variant = RT_VARIANT_SGPR | RT_VARIANT_STATIC; /* no room for the arenas: the per-pixel kernel renders the frame */
            pw.tile_order = g_diag_tile_order.load();
            pw.tile_cost = g_diag_tile_cost.load();
            static_assert(PW_G_BLOCK_WORDS * sizeof(uint32_t) == 128, "two blocks of global words and the frame description share the 512-byte header");
            static_assert(sizeof(rt::KernelFrame) <= 128, "the frame description must fit its slot of the workspace header");
            pw.frame = reinterpret_cast<const rt::KernelFrame *>(static_cast<unsigned char *>(ws.d_pwf) + 256);
            pw.arena = static_cast<unsigned char *>(ws.d_pwf) + 512;
            pw.ray_count = d_ray_count;
            pw_groups = (uint32_t)groups;
            if (ws.d_pwf != nullptr) {
                hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) (void)hipGetLastError();
                if (capturing != hipStreamCaptureStatusNone) ws.pw_always_prepare = true;
                if (pw_band_rows >= kf.rows && !ws.pw_always_prepare) { /* one launch: does it find its block zeroed and its frame description in place? */
                    rt::KernelFrame want_frame = kf;
                    want_frame.n_chunks = (kf.cols * kf.rows + 63u) / 64u; /* as launch_pwf fills it in */
                    pw_init = !ws.pw_ready || memcmp(&ws.pw_frame, &want_frame, sizeof want_frame) != 0;
                    pw_parity = ws.pw_parity;
                    ws.pw_parity ^= 1u;
                    ws.pw_frame = want_frame;
                    ws.pw_ready = true;
                } else { /* several bands, each with its own frame description: every launch prepares its own */
                    pw_init = true;
                    pw_parity = 0;
                    ws.pw_ready = false;
                }
            }
        }
#ifdef RT_DIAG_TIMELINE
        qs.timeline = g_diag_timeline;
#endif
    }
    if (g_prof_on.load() && !t_prof_off) {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (g_prof_used == g_prof_events.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            RT_HIP(hipEventCreate(&a));
            RT_HIP(hipEventCreate(&b));
            g_prof_events.emplace_back(a, b);
        }
        rt::set_main_kernel_events(g_prof_events[g_prof_used].first, g_prof_events[g_prof_used].second);
        g_prof_used += 1;
    } else {
        rt::set_main_kernel_events(nullptr, nullptr);
    }
    hipError_t e = hipSuccess;
    if (variant & RT_VARIANT_PWF) {
        /* a band that does not fit the arenas is rendered by the per-pixel kernel instead (a no-op otherwise) */
        uint32_t *const blocks = reinterpret_cast<uint32_t *>(const_cast<unsigned char *>(pw.arena) - 512);
        pw.global = blocks + pw_parity * PW_G_BLOCK_WORDS;
        pw.global_next = blocks + (pw_parity ^ 1u) * PW_G_BLOCK_WORDS;
        qs.run_if = pw.global + PW_G_OVERFLOW;
        for (uint32_t r0 = 0; e == hipSuccess && r0 < kf.rows; r0 += pw_band_rows) {
            rt::KernelFrame band = kf;
            band.y0 = kf.y0 + r0 * kf.y_step;
            band.rows = kf.rows - r0 < pw_band_rows ? kf.rows - r0 : pw_band_rows;
            float *band_rgb = d_rgb + (size_t)r0 * kf.cols * 3u;
            {   /* a stride near the golden section of the tile count scatters consecutive tile fetches over the image */
                auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
                const uint64_t tiles = ((uint64_t)band.cols * band.rows + 63u) / 64u;
                uint64_t stride = (uint64_t)((double)tiles * 0.6180339887498949);
                if (stride < 1) stride = 1;
                while (gcd(stride, tiles) != 1) stride += 1;
                pw.tile_stride = (uint32_t)stride;
            }
            const bool first = r0 == 0, last = r0 + pw_band_rows >= kf.rows;
            e = rt::launch_pwf(scene->ks, band, band_rgb, pw, pw_groups, stream, pw_init, first, last);
            if (e == hipSuccess) {
                rt::mute_main_kernel_events(true); /* the event pair brackets the persistent kernel(s), not the fallback */
                e = rt::launch_whitted(scene->ks, band, band_rgb, d_ray_count, qs, stream, (variant & RT_VARIANT_LDS) | RT_VARIANT_STATIC);
                rt::mute_main_kernel_events(false);
            }
        }
        if (e != hipSuccess) {
            rt_scene *mut = const_cast<rt_scene *>(scene);
            std::lock_guard<std::mutex> lock(mut->ws_mutex);
            mut->workspaces[stream].pw_ready = false; /* whatever state the blocks are in: the next launch prepares its own */
            return fail_hip("rt_render_whitted: launch", e);
        }
        return RT_OK;
    }
    e = rt::launch_whitted(scene->ks, kf, d_rgb, d_ray_count, qs, stream, variant);
    if (e != hipSuccess) return fail_hip("rt_render_whitted: launch", e);
    return RT_OK;
}

int rt_render_whitted_host(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float *h_rgb,
                           unsigned long long *h_ray_count) {
    if (!scene || !h_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_whitted_host: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_whitted_host: bad frame");
    const size_t bytes = (size_t)rt_frame_pixels(frame) * 3 * sizeof(float);
    float *d_rgb = nullptr;
    unsigned long long *d_cnt = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_rgb), bytes));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned long long));
    int rc = RT_OK;
    if (e == hipSuccess) {
        rc = rt_render_whitted(scene, camera, frame, d_rgb, d_cnt, nullptr);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(h_rgb, d_rgb, bytes, hipMemcpyDeviceToHost);
            unsigned long long cnt = 0;
            if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_ray_count) *h_ray_count = cnt;
        }
    }
    (void)hipFree(d_rgb);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail_hip("rt_render_whitted_host", e);
    return RT_OK;
}

} /* extern "C" */
