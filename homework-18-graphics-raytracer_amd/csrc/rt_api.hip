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
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <cmath>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"
#include "rt_kernels.h"
#include "rt_vec.h"
#include "rt_luma.h"

namespace rt {
void math_eval_host(int op, const float *x, const float *y, float *out, size_t n);
}

/* Per-(scene, stream) scratch.  Launches on one stream are ordered, so they can share it; other streams get
 * their own.  Grow-only; allocated on the first call that needs it (not inside a graph capture). */
struct Workspace {
    uint32_t *d_counters = nullptr; /* [0] the stochastic pass's chunk counter */
    void *d_pwf = nullptr; /* persistent-wavefront path: two blocks of global words, the frame description, one arena per workgroup */
    size_t pwf_bytes = 0;
    /* Launches on this workspace alternate between the two blocks of global words: a launch's last workgroup zeroes the
     * other block, so the next launch needs no preparation of its own unless its frame description differs from what is in
     * device memory (or nothing has run here yet). */
    uint32_t pw_parity = 0;
    bool pw_ready = false;
    bool pw_always_prepare = false; /* a call on this stream was captured into a graph: replays come unannounced, so from then on
                                     * every launch prepares its own block and frame description, as a captured one does */
    rt::KernelFrame pw_frame;
    void *d_split = nullptr; /* split distributed pass: requests, shades and frames of one batch of epochs */
    size_t split_bytes = 0;
};

struct rt_scene {
    int device;
    void *d_blob; /* one allocation holding every array */
    rt::KernelScene ks;
    uint32_t resident_waves; /* CUs * 4 SIMDs * RT_MIN_WAVES: the persistent grid */
    uint32_t pwf_workgroups;  /* CUs * resident workgroups of the persistent-wavefront kernel */
    std::mutex ws_mutex;
    std::map<hipStream_t, Workspace> workspaces;
};

/* Process-wide settings (rt_set_*): read by render calls on any thread, so they are atomics.  A value < 0 means "not set
 * yet": the first reader resolves it from the environment (two threads doing that at once compute the same value). */
static std::atomic<int> g_wf_nodes_per_pixel{-1};
static std::atomic<const uint32_t *> g_diag_tile_order{nullptr};
#ifndef RT_DIST_SPLIT_DEFAULT
#define RT_DIST_SPLIT_DEFAULT 1
#endif
static std::atomic<int> g_dist_split{-1}; /* -1: RT_AMD_DIST_SPLIT or the default */
extern "C" int rt_set_distributed_split(int on) { g_dist_split.store(on < 0 ? -1 : (on > 1 ? 1 : on)); return 0; } /* 2 (round 2's queued chain) is 1 now */
extern "C" void rt_diag_set_tile_order(const void *device_ptr) { g_diag_tile_order.store(static_cast<const uint32_t *>(device_ptr)); }
#ifdef RT_DIAG_TIMELINE
static unsigned long long *g_diag_timeline = nullptr;
extern "C" void rt_diag_set_timeline(void *device_ptr) { g_diag_timeline = static_cast<unsigned long long *>(device_ptr); }
#endif

static thread_local std::string g_error;
static std::atomic<int> g_variant{-1};

static int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
static int fail_hip(const char *what, hipError_t e) {
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE
           : (e == hipErrorOutOfMemory)                           ? RT_ERR_OUT_OF_MEMORY
                                                                  : RT_ERR_HIP;
}
#define RT_HIP(call)                                          \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return fail_hip(#call, e_);     \
    } while (0)

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

static bool frame_ok(const rt_frame *f) {
    return f && f->width > 0 && f->height > 0 && f->y_step >= 1 && f->x0 < f->x1 && f->y0 < f->y1 && f->x1 <= f->width &&
           f->y1 <= f->height;
}
/* the kernels and launchers index a tile's pixels with 32-bit arithmetic: a tile of 2^32 pixels or more is refused
 * rather than wrapped (65536 x 65536 would wrap to 0 and "render" nothing) */
static bool frame_fits(const rt_frame *f) {
    const uint64_t rows = ((uint64_t)f->y1 - f->y0 + f->y_step - 1) / f->y_step;
    return rows * (uint64_t)(f->x1 - f->x0) < (1ull << 32) - 64u;
}

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
static std::atomic<bool> g_prof_on{false};
/* the profiling events live on the device that was current when they were made: calls that hop between devices (rt_multi_*) are
 * not profiled (the thread-local switch is theirs) */
static thread_local bool t_prof_off = false;
struct ProfilingOff {
    bool prev;
    ProfilingOff() : prev(t_prof_off) { t_prof_off = true; }
    ~ProfilingOff() { t_prof_off = prev; }
};

int rt_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on.store(on != 0);
    g_prof_used = 0;
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

/* Everything rt_scene_create derives from the ABI arrays, on the host (no HIP call in here): the device records of
 * rt_device_scene.h.  Also behind rt_scene_describe_nodes, which lets a test look at the node array without a GPU. */
struct SceneLayout {
    std::vector<rt::DevTri> tris;
    std::vector<rt::DevTriAttr> attrs;
    std::vector<rt::DevSegment> segments;
    std::vector<rt::DevTriHead> heads;
    std::vector<rt::DevSphere> spheres;
    double scene_extent = 0.0;
};

static int layout_scene(const rt_scene_desc *desc, SceneLayout &layout) {
    if ((desc->n_triangles && !desc->triangles) || (desc->n_spheres && !desc->spheres) || (desc->n_materials && !desc->materials) ||
        (desc->n_lights && !desc->lights))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null array with non-zero count");
    if ((uint64_t)desc->n_triangles + desc->n_spheres >= 0x1fffffffull)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: too many primitives");
    if (desc->n_triangles > RT_MAX_TRIANGLES)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: more than 2^24 triangles (the path is brute force by definition: one cast tests them all)");
    for (uint32_t i = 0; i < desc->n_triangles; ++i)
        if (desc->triangles[i].object_index >= desc->n_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: triangle object_index out of range");
    for (uint32_t i = 0; i < desc->n_spheres; ++i)
        if (desc->spheres[i].object_index >= desc->n_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: sphere object_index out of range");
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind > RT_LIGHT_POINT) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: unknown light kind");
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        if (desc->materials[i].diffuse_fn > RT_DIFFUSE_STRIPE_SUM || desc->materials[i].normal_fn > RT_NORMAL_WAVE_U)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: unknown material function");

    using rt::V3;
    std::vector<rt::DevTri> &tris = layout.tris;
    std::vector<rt::DevTriAttr> &attrs = layout.attrs;
    tris.assign(desc->n_triangles, rt::DevTri());
    attrs.assign(desc->n_triangles, rt::DevTriAttr());
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        const rt_triangle &s = desc->triangles[i];
        rt::DevTri &t = tris[i];
        rt::DevTriAttr &a = attrs[i];
        memset(&t, 0, sizeof t);
        memset(&a, 0, sizeof a);
        const V3 v0 = rt::v3p(s.vertices[0].position), v1 = rt::v3p(s.vertices[1].position), v2 = rt::v3p(s.vertices[2].position);
        /* Triangle::face_normal, primitives.rs:36-42 */
        const V3 n = rt::normalize(rt::cross(v1 - v0, v2 - v1));
        t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
        t.d = rt::dot(n, v0); /* main.rs:203 */
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z;
        t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z;
        t.obj = s.object_index;
        const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0; /* main.rs:219-221 */
        t.e0[0] = e0.x; t.e0[1] = e0.y; t.e0[2] = e0.z;
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
        t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
        t.area = rt::dot(rt::cross(v1 - v0, v2 - v0), n); /* main.rs:235 */
        for (int k = 0; k < 3; ++k) {
            a.n0[k] = s.vertices[0].normal[k];
            a.n1[k] = s.vertices[1].normal[k];
            a.n2[k] = s.vertices[2].normal[k];
        }
        a.uv0x = s.vertices[0].uv[0]; a.uv0y = s.vertices[0].uv[1];
        a.uv1x = s.vertices[1].uv[0]; a.uv1y = s.vertices[1].uv[1];
        a.uv2x = s.vertices[2].uv[0]; a.uv2y = s.vertices[2].uv[1];
    }
    /* bounding spheres for the conservative rejection in the intersection loop (rt_device_scene.h) */
    double &scene_extent = layout.scene_extent;
    scene_extent = 0.0;
    for (uint32_t i = 0; i < desc->n_triangles; ++i)
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const double a = fabs((double)desc->triangles[i].vertices[v].position[k]);
                if (a > scene_extent) scene_extent = a; /* NaN never compares greater */
            }
    const bool filter_off = getenv("RT_AMD_NO_SPHERE_FILTER") != nullptr; /* A/B switch; results are the same either way */
    const char *frac_env = getenv("RT_AMD_FILTER_MAX_FRAC");
    /* a triangle as large as the scene rejects next to nothing: not worth its ten instructions */
    const double max_frac = (frac_env && *frac_env) ? atof(frac_env) : 0.5;
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        rt::DevTri &t = tris[i];
        t.bq = std::numeric_limits<float>::infinity();
        t.bcx = t.bcy = t.bcz = 0.0f;
        double P[3][3];
        bool finite = true;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                P[v][k] = (double)desc->triangles[i].vertices[v].position[k];
                finite = finite && std::isfinite(P[v][k]);
            }
        if (!finite || filter_off || !(scene_extent <= 1e10)) continue;
        auto sub = [](const double *a, const double *b, double *o) { for (int k = 0; k < 3; ++k) o[k] = a[k] - b[k]; };
        auto dotd = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
        double ab[3], ac[3], bc[3];
        sub(P[1], P[0], ab); sub(P[2], P[0], ac); sub(P[2], P[1], bc);
        const double la = dotd(bc, bc), lb = dotd(ac, ac), lc = dotd(ab, ab); /* squared sides opposite A, B, C */
        if (!(la > 0.0 && lb > 0.0 && lc > 0.0)) continue;
        /* smallest angle, from sin and cos at each vertex */
        double cr[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
        const double twice_area = sqrt(dotd(cr, cr));
        const double angA = atan2(twice_area, dotd(ab, ac));
        const double angB = atan2(twice_area, -dotd(ab, bc));
        const double angC = atan2(twice_area, dotd(ac, bc));
        const double ang_min = angA < angB ? (angA < angC ? angA : angC) : (angB < angC ? angB : angC);
        if (!(ang_min >= 0.0201)) continue; /* sin(angle/2) >= 0.01 */
        double c[3], r2;
        if (la >= lb + lc) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[1][k] + P[2][k]); r2 = 0.25 * la; }
        else if (lb >= la + lc) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[0][k] + P[2][k]); r2 = 0.25 * lb; }
        else if (lc >= la + lb) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[0][k] + P[1][k]); r2 = 0.25 * lc; }
        else { /* acute: circumcentre */
            const double wa = la * (lb + lc - la), wb = lb * (lc + la - lb), wc = lc * (la + lb - lc);
            const double w = wa + wb + wc;
            for (int k = 0; k < 3; ++k) c[k] = (wa * P[0][k] + wb * P[1][k] + wc * P[2][k]) / w;
            double d0[3];
            sub(P[0], c, d0);
            r2 = dotd(d0, d0);
        }
        /* the sphere must contain the three vertices whatever the rounding above did */
        for (int v = 0; v < 3; ++v) {
            double dv[3];
            sub(P[v], c, dv);
            const double q = dotd(dv, dv);
            if (q > r2) r2 = q;
        }
        const double radius = sqrt(r2);
        if (!(radius <= max_frac * scene_extent)) continue;
        if (!(radius >= 1e-3 * scene_extent) || !std::isfinite(radius)) continue; /* tiny against the scene: p - c would cancel */
        t.bcx = (float)c[0]; t.bcy = (float)c[1]; t.bcz = (float)c[2];
        /* 1.05 R^2, plus the float rounding of the centre (<= 1e-7 * extent per axis, far inside the margin), rounded up */
        t.bq = std::nextafter((float)(1.05 * r2 * 1.0001), std::numeric_limits<float>::infinity());
    }
    /* The triangles as NODES for the intersection loop (rt_device_scene.h "segments"): a pre-order array of leaves (runs of
     * consecutive triangles) and inner nodes over them, each with a skip pointer.  A run of one object's >= 8 triangles, all of
     * which qualify for their own bounding-sphere rejection, becomes a tree: leaves of RT_LEAF_TRIANGLES, grouped 16 by 16;
     * every node that is small against the scene gets a bounding sphere and either up to 8 representative face normals or a
     * normal cone, and can then be skipped by a wave none of whose rays can hit anything in it.  Everything else is a plain
     * leaf that is always visited. */
    std::vector<rt::DevSegment> &segments = layout.segments;
    segments.clear();
    {
        const bool clusters_off = filter_off || getenv("RT_AMD_NO_CLUSTERS") != nullptr; /* A/B switch; results are the same either way */
        const bool flat_only = getenv("RT_AMD_NO_HIERARCHY") != nullptr; /* A/B: one cluster per object run, explicit normals only (round 1) */
        uint32_t single_leaf_max = 64u; /* A/B: objects up to this many triangles stay one leaf */
        if (const char *v = getenv("RT_AMD_SINGLE_LEAF_MAX")) { if (*v) single_leaf_max = (uint32_t)atoi(v); }
        /* A plain run may only grow the leaf before it if that leaf is not inside a subtree that is already closed: an inner
         * node's skip_to jumps over everything emitted below it, so triangles appended to a leaf in there would be skipped with
         * it.  merge_barrier = the number of nodes no later run may be merged into (moved whenever a subtree or a tree ends). */
        size_t merge_barrier = 0;
        auto push_plain = [&](uint32_t first, uint32_t count) {
            if (segments.size() > merge_barrier && segments.back().n_normals == 0u && segments.back().count != 0u &&
                segments.back().first + segments.back().count == first) {
                segments.back().count += count; /* adjacent plain runs are one leaf */
                return;
            }
            rt::DevSegment g;
            memset(&g, 0, sizeof g);
            g.first = first;
            g.count = count;
            g.skip_to = (uint32_t)segments.size() + 1u;
            segments.push_back(g);
        };
        /* bounding sphere + steepness data of the triangles [lo, hi); false: the node cannot be skipped */
        auto node_stats = [&](uint32_t lo_t, uint32_t hi_t, rt::DevSegment *g) -> bool {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (uint32_t k = lo_t; k < hi_t; ++k)
                for (int v = 0; v < 3; ++v)
                    for (int a = 0; a < 3; ++a) {
                        const double x = (double)desc->triangles[k].vertices[v].position[a];
                        if (x < lo[a]) lo[a] = x;
                        if (x > hi[a]) hi[a] = x;
                    }
            for (int a = 0; a < 3; ++a) g->c[a] = (float)(0.5 * (lo[a] + hi[a]));
            double r2 = 0.0;
            for (uint32_t k = lo_t; k < hi_t; ++k) { /* the sphere must contain every triangle's own bounding sphere */
                const double dx = (double)tris[k].bcx - g->c[0], dy = (double)tris[k].bcy - g->c[1], dz = (double)tris[k].bcz - g->c[2];
                const double reach = sqrt(dx * dx + dy * dy + dz * dz) + sqrt((double)tris[k].bq);
                if (reach * reach > r2) r2 = reach * reach;
            }
            const double radius = sqrt(r2);
            if (!(std::isfinite(radius) && radius <= max_frac * scene_extent && radius >= 1e-3 * scene_extent)) return false;
            /* (1.05 R)^2 with R already holding the triangles' own 1.05 margins: generous, and rounded up */
            g->r2_hi = std::nextafter((float)(r2 * 1.0001), std::numeric_limits<float>::infinity());
            /* one representative per face plane direction: sign canonicalised, merged within 1e-4 per component */
            bool explicit_ok = true;
            g->n_normals = 0u;
            double mean[3] = {0.0, 0.0, 0.0};
            for (uint32_t k = lo_t; k < hi_t; ++k) {
                float n[3] = {tris[k].n[0], tris[k].n[1], tris[k].n[2]};
                if (!(std::isfinite(n[0]) && std::isfinite(n[1]) && std::isfinite(n[2]))) return false;
                const int lead = fabsf(n[0]) > 1e-3f ? 0 : (fabsf(n[1]) > 1e-3f ? 1 : 2);
                if (n[lead] < 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
                bool known = false;
                for (uint32_t q = 0; q < g->n_normals && !known; ++q)
                    known = fabsf(g->normals[q][0] - n[0]) <= 1e-4f && fabsf(g->normals[q][1] - n[1]) <= 1e-4f && fabsf(g->normals[q][2] - n[2]) <= 1e-4f;
                if (!known && explicit_ok) {
                    if (g->n_normals == RT_SEGMENT_NORMALS) explicit_ok = false;
                    else {
                        g->normals[g->n_normals][0] = n[0]; g->normals[g->n_normals][1] = n[1]; g->normals[g->n_normals][2] = n[2];
                        g->n_normals += 1u;
                    }
                }
            }
            if (explicit_ok && g->n_normals != 0u) return true;
            if (flat_only) return false;
            /* More than 8 plane directions: a CONE.  Axis a (unit), half-angle theta >= the angle between a and every face normal
             * or its negative.  For a unit direction d and a unit normal n within theta of +-a:
             *     |n.d| >= |a.d| cos(theta) - sin(theta),
             * so |a.d| >= (1.01e-3 + sin theta) / cos theta =: K implies |n.d| >= 1.01e-3 for every triangle below — the
             * condition the explicit normals test one by one (the 1 % covers the binary32 evaluation of both sides and of the
             * loop's own n.d).  Stored as K^2 for the test (a.d)^2 >= K^2 (d.d); cones of 60 degrees and more are useless. */
            for (uint32_t k = lo_t; k < hi_t; ++k) { /* the axis: mean of the normals, each flipped into the first one's half-space */
                const double s = ((double)tris[k].n[0] * tris[lo_t].n[0] + (double)tris[k].n[1] * tris[lo_t].n[1] + (double)tris[k].n[2] * tris[lo_t].n[2]) < 0.0 ? -1.0 : 1.0;
                for (int a = 0; a < 3; ++a) mean[a] += s * (double)tris[k].n[a];
            }
            const double ml = sqrt(mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
            if (!(ml > 1e-6)) return false;
            float ax[3];
            for (int a = 0; a < 3; ++a) ax[a] = (float)(mean[a] / ml);
            const double al = sqrt((double)ax[0] * ax[0] + (double)ax[1] * ax[1] + (double)ax[2] * ax[2]); /* of the ROUNDED axis: what the kernel uses */
            double cos_min = 1.0;
            for (uint32_t k = lo_t; k < hi_t; ++k) {
                const double nl = sqrt((double)tris[k].n[0] * tris[k].n[0] + (double)tris[k].n[1] * tris[k].n[1] + (double)tris[k].n[2] * tris[k].n[2]);
                const double c = fabs(((double)tris[k].n[0] * ax[0] + (double)tris[k].n[1] * ax[1] + (double)tris[k].n[2] * ax[2]) / (nl * al));
                if (!(c <= 1.0)) { if (c > 1.0 && c < 1.0 + 1e-9) continue; return false; }
                if (c < cos_min) cos_min = c;
            }
            const double theta = acos(cos_min) + 1e-5; /* slack for everything rounded on the way */
            if (!(theta < 1.0471975511965976)) return false; /* 60 degrees */
            const double K = (1.01e-3 + sin(theta)) / cos(theta) * 1.0001;
            if (!(K < 1.0)) return false;
            g->n_normals = RT_SEGMENT_CONE;
            /* the kernel compares (a.d)^2 with K^2 (d.d) where a is the rounded axis of length al: fold al^2 in, round up */
            g->normals[0][0] = ax[0]; g->normals[0][1] = ax[1]; g->normals[0][2] = ax[2];
            g->normals[0][3] = std::nextafter((float)(K * K * al * al * 1.0001), std::numeric_limits<float>::infinity());
            return true;
        };
        /* pre-order emission of the tree over the leaves [l0, l1) of the object run [run_lo, run_hi) */
        struct Emit {
            static void go(uint32_t l0, uint32_t l1, uint32_t run_lo, uint32_t run_hi, std::vector<rt::DevSegment> &out,
                           const std::function<bool(uint32_t, uint32_t, rt::DevSegment *)> &stats,
                           const std::function<void(uint32_t, uint32_t)> &plain, size_t *barrier) {
                const uint32_t t0 = run_lo + l0 * RT_LEAF_TRIANGLES;
                const uint32_t t1 = std::min<uint64_t>(run_hi, (uint64_t)run_lo + (uint64_t)l1 * RT_LEAF_TRIANGLES);
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                const bool ok = stats(t0, t1, &g);
                if (l1 - l0 == 1u) {
                    if (!ok) { plain(t0, t1 - t0); return; }
                    g.first = t0;
                    g.count = t1 - t0;
                    g.skip_to = (uint32_t)out.size() + 1u;
                    out.push_back(g);
                    return;
                }
                size_t at = (size_t)-1;
                if (ok) { /* an inner node: count 0, skip_to patched once its subtree is out */
                    g.first = t0;
                    g.count = 0u;
                    at = out.size();
                    out.push_back(g);
                }
                uint32_t child = 1u; /* leaves per child: the largest power of 16 below the span */
                while ((uint64_t)child * 16u < (uint64_t)(l1 - l0)) child *= 16u;
                for (uint32_t c0 = l0; c0 < l1; c0 += child) go(c0, std::min(l1, c0 + child), run_lo, run_hi, out, stats, plain, barrier);
                if (at != (size_t)-1) {
                    out[at].skip_to = (uint32_t)out.size();
                    *barrier = out.size(); /* the subtree is closed: nothing may be appended to a leaf inside it */
                }
            }
        };
        for (uint32_t i = 0; i < desc->n_triangles;) {
            uint32_t j = i;
            while (j < desc->n_triangles && desc->triangles[j].object_index == desc->triangles[i].object_index) ++j;
            bool ok = !clusters_off && j - i >= 8u;
            for (uint32_t k = i; ok && k < j; ++k) ok = std::isfinite(tris[k].bq); /* every triangle qualifies for its own rejection */
            if (!ok) {
                push_plain(i, j - i);
            } else if (flat_only || j - i <= single_leaf_max) {
                /* a small object is ONE leaf (the reference scene's dodecahedron: 36 triangles, 6 plane directions — one test per
                 * cast decides it; as a tree of three leaves it cost the bench frame 3 %) */
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                if (node_stats(i, j, &g)) { g.first = i; g.count = j - i; g.skip_to = (uint32_t)segments.size() + 1u; segments.push_back(g); }
                else if (flat_only) push_plain(i, j - i);
                else Emit::go(0u, (j - i + RT_LEAF_TRIANGLES - 1u) / RT_LEAF_TRIANGLES, i, j, segments, node_stats, push_plain, &merge_barrier);
            } else {
                const uint32_t n_leaves = (j - i + RT_LEAF_TRIANGLES - 1u) / RT_LEAF_TRIANGLES;
                Emit::go(0u, n_leaves, i, j, segments, node_stats, push_plain, &merge_barrier);
            }
            i = j;
        }
        /* Neighbouring clustered leaves whose common bounding sphere is hardly larger than the larger of their own become ONE leaf
         * (the reference scene's two glass slabs, main.rs:879-977: 12 + 12 triangles an arm's length apart, the same three plane
         * directions): a ray that needs one nearly always needs the other, and a leaf is a bounding-sphere test, a set of plane
         * directions and — pair-wise — a set-up of its own.  A leaf is any run of consecutive triangles that all qualify for their own
         * rejection, so nothing else changes.  Only leaves with the same ancestors are joined (no subtree ends between them). */
        if (!clusters_off && getenv("RT_AMD_NO_LEAF_MERGE") == nullptr) {
            for (size_t k = 0; k + 1u < segments.size();) {
                const rt::DevSegment a = segments[k], b = segments[k + 1u];
                bool ok = a.count != 0u && b.count != 0u && a.n_normals != 0u && b.n_normals != 0u && a.n_normals != RT_SEGMENT_CONE &&
                          b.n_normals != RT_SEGMENT_CONE && a.first + a.count == b.first && a.count + b.count <= 64u;
                for (size_t j = 0; ok && j < k; ++j) ok = !(segments[j].count == 0u && segments[j].skip_to == k + 1u);
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                ok = ok && node_stats(a.first, b.first + b.count, &g) && g.n_normals != RT_SEGMENT_CONE &&
                     g.r2_hi <= 1.15f * std::max(a.r2_hi, b.r2_hi);
                if (!ok) { ++k; continue; }
                g.first = a.first;
                g.count = a.count + b.count;
                g.skip_to = (uint32_t)k + 1u;
                segments[k] = g;
                segments.erase(segments.begin() + (ptrdiff_t)k + 1);
                for (rt::DevSegment &n : segments)
                    if (n.skip_to > k + 1u) n.skip_to -= 1u;
                /* and again from the same node: it may take the next one too */
            }
        }
        /* clustered leaves: how their triangles are dealt to the lanes of a pair-wise pass (rt_device_scene.h RT_SEG_PAIR_*) */
        const bool pairs_off = getenv("RT_AMD_NO_PAIRS") != nullptr; /* A/B switch; results are the same either way */
        for (rt::DevSegment &g : segments) {
            if (g.count == 0u || g.n_normals == 0u || pairs_off || g.count > 64u) continue;
            uint32_t best_k = 0u, best_ck = 0u, best_r = 0u;
            double best_fill = 0.0;
            for (uint32_t K = 1u; K <= 8u; ++K) {
                const uint32_t ck = (g.count + K - 1u) / K;
                if (ck < 4u && K > 1u) break;
                const uint32_t R = 64u / ck;
                const double fill = (double)R * g.count / K; /* pairs per full pass */
                if (fill > best_fill * 1.05) { best_fill = fill; best_k = K; best_ck = ck; best_r = R; }
            }
            if (best_k == 0u) continue;
            auto put = [](float *slot, uint32_t v) { memcpy(slot, &v, sizeof v); };
            put(&g.normals[1][3], best_ck | (best_k << 8) | (best_r << 16));
            put(&g.normals[2][3], 65535u / best_ck + 1u);
            put(&g.normals[3][3], 65535u / best_k + 1u);
        }
    }
    /* triangles on their predecessor's plane (rt_device_scene.h RT_TRI_FOLLOWS): same segment; n and d equal bit for bit, or
     * (WEAK) equal up to the signs of zero components */
    if (getenv("RT_AMD_NO_PLANE_SHARING") == nullptr && desc->n_materials <= RT_TRI_OBJ_MASK) { /* A/B switch; results are the same either way */
        const bool weak_ok = getenv("RT_AMD_NO_WEAK_PLANE_SHARING") == nullptr;
        /* any two consecutive triangles of one object: a call of the loop covers consecutive records and treats its first triangle
         * as a leader whatever its flag says, so a pair may straddle leaves */
        for (uint32_t i = 1u; i < desc->n_triangles; ++i) {
            if (desc->triangles[i].object_index != desc->triangles[i - 1u].object_index) continue;
            {
                const float a[4] = {tris[i - 1u].n[0], tris[i - 1u].n[1], tris[i - 1u].n[2], tris[i - 1u].d};
                const float b[4] = {tris[i].n[0], tris[i].n[1], tris[i].n[2], tris[i].d};
                bool exact = true, weak = true;
                for (int k = 0; k < 4; ++k) {
                    const bool same_bits = memcmp(&a[k], &b[k], sizeof(float)) == 0;
                    exact = exact && same_bits;
                    weak = weak && (same_bits || (a[k] == 0.0f && b[k] == 0.0f));
                }
                if (exact) tris[i].obj |= RT_TRI_FOLLOWS;
                else if (weak && weak_ok) tris[i].obj |= RT_TRI_FOLLOWS | RT_TRI_FOLLOWS_WEAK;
            }
        }
    }
    std::vector<rt::DevTriHead> &heads = layout.heads;
    heads.assign(desc->n_triangles, rt::DevTriHead());
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        rt::DevTriHead &h = heads[i];
        const rt::DevTri &t = tris[i];
        h.n[0] = t.n[0]; h.n[1] = t.n[1]; h.n[2] = t.n[2]; h.d = t.d;
        h.bc[0] = t.bcx; h.bc[1] = t.bcy; h.bc[2] = t.bcz; h.bq = t.bq;
    }
    std::vector<rt::DevSphere> &spheres = layout.spheres;
    spheres.assign(desc->n_spheres, rt::DevSphere());
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const rt_sphere &s = desc->spheres[i];
        rt::DevSphere &d = spheres[i];
        memset(&d, 0, sizeof d);
        d.c[0] = s.center[0]; d.c[1] = s.center[1]; d.c[2] = s.center[2];
        d.radius = s.radius;
        d.r2 = s.radius * s.radius; /* radius.powi(2), main.rs:272 */
        d.obj = s.object_index;
    }
    return RT_OK;
}

/* Diagnostics: the node array (rt_device_scene.h) rt_scene_create would build for `desc`, six words per node — first, count,
 * n_normals, skip_to, the pair-wise dealing word, 0 — without touching a device. */
int rt_scene_describe_nodes(const rt_scene_desc *desc, uint32_t *out_words, uint32_t cap_nodes, uint32_t *n_nodes) {
    if (!desc || !n_nodes || (cap_nodes && !out_words)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_describe_nodes: null argument");
    SceneLayout layout;
    const int rc = layout_scene(desc, layout);
    if (rc != RT_OK) return rc;
    *n_nodes = (uint32_t)layout.segments.size();
    for (uint32_t k = 0; k < *n_nodes && k < cap_nodes; ++k) {
        const rt::DevSegment &g = layout.segments[k];
        uint32_t pair_word;
        memcpy(&pair_word, &g.normals[1][3], sizeof pair_word);
        uint32_t *o = out_words + (size_t)k * 6u;
        o[0] = g.first; o[1] = g.count; o[2] = g.n_normals; o[3] = g.skip_to; o[4] = pair_word; o[5] = 0u;
    }
    return RT_OK;
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
    const size_t total = off_light_aux + up(desc->n_lights * sizeof(rt::LightAux)) + 256;
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
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device);
    if (e != hipSuccess || cus <= 0) cus = 256;
    sc->resident_waves = (uint32_t)cus * 4u * (uint32_t)RT_MIN_WAVES;
    sc->pwf_workgroups = (uint32_t)cus * (uint32_t)rt::pwf_workgroups_per_cu(7168u, 8192u);
    *out_scene = sc;
    return RT_OK;
}

int rt_scene_destroy(rt_scene *scene) {
    if (!scene) return RT_OK;
    hipError_t e = hipSuccess;
    for (auto &kv : scene->workspaces) {
        if (kv.second.d_counters) (void)hipFree(kv.second.d_counters);
        if (kv.second.d_pwf) (void)hipFree(kv.second.d_pwf);
        if (kv.second.d_split) (void)hipFree(kv.second.d_split);
    }
    if (scene->d_blob) e = hipFree(scene->d_blob);
    delete scene;
    if (e != hipSuccess) return fail_hip("rt_scene_destroy: hipFree", e);
    return RT_OK;
}

static int make_kernel_frame(const rt_camera *camera, const rt_frame *frame, rt::KernelFrame *kf) {
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
        if (!ws.d_counters) RT_HIP(hipMalloc(reinterpret_cast<void **>(&ws.d_counters), 256));
        if (variant & RT_VARIANT_PWF) {
            /* The kernel keeps one LDS word per 64 ring slots and per 64 nodes, so an arena holds 64 K ring slots at most.
             * A tile too large for that at the budget asked for (beyond ~8 Mpixel at 6 nodes per pixel) is rendered as
             * several bands of rows, one launch each, through the same workspace. */
            const uint64_t ring_max = 1ull << 16;
            const uint64_t pixels = (uint64_t)kf.cols * kf.rows;
            uint64_t groups = (pixels + 63u) / 64u; /* a workgroup fetches one to eight tiles at a time */
            if (groups > scene->pwf_workgroups) groups = scene->pwf_workgroups;
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
                const char *refuse = getenv("RT_AMD_DIAG_WS_REFUSE"); /* test hook: pretend the allocation fails */
                if ((refuse && atoi(refuse) > 0) || hipMalloc(&ws.d_pwf, need) != hipSuccess) {
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
            if (ws.d_pwf == nullptr) variant = RT_VARIANT_SGPR | RT_VARIANT_STATIC; /* no room for the arenas: the per-pixel kernel renders the frame */
            pw.tile_order = g_diag_tile_order.load();
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

/* ---- distributed pass ------------------------------------------------------- */

struct rt_rng {
    int device;
    uint32_t *d_states; /* RT_RNG_DEVICE_WORDS per pixel */
    uint32_t *d_list;   /* scratch of the look-ahead pass: 1 + pixels words */
    uint32_t compute_units;
    /* The look-ahead for the NEXT batch runs on a stream of its own next to this batch's shade kernel (nothing after
     * the chain kernel touches the records).  ahead = every pixel has its next block, as of the work enqueued so far. */
    hipStream_t aux;
    hipEvent_t ev_chain, ev_prepared;
    bool ahead;
    /* The shade and unwind kernels of a batch run on a third stream, beside the NEXT batch's chain kernel (two workspaces, used
     * in turn): they fill what its tail leaves idle.  ev_tail[b]: the unwind that read workspace b has finished. */
    hipStream_t tail;
    hipEvent_t ev_tail[2];
    /* the chain kernel's pixels grouped by what their samples cost (rt_kernels.h DistParams::pixel_order): per pixel its cost in the
     * last batch unwound | two orders, one per workspace of a pipelined call | 512 words of scratch */
    uint32_t *d_pix;
    bool order_valid[2];
    hipStream_t main_stream; /* of the call in progress (for the after-chain hook) */
    uint32_t cols, rows, x0, y0, y_step;
};

int rt_rng_state_words(void) { return (int)RT_RNG_STATE_WORDS; }

int rt_rng_create(const rt_frame *frame, rt_rng **out_rng) {
    if (!out_rng) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_create: null argument");
    *out_rng = nullptr;
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_create: bad frame");
    if (!frame_fits(frame)) return fail(RT_ERR_UNSUPPORTED, "rt_rng_create: tile of 2^32 pixels or more");
    rt_rng *r = new (std::nothrow) rt_rng();
    if (!r) return fail(RT_ERR_OUT_OF_MEMORY, "rt_rng_create: host allocation failed");
    r->cols = frame->x1 - frame->x0;
    r->rows = rt_frame_rows(frame);
    r->x0 = frame->x0;
    r->y0 = frame->y0;
    r->y_step = frame->y_step;
    r->d_states = nullptr;
    r->d_list = nullptr;
    r->compute_units = 256;
    r->aux = nullptr;
    r->ev_chain = r->ev_prepared = nullptr;
    r->tail = nullptr;
    r->ev_tail[0] = r->ev_tail[1] = nullptr;
    r->d_pix = nullptr;
    r->order_valid[0] = r->order_valid[1] = false;
    r->ahead = false;
    r->main_stream = nullptr;
    const size_t bytes = (size_t)r->cols * r->rows * RT_RNG_DEVICE_WORDS * sizeof(uint32_t);
    hipError_t e = hipGetDevice(&r->device);
    if (e == hipSuccess) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r->device) == hipSuccess && cus > 0) r->compute_units = (uint32_t)cus;
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_states), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_list), ((size_t)r->cols * r->rows + 1u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_pix), ((size_t)r->cols * r->rows * 3u + 512u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(r->d_pix, 0, ((size_t)r->cols * r->rows * 3u + 512u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_chain, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_prepared, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->tail, hipStreamNonBlocking); /* (at the lowest stream priority: no different, 1 226 against 1 229 Msamples/s) */
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_tail[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_tail[1], hipEventDisableTiming);
    if (e == hipSuccess) {
        rt::KernelFrame kf;
        memset(&kf, 0, sizeof kf);
        kf.cols = r->cols; kf.rows = r->rows; kf.x0 = r->x0; kf.y0 = r->y0; kf.y_step = r->y_step;
        e = rt::launch_rng_seed(r->d_states, kf, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (r->d_states) (void)hipFree(r->d_states);
        if (r->d_list) (void)hipFree(r->d_list);
        if (r->d_pix) (void)hipFree(r->d_pix);
        if (r->ev_chain) (void)hipEventDestroy(r->ev_chain);
        if (r->ev_prepared) (void)hipEventDestroy(r->ev_prepared);
        if (r->aux) (void)hipStreamDestroy(r->aux);
        for (int b = 0; b < 2; ++b) if (r->ev_tail[b]) (void)hipEventDestroy(r->ev_tail[b]);
        if (r->tail) (void)hipStreamDestroy(r->tail);
        delete r;
        return fail_hip("rt_rng_create", e);
    }
    *out_rng = r;
    return RT_OK;
}

int rt_rng_destroy(rt_rng *rng) {
    if (!rng) return RT_OK;
    if (rng->aux) (void)hipStreamSynchronize(rng->aux); /* a look-ahead pass may still be writing the records */
    if (rng->tail) (void)hipStreamSynchronize(rng->tail);
    hipError_t e = rng->d_states ? hipFree(rng->d_states) : hipSuccess;
    if (rng->d_list) (void)hipFree(rng->d_list);
    if (rng->d_pix) (void)hipFree(rng->d_pix);
    if (rng->ev_chain) (void)hipEventDestroy(rng->ev_chain);
    if (rng->ev_prepared) (void)hipEventDestroy(rng->ev_prepared);
    if (rng->aux) (void)hipStreamDestroy(rng->aux);
    for (int b = 0; b < 2; ++b) if (rng->ev_tail[b]) (void)hipEventDestroy(rng->ev_tail[b]);
    if (rng->tail) (void)hipStreamDestroy(rng->tail);
    delete rng;
    if (e != hipSuccess) return fail_hip("rt_rng_destroy: hipFree", e);
    return RT_OK;
}

int rt_rng_download(const rt_rng *rng, uint32_t *h_states) {
    if (!rng || !h_states) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_download: null argument");
    /* the device keeps two banks per pixel (the block in use and the next one, generated ahead); what leaves is the
     * reference's record: the bank in use + the position */
    const size_t bytes = (size_t)rng->cols * rng->rows * RT_RNG_STATE_WORDS * sizeof(uint32_t);
    if (bytes == 0) return RT_OK;
    RT_HIP(hipDeviceSynchronize());
    uint32_t *d_tmp = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_tmp), bytes));
    hipError_t e = rt::launch_rng_export(rng->d_states, rng->cols * rng->rows, d_tmp, nullptr);
    if (e == hipSuccess) e = hipMemcpy(h_states, d_tmp, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail_hip("rt_rng_download", e);
    return RT_OK;
}

/* after the chain kernel of a batch: look-ahead for the next batch on the aux stream */
static hipError_t lookahead_after_chain(void *ctx) {
    rt_rng *rng = static_cast<rt_rng *>(ctx);
    hipError_t e = hipEventRecord(rng->ev_chain, rng->main_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(rng->aux, rng->ev_chain, 0);
    if (e == hipSuccess) e = rt::launch_rng_prepare(rng->d_states, rng->cols * rng->rows, rng->d_list, rng->compute_units, rng->aux);
    if (e == hipSuccess) e = hipEventRecord(rng->ev_prepared, rng->aux);
    if (e == hipSuccess) rng->ahead = true;
    return e;
}

int rt_render_distributed(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                          rt_rng *rng, uint32_t n_epochs, float *d_accum, float *d_samples, unsigned char *d_valid,
                          unsigned long long *d_ray_count, void *hip_stream) {
    if (!scene || !rng) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: null argument");
    if (!d_accum && !d_samples) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: need d_accum or d_samples");
    rt::KernelFrame kf;
    int rc = make_kernel_frame(camera, frame, &kf);
    if (rc != RT_OK) return rc;
    if (kf.cols != rng->cols || kf.rows != rng->rows || kf.x0 != rng->x0 || kf.y0 != rng->y0 || kf.y_step != rng->y_step)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: the RNG was created for a different tile");
    rt::DistParams dp;
    dp.rng_states = rng->d_states;
    dp.n_epochs = n_epochs;
    dp.focus = focus;
    dp.blur = blur;
    dp.accum = d_accum;
    dp.samples = d_samples;
    dp.valid = d_valid;
    dp.ray_count = d_ray_count;
    dp.work_queue = nullptr;
    dp.pixel_order = nullptr;
    dp.pixel_cost = nullptr;
    dp.own_first_chunk = 0u;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint32_t dist_waves = scene->resident_waves;
    int split = g_dist_split.load();
    if (split < 0) {
        const char *v = getenv("RT_AMD_DIST_SPLIT");
        split = v && *v ? (*v == '0' ? 0 : (*v == '1' ? 1 : 2)) : RT_DIST_SPLIT_DEFAULT;
    }
    const size_t n_pixels = (size_t)kf.cols * kf.rows;
    if (n_pixels == 0 || n_epochs == 0) return RT_OK;
    bool lookahead = true; /* A/B: RT_AMD_RNG_LOOKAHEAD=0 leaves every IsaacCore::generate to the render kernels */
    if (const char *v = getenv("RT_AMD_RNG_LOOKAHEAD")) lookahead = !(*v == '0');
    bool overlap = true; /* A/B: RT_AMD_RNG_OVERLAP=0 runs the look-ahead in line, before each chain kernel */
    if (const char *v = getenv("RT_AMD_RNG_OVERLAP")) overlap = !(*v == '0');
    rt_scene *mut = const_cast<rt_scene *>(scene);
    if (split && kf.max_depth <= 254) {
        /* chain / shade / unwind kernels over batches of epochs (rt_distributed.hip "the split pass"); a batch is as
         * many epochs as fit the workspace cap (RT_AMD_DIST_WS_MB, default 32 GiB for the two workspaces of a call of several
         * batches; one epoch at least) and never more than 16 — a visit of a pixel should not need more random words than the block
         * in use plus the one prepared ahead */
        const uint32_t slots = (uint32_t)(kf.max_depth > 0 ? kf.max_depth : 0) + 1u;
        const size_t per_epoch = rt::distributed_split_bytes_per_sample(kf.max_depth) * n_pixels + 4096;
        /* Two workspaces, used in turn, when the call has more than one batch: batch k's shade and unwind kernels then run on a
         * stream of their own beside batch k+1's chain kernel (A/B: RT_AMD_DIST_PIPELINE=0: one workspace, everything in line) */
        bool pipeline = true;
        if (const char *v = getenv("RT_AMD_DIST_PIPELINE")) pipeline = !(*v == '0');
        /* the chain kernel's pixels grouped by cost when a lane gets two of them at most (rt_kernels.h DistParams::pixel_order);
         * A/B: RT_AMD_DIST_BY_COST=0 never, =1 always */
        bool by_cost = (n_pixels + 63u) / 64u <= 2u * (size_t)rt::dist_chain_waves(dist_waves);
        if (const char *v = getenv("RT_AMD_DIST_BY_COST")) by_cost = !(*v == '0');
        dp.own_first_chunk = (n_pixels + 63u) / 64u <= 2u * (size_t)rt::dist_chain_waves(dist_waves) ? 1u : 0u; /* rt_kernels.h */
        if (const char *v = getenv("RT_AMD_DIST_OWN_FIRST")) dp.own_first_chunk = *v == '0' ? 0u : 1u; /* A/B */
        bool prep_first = true; /* A/B: RT_AMD_DIST_PREP_FIRST=0: shade kernel and look-ahead start together */
        if (const char *v = getenv("RT_AMD_DIST_PREP_FIRST")) prep_first = !(*v == '0');
        size_t cap = (size_t)(pipeline ? 32768 : 16384) << 20;
        if (const char *v = getenv("RT_AMD_DIST_WS_MB")) {
            if (*v) cap = (size_t)strtoull(v, nullptr, 10) << 20;
        }
        uint32_t batch = (uint32_t)std::min<size_t>(std::min<size_t>(n_epochs, 16), std::max<size_t>(1, cap / per_epoch));
        if (pipeline && batch < n_epochs) /* more than one batch: each workspace gets half the cap */
            batch = (uint32_t)std::min<size_t>(batch, std::max<size_t>(1, cap / 2u / per_epoch));
        if (batch < n_epochs) batch = (n_epochs + (n_epochs + batch - 1u) / batch - 1u) / ((n_epochs + batch - 1u) / batch); /* as many batches, of equal size */
        uint32_t n_buf = pipeline && batch < n_epochs ? 2u : 1u;
        size_t o_hdr = 0, o_req = 0, o_shade = 0, o_frame = 0;
        auto layout = [&](uint32_t epochs) { /* -> bytes of ONE workspace */
            auto carve = [](size_t &off, size_t bytes) { const size_t at = off; off = (off + bytes + 255u) & ~(size_t)255u; return at; };
            const size_t n_samples = n_pixels * epochs;
            size_t off = 0;
            o_hdr = carve(off, n_samples * sizeof(uint32_t));
            o_req = carve(off, n_samples * slots * 4u * sizeof(uint4));
            o_shade = carve(off, n_samples * slots * sizeof(float4));
            o_frame = carve(off, n_samples * (slots - 1u) * sizeof(float4));
            return off;
        };
        char *base = nullptr;
        size_t buf_stride = 0;
        {
            std::lock_guard<std::mutex> lock(mut->ws_mutex);
            Workspace &ws = mut->workspaces[stream];
            if (!ws.d_counters) RT_HIP(hipMalloc(reinterpret_cast<void **>(&ws.d_counters), 256));
            size_t need = layout(batch) * n_buf;
            if (ws.d_split && ws.split_bytes < need) {
                /* A workspace that holds at least half the batch wanted is used as it is: giving back and obtaining
                 * gigabytes costs far more than the shorter batches do (measured: 0.65 s to replace a 14 GB workspace
                 * by a 16 GB one, against 0.15 s for the 64 epochs the call was made for; profiles/README.md) */
                uint32_t fit = batch;
                while (fit > 1u && layout(fit) * n_buf > ws.split_bytes) fit -= 1u;
                if (layout(fit) * n_buf <= ws.split_bytes && fit * 2u >= batch) batch = fit;
                need = layout(batch) * n_buf;
            }
            if (ws.split_bytes < need) {
                if (ws.d_split) {
                    RT_HIP(hipStreamSynchronize(stream));
                    RT_HIP(hipFree(ws.d_split));
                    ws.d_split = nullptr;
                    ws.split_bytes = 0;
                }
                /* no room for the batch the cap allows: halve it; no room for one epoch: the one-kernel organisation */
                int refuse = 0; /* test hook: pretend the first n allocations fail (tests/test_gpu_distributed_parity.py) */
                if (const char *v = getenv("RT_AMD_DIAG_WS_REFUSE")) refuse = atoi(v);
                while (refuse-- > 0 || hipMalloc(&ws.d_split, need) != hipSuccess) {
                    (void)hipGetLastError();
                    ws.d_split = nullptr;
                    if (batch == 1u && n_buf == 1u) break;
                    if (batch == 1u) n_buf = 1u;
                    else batch = (batch + 1u) / 2u;
                    need = layout(batch) * n_buf;
                }
                ws.split_bytes = ws.d_split ? need : 0;
            }
            dp.work_queue = ws.d_counters;
            base = static_cast<char *>(ws.d_split);
            buf_stride = layout(batch); /* sets the offsets for the batch size settled on */
        }
        if (base == nullptr) goto one_kernel;
        if (batch >= n_epochs) n_buf = 1u;
        dp.sp_slots = slots;
        uint32_t k = 0;
        bool tail_used[2] = {false, false};
        hipError_t e = hipSuccess;
        for (uint32_t e0 = 0; e0 < n_epochs && e == hipSuccess; e0 += batch, ++k) {
            /* the layout is [slot][sample of THIS batch]: a short last batch just uses a prefix of every array */
            const uint32_t b = n_buf == 2u ? (k & 1u) : 0u;
            char *const ws_base = base + (size_t)b * buf_stride;
            dp.sp_hdr = reinterpret_cast<uint32_t *>(ws_base + o_hdr);
            dp.sp_req = reinterpret_cast<uint4 *>(ws_base + o_req);
            dp.sp_shade = reinterpret_cast<float4 *>(ws_base + o_shade);
            dp.sp_frame = reinterpret_cast<float4 *>(ws_base + o_frame);
            dp.epoch0 = e0;
            dp.n_epochs = std::min(batch, n_epochs - e0);
            e = hipMemsetAsync(dp.work_queue, 0, sizeof(uint32_t), stream);
            if (e == hipSuccess && tail_used[b]) e = hipStreamWaitEvent(stream, rng->ev_tail[b], 0); /* the unwind two batches ago has read this workspace */
            if (e == hipSuccess && lookahead && !rng->ahead) e = rt::launch_rng_prepare(rng->d_states, (uint32_t)n_pixels, rng->d_list, rng->compute_units, stream);
            rng->ahead = false; /* the chain kernel uses blocks up */
            rng->main_stream = stream;
            /* the pixels in the order of what they cost in the batch that used this workspace last (two batches ago in a pipelined
             * call, the last one else): rt_kernels.h DistParams::pixel_order */
            uint32_t *const pix_cost = rng->d_pix, *const pix_order = rng->d_pix + (size_t)(1u + b) * n_pixels, *const pix_scratch = rng->d_pix + 3u * n_pixels;
            dp.pixel_cost = by_cost ? pix_cost : nullptr;
            dp.pixel_order = by_cost && rng->order_valid[b] ? pix_order : nullptr;
            if (e == hipSuccess) e = rt::launch_dist_chain(scene->ks, kf, dp, dist_waves, stream);
            /* from here on this batch does not touch the RNG records: the look-ahead for the next one, on its own stream */
            if (e == hipSuccess && lookahead && overlap) e = lookahead_after_chain(rng);
            if (n_buf == 2u) {
                if (e == hipSuccess) e = hipEventRecord(rng->ev_tail[b], stream); /* first: the chain kernel has written workspace b ... */
                if (e == hipSuccess) e = hipStreamWaitEvent(rng->tail, rng->ev_tail[b], 0);
                /* the next chain kernel waits for the look-ahead, and the look-ahead's workgroups need 64 KB of LDS each: the shade
                 * kernel starts after it instead of taking that LDS first (between two chain kernels of a 1/8 share of the 1080p
                 * frame 1.1 -> 0.3 ms: 0.38 -> 0.365 ms per epoch, a 1/4 share 0.553 -> 0.517, the whole frame 1.685 -> 1.669) */
                if (e == hipSuccess && prep_first && rng->ahead) e = hipStreamWaitEvent(rng->tail, rng->ev_prepared, 0);
                if (e == hipSuccess) e = rt::launch_dist_shade_unwind(scene->ks, kf, dp, rng->tail);
                if (e == hipSuccess && by_cost) {
                    e = rt::launch_dist_pixel_order(pix_cost, pix_order, (uint32_t)n_pixels, pix_scratch, rng->tail);
                    rng->order_valid[b] = e == hipSuccess;
                }
                if (e == hipSuccess) e = hipEventRecord(rng->ev_tail[b], rng->tail); /* ... then: and the unwind has read it */
                tail_used[b] = e == hipSuccess;
            } else if (e == hipSuccess) {
                e = rt::launch_dist_shade_unwind(scene->ks, kf, dp, stream);
                if (e == hipSuccess && by_cost) {
                    e = rt::launch_dist_pixel_order(pix_cost, pix_order, (uint32_t)n_pixels, pix_scratch, stream);
                    rng->order_valid[b] = e == hipSuccess;
                }
            }
            /* the next chain kernel — of this call or, on whatever stream is ordered after this one, of the next — needs the prepared blocks */
            if (e == hipSuccess && rng->ahead) e = hipStreamWaitEvent(stream, rng->ev_prepared, 0);
        }
        /* everything the call started is behind the caller's stream again */
        for (uint32_t b = 0; b < 2u; ++b)
            if (tail_used[b]) { const hipError_t e2 = hipStreamWaitEvent(stream, rng->ev_tail[b], 0); if (e == hipSuccess) e = e2; }
        if (e != hipSuccess) return fail_hip("rt_render_distributed: launch", e);
        return RT_OK;
    }
one_kernel:
    dp.n_epochs = n_epochs;
    dp.epoch0 = 0;
    dp.sp_hdr = nullptr; dp.sp_req = nullptr; dp.sp_shade = nullptr; dp.sp_frame = nullptr; dp.sp_slots = 0;
    {
        const char *v = getenv("RT_AMD_DIST_STATIC"); /* A/B: one 64-pixel chunk per wave instead of persistent lanes */
        if (!(v && *v == '1')) {
            std::lock_guard<std::mutex> lock(mut->ws_mutex);
            Workspace &ws = mut->workspaces[stream];
            if (!ws.d_counters) RT_HIP(hipMalloc(reinterpret_cast<void **>(&ws.d_counters), 256));
            dp.work_queue = ws.d_counters;
        }
    }
    hipError_t e = hipSuccess;
    if (dp.work_queue) e = hipMemsetAsync(dp.work_queue, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess && lookahead && !rng->ahead) e = rt::launch_rng_prepare(rng->d_states, (uint32_t)n_pixels, rng->d_list, rng->compute_units, stream);
    rng->ahead = false;
    if (e == hipSuccess) e = rt::launch_distributed(scene->ks, kf, dp, dist_waves, stream);
    if (e != hipSuccess) return fail_hip("rt_render_distributed: launch", e);
    return RT_OK;
}

int rt_render_distributed_host(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                               rt_rng *rng, uint32_t n_epochs, float *h_accum, unsigned long long *h_ray_count) {
    if (!scene || !rng || !h_accum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed_host: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed_host: bad frame");
    const size_t bytes = (size_t)rt_frame_pixels(frame) * 3 * sizeof(float);
    float *d_accum = nullptr;
    unsigned long long *d_cnt = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_accum), bytes));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpy(d_accum, h_accum, bytes, hipMemcpyHostToDevice); /* img continues from the caller's sums */
    int rc = RT_OK;
    if (e == hipSuccess) {
        rc = rt_render_distributed(scene, camera, frame, focus, blur, rng, n_epochs, d_accum, nullptr, nullptr, d_cnt, nullptr);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(h_accum, d_accum, bytes, hipMemcpyDeviceToHost);
            unsigned long long cnt = 0;
            if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_ray_count) *h_ray_count = cnt;
        }
    }
    (void)hipFree(d_accum);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail_hip("rt_render_distributed_host", e);
    return RT_OK;
}

/* ---- several devices from one process ------------------------------------------------
 * The sharding of dist.py (interleaved row bands, the scene replicated, no data-path collective, one gather of
 * the bands) for a host that is neither Python nor MPI: the Rust main() the boundary is designed for.  One scene copy, one
 * stream and one band buffer per entry of `devices`; an entry may repeat (several bands on one GPU: how the one-GPU test
 * box exercises this).  Bands are rendered concurrently, copied to pinned host memory and de-interleaved on the host —
 * the destination is a host image anyway. */
/* the caller's current device, put back on every way out of an rt_multi_* call */
struct DeviceRestore {
    int prev = 0;
    DeviceRestore() { (void)hipGetDevice(&prev); }
    ~DeviceRestore() { (void)hipSetDevice(prev); }
};

struct rt_multi {
    struct Part {
        int device = 0;
        rt_scene *scene = nullptr;
        hipStream_t stream = nullptr;
        float *d_band = nullptr;
        size_t band_floats = 0;
        unsigned long long *d_count = nullptr;
        float *h_band = nullptr; /* pinned */
        size_t h_floats = 0;
        rt_rng *rng = nullptr; /* stochastic pass: the streams of this part's rows */
        rt_frame rng_frame;
        hipEvent_t done = nullptr;   /* recorded on `stream` after a band is rendered (the device-resident entry points) */
        float *d_stage = nullptr;    /* on parts[0].device: where a band of another device lands before it is de-interleaved */
        size_t stage_floats = 0;
        unsigned long long *d_stage_count = nullptr; /* likewise its cast count */
    };
    hipEvent_t ready = nullptr; /* recorded on the caller's stream: the parts' streams wait for it before touching the image */
    std::vector<Part> parts;
    rt_frame rng_for; /* the frame the generators were created for */
    bool have_rng = false;
};

static void multi_part_frame(const rt_frame *f, int r, int n, rt_frame *out) {
    *out = *f;
    out->y0 = f->y0 + (uint32_t)r * f->y_step;
    out->y_step = f->y_step * (uint32_t)n;
}

int rt_multi_destroy(rt_multi *m) {
    if (!m) return RT_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto &p : m->parts) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.rng) (void)rt_rng_destroy(p.rng);
        if (p.scene) (void)rt_scene_destroy(p.scene);
        if (p.d_band) (void)hipFree(p.d_band);
        if (p.d_count) (void)hipFree(p.d_count);
        if (p.h_band) (void)hipHostFree(p.h_band);
        if (p.done) (void)hipEventDestroy(p.done);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    if (!m->parts.empty()) {
        (void)hipSetDevice(m->parts[0].device);
        for (auto &p : m->parts) {
            if (p.d_stage) (void)hipFree(p.d_stage);
            if (p.d_stage_count) (void)hipFree(p.d_stage_count);
        }
        if (m->ready) (void)hipEventDestroy(m->ready);
    }
    (void)hipSetDevice(prev);
    delete m;
    return RT_OK;
}

int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, rt_multi **out) {
    if (!desc || !devices || !out || n_devices < 1) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: null argument or no devices");
    *out = nullptr;
    int n_visible = 0;
    RT_HIP(hipGetDeviceCount(&n_visible));
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= n_visible) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: device index out of range");
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!m) return fail(RT_ERR_OUT_OF_MEMORY, "rt_multi_create: host allocation failed");
    int prev = 0;
    (void)hipGetDevice(&prev);
    m->parts.resize((size_t)n_devices);
    int rc = RT_OK;
    for (int i = 0; i < n_devices && rc == RT_OK; ++i) {
        rt_multi::Part &p = m->parts[(size_t)i];
        p.device = devices[i];
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p.d_count), sizeof(unsigned long long));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p.done, hipEventDisableTiming);
        if (e != hipSuccess) { rc = fail_hip("rt_multi_create", e); break; }
        rc = rt_scene_create(desc, &p.scene);
    }
    if (rc == RT_OK) { /* on the first entry's device: the staging cast counts and the caller-stream event */
        hipError_t e = hipSetDevice(m->parts[0].device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ready, hipEventDisableTiming);
        for (int i = 0; i < n_devices && e == hipSuccess; ++i)
            e = hipMalloc(reinterpret_cast<void **>(&m->parts[(size_t)i].d_stage_count), sizeof(unsigned long long));
        if (e != hipSuccess) rc = fail_hip("rt_multi_create", e);
    }
    (void)hipSetDevice(prev);
    if (rc != RT_OK) {
        const std::string msg = g_error;
        (void)rt_multi_destroy(m);
        g_error = msg;
        return rc;
    }
    *out = m;
    return RT_OK;
}

/* band buffers of the parts for `frame` (grow-only); returns the rows of part r in rows_out[r] */
static int multi_prepare(rt_multi *m, const rt_frame *frame, std::vector<rt_frame> *frames, bool host_bands) {
    const int n = (int)m->parts.size();
    frames->resize((size_t)n);
    for (int r = 0; r < n; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        multi_part_frame(frame, r, n, &(*frames)[(size_t)r]);
        const rt_frame &pf = (*frames)[(size_t)r];
        const size_t floats = pf.y0 < pf.y1 ? (size_t)rt_frame_pixels(&pf) * 3u : 0u;
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess && floats > p.band_floats) {
            if (p.d_band) (void)hipFree(p.d_band);
            p.d_band = nullptr;
            p.band_floats = 0;
            e = hipMalloc(reinterpret_cast<void **>(&p.d_band), floats * sizeof(float));
            if (e == hipSuccess) p.band_floats = floats;
        }
        if (e == hipSuccess && host_bands && floats > p.h_floats) {
            if (p.h_band) (void)hipHostFree(p.h_band);
            p.h_band = nullptr;
            p.h_floats = 0;
            e = hipHostMalloc(reinterpret_cast<void **>(&p.h_band), floats * sizeof(float), hipHostMallocDefault);
            if (e == hipSuccess) p.h_floats = floats;
        }
        if (e == hipSuccess && !host_bands && floats > p.stage_floats && (p.device != m->parts[0].device || getenv("RT_AMD_MULTI_FORCE_STAGE") != nullptr)) {
            e = hipSetDevice(m->parts[0].device);
            if (e == hipSuccess && p.d_stage) (void)hipFree(p.d_stage);
            p.d_stage = nullptr;
            p.stage_floats = 0;
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p.d_stage), floats * sizeof(float));
            if (e == hipSuccess) p.stage_floats = floats;
        }
        if (e != hipSuccess) return fail_hip("rt_multi: band buffers", e);
    }
    return RT_OK;
}

/* image row k of the tile (k-th rendered row) belongs to part k % n, its row k / n */
static void multi_deinterleave(const rt_multi *m, const rt_frame *frame, float *h_rgb) {
    const size_t n = m->parts.size();
    const size_t cols = frame->x1 - frame->x0;
    const size_t rows = rt_frame_rows(frame);
    for (size_t k = 0; k < rows; ++k) {
        const rt_multi::Part &p = m->parts[k % n];
        const float *src = p.h_band + (k / n) * cols * 3u;
        float *dst = h_rgb + k * cols * 3u;
        memcpy(dst, src, cols * 3u * sizeof(float));
    }
}

/* the generators: created on the first call for a frame and kept (the streams continue from call to call, main.rs:1131); a
 * different frame starts new ones, as a new rt_rng_create would */
static int multi_generators(rt_multi *m, const rt_frame *frame, const std::vector<rt_frame> &frames) {
    if (m->have_rng && memcmp(&m->rng_for, frame, sizeof *frame) == 0) return RT_OK;
    const int n = (int)m->parts.size();
    int rc = RT_OK;
    for (int r = 0; r < n && rc == RT_OK; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        const hipError_t e = hipSetDevice(p.device);
        if (e != hipSuccess) { rc = fail_hip("rt_multi: generators", e); break; }
        if (p.rng) { (void)rt_rng_destroy(p.rng); p.rng = nullptr; }
        if (frames[(size_t)r].y0 < frames[(size_t)r].y1) rc = rt_rng_create(&frames[(size_t)r], &p.rng);
    }
    m->rng_for = *frame;
    m->have_rng = rc == RT_OK;
    return rc;
}

int rt_multi_render_whitted_host(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float *h_rgb, unsigned long long *h_ray_count) {
    if (!m || !camera || !h_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_whitted_host: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_whitted_host: bad frame");
    DeviceRestore restore;
    ProfilingOff no_profiling;
    std::vector<rt_frame> frames;
    int rc = multi_prepare(m, frame, &frames, true);
    const int n = (int)m->parts.size();
    for (int r = 0; r < n && rc == RT_OK; ++r) { /* every part's band is in flight before the first is waited for */
        rt_multi::Part &p = m->parts[(size_t)r];
        const rt_frame &pf = frames[(size_t)r];
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemsetAsync(p.d_count, 0, sizeof(unsigned long long), p.stream);
        if (e != hipSuccess) { rc = fail_hip("rt_multi_render_whitted_host", e); break; }
        if (!(pf.y0 < pf.y1)) continue; /* more parts than rows */
        rc = rt_render_whitted(p.scene, camera, &pf, p.d_band, p.d_count, p.stream);
        if (rc == RT_OK) {
            e = hipMemcpyAsync(p.h_band, p.d_band, (size_t)rt_frame_pixels(&pf) * 3u * sizeof(float), hipMemcpyDeviceToHost, p.stream);
            if (e != hipSuccess) rc = fail_hip("rt_multi_render_whitted_host: copy", e);
        }
    }
    unsigned long long total = 0;
    for (int r = 0; r < n; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        (void)hipSetDevice(p.device);
        hipError_t e = hipStreamSynchronize(p.stream);
        unsigned long long c = 0;
        if (e == hipSuccess) e = hipMemcpy(&c, p.d_count, sizeof c, hipMemcpyDeviceToHost);
        if (e != hipSuccess && rc == RT_OK) rc = fail_hip("rt_multi_render_whitted_host: synchronize", e);
        total += c;
    }
    if (rc != RT_OK) return rc;
    multi_deinterleave(m, frame, h_rgb);
    if (h_ray_count) *h_ray_count = total;
    return RT_OK;
}

int rt_multi_render_distributed_host(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float focus, float blur, uint32_t n_epochs,
                                     float *h_accum, unsigned long long *h_ray_count) {
    if (!m || !camera || !h_accum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_distributed_host: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_distributed_host: bad frame");
    DeviceRestore restore;
    ProfilingOff no_profiling;
    std::vector<rt_frame> frames;
    int rc = multi_prepare(m, frame, &frames, true);
    const int n = (int)m->parts.size();
    rc = rc == RT_OK ? multi_generators(m, frame, frames) : rc;
    const size_t cols = frame->x1 - frame->x0;
    const size_t rows = rt_frame_rows(frame);
    for (int r = 0; r < n && rc == RT_OK; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        const rt_frame &pf = frames[(size_t)r];
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemsetAsync(p.d_count, 0, sizeof(unsigned long long), p.stream);
        if (e != hipSuccess) { rc = fail_hip("rt_multi_render_distributed_host", e); break; }
        if (!(pf.y0 < pf.y1)) continue;
        /* img continues from the caller's sums: this part's rows of it, interleaved -> band */
        for (size_t k = (size_t)r, b = 0; k < rows; k += (size_t)n, ++b) memcpy(p.h_band + b * cols * 3u, h_accum + k * cols * 3u, cols * 3u * sizeof(float));
        const size_t bytes = (size_t)rt_frame_pixels(&pf) * 3u * sizeof(float);
        e = hipMemcpyAsync(p.d_band, p.h_band, bytes, hipMemcpyHostToDevice, p.stream);
        if (e != hipSuccess) { rc = fail_hip("rt_multi_render_distributed_host: upload", e); break; }
        rc = rt_render_distributed(p.scene, camera, &pf, focus, blur, p.rng, n_epochs, p.d_band, nullptr, nullptr, p.d_count, p.stream);
        if (rc == RT_OK) {
            e = hipMemcpyAsync(p.h_band, p.d_band, bytes, hipMemcpyDeviceToHost, p.stream);
            if (e != hipSuccess) rc = fail_hip("rt_multi_render_distributed_host: copy", e);
        }
    }
    unsigned long long total = 0;
    for (int r = 0; r < n; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        (void)hipSetDevice(p.device);
        hipError_t e = hipStreamSynchronize(p.stream);
        unsigned long long c = 0;
        if (e == hipSuccess) e = hipMemcpy(&c, p.d_count, sizeof c, hipMemcpyDeviceToHost);
        if (e != hipSuccess && rc == RT_OK) rc = fail_hip("rt_multi_render_distributed_host: synchronize", e);
        total += c;
    }
    if (rc != RT_OK) return rc;
    multi_deinterleave(m, frame, h_accum); /* the bands hold img + samples */
    if (h_ray_count) *h_ray_count = total;
    return RT_OK;
}

/* ---- the same, device-resident: the frame is assembled in device memory on the first entry's device ------------------
 * (a host that is not Python should be able to go render -> rt_post_process_device -> rt_encode_srgb8_device
 * over several GPUs without the bands passing through host memory.)  Every part renders its band on its own stream; the
 * caller's stream — on parts[0].device — waits for each band's event and copies it into its rows of the image: straight
 * from the band buffer when that lives on the same device, else through a staging buffer filled by hipMemcpyPeerAsync
 * (xGMI between two MI355X; no host memory either way).  The copy into the image is strided (hipMemcpy2DAsync: band row b
 * is image row r + b n).  Asynchronous like rt_render_whitted: when the call returns everything is enqueued, and work on
 * `hip_stream` after it sees the finished frame. */
__global__ void multi_add_count_kernel(unsigned long long *dst, const unsigned long long *src) { atomicAdd(dst, *src); }

static int multi_gather_band(rt_multi *m, int r, const rt_frame &pf, const rt_frame *frame, float *d_image, unsigned long long *d_ray_count, hipStream_t stream) {
    rt_multi::Part &p = m->parts[(size_t)r];
    const int n = (int)m->parts.size();
    const size_t cols = frame->x1 - frame->x0;
    const size_t band_rows = rt_frame_rows(&pf);
    const size_t row_bytes = cols * 3u * sizeof(float);
    hipError_t e = hipSetDevice(m->parts[0].device);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, p.done, 0);
    const float *src = p.d_band;
    const unsigned long long *src_count = p.d_count;
    if (e == hipSuccess && p.d_stage != nullptr) { /* another device's band (or the test hook): over the link into the staging buffer */
        e = hipMemcpyPeerAsync(p.d_stage, m->parts[0].device, p.d_band, p.device, band_rows * row_bytes, stream);
        src = p.d_stage;
    }
    if (e == hipSuccess && d_ray_count != nullptr && (p.device != m->parts[0].device || p.d_stage != nullptr)) {
        e = hipMemcpyPeerAsync(p.d_stage_count, m->parts[0].device, p.d_count, p.device, sizeof(unsigned long long), stream);
        src_count = p.d_stage_count;
    }
    if (e == hipSuccess && band_rows != 0u)
        e = hipMemcpy2DAsync(d_image + (size_t)r * cols * 3u, (size_t)n * row_bytes, src, row_bytes, row_bytes, band_rows, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess && d_ray_count != nullptr) {
        hipLaunchKernelGGL(multi_add_count_kernel, dim3(1), dim3(1), 0, stream, d_ray_count, src_count);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail_hip("rt_multi: gathering a band", e);
    return RT_OK;
}

int rt_multi_render_whitted(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float *d_rgb, unsigned long long *d_ray_count, void *hip_stream) {
    if (!m || !camera || !d_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_whitted: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_whitted: bad frame");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    DeviceRestore restore;
    ProfilingOff no_profiling;
    std::vector<rt_frame> frames;
    int rc = multi_prepare(m, frame, &frames, false);
    const int n = (int)m->parts.size();
    for (int r = 0; r < n && rc == RT_OK; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        const rt_frame &pf = frames[(size_t)r];
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemsetAsync(p.d_count, 0, sizeof(unsigned long long), p.stream);
        if (e != hipSuccess) { rc = fail_hip("rt_multi_render_whitted", e); break; }
        if (pf.y0 < pf.y1) rc = rt_render_whitted(p.scene, camera, &pf, p.d_band, p.d_count, p.stream);
        if (rc == RT_OK && (e = hipEventRecord(p.done, p.stream)) != hipSuccess) rc = fail_hip("rt_multi_render_whitted", e);
    }
    for (int r = 0; r < n && rc == RT_OK; ++r) rc = multi_gather_band(m, r, frames[(size_t)r], frame, d_rgb, d_ray_count, stream);
    return rc;
}

int rt_multi_render_distributed(rt_multi *m, const rt_camera *camera, const rt_frame *frame, float focus, float blur, uint32_t n_epochs,
                                float *d_accum, unsigned long long *d_ray_count, void *hip_stream) {
    if (!m || !camera || !d_accum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_distributed: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_distributed: bad frame");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    DeviceRestore restore;
    ProfilingOff no_profiling;
    std::vector<rt_frame> frames;
    int rc = multi_prepare(m, frame, &frames, false);
    rc = rc == RT_OK ? multi_generators(m, frame, frames) : rc;
    const int n = (int)m->parts.size();
    const size_t cols = frame->x1 - frame->x0;
    const size_t row_bytes = cols * 3u * sizeof(float);
    /* img continues from the caller's sums (main.rs:1165): each part first takes its rows of it, once the caller's stream has
     * them ready */
    hipError_t e0 = hipSetDevice(m->parts[0].device);
    if (e0 == hipSuccess) e0 = hipEventRecord(m->ready, stream);
    if (rc == RT_OK && e0 != hipSuccess) rc = fail_hip("rt_multi_render_distributed", e0);
    for (int r = 0; r < n && rc == RT_OK; ++r) {
        rt_multi::Part &p = m->parts[(size_t)r];
        const rt_frame &pf = frames[(size_t)r];
        const size_t band_rows = pf.y0 < pf.y1 ? rt_frame_rows(&pf) : 0u;
        hipError_t e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemsetAsync(p.d_count, 0, sizeof(unsigned long long), p.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.stream, m->ready, 0);
        if (e == hipSuccess && band_rows != 0u) {
            if (p.d_stage != nullptr) { /* rows of the image -> staging (on the image's device), then over the link */
                e = hipSetDevice(m->parts[0].device);
                if (e == hipSuccess)
                    e = hipMemcpy2DAsync(p.d_stage, row_bytes, d_accum + (size_t)r * cols * 3u, (size_t)n * row_bytes, row_bytes, band_rows, hipMemcpyDeviceToDevice, stream);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(p.d_band, p.device, p.d_stage, m->parts[0].device, band_rows * row_bytes, stream);
                if (e == hipSuccess) e = hipEventRecord(p.done, stream);
                if (e == hipSuccess) e = hipSetDevice(p.device);
                if (e == hipSuccess) e = hipStreamWaitEvent(p.stream, p.done, 0);
            } else {
                e = hipMemcpy2DAsync(p.d_band, row_bytes, d_accum + (size_t)r * cols * 3u, (size_t)n * row_bytes, row_bytes, band_rows, hipMemcpyDeviceToDevice, p.stream);
            }
        }
        if (e != hipSuccess) { rc = fail_hip("rt_multi_render_distributed", e); break; }
        if (band_rows != 0u) rc = rt_render_distributed(p.scene, camera, &pf, focus, blur, p.rng, n_epochs, p.d_band, nullptr, nullptr, p.d_count, p.stream);
        if (rc == RT_OK && (e = hipEventRecord(p.done, p.stream)) != hipSuccess) rc = fail_hip("rt_multi_render_distributed", e);
    }
    for (int r = 0; r < n && rc == RT_OK; ++r) rc = multi_gather_band(m, r, frames[(size_t)r], frame, d_accum, d_ray_count, stream);
    return rc;
}

/* ---- post_process / encode on the device ----------------------------------------- */

struct PostWs {
    uint32_t *d_keys = nullptr;
    size_t n = 0;
    uint32_t *d_state = nullptr;
};
/* keyed by (device, stream): the default stream is nullptr on every device, and a buffer allocated on one device must
 * not serve a launch on another.  Grow-only; rt_post_release() frees the buffers of the current device. */
static std::mutex g_post_mutex;
static std::map<std::pair<int, hipStream_t>, PostWs> g_post_ws;

int rt_post_release(void) {
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    RT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_post_mutex);
    for (auto it = g_post_ws.begin(); it != g_post_ws.end();) {
        if (it->first.first == device) {
            if (it->second.d_keys) (void)hipFree(it->second.d_keys);
            if (it->second.d_state) (void)hipFree(it->second.d_state);
            it = g_post_ws.erase(it);
        } else {
            ++it;
        }
    }
    return RT_OK;
}

int rt_post_process_device(float *d_rgb, size_t n_pixels, float *d_divisor, void *hip_stream) {
    if (!d_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_process_device: null argument");
    if (n_pixels == 0) return RT_OK;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint32_t *keys = nullptr, *state = nullptr;
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    {
        std::lock_guard<std::mutex> lock(g_post_mutex);
        PostWs &ws = g_post_ws[std::make_pair(device, stream)];
        if (!ws.d_state) RT_HIP(hipMalloc(reinterpret_cast<void **>(&ws.d_state), 260 * sizeof(uint32_t)));
        if (n_pixels > ws.n) {
            if (ws.d_keys) (void)hipFree(ws.d_keys);
            ws.d_keys = nullptr;
            ws.n = 0;
            RT_HIP(hipMalloc(reinterpret_cast<void **>(&ws.d_keys), n_pixels * sizeof(uint32_t)));
            ws.n = n_pixels;
        }
        keys = ws.d_keys;
        state = ws.d_state;
    }
    float row[3];
    rt::luma_row(row);
    hipError_t e = rt::launch_post_process(d_rgb, n_pixels, row, keys, state, d_divisor, stream);
    if (e != hipSuccess) return fail_hip("rt_post_process_device: launch", e);
    return RT_OK;
}

int rt_accumulate_device(const float *d_samples, const unsigned char *d_valid, uint32_t n_epochs, size_t n_pixels, float *d_sum,
                         float *d_weight, void *hip_stream) {
    if (!d_samples || !d_valid || !d_sum || !d_weight) return fail(RT_ERR_INVALID_ARGUMENT, "rt_accumulate_device: null argument");
    const hipError_t e = rt::launch_accumulate(d_samples, d_valid, n_epochs, n_pixels, d_sum, d_weight, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_accumulate_device: launch", e);
    return RT_OK;
}

int rt_accumulator_resolve_device(const float *d_sum, const float *d_weight, size_t n_pixels, float *d_rgb, void *hip_stream) {
    if (!d_sum || !d_weight || !d_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "rt_accumulator_resolve_device: null argument");
    const hipError_t e = rt::launch_accumulator_resolve(d_sum, d_weight, n_pixels, d_rgb, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_accumulator_resolve_device: launch", e);
    return RT_OK;
}

int rt_encode_srgb8_device(const float *d_rgb, size_t n_values, unsigned char *d_out, void *hip_stream) {
    if (!d_rgb || !d_out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_encode_srgb8_device: null argument");
    hipError_t e = rt::launch_encode_srgb8(d_rgb, n_values, d_out, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_encode_srgb8_device: launch", e);
    return RT_OK;
}

int rt_math_eval_host(int op, const float *x, const float *y, float *out, size_t n) {
    if (!x || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_math_eval_host: null argument");
    rt::math_eval_host(op, x, y, out, n);
    return RT_OK;
}

int rt_math_eval_device(int op, const float *h_x, const float *h_y, float *h_out, size_t n) {
    if (!h_x || !h_out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_math_eval_device: null argument");
    if (n == 0) return RT_OK;
    float *d_x = nullptr, *d_y = nullptr, *d_o = nullptr;
    const size_t bytes = n * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_x), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_y), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_o), bytes);
    if (e == hipSuccess) e = hipMemcpy(d_x, h_x, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = h_y ? hipMemcpy(d_y, h_y, bytes, hipMemcpyHostToDevice) : hipMemset(d_y, 0, bytes);
    if (e == hipSuccess) e = rt::launch_math_eval(op, d_x, d_y, d_o, n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_o, bytes, hipMemcpyDeviceToHost);
    if (d_x) (void)hipFree(d_x);
    if (d_y) (void)hipFree(d_y);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) return fail_hip("rt_math_eval_device", e);
    return RT_OK;
}

} /* extern "C" */
