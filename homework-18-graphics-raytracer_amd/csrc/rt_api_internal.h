/*
 * rt_api_internal.h — what the translation units behind include/rt_amd.h share: the scene handle with its per-stream
 * workspaces, the status/error helpers, the frame checks.  Internal (hidden visibility): the library's exports are the
 * extern "C" entry points of include/rt_amd.h and nothing else.
 *
 *   rt_api.hip         errors, settings, profiling, rt_scene_create/destroy, rt_render_whitted (the per-stream arenas)
 *   rt_api_layout.hip  the device records and the node tree built from the ABI arrays — host only, no HIP call
 *   rt_api_dist.hip    rt_rng_* and rt_render_distributed: batches, the two workspaces, the streams of a pipelined call
 *   rt_api_multi.hip   rt_multi_*: a device list from one process
 *   rt_api_post.hip    post_process / sRGB / accumulator / rt_math_eval entry points
 */
#ifndef RT_API_INTERNAL_H
#define RT_API_INTERNAL_H

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

#define RT_API_HIDDEN __attribute__((visibility("hidden")))

/* Per-(scene, stream) scratch.  Launches on one stream are ordered, so they can share it; other streams get
 * their own.  Grow-only; allocated on the first call that needs it (not inside a graph capture). */
#define RT_WS_COUNTER_BYTES 256u
struct Workspace {
    uint32_t *d_counters = nullptr; /* RT_WS_COUNTER_BYTES, zeroed once: [0] the stochastic pass's chunk counter */
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
    uint32_t *d_bfs = nullptr; /* persistent-wavefront path on a scene walked breadth-first: item and job lists per wave (rt_kernels.h PwParams) */
    size_t bfs_words = 0;
    void *d_split = nullptr; /* split distributed pass: requests, shades and frames of one batch of epochs */
    size_t split_bytes = 0;
};

inline hipError_t ensure_counters(Workspace &ws) {
    if (ws.d_counters) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ws.d_counters), RT_WS_COUNTER_BYTES);
    if (e == hipSuccess) e = hipMemset(ws.d_counters, 0, RT_WS_COUNTER_BYTES);
    return e;
}

struct rt_scene {
    int device;
    void *d_blob; /* one allocation holding every array */
    rt::KernelScene ks;
    uint32_t resident_waves; /* CUs * 4 SIMDs * RT_MIN_WAVES: the persistent grid */
    uint32_t pwf_workgroups;  /* CUs * resident workgroups of the persistent-wavefront kernel */
    std::mutex ws_mutex;
    std::map<hipStream_t, Workspace> workspaces;
};

/* rt_last_error() of the calling thread.  (Thread-local state is reached through functions of the translation unit that defines
 * it: an `extern thread_local` of hidden visibility makes the other units call its weak, undefined initialisation function through
 * a PC-relative address that is not null — a jump to the library's first byte.) */
RT_API_HIDDEN std::string &last_error();
RT_API_HIDDEN int fail(int code, const std::string &msg);
RT_API_HIDDEN int fail_hip(const char *what, hipError_t e);
#define RT_HIP(call)                                          \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return fail_hip(#call, e_);     \
    } while (0)

RT_API_HIDDEN bool frame_ok(const rt_frame *f);
RT_API_HIDDEN bool frame_fits(const rt_frame *f);
RT_API_HIDDEN int make_kernel_frame(const rt_camera *camera, const rt_frame *frame, rt::KernelFrame *kf);

/* the profiling events live on the device that was current when they were made: calls that hop between devices (rt_multi_*) are
 * not profiled (the thread-local switch is theirs) */
RT_API_HIDDEN bool &profiling_off_flag(); /* of the calling thread */
struct ProfilingOff {
    bool prev;
    ProfilingOff() : prev(profiling_off_flag()) { profiling_off_flag() = true; }
    ~ProfilingOff() { profiling_off_flag() = prev; }
};

/* rt_profile_enable also times the depth-of-field pass's kernels (rt_api_dist.hip): is it on for this thread's calls, and the hooks
 * rt_profile_enable / rt_profile_read_distributed are made of */
RT_API_HIDDEN bool profiling_on();
RT_API_HIDDEN void dist_profile_reset();

/* Everything rt_scene_create derives from the ABI arrays, on the host (no HIP call in there): the device records of
 * rt_device_scene.h.  Also behind rt_scene_describe_nodes, which lets a test look at the node array without a GPU. */
struct SceneLayout {
    std::vector<rt::DevTri> tris;
    std::vector<rt::DevTriAttr> attrs;
    std::vector<rt::DevSegment> segments;
    std::vector<rt::DevTriHead> heads;
    std::vector<rt::DevSphere> spheres;
    double scene_extent = 0.0;
};
RT_API_HIDDEN int layout_scene(const rt_scene_desc *desc, SceneLayout &layout);

#endif /* RT_API_INTERNAL_H */
