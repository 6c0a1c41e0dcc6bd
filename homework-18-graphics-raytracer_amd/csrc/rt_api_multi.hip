/*
 * rt_api_multi.hip — rt_multi_*: one process, a list of devices (include/rt_amd.h), host-image and device-resident forms.
 */
#include "rt_api_internal.h"

extern "C" {

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
        hipEvent_t staged = nullptr; /* on parts[0].device, recorded on the CALLER's stream: this part's rows of the image are on their way to it
                                      * (an event is recorded on a stream of its own device only: `done` belongs to `device`, the caller's stream to parts[0].device) */
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
            if (p.staged) (void)hipEventDestroy(p.staged);
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
        for (int i = 0; i < n_devices && e == hipSuccess; ++i) {
            e = hipMalloc(reinterpret_cast<void **>(&m->parts[(size_t)i].d_stage_count), sizeof(unsigned long long));
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m->parts[(size_t)i].staged, hipEventDisableTiming);
        }
        if (e != hipSuccess) rc = fail_hip("rt_multi_create", e);
    }
    (void)hipSetDevice(prev);
    if (rc != RT_OK) {
        const std::string msg = last_error();
        (void)rt_multi_destroy(m);
        last_error() = msg;
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
        if (e == hipSuccess && !host_bands && floats > p.stage_floats && (p.device != m->parts[0].device || rt::option(rt::OPT_MULTI_FORCE_STAGE, 0) != 0)) {
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
                if (e == hipSuccess) e = hipEventRecord(p.staged, stream); /* an event of the caller's stream's device (`done` is p.device's) */
                if (e == hipSuccess) e = hipSetDevice(p.device);
                if (e == hipSuccess) e = hipStreamWaitEvent(p.stream, p.staged, 0);
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

} /* extern "C" */
