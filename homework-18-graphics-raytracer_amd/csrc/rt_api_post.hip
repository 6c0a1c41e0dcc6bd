/*
 * rt_api_post.hip — what follows the render path on the device (main.rs:748-762, image.rs:55-66, photon.rs): post_process, the
 * sRGB / u8 encode, the accumulator, and the rt_math_eval diagnostics.  Kernels: rt_post.hip.
 */
#include "rt_api_internal.h"

namespace rt {
void math_eval_host(int op, const float *x, const float *y, float *out, size_t n);
}

extern "C" {

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

/* the passes of rt_post_process_device one by one, on the caller's device memory (include/rt_amd.h) */
int rt_post_keys_device(const float *d_rgb, size_t n_pixels, uint32_t *d_keys, uint32_t *d_state, void *hip_stream) {
    if (!d_state || (n_pixels && (!d_rgb || !d_keys))) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_keys_device: null argument");
    float row[3];
    rt::luma_row(row);
    const hipError_t e = rt::launch_post_keys(d_rgb, n_pixels, row, d_keys, d_state, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_post_keys_device: launch", e);
    return RT_OK;
}

int rt_post_hist_device(const uint32_t *d_keys, size_t n_pixels, int pass, uint32_t *d_state, void *hip_stream) {
    if (!d_state || (n_pixels && !d_keys)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_hist_device: null argument");
    if (pass < 0 || pass > 3) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_hist_device: pass 0..3");
    const hipError_t e = rt::launch_post_hist(d_keys, n_pixels, pass, d_state, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_post_hist_device: launch", e);
    return RT_OK;
}

int rt_post_pick_device(int pass, uint32_t *d_state, void *hip_stream) {
    if (!d_state) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_pick_device: null argument");
    if (pass < 0 || pass > 3) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_pick_device: pass 0..3");
    const hipError_t e = rt::launch_post_pick(pass, d_state, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_post_pick_device: launch", e);
    return RT_OK;
}

int rt_post_scale_device(float *d_rgb, size_t n_pixels, const uint32_t *d_state, float *d_divisor, void *hip_stream) {
    if (!d_state || (n_pixels && !d_rgb)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_post_scale_device: null argument");
    const hipError_t e = rt::launch_post_scale(d_rgb, n_pixels, d_state, d_divisor, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail_hip("rt_post_scale_device: launch", e);
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
