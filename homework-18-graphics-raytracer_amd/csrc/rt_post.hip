/*
 * rt_post.hip — the step right after the render path, on the device:
 *   post_process   src/main.rs:748-762   divide the image by the 99th-percentile luma of its normal lumas
 *   convert_from   src/image.rs:55-66    linear -> sRGB transfer, f32 -> u8
 *
 * The reference sorts all lumas to index one element; selecting the k-th smallest is order independent, so an
 * exact radix select gives the same value: 4 passes over order-preserving 32-bit keys, 8 bits per pass, each pass
 * a histogram of the keys that still match the chosen prefix.  HBM-bound byte work: one coalesced read of the
 * key array per pass (8 MB at 1080p), no sort.  The arithmetic (luma row, divide, transfer function, truncating
 * u8 cast) is the host's (rt_host.cpp) operation for operation, so results are bit-identical to it and to the oracle.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_detmath.h"
#include "rt_kernels.h"

namespace rt {

#define KEY_INVALID 0xffffffffu

/* monotone map f32 -> u32 (negative floats reversed below the positives) */
__device__ __forceinline__ uint32_t order_key(float x) {
    const uint32_t b = rtdm::f32_bits(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return rtdm::f32_from_bits(b);
}

/* state[0] = number of normal lumas, state[1] = k (0-based rank wanted), state[2] = chosen key prefix,
 * state[3] = result key, state[4..259] = histogram */
__global__ void post_keys_kernel(const float *__restrict__ rgb, uint32_t *__restrict__ keys, size_t n, float r0, float r1, float r2,
                                 uint32_t *__restrict__ state) {
    uint32_t local = 0u;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        /* into_luma: the Y row of palette's rgb->xyz matrix, (c3*r + c4*g) + c5*b */
        const float l = (r0 * rgb[3 * i] + r1 * rgb[3 * i + 1]) + r2 * rgb[3 * i + 2];
        const bool ok = rtdm::is_normal(l); /* main.rs:751 */
        keys[i] = ok ? order_key(l) : KEY_INVALID;
        local += ok ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63u) == 0u && local != 0u) atomicAdd(&state[0], local);
}

__global__ void post_rank_kernel(uint32_t *state) {
    /* (luma_cumulative.len() as f32 * 0.99) as usize, main.rs:754 */
    const uint32_t len = state[0];
    uint32_t k = (uint32_t)((float)len * 0.99f);
    if (len != 0u && k >= len) k = len - 1u;
    state[1] = k;
    state[2] = 0u;
    state[3] = KEY_INVALID;
}

__global__ void post_hist_kernel(const uint32_t *__restrict__ keys, size_t n, int pass, uint32_t *__restrict__ state) {
    __shared__ uint32_t hist[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const uint32_t prefix = state[2];
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
        if (k != KEY_INVALID && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x)
        if (hist[i] != 0u) atomicAdd(&state[4 + i], hist[i]);
}

__global__ void post_pick_kernel(int pass, uint32_t *state) {
    if (threadIdx.x != 0) return;
    if (state[0] == 0u) return;
    const int shift = 24 - 8 * pass;
    uint32_t k = state[1];
    uint32_t digit = 255u;
    for (uint32_t d = 0; d < 256u; ++d) {
        const uint32_t c = state[4 + d];
        if (k < c) { digit = d; break; }
        k -= c;
    }
    state[1] = k;
    state[2] |= digit << shift;
    for (uint32_t d = 0; d < 256u; ++d) state[4 + d] = 0u;
    if (pass == 3) state[3] = state[2];
}

__global__ void post_scale_kernel(float *__restrict__ rgb, size_t n_values, const uint32_t *__restrict__ state, float *__restrict__ divisor_out) {
    const bool have = state[0] != 0u;
    const float p98 = have ? key_value(state[3]) : 0.0f;
    const bool apply = have && p98 > 1.1920928955078125e-7f; /* main.rs:755 */
    if (blockIdx.x == 0 && threadIdx.x == 0 && divisor_out != nullptr) *divisor_out = apply ? p98 : 0.0f;
    if (!apply) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) rgb[i] = rgb[i] / p98;
}

__global__ void encode_srgb8_kernel(const float *__restrict__ rgb, size_t n_values, unsigned char *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) {
        const float x = rgb[i];
        const float e = (x <= 0.0031308f) ? 12.92f * x : 1.055f * rtdm::powf(x, 1.0f / 2.4f) - 0.055f;
        float r = e * 255.0f; /* palette 0.4 f32 -> u8: scale, clamp, truncate */
        if (!(r > 0.0f)) r = 0.0f;
        if (r > 255.0f) r = 255.0f;
        out[i] = (unsigned char)r;
    }
}

/* PhotonAccumulator::accumulate (photon.rs:25-28) for the samples that passed the filter, epochs in order; one thread
 * per pixel, so a pixel's additions happen in the reference's order */
__global__ void accumulate_kernel(const float *__restrict__ samples, const unsigned char *__restrict__ valid, uint32_t n_epochs,
                                  size_t n_pixels, float *__restrict__ sum, float *__restrict__ weight) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += stride) {
        float s0 = sum[3 * i], s1 = sum[3 * i + 1], s2 = sum[3 * i + 2], w = weight[i];
        for (uint32_t e = 0; e < n_epochs; ++e) {
            if (valid[(size_t)e * n_pixels + i] == 0u) continue;
            const float *ph = samples + ((size_t)e * n_pixels + i) * 3u;
            s0 = s0 + ph[0];
            s1 = s1 + ph[1];
            s2 = s2 + ph[2];
            w += 1.0f;
        }
        sum[3 * i] = s0; sum[3 * i + 1] = s1; sum[3 * i + 2] = s2;
        weight[i] = w;
    }
}

/* into_rgb_internal (photon.rs:15-23) */
__global__ void accumulator_resolve_kernel(const float *__restrict__ sum, const float *__restrict__ weight, size_t n_pixels,
                                           float *__restrict__ rgb) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += stride) {
        const float w = weight[i];
        const bool empty = w < 1.1920928955078125e-7f; /* std::f32::EPSILON */
        rgb[3 * i] = empty ? 0.0f : sum[3 * i] / w;
        rgb[3 * i + 1] = empty ? 0.0f : sum[3 * i + 1] / w;
        rgb[3 * i + 2] = empty ? 0.0f : sum[3 * i + 2] / w;
    }
}

static unsigned grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 2048) b = 2048; /* grid-stride the rest (memory-bound: 256 CUs x 8 blocks) */
    if (b == 0) b = 1;
    return (unsigned)b;
}

/* The passes one by one, on caller-owned memory (keys: n_pixels u32; state: RT_POST_STATE_WORDS u32) — so that a frame whose bands
 * live on several ranks can sum `state` over them between the passes (dist.post_process_sharded: all-reduce of state[0] after the
 * keys, of state[4..259] after each histogram) with everything stream-ordered and no host round trip:
 *   keys   zeroes state; keys[i] <- order key of pixel i's luma; state[0] <- this band's count of normal lumas
 *   hist   (pass 0 first turns state[0] — by now the count over all bands — into the rank wanted, main.rs:754) state[4..259] += the
 *          histogram of this band's keys that match the prefix chosen so far
 *   pick   the digit of the wanted rank from the (summed) histogram; clears it
 *   scale  divide by the selected luma when it is above f32::EPSILON (main.rs:755-760) */
hipError_t launch_post_keys(const float *rgb, size_t n_pixels, const float luma_row[3], uint32_t *keys, uint32_t *state, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(state, 0, 260 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (n_pixels != 0)
        hipLaunchKernelGGL(post_keys_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, stream, rgb, keys, n_pixels, luma_row[0], luma_row[1],
                           luma_row[2], state);
    return hipGetLastError();
}
hipError_t launch_post_hist(const uint32_t *keys, size_t n_pixels, int pass, uint32_t *state, hipStream_t stream) {
    if (pass == 0) hipLaunchKernelGGL(post_rank_kernel, dim3(1), dim3(1), 0, stream, state);
    if (n_pixels != 0) hipLaunchKernelGGL(post_hist_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, stream, keys, n_pixels, pass, state);
    return hipGetLastError();
}
hipError_t launch_post_pick(int pass, uint32_t *state, hipStream_t stream) {
    hipLaunchKernelGGL(post_pick_kernel, dim3(1), dim3(64), 0, stream, pass, state);
    return hipGetLastError();
}
hipError_t launch_post_scale(float *rgb, size_t n_pixels, const uint32_t *state, float *divisor_out, hipStream_t stream) {
    hipLaunchKernelGGL(post_scale_kernel, dim3(grid_for(n_pixels * 3)), dim3(256), 0, stream, rgb, n_pixels * 3, state, divisor_out);
    return hipGetLastError();
}

/* all of it on one device: keys: n_pixels u32; state: 260 u32 (zeroed here) */
hipError_t launch_post_process(float *rgb, size_t n_pixels, const float luma_row[3], uint32_t *keys, uint32_t *state,
                               float *divisor_out, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    hipError_t e = launch_post_keys(rgb, n_pixels, luma_row, keys, state, stream);
    for (int pass = 0; pass < 4 && e == hipSuccess; ++pass) {
        e = launch_post_hist(keys, n_pixels, pass, state, stream);
        if (e == hipSuccess) e = launch_post_pick(pass, state, stream);
    }
    if (e == hipSuccess) e = launch_post_scale(rgb, n_pixels, state, divisor_out, stream);
    return e;
}

hipError_t launch_accumulate(const float *samples, const unsigned char *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight,
                             hipStream_t stream) {
    if (n_pixels == 0 || n_epochs == 0) return hipSuccess;
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, stream, samples, valid, n_epochs, n_pixels, sum, weight);
    return hipGetLastError();
}

hipError_t launch_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(accumulator_resolve_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, stream, sum, weight, n_pixels, rgb);
    return hipGetLastError();
}

hipError_t launch_encode_srgb8(const float *rgb, size_t n_values, unsigned char *out, hipStream_t stream) {
    if (n_values == 0) return hipSuccess;
    hipLaunchKernelGGL(encode_srgb8_kernel, dim3(grid_for(n_values)), dim3(256), 0, stream, rgb, n_values, out);
    return hipGetLastError();
}

} /* namespace rt */
