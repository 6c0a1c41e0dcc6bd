/*
 * rt_pwf_common.h — item encodings and small helpers of the persistent-wavefront kernel (rt_pwf.hip).  Device code only.
 */
#ifndef RT_PWF_COMMON_H
#define RT_PWF_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_shade.h"
#include "rt_kernels.h"
#include "rt_cast.h"

namespace rt {

#define PW_NO_CHILD 0xffffffffu
#define PW_FINAL 0xfffffffeu  /* record.cr: the stored value is final (miss, or the unscaled shade at depth 0) */
/* the exclusion word of a queued ray also carries: bits 21-26 the depth left for the node, bits 27-28 the ray's face mode */
#define PW_DEPTH_SHIFT 21u
#define PW_MODE_SHIFT 27u
#define PW_EXCL_MASK (0xe0000000u | ((1u << PW_DEPTH_SHIFT) - 1u))


__device__ __forceinline__ uint32_t pfu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float puf(uint32_t x) { return __uint_as_float(x); }

/* wave-aggregated reservation on an LDS counter: lanes with `want` get consecutive positions */
__device__ __forceinline__ uint32_t lds_append(uint32_t *counter, bool want) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return 0u;
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    const int leader = (int)__builtin_ctzll(mask);
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void pw_slot_to_pixel(const KernelFrame &fr, uint32_t slot, uint32_t *row, uint32_t *col) {
    const uint32_t band_slots = fr.cols << 3;
    const uint32_t band = slot / band_slots;
    const uint32_t r = slot - band * band_slots;
    const uint32_t rows_left = fr.rows - (band << 3);
    const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
    *col = r / band_rows;
    *row = (band << 3) + (r - *col * band_rows);
}

/* get_shade's `for light in &self.lights` up to the next shadow cast (main.rs:413-433): advance *light_i to the first
 * light from *light_i on that needs one; false when the loop is over */
__device__ __forceinline__ bool next_shadow_ray(const KernelScene &sc, uint32_t *light_i, V3 pos, V3 adj_n, DirLight *dl) {
    while (*light_i < sc.n_lights) {
        if (approximate_into_directional(sc.lights[*light_i], pos, dl)) {
            const float cosine = -dot(dl->direction, adj_n);
            if (!(cosine <= 0.0f)) return true;
        }
        *light_i += 1u;
    }
    return false;
}

} /* namespace rt */

#endif /* RT_PWF_COMMON_H */
