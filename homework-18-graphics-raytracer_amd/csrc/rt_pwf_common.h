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

template <class Frame>
__device__ __forceinline__ void pw_slot_to_pixel(const Frame &fr, uint32_t slot, uint32_t *row, uint32_t *col) {
    const uint32_t band_slots = fr.cols << 3;
    const uint32_t band = slot / band_slots;
    const uint32_t r = slot - band * band_slots;
    const uint32_t rows_left = fr.rows - (band << 3);
    const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
    *col = r / band_rows;
    *row = (band << 3) + (r - *col * band_rows);
}

/* ---- workgroup-local queues of work items (used by rt_pwf.hip) ----
 * Positions are reserved with one wave-aggregated atomic on `alloc` (lds_append), the items are written, and then counted
 * into `ready[page]` (pa_publish); consumers claim whole pages of 64 positions in order (pa_claim), full or — once nothing
 * fuller is to be had — SEALED: `alloc` is moved to the page boundary by compare-and-swap so that no reservation can slip
 * in, and the page is taken with the items it has.  All bookkeeping is in LDS; see the header of rt_pwf.hip. */
#define PA_SEALED 0x80000000u
/* ready[]: one entry per page (its item count, <= 64, and a sealed flag): a word each, or — PACKED — 16 bits each, two to a
 * word.  LDS is what limits the number of resident workgroups, and a frame of several megapixels has thousands of pages
 * (12 KB of words at 8 Mpixel: one workgroup per CU less); below that the word form is used, which is the faster one (the
 * claims run on one lane, every instruction of theirs at a wave's price).  pa_ready_packed is the rule kernel and launcher
 * share. */
#define PA_SEALED16 0x8000u
#define PA_READY_WORDS(pages, packed) ((packed) ? ((pages) + 1u) / 2u : (pages))
#define PA_READY_UNPACKED_LIMIT 8192u /* bytes: with the kernel's 45 KB of item pages, what leaves room for three workgroups per CU */
__host__ __device__ __forceinline__ bool pa_ready_packed(uint32_t node_cap, uint32_t ring_cap) {
    return ((node_cap + 63u) / 64u + 2u * (ring_cap / 64u)) * 4u > PA_READY_UNPACKED_LIMIT;
}
__device__ __forceinline__ uint32_t pa_ready_shift(uint32_t idx) { return (idx & 1u) << 4; }
/* the queue helpers: inlined at every site (46 % of pwf_kernel's instructions); -DPA_OUTLINE makes them functions (A/B:
 * profiles/r04_ab6.txt) */
#ifdef PA_OUTLINE
#define PA_HELPER __device__ __noinline__
#else
#define PA_HELPER __device__ __forceinline__
#endif
struct PaQueue {
    uint32_t alloc; /* next position to reserve */
    uint32_t taken; /* next PAGE to claim */
};


/* Items and records are stored field-major within pages of 64: field f of entry e sits at ((e >> 6) * F + f) * 64 + (e & 63)
 * (in uint4s).  The 64 lanes of a chunk hold 64 consecutive entries, so every load and store of a field is one contiguous
 * kilobyte (16 cache lines) instead of 64 pieces a record apart (64 lines): the queues' traffic is the same, the number of
 * lines the vector-memory pipeline touches a quarter.  pa_entry() returns the address of field 0; field f is 64 * f further. */
#define PA_F(f) ((f) * 64u)
__device__ __forceinline__ size_t pa_entry(uint32_t e, uint32_t fields) { return ((size_t)(e >> 6) * fields << 6) + (e & 63u); }

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/* lane 0 only.  Try to claim the next page of a queue; on success *start is its first position and the return value its
 * item count.  ready[] holds one word per page (index masked for the rings). */
template <bool PACKED = false>
PA_HELPER uint32_t pa_claim(PaQueue *q, uint32_t *ready, uint32_t page_mask, uint32_t min_partial, uint32_t *start) {
    if (PACKED) {
        for (int tries = 0; tries < 4; ++tries) {
            const uint32_t page = lds_load(&q->taken);
            const uint32_t a = lds_load(&q->alloc);
            if (a <= page * 64u) return 0u; /* empty */
            const uint32_t idx = page & page_mask, sh = pa_ready_shift(idx);
            const uint32_t w = (lds_load(&ready[idx >> 1]) >> sh) & 0xffffu;
            const uint32_t c = w & 0x7fffu;
            if (c == 64u || (w & PA_SEALED16) != 0u) {
                if (atomicCAS(&q->taken, page, page + 1u) == page) {
                    atomicAnd(&ready[idx >> 1], ~(0xffffu << sh));
                    *start = page * 64u;
                    return c;
                }
                continue;
            }
            if (min_partial == 0u || a >= (page + 1u) * 64u || c < min_partial || c != a - page * 64u) return 0u;
            if (atomicCAS(&q->alloc, a, (page + 1u) * 64u) == a) atomicOr(&ready[idx >> 1], PA_SEALED16 << sh);
        }
        return 0u;
    }
    for (int tries = 0; tries < 4; ++tries) {
        const uint32_t page = lds_load(&q->taken);
        const uint32_t a = lds_load(&q->alloc);
        if (a <= page * 64u) return 0u; /* empty */
        const uint32_t w = lds_load(&ready[page & page_mask]);
        const uint32_t c = w & 0xffffu;
        if (c == 64u || (w & PA_SEALED) != 0u) {
            if (atomicCAS(&q->taken, page, page + 1u) == page) {
                ready[page & page_mask] = 0u; /* the slot is reused one lap later at the earliest */
                *start = page * 64u;
                return c;
            }
            continue; /* somebody else took it: look at the next page */
        }
        if (min_partial == 0u || a >= (page + 1u) * 64u || c < min_partial || c != a - page * 64u) return 0u; /* still filling */
        /* a partly filled last page whose reserved positions are all written: close it against further reservations */
        if (atomicCAS(&q->alloc, a, (page + 1u) * 64u) == a) atomicOr(&ready[page & page_mask], PA_SEALED);
    }
    return 0u;
}

/* all lanes.  lds_append for a queue of bounded capacity (`capacity` positions, a multiple of 64; `released` counts the pages
 * whose slots may be written again): *ok is false — for the whole wave — when the items do not fit right now. */
PA_HELPER uint32_t pa_try_append(PaQueue *q, const uint32_t *released, uint32_t capacity, bool want, bool *ok) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    *ok = true;
    if (mask == 0ull) return 0u;
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    const int leader = (int)__builtin_ctzll(mask);
    uint32_t base = 0xffffffffu;
    if ((int)(threadIdx.x & 63u) == leader) {
        for (int tries = 0; tries < 4; ++tries) {
            const uint32_t a = lds_load(&q->alloc);
            if (a + n - (lds_load(released) << 6) > capacity) break; /* full */
            if (atomicCAS(&q->alloc, a, a + n) == a) { base = a; break; }
        }
    }
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (base == 0xffffffffu) { *ok = false; return 0u; }
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

/* all lanes.  The same, taking as many of the wave's items as there is room for: *ok is per lane (the first ones in lane order fit) */
PA_HELPER uint32_t pa_try_append_some(PaQueue *q, const uint32_t *released, uint32_t capacity, bool want, bool *ok) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    *ok = false;
    if (mask == 0ull) return 0u;
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    const int leader = (int)__builtin_ctzll(mask);
    uint32_t base = 0u, take = 0u;
    if ((int)(threadIdx.x & 63u) == leader) {
        for (int tries = 0; tries < 4; ++tries) {
            const uint32_t a = lds_load(&q->alloc);
            const uint32_t used = a - (lds_load(released) << 6);
            const uint32_t t = used >= capacity ? 0u : (capacity - used < n ? capacity - used : n);
            if (t == 0u) break; /* full */
            if (atomicCAS(&q->alloc, a, a + t) == a) { base = a; take = t; break; }
        }
    }
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    take = (uint32_t)__builtin_amdgcn_readlane((int)take, leader);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    *ok = want && rank < take;
    return base + rank;
}

/* all lanes.  Publish `want` items written at positions pos.. (as returned by lds_append): add the per-page counts. */
template <bool PACKED = false>
PA_HELPER void pa_publish(uint32_t *ready, uint32_t page_mask, bool want, uint32_t pos, uint32_t *gen) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* the items are written before they are counted */
    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
    const int leader = (int)__builtin_ctzll(mask);
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)pos, leader); /* the leader holds the lowest position */
    if ((int)(threadIdx.x & 63u) == leader) {
        const uint32_t p0 = first >> 6, p1 = (first + n - 1u) >> 6;
        if (PACKED) {
            const uint32_t i0 = p0 & page_mask, i1 = p1 & page_mask;
            const uint32_t n0 = p0 == p1 ? n : (p1 << 6) - first;
            atomicAdd(&ready[i0 >> 1], n0 << pa_ready_shift(i0));
            if (p0 != p1) atomicAdd(&ready[i1 >> 1], (n - n0) << pa_ready_shift(i1));
        } else if (p0 == p1) {
            atomicAdd(&ready[p0 & page_mask], n);
        } else {
            const uint32_t n0 = (p1 << 6) - first;
            atomicAdd(&ready[p0 & page_mask], n0);
            atomicAdd(&ready[p1 & page_mask], n - n0);
        }
        atomicAdd(gen, 1u); /* wakes the sleepers */
    }
}

/* get_shade's `for light in &self.lights` up to the next shadow cast (main.rs:413-433): advance *light_i to the first
 * light from *light_i on that needs one; false when the loop is over */
template <class Scene> /* KernelScene, by value or in the kernel-argument segment (constant address space) */
__device__ __forceinline__ bool next_shadow_ray(const Scene &sc, uint32_t *light_i, V3 pos, V3 adj_n, DirLight *dl) {
    while (*light_i < sc.n_lights) {
        V3 direction;
        if (light_asks(sc.lights[*light_i], sc.light_aux[*light_i], pos, adj_n, &direction)) {
            (void)approximate_into_directional(sc.lights[*light_i], pos, dl); /* the light asks: its colour too */
            return true;
        }
        *light_i += 1u;
    }
    return false;
}

/* all lanes.  The same search with the wave in step: every lane that is still `searching` asks light `first`, then first + 1,
 * ... (first is wave-uniform, so each light's record comes through scalar loads and only the code for its kind runs) */
template <class Scene> /* KernelScene, by value or in the kernel-argument segment (constant address space) */
__device__ __forceinline__ bool next_shadow_ray_in_step(const Scene &sc, uint32_t first, bool searching, uint32_t *light_i, V3 pos, V3 adj_n) {
    bool found = false;
    for (uint32_t li = first; li < sc.n_lights; ++li) {
        if (__builtin_amdgcn_ballot_w64(searching) == 0ull) break;
        V3 direction;
        const bool asks = light_asks(uniform_ref(sc.lights + li), uniform_ref(sc.light_aux + li), pos, adj_n, &direction);
        if (searching && asks) {
            *light_i = li;
            found = true;
            searching = false;
        }
    }
    return found;
}

} /* namespace rt */

#endif /* RT_PWF_COMMON_H */
