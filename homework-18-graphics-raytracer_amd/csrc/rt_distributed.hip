/*
 * rt_distributed.hip — the stochastic ("distributed") pass as a HIP kernel for gfx950.
 *
 * Replaces the par_iter_mut closure at src/main.rs:1131-1156: per pixel and epoch,
 *   shoot_focus (2 Gaussian draws, main.rs:101-127) -> cast -> distributed_ray_trace (main.rs:521-614),
 * with the per-pixel IsaacRng (seeded y*2^33 + x, main.rs:1117-1127) resident in HBM and its stream
 * continuing across epochs.  The sample filter of main.rs:1157-1160 (drop unless all three channels
 * are is_normal) and `img[at] += photon` (main.rs:1165) are fused into the kernel.
 *
 * distributed_ray_trace is a chain, not a tree (one scattered ray per level), so the per-lane state
 * machine is: cast -> [select + scatter] -> cast next -> shade next (3 shadow casts) -> descend ...,
 * then unwind `get_shade(next).mix(x * brdf, 0.5)` / `(x + get_shade(next)) * decay` through a small
 * frame stack.  Every cast of every lane goes through the same wave-convergent intersection loop as the
 * Whitted kernel (rt_cast.h).  get_shade(&hit) at main.rs:524 is pure and only used at depth <= 0; it is
 * evaluated there only (the oracle follows the same plan, so cast counts agree).
 *
 * rand 0.5 (ISAAC-32, Uniform<f32>, ziggurat Normal) is restated from the crate's published algorithms (the
 * crate is not in the build image).  Pinned: bit-for-bit against oracle/rt_oracle.cpp on every sample, flag and RNG
 * record, and — through main()'s whole progressive loop at 7 epochs — per pixel against the reference's own
 * report/out.png (blur 0.04) and report/out_small_blur.png (blur 0.02): tests/test_gpu_reference_pins.py.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "rt_cast.h"
#include "rt_pwf_common.h"
#include "rt_ziggurat_tables.h"

namespace rt {

/* ---- per-pixel RNG record in HBM ----------------------------------------------------------------------------------
 * Two BANKS of the oracle's layout (mem[256], a, b, c, results[256], one spare word).  Bank `cur` is the generator's
 * state as the reference has it (IsaacRng: the block in use + the position in it); the other bank, when `prepared`, holds
 * the state after the NEXT IsaacCore::generate, computed ahead of time by rng_prepare_kernel.  A lane that runs dry then
 * just switches banks.  Why: generate is 256 steps with two address-dependent loads each — tens of microseconds for one
 * lane with the 63 others of its wave waiting, and after a few dozen epochs the pixels' streams are out of step, so in
 * the render kernels it is always ONE lane (measured: a third of the chain kernel's wave time).  In the prepare pass all
 * the lanes of a wave generate together.  rt_rng_download exports bank `cur` + the position: the oracle's record. */
enum : uint32_t {
    RNG_MEM = 0u, RNG_A = 256u, RNG_B = 257u, RNG_C = 258u, RNG_RESULTS = 259u, RNG_SPARE = 515u, RNG_BANK_WORDS = 516u,
    RNG_INDEX = RNG_SPARE,                      /* bank 0's spare word: position in the current block (256 = used up) */
    RNG_FLAGS = RNG_BANK_WORDS + RNG_SPARE,     /* bank 1's spare word: bit 0 = cur, bit 1 = prepared */
    RNG_WORDS = 2u * RNG_BANK_WORDS
};
static_assert(RNG_WORDS == RT_RNG_DEVICE_WORDS && RNG_BANK_WORDS == RT_RNG_STATE_WORDS, "rt_kernels.h");

__device__ const double ZIG_X[257] = RT_ZIG_NORM_X;
__device__ const double ZIG_F[257] = RT_ZIG_NORM_F;

/* IsaacCore::init(key, rounds = 1) as called by IsaacRng::new_from_u64(seed) */
__device__ void isaac_seed(uint32_t *st, unsigned long long seed) {
    for (uint32_t i = 0; i < 256u; ++i) st[RNG_MEM + i] = 0u;
    st[RNG_MEM + 0] = (uint32_t)seed;
    st[RNG_MEM + 1] = (uint32_t)(seed >> 32);
    uint32_t a = 0x1367df5au, b = 0x95d90059u, c = 0xc3163e4bu, d = 0x0f421ad8u;
    uint32_t e = 0xd92a4a78u, f = 0xa51a3c49u, g = 0xc4efea1bu, h = 0x30609119u;
    for (uint32_t i = 0; i < 256u; i += 8u) {
        a += st[i]; b += st[i + 1]; c += st[i + 2]; d += st[i + 3];
        e += st[i + 4]; f += st[i + 5]; g += st[i + 6]; h += st[i + 7];
        a ^= b << 11; d += a; b += c;
        b ^= c >> 2;  e += b; c += d;
        c ^= d << 8;  f += c; d += e;
        d ^= e >> 16; g += d; e += f;
        e ^= f << 10; h += e; f += g;
        f ^= g >> 4;  a += f; g += h;
        g ^= h << 8;  b += g; h += a;
        h ^= a >> 9;  c += h; a += b;
        st[i] = a; st[i + 1] = b; st[i + 2] = c; st[i + 3] = d;
        st[i + 4] = e; st[i + 5] = f; st[i + 6] = g; st[i + 7] = h;
    }
    st[RNG_A] = 0u;
    st[RNG_B] = 0u;
    st[RNG_C] = 0u;
    for (uint32_t i = 0; i < 256u; ++i) st[RNG_RESULTS + i] = 0u;
    st[RNG_INDEX] = 256u;
    st[RNG_FLAGS] = 0u; /* bank 0 is current, nothing prepared */
}

/* IsaacCore::generate from bank `src` into bank `dst` (results stored backwards: read forwards = the reference
 * implementation's order) */
__device__ void isaac_generate(const uint32_t *src, uint32_t *dst) {
    for (uint32_t i = 0; i < 256u; i += 4u) *reinterpret_cast<uint4 *>(dst + RNG_MEM + i) = *reinterpret_cast<const uint4 *>(src + RNG_MEM + i);
    const uint32_t cc = src[RNG_C] + 1u;
    dst[RNG_C] = cc;
    uint32_t a = src[RNG_A], b = src[RNG_B] + cc;
    for (uint32_t i = 0; i < 256u; ++i) {
        const uint32_t x = dst[RNG_MEM + i];
        const uint32_t sel = i & 3u;
        const uint32_t mixv = sel == 0u ? (a ^ (a << 13)) : sel == 1u ? (a ^ (a >> 6)) : sel == 2u ? (a ^ (a << 2)) : (a ^ (a >> 16));
        a = mixv + dst[RNG_MEM + ((i + 128u) & 255u)];
        const uint32_t y = a + b + dst[RNG_MEM + ((x >> 2) & 255u)];
        dst[RNG_MEM + i] = y;
        b = x + dst[RNG_MEM + ((y >> 10) & 255u)];
        dst[RNG_RESULTS + 255u - i] = b;
    }
    dst[RNG_A] = a;
    dst[RNG_B] = b;
}

/* The steps of generate on a mem[] staged in LDS, slot-interleaved (word i of slot k at i * SLOTS + k: the lanes'
 * accesses to the same i fall in different banks); results go straight to `dst`. */
#define RNG_LDS_SLOTS 8u
__device__ __forceinline__ void isaac_steps_lds(uint32_t *m, const uint32_t *src, uint32_t *dst) {
    const uint32_t cc = src[RNG_C] + 1u;
    dst[RNG_C] = cc;
    uint32_t a = src[RNG_A], b = src[RNG_B] + cc;
#define RT_ISAAC_STEP(I, MIX)                                                     \
    {                                                                             \
        const uint32_t x = m[(I) * RNG_LDS_SLOTS];                                \
        a = (a ^ (MIX)) + m[(((I) + 128u) & 255u) * RNG_LDS_SLOTS];               \
        const uint32_t y = a + b + m[((x >> 2) & 255u) * RNG_LDS_SLOTS];          \
        m[(I) * RNG_LDS_SLOTS] = y;                                               \
        b = x + m[((y >> 10) & 255u) * RNG_LDS_SLOTS];                            \
        dst[RNG_RESULTS + 255u - (I)] = b;                                        \
    }
    for (uint32_t i = 0; i < 256u; i += 4u) {
        RT_ISAAC_STEP(i, a << 13)
        RT_ISAAC_STEP(i + 1u, a >> 6)
        RT_ISAAC_STEP(i + 2u, a << 2)
        RT_ISAAC_STEP(i + 3u, a >> 16)
    }
#undef RT_ISAAC_STEP
    dst[RNG_A] = a;
    dst[RNG_B] = b;
}

/* generate for the lanes of a render wave that run dry WITHOUT a prepared bank (a pixel that used more than a whole
 * block within one visit — rare): a divergent branch, usually one lane.  mem[] goes through LDS, up to RNG_LDS_SLOTS
 * lanes at a time; more than 2 * SLOTS lanes at once take the HBM version, all in parallel. */
__device__ void isaac_generate_staged(const uint32_t *src, uint32_t *dst, uint32_t *lds) {
    const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
    if (lds == nullptr || (uint32_t)__builtin_popcountll(all) > 2u * RNG_LDS_SLOTS) {
        isaac_generate(src, dst);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__builtin_popcountll(all & ((1ull << lane) - 1ull));
    for (uint32_t first = 0u; first < 2u * RNG_LDS_SLOTS; first += RNG_LDS_SLOTS) {
        if (rank < first || rank >= first + RNG_LDS_SLOTS) continue;
        uint32_t *m = lds + (rank - first);
        for (uint32_t i = 0; i < 256u; i += 4u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + RNG_MEM + i);
            m[(i + 0u) * RNG_LDS_SLOTS] = v.x;
            m[(i + 1u) * RNG_LDS_SLOTS] = v.y;
            m[(i + 2u) * RNG_LDS_SLOTS] = v.z;
            m[(i + 3u) * RNG_LDS_SLOTS] = v.w;
        }
        isaac_steps_lds(m, src, dst);
        for (uint32_t i = 0; i < 256u; i += 4u) {
            uint4 v;
            v.x = m[(i + 0u) * RNG_LDS_SLOTS];
            v.y = m[(i + 1u) * RNG_LDS_SLOTS];
            v.z = m[(i + 2u) * RNG_LDS_SLOTS];
            v.w = m[(i + 3u) * RNG_LDS_SLOTS];
            *reinterpret_cast<uint4 *>(dst + RNG_MEM + i) = v;
        }
    }
}

/* BlockRng over the record; position and flags live in registers while a lane works on the pixel */
struct Rng {
    uint32_t *rec;  /* the pixel's record */
    uint32_t *st;   /* its current bank */
    uint32_t index;
    uint32_t flags; /* bit 0 = cur, bit 1 = prepared */
    uint32_t *lds;  /* the wave's RNG_LDS_SLOTS x 256 words of staging, or nullptr */
};
__device__ __forceinline__ void rng_open(Rng &r, uint32_t *rec) {
    r.rec = rec;
    r.index = rec[RNG_INDEX];
    r.flags = rec[RNG_FLAGS];
    r.st = rec + (r.flags & 1u) * RNG_BANK_WORDS;
}
__device__ __forceinline__ void rng_park(Rng &r) {
    r.rec[RNG_INDEX] = r.index;
    r.rec[RNG_FLAGS] = r.flags;
}
/* the current block is used up: move on to the next one (IsaacCore::generate, BlockRng::generate_and_set) */
__device__ __forceinline__ void rng_refill(Rng &r) {
    uint32_t *other = r.rec + ((r.flags & 1u) ^ 1u) * RNG_BANK_WORDS;
    if ((r.flags & 2u) == 0u) isaac_generate_staged(r.st, other, r.lds);
    r.st = other;
    r.flags = (r.flags & 1u) ^ 1u;
}
__device__ __forceinline__ uint32_t next_u32(Rng &r) {
    if (r.index >= 256u) { rng_refill(r); r.index = 0u; }
    return r.st[RNG_RESULTS + r.index++];
}
__device__ __forceinline__ unsigned long long next_u64(Rng &r) {
    if (r.index < 255u) {
        const unsigned long long x = r.st[RNG_RESULTS + r.index], y = r.st[RNG_RESULTS + r.index + 1u];
        r.index += 2u;
        return (y << 32) | x;
    } else if (r.index >= 256u) {
        rng_refill(r);
        r.index = 2u;
        return ((unsigned long long)r.st[RNG_RESULTS + 1] << 32) | r.st[RNG_RESULTS + 0];
    } else {
        const unsigned long long x = r.st[RNG_RESULTS + 255];
        rng_refill(r);
        r.index = 1u;
        return ((unsigned long long)r.st[RNG_RESULTS + 0] << 32) | x;
    }
}
/* rand 0.5 UniformFloat<f32>::sample_single: 23 random bits -> [1,2), then * scale + offset */
__device__ __forceinline__ float gen_range_f32(Rng &r, float low, float high) {
    const float scale = high - low;
    const float offset = low - scale;
    const float value1_2 = rtdm::f32_from_bits((next_u32(r) >> 9) | 0x3f800000u);
    return value1_2 * scale + offset;
}
/* the same from a word already drawn */
__device__ __forceinline__ float range_f32_of(uint32_t word, float low, float high) {
    const float scale = high - low;
    const float offset = low - scale;
    const float value1_2 = rtdm::f32_from_bits((word >> 9) | 0x3f800000u);
    return value1_2 * scale + offset;
}
/* the next three words of the stream: when they are in the current block, three loads in flight together instead of three
 * load latencies one after the other (a pixel's block is out of the caches again between two visits) */
__device__ __forceinline__ void next_u32x3(Rng &r, uint32_t *w0, uint32_t *w1, uint32_t *w2) {
    if (r.index <= 253u) {
        const uint32_t *p = r.st + RNG_RESULTS + r.index;
        *w0 = p[0];
        *w1 = p[1];
        *w2 = p[2];
        r.index += 3u;
    } else {
        *w0 = next_u32(r);
        *w1 = next_u32(r);
        *w2 = next_u32(r);
    }
}
__device__ __forceinline__ double open01_f64(Rng &r) {
    const unsigned long long fraction = next_u64(r) >> 12;
    return rtdm::f64_from_bits(fraction | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
}
__device__ __forceinline__ double standard_f64(Rng &r) { return (1.0 / 9007199254740992.0) * (double)(next_u64(r) >> 11); }

/* StandardNormal: ziggurat(symmetric), rand 0.5 distributions/mod.rs */
__device__ double standard_normal(Rng &r) {
    for (;;) {
        const unsigned long long bits = next_u64(r);
        const uint32_t i = (uint32_t)(bits & 0xffull);
        const double u = rtdm::f64_from_bits((bits >> 12) | 0x4000000000000000ull) - 3.0;
        const double x = u * ZIG_X[i];
        const double test_x = x < 0.0 ? -x : x;
        if (test_x < ZIG_X[i + 1u]) return x;
        if (i == 0u) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                const double x_ = open01_f64(r);
                const double y_ = open01_f64(r);
                xx = rtdm::log_pos(x_) / RT_ZIG_NORM_R;
                yy = rtdm::log_pos(y_);
            }
            return u < 0.0 ? xx - RT_ZIG_NORM_R : RT_ZIG_NORM_R - xx;
        }
        const double z = -x * x / 2.0;
        const double pdf = z < -700.0 ? 0.0 : rtdm::exp_mid(z);
        if (ZIG_F[i + 1u] + (ZIG_F[i] - ZIG_F[i + 1u]) * standard_f64(r) < pdf) return x;
    }
}

/* the ziggurat's immediate accept (98.8 % of the draws) on 64 bits already drawn; false: the caller takes standard_normal */
__device__ __forceinline__ bool standard_normal_fast(unsigned long long bits, double *out) {
    const uint32_t i = (uint32_t)(bits & 0xffull);
    const double u = rtdm::f64_from_bits((bits >> 12) | 0x4000000000000000ull) - 3.0;
    const double x = u * ZIG_X[i];
    const double test_x = x < 0.0 ? -x : x;
    *out = x;
    return test_x < ZIG_X[i + 1u];
}

/* two StandardNormal draws in stream order.  Usually both accept at once and their four words are in the current block:
 * then the four loads travel together; otherwise redo both the ordinary way from the same position (same draws). */
__device__ __forceinline__ void standard_normal_x2(Rng &r, double *n0, double *n1) {
    if (r.index <= 252u) {
        const uint32_t *p = r.st + RNG_RESULTS + r.index;
        const unsigned long long a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
        double x0, x1;
        const bool f0 = standard_normal_fast((a1 << 32) | a0, &x0), f1 = standard_normal_fast((a3 << 32) | a2, &x1);
        if (f0 && f1) {
            *n0 = x0;
            *n1 = x1;
            r.index += 4u;
            return;
        }
    }
    *n0 = standard_normal(r);
    *n1 = standard_normal(r);
}

__global__ __launch_bounds__(256) void rng_seed_kernel(uint32_t *states, uint32_t cols, uint32_t rows, uint32_t x0, uint32_t y0, uint32_t y_step) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cols * rows) return;
    const uint32_t row = p / cols, col = p - row * cols;
    const unsigned long long y = y0 + (unsigned long long)row * y_step, x = x0 + col;
    isaac_seed(states + (size_t)p * RNG_WORDS, y * (2ull << 32) + x); /* main.rs:1119 */
}

hipError_t launch_rng_seed(uint32_t *states, const KernelFrame &fr, hipStream_t stream) {
    const uint32_t n = fr.cols * fr.rows;
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(rng_seed_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, states, fr.cols, fr.rows, fr.x0, fr.y0, fr.y_step);
    return hipGetLastError();
}

/* ---- the look-ahead pass: every pixel gets its next block before a render kernel may need it ---- */

/* list[1 + k] = the pixels without a prepared bank; list[0] = how many (zeroed by the launcher) */
__global__ __launch_bounds__(256) void rng_scan_kernel(const uint32_t *states, uint32_t n_pixels, uint32_t *list) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool want = p < n_pixels && (states[(size_t)p * RNG_WORDS + RNG_FLAGS] & 2u) == 0u;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0u;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(list, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
    if (want) list[1u + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = p;
}

/* Every lane generates the next block of ONE record, its mem[] in a column of LDS of its own: word i of the lane's record
 * at stage[i * SLOTS + lane], SLOTS a multiple of 32, so a lane stays in its bank whatever i it asks for — the two
 * data-dependent reads of a step never conflict.  Nothing is shared between lanes: no barriers.  mem[i] and mem[i + 128] of
 * the next four steps are read ahead of the chain (step j only ever writes mem[j], so they cannot be stale), which leaves ONE
 * LDS latency per step on it (b needs mem[(y >> 10) & 255], the next y needs b), and the four results of a group leave as
 * one 16-byte store.  Two one-wave workgroups of 64 records (64 KB) per CU.  Measured alone over 2^20 records
 * (tools/diag_prepare.py, profiles/r03_lookahead_kernel.txt): 1.37 ms against 2.04 ms for round 2's form (8 records per wave
 * staged by all its lanes, the steps on 8 of the 64, twenty waves per CU: the LDS pipe served 64-lane instructions for 8
 * lanes' worth of work) — 0.78 ms of steps (~230 clocks each) and 0.92 ms of copies (2.2 TB/s in 16-byte pieces), partly
 * overlapped.  What the pass gains is more than that: the look-ahead runs beside the shade kernel, and two waves per CU
 * take far less from it than twenty did (shade kernel 5.97 -> 4.50 ms per batch, the pass 1 073 -> 1 155 Msamples/s).  One
 * workgroup of 160 lanes with all 160 KB of a CU is as fast alone (1.39 ms) but cannot share a CU with the shade kernel's
 * workgroups: 1 118 Msamples/s. */
struct __attribute__((packed, aligned(4))) RngU4 { uint32_t x, y, z, w; }; /* 16 bytes at a word boundary: results[] starts at word 259 */

template <uint32_t SLOTS>
__global__ __launch_bounds__(64) void rng_prepare_kernel(uint32_t *states, const uint32_t *list) {
    extern __shared__ uint32_t rng_prep_stage[]; /* SLOTS x 256 words */
    const uint32_t lane = threadIdx.x;
    const uint32_t count = list[0];
    uint32_t *m = rng_prep_stage + lane;
    for (uint32_t first = blockIdx.x * SLOTS; first < count; first += gridDim.x * SLOTS) {
        if (lane >= SLOTS || first + lane >= count) continue;
        uint32_t *rec = states + (size_t)list[1u + first + lane] * RNG_WORDS;
        const uint32_t cur = rec[RNG_FLAGS] & 1u;
        const uint32_t *src = rec + cur * RNG_BANK_WORDS;
        uint32_t *dst = rec + (cur ^ 1u) * RNG_BANK_WORDS;
        for (uint32_t i = 0; i < 256u; i += 32u) { /* eight loads in flight */
            uint4 v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) v[j] = *reinterpret_cast<const uint4 *>(src + RNG_MEM + i + 4u * j);
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                m[(i + 4u * j + 0u) * SLOTS] = v[j].x;
                m[(i + 4u * j + 1u) * SLOTS] = v[j].y;
                m[(i + 4u * j + 2u) * SLOTS] = v[j].z;
                m[(i + 4u * j + 3u) * SLOTS] = v[j].w;
            }
        }
        const uint32_t cc = src[RNG_C] + 1u;
        uint32_t a = src[RNG_A], b = src[RNG_B] + cc;
        for (uint32_t i = 0; i < 256u; i += 4u) {
            const uint32_t x0 = m[(i + 0u) * SLOTS], x1 = m[(i + 1u) * SLOTS], x2 = m[(i + 2u) * SLOTS], x3 = m[(i + 3u) * SLOTS];
            const uint32_t h0 = m[((i + 128u) & 255u) * SLOTS], h1 = m[((i + 129u) & 255u) * SLOTS], h2 = m[((i + 130u) & 255u) * SLOTS],
                           h3 = m[((i + 131u) & 255u) * SLOTS];
            RngU4 r;
#define RT_ISAAC_WIDE_STEP(K, X, H, MIX, OUT)                         \
            {                                                         \
                a = (a ^ (MIX)) + (H);                                \
                const uint32_t y = a + b + m[(((X) >> 2) & 255u) * SLOTS]; \
                m[(i + (K)) * SLOTS] = y;                             \
                b = (X) + m[((y >> 10) & 255u) * SLOTS];              \
                OUT = b;                                              \
            }
            RT_ISAAC_WIDE_STEP(0u, x0, h0, a << 13, r.w)
            RT_ISAAC_WIDE_STEP(1u, x1, h1, a >> 6, r.z)
            RT_ISAAC_WIDE_STEP(2u, x2, h2, a << 2, r.y)
            RT_ISAAC_WIDE_STEP(3u, x3, h3, a >> 16, r.x)
#undef RT_ISAAC_WIDE_STEP
            *reinterpret_cast<RngU4 *>(dst + RNG_RESULTS + 252u - i) = r; /* results[255 - i] = step i's word: read forwards */
        }
        for (uint32_t i = 0; i < 256u; i += 32u) {
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                uint4 v;
                v.x = m[(i + 4u * j + 0u) * SLOTS];
                v.y = m[(i + 4u * j + 1u) * SLOTS];
                v.z = m[(i + 4u * j + 2u) * SLOTS];
                v.w = m[(i + 4u * j + 3u) * SLOTS];
                *reinterpret_cast<uint4 *>(dst + RNG_MEM + i + 4u * j) = v;
            }
        }
        dst[RNG_A] = a;
        dst[RNG_B] = b;
        dst[RNG_C] = cc;
        rec[RNG_FLAGS] = cur | 2u;
    }
}

#ifndef RNG_PREP_SLOTS
#define RNG_PREP_SLOTS 64u /* records (and KB of LDS) of a one-wave workgroup; a multiple of 32, at most 64 */
#endif

hipError_t launch_rng_prepare(uint32_t *states, uint32_t n_pixels, uint32_t *list, uint32_t compute_units, hipStream_t stream) {
    if (n_pixels == 0u) return hipSuccess;
    hipError_t e = hipMemsetAsync(list, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rng_scan_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, states, n_pixels, list);
    const uint32_t groups = std::min((n_pixels + RNG_PREP_SLOTS - 1u) / RNG_PREP_SLOTS, compute_units * (160u / RNG_PREP_SLOTS));
    hipLaunchKernelGGL((rng_prepare_kernel<RNG_PREP_SLOTS>), dim3(groups), dim3(64), RNG_PREP_SLOTS * 1024u, stream, states, list);
    return hipGetLastError();
}

/* the oracle's record of every pixel: bank `cur` + the position (rt_rng_download) */
__global__ __launch_bounds__(256) void rng_export_kernel(const uint32_t *states, uint32_t n_pixels, uint32_t *out) {
    const size_t n_words = (size_t)n_pixels * RNG_BANK_WORDS;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (size_t)gridDim.x * blockDim.x) {
        const size_t p = w / RNG_BANK_WORDS;
        const uint32_t i = (uint32_t)(w - p * RNG_BANK_WORDS);
        const uint32_t *rec = states + p * RNG_WORDS;
        out[w] = i == RNG_SPARE ? rec[RNG_INDEX] : rec[(rec[RNG_FLAGS] & 1u) * RNG_BANK_WORDS + i];
    }
}

hipError_t launch_rng_export(const uint32_t *states, uint32_t n_pixels, uint32_t *out, hipStream_t stream) {
    if (n_pixels == 0u) return hipSuccess;
    hipLaunchKernelGGL(rng_export_kernel, dim3(4096), dim3(256), 0, stream, states, n_pixels, out);
    return hipGetLastError();
}

/* ---- the kernel ------------------------------------------------------------------------------- */

enum : uint32_t {
    DP_DONE = 0u,
    DP_PRIMARY = 1u,     /* cast of the shoot_focus ray                        (main.rs:1150) */
    DP_NEXT = 2u,        /* cast of the reflected / escape ray of a level      (main.rs:564, 583, 603) */
    DP_SHADOW = 3u,      /* a shadow ray of get_shade                          (main.rs:435) */
    DP_REFR_INSIDE = 4u, /* get_refract's first inside cast                    (main.rs:371) */
    DP_REFR_BOUNCE = 5u, /* a total-internal-reflection bounce                 (main.rs:381) */
    DP_START = 6u        /* chain kernel: the next epoch's shoot_focus is due */
};

struct DFrame {
    V3 shade;        /* get_shade(&next_hit) */
    V3 factor;       /* brdf (kinds 0, 1) or (decay, -, -) (kind 2) */
    uint32_t kind;   /* 0 Diffuse, 1 Reflection, 2 Refraction */
};

#ifndef RT_DIST_MIN_WAVES
#define RT_DIST_MIN_WAVES 3 /* waves per SIMD the register allocation aims for (profiles/README.md) */
#endif
/* BFS: the casts as a breadth-first walk of the node tree (rt_cast_bfs.h cast_bfs), for scenes beyond the caches (KernelScene::bfs_walk):
 * 256 VGPRs, two waves per SIMD, 5 KB of LDS and a set of record lists per wave */
#define RT_DIST_BFS_WAVES 2
template <int MAXD, bool BFS = false>
__global__ __launch_bounds__(64, BFS ? RT_DIST_BFS_WAVES : RT_DIST_MIN_WAVES) void distributed_kernel(const KernelScene sc, const KernelFrame fr, const DistParams dp) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    BfsLds *bfs_lds = nullptr;
    BfsScratch bfs_ws = {nullptr, nullptr, nullptr, 0u, 0u};
    if constexpr (BFS) {
        __shared__ BfsLds bfs_lds_one;
        bfs_lds = &bfs_lds_one;
        uint2 *const mine = reinterpret_cast<uint2 *>(dp.bfs_scratch) + (size_t)wave * (2u * (size_t)dp.bfs_items_cap + dp.bfs_jobs_cap);
        bfs_ws.items_a = mine;
        bfs_ws.items_b = mine + dp.bfs_items_cap;
        bfs_ws.jobs = mine + 2u * (size_t)dp.bfs_items_cap;
        bfs_ws.items_cap = dp.bfs_items_cap;
        bfs_ws.jobs_cap = dp.bfs_jobs_cap;
    }
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t band_slots = fr.cols << 3;
    /* Persistent lanes: a lane takes a pixel, runs ALL of this call's epochs for it (the pixel's random stream
     * is sequential), then takes the next pixel.  The scattered rays are incoherent whatever the assignment, so
     * nothing is lost by mixing pixels in a wave, and every lane stays busy until the tile runs dry.  Pixels are
     * handed out as in the Whitted kernel: 64-slot chunks from a global counter (one atomic per chunk per wave),
     * slots inside a chunk by ballot + prefix count.  dp.work_queue == nullptr selects the static assignment
     * (wave w = chunk w), kept for A/B. */
    const bool persistent = dp.work_queue != nullptr;
    uint32_t q_next = 0u, q_end = 0u;
    bool exhausted = false;
    if (!persistent) {
        q_next = wave * 64u < total_slots ? wave * 64u : total_slots;
        q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
        exhausted = true;
    }
    uint32_t out_index = 0u;
    float clip_x = 0.0f, clip_y = 0.0f;
    const V3 cam_x = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
    const V3 cam_y = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
    const V3 cam_t = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
    const V3 cam_o = v3(fr.cam_origin_focus[0], fr.cam_origin_focus[1], fr.cam_origin_focus[2]);
    const size_t n_pixels = total_slots;

    Rng rng;
    rng.rec = rng.st = dp.rng_states;
    rng.index = 256u;
    rng.flags = 0u;
    rng.lds = nullptr;

    uint32_t phase = DP_DONE;
    uint32_t epoch = 0u;
    Ray req;
    req.o = v3(0.0f, 0.0f, 0.0f);
    req.d = v3(0.0f, 0.0f, 1.0f);
    req.mode = FACE_FRONT;
    req.excl = 0u;
    uint32_t casts = 0u;

    HitGeom h; /* the hit of the current level */
    h.pos = h.normal = v3(0.0f, 0.0f, 0.0f);
    h.u = h.v = 0.0f;
    h.prim = h.bf = h.obj = 0u;
    V3 h_in_dir = v3(0.0f, 0.0f, 0.0f); /* hit.ray.direction */
    uint32_t h_in_mode = FACE_FRONT;     /* hit.ray.face_direction */
    V3 sdir = v3(0.0f, 0.0f, 0.0f);     /* scattered_hit.ray.direction */
    uint32_t kind = 0u;                  /* RayType selected at this level */
    V3 view_dir_in = v3(0.0f, 0.0f, 0.0f); /* ray direction used as `hit.ray` by the get_shade in progress */
    uint32_t shade_then = 0u;            /* 0: result is the level's value (return), 1: result is get_shade(next_hit) (descend) */
    int32_t sp = 0;
    V3 sum = v3(0.0f, 0.0f, 0.0f), adj_n = v3(0.0f, 0.0f, 0.0f), l_color = v3(0.0f, 0.0f, 0.0f);
    uint32_t light_i = 0u;
    float travel = 0.0f;
    int32_t retry = 0;
    V3 accum = v3(0.0f, 0.0f, 0.0f);
    DFrame stack[MAXD];

    /* Camera::shoot_focus for the next epoch of this pixel (main.rs:101-127) */
    auto start_epoch = [&]() {
        const V3 direction = normalize(clip_x * cam_x + clip_y * cam_y + cam_t);
        const float xoffset = (float)(0.0 + (double)dp.blur * standard_normal(rng)); /* Normal::new(0.0, blur as f64) */
        const float yoffset = (float)(0.0 + (double)dp.blur * standard_normal(rng));
        req.d = normalize(direction * dp.focus + cam_x * xoffset + cam_y * yoffset);
        /* center + toward.normalize() * near - (x*xo + y*yo); the first two terms are per-frame (rt_api.hip) */
        req.o = cam_o - (cam_x * xoffset + cam_y * yoffset);
        req.mode = FACE_FRONT;
        req.excl = 0u;
        sp = 0;
        phase = DP_PRIMARY;
    };

    for (;;) {
        /* ---- idle lanes take the next pixels ---- */
        unsigned long long need = __builtin_amdgcn_ballot_w64(phase == DP_DONE);
        if (dp.n_epochs == 0u) need = 0ull;
        while (need != 0ull) {
            if (q_next == q_end) { /* wave-uniform */
                if (exhausted) break;
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(dp.work_queue, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c * 64u >= total_slots) { exhausted = true; break; }
                q_next = c * 64u;
                q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t avail = q_end - q_next;
            if (phase == DP_DONE && rank < avail) {
                /* same slot -> pixel mapping as the Whitted kernel: 8-row bands, column-major inside a band */
                const uint32_t slot = q_next + rank;
                const uint32_t band = slot / band_slots;
                const uint32_t r = slot - band * band_slots;
                const uint32_t rows_left = fr.rows - (band << 3);
                const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
                const uint32_t col = r / band_rows;
                const uint32_t row = (band << 3) + (r - col * band_rows);
                out_index = row * fr.cols + col;
                const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                clip_y = (fr.half_height - (float)y) / fr.height_f; /* main.rs:1134-1135 */
                clip_x = ((float)x - fr.half_width) / fr.height_f;
                rng_open(rng, dp.rng_states + (size_t)out_index * RNG_WORDS);
                accum = v3(0.0f, 0.0f, 0.0f);
                if (dp.accum != nullptr) accum = v3(dp.accum[(size_t)out_index * 3u], dp.accum[(size_t)out_index * 3u + 1u], dp.accum[(size_t)out_index * 3u + 2u]);
                epoch = 0u;
                start_epoch();
            }
            const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
            q_next += n_need < avail ? n_need : avail;
            need = __builtin_amdgcn_ballot_w64(phase == DP_DONE);
        }
        if (__builtin_amdgcn_ballot_w64(phase != DP_DONE) == 0ull) break;

        CastResult cr;
        cr.prim = -1;
        cr.t = 0.0f;
        cr.bf = 0u;
        cr.a0 = cr.a1 = cr.a2 = 0.0f;
        if constexpr (BFS) {
            cr = cast_bfs(sc, req, phase != DP_DONE, bfs_lds, bfs_ws); /* all lanes: those without a ray help */
            if (phase != DP_DONE) casts += 1u;
        } else if (phase != DP_DONE) {
            cr = cast_asm(sc, req);
            casts += 1u;
        }
        if (phase == DP_DONE) continue;

        enum { GO_LEVEL, GO_START_SHADE, GO_NEXT_LIGHT, GO_SHADE_DONE, GO_TRY_EXIT, GO_RETURN } go = GO_RETURN;
        V3 value = v3(0.0f, 0.0f, 0.0f);
        HitGeom ih = h; /* inside hit of get_refract, live within this step only */
        V3 i_in_dir = req.d;
        uint32_t i_in_mode = req.mode;

        if (phase == DP_PRIMARY) {
            if (cr.prim < 0) {
                value = v3(0.0f, 0.0f, 0.0f); /* main.rs:1154 */
                go = GO_RETURN;
            } else {
                h = finish_hit(sc, req, cr, false);
                h_in_dir = req.d;
                h_in_mode = req.mode;
                go = GO_LEVEL;
            }
        } else if (phase == DP_NEXT) {
            if (cr.prim < 0) {
                if (kind == 2u) { /* main.rs:606-608 */
                    value = v3(0.0f, 0.0f, 0.0f);
                    go = GO_RETURN;
                } else { /* get_shade(&scattered_hit): the same hit, seen along the scattered direction (main.rs:573, 592) */
                    view_dir_in = sdir;
                    shade_then = 0u;
                    go = GO_START_SHADE;
                }
            } else {
                /* brdf of the CURRENT level, before `h` moves on to the next hit (main.rs:566-570, 585-589) */
                DFrame f;
                f.kind = kind;
                f.shade = v3(0.0f, 0.0f, 0.0f);
                if (kind == 2u) {
                    f.factor = v3(rtdm::powf(sc.materials[h.obj].opaque_decay, travel), 0.0f, 0.0f); /* main.rs:605 */
                } else {
                    const Mat m = material_approx(sc.materials[h.obj], h.u, h.v);
                    const V3 view = -h_in_dir;
                    f.factor = kind == 0u ? get_diffuse(m, h.normal, req.d) : get_specular(m, h.normal, view, req.d);
                }
                stack[sp] = f;
                h = finish_hit(sc, req, cr, false);
                h_in_dir = req.d;
                h_in_mode = req.mode;
                view_dir_in = req.d;
                shade_then = 1u;
                go = GO_START_SHADE;
            }
        } else if (phase == DP_SHADOW) {
            const rt_light &L = sc.lights[light_i];
            bool lit = true;
            if (cr.prim >= 0) {
                const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                if (has_origin) {
                    const V3 occ = req.o + req.d * cr.t;
                    if (distance(h.pos, occ) < distance(h.pos, v3(L.origin[0], L.origin[1], L.origin[2]))) lit = false;
                } else {
                    lit = false;
                }
            }
            if (lit) {
                const Mat m = material_approx(sc.materials[h.obj], h.u, h.v);
                const V3 light_direction = req.d;
                const V3 diffuse = get_diffuse(m, adj_n, light_direction) * l_color;
                const V3 specular = get_specular(m, adj_n, -view_dir_in, light_direction) * l_color;
                sum = sum + diffuse * (1.0f - m.shiness) + specular * m.shiness;
            }
            light_i += 1u;
            go = GO_NEXT_LIGHT;
        } else { /* DP_REFR_INSIDE / DP_REFR_BOUNCE */
            if (cr.prim < 0) {
                value = v3(0.0f, 0.0f, 0.0f); /* Refraction::Infinite -> black (main.rs:610) */
                go = GO_RETURN;
            } else {
                ih = finish_hit(sc, req, cr, false);
                i_in_dir = req.d;
                i_in_mode = req.mode;
                if (phase == DP_REFR_INSIDE) {
                    travel = distance(ih.pos, h.pos);
                    retry = 0;
                } else {
                    travel += distance(req.o, ih.pos);
                    retry += 1;
                }
                go = GO_TRY_EXIT;
            }
        }

        for (;;) {
            if (go == GO_LEVEL) {
                /* distributed_ray_trace(state, &h), depth = max_depth - sp */
                const int32_t depth = fr.max_depth - sp;
                if (depth <= 0) { /* main.rs:524-527 */
                    view_dir_in = h_in_dir;
                    shade_then = 0u;
                    go = GO_START_SHADE;
                    continue;
                }
                const rt_material &rm = sc.materials[h.obj];
                /* weighted_select (main.rs:652-666) */
                const float w0 = (1.0f - rm.shiness) * (1.0f - rm.transparency);
                const float w1 = rm.shiness * (1.0f - rm.transparency);
                const float w2 = rm.transparency;
                float wsum = 0.0f;
                wsum = wsum + w0;
                wsum = wsum + w1;
                wsum = wsum + w2;
                const float rsel = gen_range_f32(rng, 0.0f, wsum);
                float acc = 0.0f;
                acc += w0;
                kind = 2u;
                if (rsel < acc) kind = 0u;
                else {
                    acc += w1;
                    if (rsel < acc) kind = 1u;
                }
                /* scatter_hit (main.rs:539-554) */
                const float exponent = kind == 0u ? 1.0f : rm.smoothness;
                const V3 lobe = kind == 0u ? -h.normal : h_in_dir;
                const float phi = rtdm::acosf(rtdm::powf(1.0f - gen_range_f32(rng, 0.0f, 1.0f), exponent));
                const float theta = gen_range_f32(rng, -RT_F_PI, RT_F_PI);
                float sphi, cphi, stheta, ctheta;
                rtdm::sincosf(phi, &sphi, &cphi);
                rtdm::sincosf(theta, &stheta, &ctheta);
                sdir = adjust_normal(v3(sphi * ctheta, sphi * stheta, cphi), normalize(lobe));
                const float cosine = -dot(h.normal, sdir);
                if (cosine <= 0.0f) { /* main.rs:560, 579, 598 */
                    value = v3(0.0f, 0.0f, 0.0f);
                    go = GO_RETURN;
                    continue;
                }
                if (kind != 2u) {
                    /* get_reflect(&scattered_hit) (main.rs:328-341) */
                    req.o = h.pos;
                    req.d = reflect_dir(h.normal, sdir);
                    req.mode = h_in_mode;
                    req.excl = pack_excl(h.prim, h.bf ? FACE_FRONT : FACE_BACK);
                    phase = DP_NEXT;
                    break;
                }
                /* get_refract(&scattered_hit, 100.0) (main.rs:343-405) */
                V3 refract_in;
                if (refract_dir(h.normal, sdir, rm.refraction_index, &refract_in)) {
                    req.o = h.pos;
                    req.d = normalize(refract_in);
                    req.mode = FACE_BACK;
                    req.excl = pack_excl(h.prim, FACE_FRONT);
                    phase = DP_REFR_INSIDE;
                    break;
                }
                value = v3(0.0f, 0.0f, 0.0f); /* Trapped */
                go = GO_RETURN;
            } else if (go == GO_TRY_EXIT) {
                const rt_material &rm = sc.materials[h.obj];
                V3 out_dir;
                const bool have_out = refract_dir(ih.normal, i_in_dir, 1.0f / rm.refraction_index, &out_dir);
                if (!have_out && travel <= 100.0f && retry < 10) {
                    req.o = ih.pos;
                    req.d = reflect_dir(ih.normal, i_in_dir);
                    req.mode = i_in_mode;
                    req.excl = pack_excl(ih.prim, ih.bf ? FACE_FRONT : FACE_BACK);
                    phase = DP_REFR_BOUNCE;
                    break;
                }
                if (!have_out) { /* Trapped */
                    value = v3(0.0f, 0.0f, 0.0f);
                    go = GO_RETURN;
                    continue;
                }
                req.o = ih.pos; /* escape ray, main.rs:393-401 */
                req.d = normalize(out_dir);
                req.mode = FACE_FRONT;
                req.excl = pack_excl(ih.prim, FACE_BACK);
                phase = DP_NEXT;
                break;
            } else if (go == GO_START_SHADE) {
                /* get_shade(&h') where h' = h with ray.direction = view_dir_in (main.rs:407-412) */
                const Mat m = material_approx(sc.materials[h.obj], h.u, h.v);
                adj_n = adjust_normal(m.normal, h.normal);
                sum = v3(0.0f, 0.0f, 0.0f);
                light_i = 0u;
                go = GO_NEXT_LIGHT;
            } else if (go == GO_NEXT_LIGHT) {
                bool issued = false;
                while (light_i < sc.n_lights) {
                    DirLight dl;
                    if (approximate_into_directional(sc.lights[light_i], h.pos, &dl)) {
                        const float cosine = -dot(dl.direction, adj_n);
                        if (!(cosine <= 0.0f)) {
                            req.o = h.pos;
                            req.d = -dl.direction;
                            req.mode = FACE_BACK;
                            req.excl = pack_excl(h.prim, FACE_BACK);
                            l_color = dl.color;
                            phase = DP_SHADOW;
                            issued = true;
                            break;
                        }
                    }
                    light_i += 1u;
                }
                if (issued) break;
                go = GO_SHADE_DONE;
            } else if (go == GO_SHADE_DONE) {
                if (shade_then == 1u) { /* that was get_shade(&next_hit): keep it and descend (main.rs:565, 584, 604) */
                    stack[sp].shade = sum;
                    sp += 1;
                    go = GO_LEVEL;
                } else {
                    value = sum;
                    go = GO_RETURN;
                }
            } else { /* GO_RETURN */
                if (sp > 0) {
                    const DFrame f = stack[sp - 1];
                    if (f.kind == 2u) {
                        value = (value + f.shade) * f.factor.x; /* main.rs:605 */
                    } else {
                        const V3 s = value * f.factor;           /* main.rs:566, 585 */
                        value = f.shade + (s - f.shade) * 0.5f;  /* palette Mix::mix(&s, 0.5), main.rs:571, 590 */
                    }
                    sp -= 1;
                    continue;
                }
                /* the sample of this (pixel, epoch): filter (main.rs:1157-1160) and accumulate (main.rs:1165) */
                const bool ok = rtdm::is_normal(value.x) && rtdm::is_normal(value.y) && rtdm::is_normal(value.z);
                if (dp.samples != nullptr) {
                    float *o = dp.samples + ((size_t)epoch * n_pixels + out_index) * 3u;
                    o[0] = value.x; o[1] = value.y; o[2] = value.z;
                }
                if (dp.valid != nullptr) dp.valid[(size_t)epoch * n_pixels + out_index] = ok ? 1 : 0;
                if (ok) accum = accum + value;
                epoch += 1u;
                if (epoch < dp.n_epochs) {
                    start_epoch();
                } else { /* this pixel is finished for this call: park its stream position and its sum */
                    rng_park(rng);
                    if (dp.accum != nullptr) {
                        float *o = dp.accum + (size_t)out_index * 3u;
                        o[0] = accum.x; o[1] = accum.y; o[2] = accum.z;
                    }
                    phase = DP_DONE;
                }
                break;
            }
        }
    }

    if (dp.ray_count != nullptr) {
        uint32_t c = casts;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0u && c != 0u) atomicAdd(dp.ray_count, (unsigned long long)c);
    }
}

hipError_t launch_distributed(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, uint32_t resident_waves, hipStream_t stream) {
    const uint32_t total = fr.cols * fr.rows;
    uint32_t waves = (total + 63u) / 64u;
    if (waves == 0u) return hipSuccess;
    if (dp.work_queue != nullptr && waves > resident_waves) waves = resident_waves; /* persistent lanes: fill the chip once */
    if (sc.bfs_walk != 0u && dp.bfs_scratch != nullptr && dp.work_queue != nullptr) { /* (the caller sized resident_waves by dist_bfs_waves) */
        if (fr.max_depth <= 8) hipLaunchKernelGGL((distributed_kernel<9, true>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
        else hipLaunchKernelGGL((distributed_kernel<RT_MAX_DEPTH + 1, true>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
        return hipGetLastError();
    }
    if (fr.max_depth <= 8) {
        hipLaunchKernelGGL((distributed_kernel<9>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
    } else {
        hipLaunchKernelGGL((distributed_kernel<RT_MAX_DEPTH + 1>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
    }
    return hipGetLastError();
}
uint32_t dist_bfs_waves(uint32_t compute_units) { return compute_units * 4u * (uint32_t)RT_DIST_BFS_WAVES; }


/* ---- the split pass ----------------------------------------------------------------------------------------------
 * distributed_kernel above keeps a lane on one pixel through everything a sample needs: the scatter chain, every
 * get_shade along it (three shadow casts each) and the unwind — 168 VGPRs with 76 of them spilled, three waves per
 * SIMD, lanes in six different phases.  But get_shade draws no random numbers and the chain does not wait for it:
 * get_shade(&next_hit) is only needed when the level returns (main.rs:565-571, 584-590, 604-605).  So the pass is cut
 * in three:
 *
 *   dist_chain_kernel   the chain alone — shoot_focus, the casts of the scattered/refracted rays, weighted_select,
 *                       scatter_hit, the per-level factor (brdf or decay) — all the random draws, in the reference's
 *                       order; per sample it records the get_shade REQUESTS (hit + view direction) and the frames;
 *   dist_shade_kernel   every request of the batch: get_shade (main.rs:407-464) with a wave-uniform light index — the
 *                       organisation of the Whitted path's SHADE items;
 *   dist_unwind_kernel  per pixel, the batch's epochs in order: the unwind of main.rs:571/590/605 over the recorded
 *                       frames, the sample filter (main.rs:1157-1160) and `img += photon` (main.rs:1165).
 *
 * Same operations on the same values as the fused kernel, so samples, flags, RNG states and cast counts are
 * bit-identical to it and to the oracle (tests/test_gpu_distributed_parity.py runs both). */

/* The shade kernel keeps the wave-uniform cast: pair-wise (RT_DIST_SHADE_PAIRS) it is slower, 5.97 against 5.15 ms per 8-epoch batch —
 * its shadow rays need the clusters less sparsely (half of the leaf visits have 8 lanes or more, a quarter 34 or more, against 7 and
 * 25 in the chain kernel) and the 19 KB of PairLds per workgroup cost it two of its six waves per SIMD (profiles/r03p7_*). */
#if !defined(RT_DIST_SHADE_PAIRS) && !defined(RT_DIST_SHADE_NO_PAIRS)
#define RT_DIST_SHADE_NO_PAIRS
#endif
#ifndef RT_DIST_CHAIN_MIN_WAVES
#define RT_DIST_CHAIN_MIN_WAVES 5 /* 96 VGPRs, no more scratch than at 4 (112); 3 / 4 / 5: 709 / 853 / 868 Msamples/s */
#endif

__device__ __forceinline__ uint32_t dfu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float duf(uint32_t x) { return __uint_as_float(x); }

template <int DUMMY>
__global__ __launch_bounds__(64, RT_DIST_CHAIN_MIN_WAVES) void dist_chain_kernel(const KernelScene sc, const KernelFrame fr, const DistParams dp) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_slots = fr.cols * fr.rows;
    const uint32_t band_slots = fr.cols << 3;
    const size_t n_pixels = total_slots;
    const size_t n_samples = n_pixels * dp.n_epochs; /* of this batch */
    /* dp.own_first_chunk: a wave's first chunk is its own — chunk w for wave w: the grid never has more waves than there are chunks */
    uint32_t q_next = dp.own_first_chunk != 0u ? blockIdx.x * 64u : 0u;
    uint32_t q_end = dp.own_first_chunk != 0u ? (q_next + 64u < total_slots ? q_next + 64u : total_slots) : 0u;
    bool exhausted = false;
    uint32_t out_index = 0u;
    float clip_x = 0.0f, clip_y = 0.0f;
    const V3 cam_x = v3(fr.cam_x[0], fr.cam_x[1], fr.cam_x[2]);
    const V3 cam_y = v3(fr.cam_y[0], fr.cam_y[1], fr.cam_y[2]);
    const V3 cam_t = v3(fr.cam_toward[0], fr.cam_toward[1], fr.cam_toward[2]);
    const V3 cam_o = v3(fr.cam_origin_focus[0], fr.cam_origin_focus[1], fr.cam_origin_focus[2]);

    /* one wave per workgroup.  The pair-wise cast's scratch SHARES the refill's staging area — one union, so that the compiler sees the
     * two views alias — because a generator is refilled between casts, never during one: every refill (rng_open /
     * refill in start_epoch and rt_dist_advance.inc) completes, LDS traffic included, before the step's cast_pairs call, and the
     * cast has returned before the next one.  With 8 KB per wave the kernel sits exactly at five waves per SIMD; anything on top
     * would cost the fifth. */
    union ChainLds {
        uint32_t stage[RNG_LDS_SLOTS * 256u];
#ifndef RT_DIST_NO_PAIRS
        PairLds pairs;
#endif
    };
    __shared__ __attribute__((aligned(16))) ChainLds chain_lds;
    uint32_t *const rng_stage = chain_lds.stage;
#ifndef RT_DIST_NO_PAIRS
    static_assert(sizeof(PairLds) <= sizeof(chain_lds.stage), "PairLds must fit the staging area");
    PairLds *const pair_lds = &chain_lds.pairs;
#endif
    Rng rng;
    rng.rec = rng.st = dp.rng_states;
    rng.index = 256u;
    rng.flags = 0u;
#ifdef RT_DIST_NO_STAGE /* A/B */
    rng.lds = nullptr;
#else
    rng.lds = rng_stage;
#endif
    uint32_t phase = DP_DONE;
    uint32_t epoch = 0u;
    Ray req;
    req.o = v3(0.0f, 0.0f, 0.0f);
    req.d = v3(0.0f, 0.0f, 1.0f);
    req.mode = FACE_FRONT;
    req.excl = 0u;
    uint32_t casts = 0u;
    HitGeom h;
    h.pos = h.normal = v3(0.0f, 0.0f, 0.0f);
    h.u = h.v = 0.0f;
    h.prim = h.bf = h.obj = 0u;
    V3 h_in_dir = v3(0.0f, 0.0f, 0.0f);
    uint32_t h_in_mode = FACE_FRONT;
    V3 sdir = v3(0.0f, 0.0f, 0.0f);
    uint32_t kind = 0u;
    int32_t sp = 0; /* frames recorded so far = the level */
    float travel = 0.0f;
    int32_t retry = 0;

    auto start_epoch = [&]() { /* Camera::shoot_focus (main.rs:101-127) */
        const V3 direction = normalize(clip_x * cam_x + clip_y * cam_y + cam_t);
        double nx, ny;
        standard_normal_x2(rng, &nx, &ny);
        const float xoffset = (float)(0.0 + (double)dp.blur * nx);
        const float yoffset = (float)(0.0 + (double)dp.blur * ny);
        req.d = normalize(direction * dp.focus + cam_x * xoffset + cam_y * yoffset);
        req.o = cam_o - (cam_x * xoffset + cam_y * yoffset);
        req.mode = FACE_FRONT;
        req.excl = 0u;
        sp = 0;
        phase = DP_PRIMARY;
    };
    /* get_shade(&h') with h' = h seen along `view` is due: leave it to dist_shade_kernel, in request slot sp */
    auto emit_request = [&](V3 view) {
        const size_t s = (size_t)epoch * n_pixels + out_index;
        uint4 *r = dp.sp_req + ((size_t)sp * n_samples + s) * 4u;
        r[0] = make_uint4(dfu(h.pos.x), dfu(h.pos.y), dfu(h.pos.z), dfu(h.u));
        r[1] = make_uint4(dfu(h.normal.x), dfu(h.normal.y), dfu(h.normal.z), dfu(h.v));
        r[2] = make_uint4(dfu(view.x), dfu(view.y), dfu(view.z), h.obj);
        r[3] = make_uint4(h.prim, 0u, 0u, 0u);
    };

#ifdef RT_DIAG_PAIR_TIME /* per wave: [0] steps, [1..7] rt_cast.h, [8] the kernel, [9] fetching work + shoot_focus, [10] after the cast: the hit and the
                          * level's factor, [11] the level's draws and scatter_hit, [12] get_refract's exit, [13] the rest of the step */
    unsigned long long diag_dt[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; /* [14] lanes at work, summed over the steps; [15] steps after the work queue ran dry */
    const unsigned long long diag_t0 = __builtin_readcyclecounter();
    unsigned long long diag_tick = diag_t0;
#define RT_STEP_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); diag_dt[k] += now_ - diag_tick; diag_tick = now_; }
#else
#define RT_STEP_TICK(k)
#endif
    for (;;) {
        unsigned long long need = __builtin_amdgcn_ballot_w64(phase == DP_DONE);
        if (dp.n_epochs == 0u) need = 0ull;
        while (need != 0ull) {
            if (q_next == q_end) {
                if (exhausted) break;
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(dp.work_queue, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c) + (dp.own_first_chunk != 0u ? gridDim.x : 0u); /* with own first chunks, 0 .. G - 1 are taken */
                if (c * 64u >= total_slots) { exhausted = true; break; }
                q_next = c * 64u;
                q_end = q_next + 64u < total_slots ? q_next + 64u : total_slots;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t avail = q_end - q_next;
            if (phase == DP_DONE && rank < avail) {
                const uint32_t slot = q_next + rank;
                uint32_t row, col;
                if (DUMMY != 0) { /* instantiation 1: the pixels grouped by cost (rt_kernels.h); its own, so that instantiation 0 does not carry it */
                    out_index = dp.pixel_order[slot];
                    row = out_index / fr.cols;
                    col = out_index - row * fr.cols;
                } else { /* the image in 8-row bands, column by column: a chunk is an 8 x 8 tile */
                    const uint32_t band = slot / band_slots;
                    const uint32_t r = slot - band * band_slots;
                    const uint32_t rows_left = fr.rows - (band << 3);
                    const uint32_t band_rows = rows_left < 8u ? rows_left : 8u;
                    col = r / band_rows;
                    row = (band << 3) + (r - col * band_rows);
                    out_index = row * fr.cols + col;
                }
                const uint32_t x = fr.x0 + col, y = fr.y0 + row * fr.y_step;
                clip_y = (fr.half_height - (float)y) / fr.height_f;
                clip_x = ((float)x - fr.half_width) / fr.height_f;
                rng_open(rng, dp.rng_states + (size_t)out_index * RNG_WORDS);
                epoch = 0u;
                phase = DP_START;
            }
            const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
            q_next += n_need < avail ? n_need : avail;
            need = __builtin_amdgcn_ballot_w64(phase == DP_DONE);
        }
        if (__builtin_amdgcn_ballot_w64(phase != DP_DONE) == 0ull) break;
        /* one copy of each expensive piece per iteration — the lanes of a wave are in all phases at once, so whatever
         * appears in several branches is executed several times over */
        if (phase == DP_START) start_epoch();

        CastResult cr;
#ifdef RT_DIST_NO_PAIRS /* A/B: every leaf wave-uniformly (round 2) */
        cr.prim = -1;
        cr.t = 0.0f;
        cr.bf = 0u;
        cr.a0 = cr.a1 = cr.a2 = 0.0f;
        if (phase != DP_DONE) cr = cast_asm(sc, req);
#else
#ifdef RT_DIAG_PAIR_TIME
        RT_STEP_TICK(9)
        cr = cast_pairs(sc, req, phase != DP_DONE, pair_lds, diag_dt);
        diag_dt[0] += 1ull;
        diag_dt[14] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(phase != DP_DONE));
        if (exhausted) diag_dt[15] += 1ull;
        diag_tick = __builtin_readcyclecounter();
#else
        cr = cast_pairs(sc, req, phase != DP_DONE, pair_lds); /* all lanes: the idle ones help with the others' pairs */
#endif
#endif
        if (phase == DP_DONE) continue;
        casts += 1u;

#include "rt_dist_advance.inc"
        if (!casting) { /* the chain of this (pixel, epoch) is over */
            dp.sp_hdr[(size_t)epoch * n_pixels + out_index] = (uint32_t)sp | (terminal ? 0x100u : 0u);
            epoch += 1u;
            if (epoch < dp.n_epochs) {
                phase = DP_START;
            } else {
                rng_park(rng); /* the stream position */
                phase = DP_DONE;
            }
        }
        RT_STEP_TICK(13)
    }

#ifdef RT_DIAG_PAIR_TIME
    diag_dt[8] = __builtin_readcyclecounter() - diag_t0;
    if (lane == 0u) {
        for (int q = 0; q < 16; ++q) atomicAdd(&g_pair_time[q], diag_dt[q]);
        atomicMax(&g_chain_critical[0], diag_dt[0]); /* the wave that made the most steps, the wave that ran longest: what a launch cannot be shorter than */
        atomicMax(&g_chain_critical[1], diag_dt[8]);
        atomicAdd(&g_chain_critical[2], 1ull);
    }
#endif
    if (dp.ray_count != nullptr) {
        uint32_t c = casts;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0u && c != 0u) atomicAdd(dp.ray_count, (unsigned long long)c);
    }
}

/* (Round 2 also had the chain as a kernel of workgroup-local queues — every cast a work item, START / NEXT / REFR queues, phase-
 * coherent chunks, the organisation of the Whitted path.  It lost to the persistent lanes above twice: 2.35 against 2.28 ms per epoch
 * in round 2, 1 015 against 1 136 Msamples/s in round 3 with the pair-wise cast in both (profiles/README.md), and is gone.) */

/* get_shade (main.rs:407-464) for every request of the batch.  The request arrays are sparse — slot k of a sample is in
 * use only if its chain got that far (90 % at slot 0, a few per cent at slot 8) — so a workgroup first lists the live
 * (slot, sample) pairs of its `tile` samples in LDS and its waves then work through the list 64 at a time with
 * every lane busy.  No global atomics; the order within the list does not matter (each result has its own address) —
 * which is used a second time: a light behind the surface needs no shadow cast (main.rs:431), on average one of the
 * scene's three, and a wave skips a light only when none of its lanes needs it.  So the list is bucket-sorted by which of
 * the first three lights face the GEOMETRIC normal (a guess at the reference's test, which uses the material's adjusted
 * normal: it only orders the work), and most waves then hold requests that need the same lights. */
#define DIST_SHADE_BUCKETS 8u /* 3 bits: which of the first three lights face the surface */
#define DIST_SHADE_HDR (1u + 2u * DIST_SHADE_BUCKETS) /* words before the lists: [0] requests, bucket sizes -> starts, bucket cursors */
#ifndef RT_DIST_SHADE_MIN_WAVES
#define RT_DIST_SHADE_MIN_WAVES 6 /* 80 VGPRs; 4 / 5 / 8 measured in profiles/README.md */
#endif
#ifndef RT_DIST_SHADE_THREADS
#define RT_DIST_SHADE_THREADS 256
#endif
#ifdef RT_DIAG_PAIR_TIME
static __device__ unsigned long long g_shade_time[8]; /* wave ticks: [0] list building + sort, [1] request load + material, [2] lights -> directional + facing, [3] the cast, [4] diffuse / specular, [5] the kernel, [6] wave-casts */
extern "C" int rt_diag_read_shade_time(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(rt::g_shade_time), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_shade_time), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#define RT_SHADE_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); sdt[k] += now_ - stick; stick = now_; }
#else
#define RT_SHADE_TICK(k)
#endif
__global__ __launch_bounds__(RT_DIST_SHADE_THREADS, RT_DIST_SHADE_MIN_WAVES) void dist_shade_kernel(const KernelScene sc, const DistParams dp, const size_t n_samples, const uint32_t tile, const uint32_t list_cap, const uint32_t sort) {
    extern __shared__ uint32_t shade_lds[];
#ifndef RT_DIST_SHADE_NO_PAIRS
    __shared__ PairLdsSlim pair_lds_all[RT_DIST_SHADE_THREADS / 64u];
    PairLdsSlim *const pair_lds = &pair_lds_all[threadIdx.x >> 6];
#endif
    uint32_t *const bucket_start = shade_lds + 1u, *const bucket_cursor = bucket_start + DIST_SHADE_BUCKETS;
    uint32_t *const unsorted = shade_lds + DIST_SHADE_HDR;       /* slot << 24 | bucket << 16 | sample - tile0 */
    uint32_t *const shade_list = unsorted + list_cap;            /* the same, bucket by bucket */
    const uint32_t lane = threadIdx.x & 63u;
    const size_t tile0 = (size_t)blockIdx.x * tile;
#ifdef RT_DIAG_PAIR_TIME
    unsigned long long sdt[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long stick = __builtin_readcyclecounter();
    const unsigned long long stick0 = stick;
#endif
    for (uint32_t k = threadIdx.x; k < DIST_SHADE_HDR; k += blockDim.x) shade_lds[k] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < tile; i += blockDim.x) {
        if (tile0 + i < n_samples) {
            const uint32_t hdr = dp.sp_hdr[tile0 + i];
            const uint32_t cnt = (hdr & 0xffu) + ((hdr >> 8) & 1u);
            if (cnt != 0u) {
                const uint32_t base = atomicAdd(&shade_lds[0], cnt);
                for (uint32_t k = 0; k < cnt; ++k) unsorted[base + k] = (k << 24) | i;
            }
        }
    }
    __syncthreads();
    const uint32_t total = shade_lds[0];
    const uint32_t key_lights = sc.n_lights < 3u ? sc.n_lights : 3u;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t entry = unsorted[e];
        uint32_t key = 0u;
        if (sort) {
            const uint4 *r = dp.sp_req + ((size_t)(entry >> 24) * n_samples + tile0 + (entry & 0xffffu)) * 4u;
            const uint4 a = r[0], b = r[1];
            const V3 pos = v3(duf(a.x), duf(a.y), duf(a.z)), normal = v3(duf(b.x), duf(b.y), duf(b.z));
            for (uint32_t l = 0; l < key_lights; ++l) {
                const auto &L = uniform_ref(sc.lights + l);
                const V3 toward = L.kind == RT_LIGHT_DIRECTIONAL ? v3(L.direction[0], L.direction[1], L.direction[2])
                                                                 : pos - v3(L.origin[0], L.origin[1], L.origin[2]);
                if (dot(toward, normal) < 0.0f) key |= 1u << l;
            }
        }
        unsorted[e] = entry | (key << 16);
        atomicAdd(&bucket_start[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < DIST_SHADE_BUCKETS) { /* sizes -> starts */
        uint32_t before = 0u;
        for (uint32_t k = 0; k < threadIdx.x; ++k) before += bucket_start[k];
        bucket_cursor[threadIdx.x] = before;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t entry = unsorted[e];
        shade_list[atomicAdd(&bucket_cursor[(entry >> 16) & (DIST_SHADE_BUCKETS - 1u)], 1u)] = entry;
    }
    __syncthreads();
    uint32_t casts = 0u;
    RT_SHADE_TICK(0)
#if defined(RT_AB_SHADE_STUB) && RT_AB_SHADE_STUB == 3
    if (shade_list[0] != 0xffffffffu) return;
#endif
    for (uint32_t first = (threadIdx.x >> 6) * 64u; first < total; first += blockDim.x) {
        const bool active = first + lane < total;
        V3 pos = v3(0.0f, 0.0f, 0.0f), normal = v3(0.0f, 0.0f, 1.0f), view = v3(0.0f, 0.0f, 1.0f);
        float u = 0.0f, v = 0.0f;
        uint32_t obj = 0u, prim = 0u;
        size_t at = 0;
        if (active) {
            const uint32_t e = shade_list[first + lane];
            at = (size_t)(e >> 24) * n_samples + tile0 + (e & 0xffffu);
            const uint4 *r = dp.sp_req + at * 4u;
            const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
            pos = v3(duf(a.x), duf(a.y), duf(a.z)); u = duf(a.w);
            normal = v3(duf(b.x), duf(b.y), duf(b.z)); v = duf(b.w);
            view = v3(duf(c.x), duf(c.y), duf(c.z)); obj = c.w;
            prim = d.x;
        }
        const Mat m = material_approx(sc.materials[obj], u, v);
        const V3 adj_n = adjust_normal(m.normal, normal); /* main.rs:410 */
        V3 sum = v3(0.0f, 0.0f, 0.0f);
        RT_SHADE_TICK(1)
#if defined(RT_AB_SHADE_STUB) && RT_AB_SHADE_STUB == 2
        sum = adj_n + m.diffuse;
        for (uint32_t light_i = 0; light_i < 0u; ++light_i) {
#else
        for (uint32_t light_i = 0; light_i < sc.n_lights; ++light_i) { /* wave-uniform */
#endif
            const auto &L = uniform_ref(sc.lights + light_i);
            /* does the light ask for a shadow cast (main.rs:413-433)?  light_asks answers without a spot light's acos wherever the
             * angle is clear of the cone's edge (rt_shade.h); the light's colour — a spot light's powf — waits for the lit lanes */
            V3 l_direction = v3(0.0f, 0.0f, 0.0f);
#ifdef RT_SHADE_NO_LIGHT_ASKS /* A/B (round 3): the light evaluated in full for every active lane */
            DirLight dl0;
            dl0.direction = dl0.color = v3(0.0f, 0.0f, 0.0f);
            bool need = false;
            if (active && approximate_into_directional(L, pos, &dl0)) {
                const float cosine = -dot(dl0.direction, adj_n);
                need = !(cosine <= 0.0f);
            }
            l_direction = dl0.direction;
#else
            const bool need = light_asks(L, uniform_ref(sc.light_aux + light_i), pos, adj_n, &l_direction) && active;
#endif
            RT_SHADE_TICK(2)
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
            Ray req;
            req.o = pos;
            req.d = -l_direction;
            req.mode = FACE_BACK;
            req.excl = pack_excl(prim, FACE_BACK);
#ifdef RT_DIST_SHADE_NO_PAIRS /* A/B */
            CastResult cr;
            cr.prim = -1;
            cr.t = 0.0f;
            cr.bf = 0u;
            cr.a0 = cr.a1 = cr.a2 = 0.0f;
#if !defined(RT_AB_SHADE_STUB) || RT_AB_SHADE_STUB != 1 /* attribution builds (profiles/r04_shade_attribution.txt): 1 no cast, 2 no light loop, 3 the lists only */
            if (need) cr = cast_asm(sc, req);
#endif
#ifdef RT_DIAG_PAIR_TIME
            RT_SHADE_TICK(3)
            sdt[6] += 1ull;
#endif
#elif defined(RT_DIAG_PAIR_TIME)
            unsigned long long diag_dt[9];
            const CastResult cr = cast_pairs(sc, req, need, pair_lds, diag_dt);
#else
            const CastResult cr = cast_pairs(sc, req, need, pair_lds); /* all lanes: those without a shadow ray help with the others' pairs */
#endif
            if (need) {
                casts += 1u;
                bool lit = true;
                if (cr.prim >= 0) {
                    const bool has_origin = (L.kind != RT_LIGHT_DIRECTIONAL) || (L.has_origin != 0u);
                    if (has_origin) {
                        const V3 occ = req.o + req.d * cr.t;
                        if (distance(pos, occ) < distance(pos, v3(L.origin[0], L.origin[1], L.origin[2]))) lit = false;
                    } else {
                        lit = false;
                    }
                }
                if (lit) {
                    DirLight dl;
                    dl.direction = dl.color = v3(0.0f, 0.0f, 0.0f);
                    (void)approximate_into_directional(L, pos, &dl); /* the light asked: Some */
                    const V3 light_direction = req.d;
                    const V3 diffuse = get_diffuse(m, adj_n, light_direction) * dl.color;
                    const V3 specular = get_specular(m, adj_n, -view, light_direction) * dl.color;
                    sum = sum + diffuse * (1.0f - m.shiness) + specular * m.shiness;
                }
            }
            RT_SHADE_TICK(4)
        }
        if (active) dp.sp_shade[at] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    }
#ifdef RT_DIAG_PAIR_TIME
    sdt[5] = __builtin_readcyclecounter() - stick0;
    if (lane == 0u) for (int q = 0; q < 7; ++q) atomicAdd(&g_shade_time[q], sdt[q]);
#endif
    if (dp.ray_count != nullptr) {
        for (int off = 32; off > 0; off >>= 1) casts += __shfl_down(casts, off, 64);
        if (lane == 0u && casts != 0u) atomicAdd(dp.ray_count, (unsigned long long)casts);
    }
}

/* the unwind (main.rs:571, 590, 605), the sample filter (main.rs:1157-1160) and the accumulation (main.rs:1165) */
__global__ __launch_bounds__(256) void dist_unwind_kernel(const DistParams dp, const size_t n_pixels, const size_t call_pixels_stride) {
    const size_t n_samples = n_pixels * dp.n_epochs;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += (size_t)gridDim.x * blockDim.x) {
        V3 accum = v3(0.0f, 0.0f, 0.0f);
        uint32_t cost = 0u; /* of the pixel's samples of this batch: a cast per level and the primary one */
        if (dp.accum != nullptr) accum = v3(dp.accum[p * 3u], dp.accum[p * 3u + 1u], dp.accum[p * 3u + 2u]);
        for (uint32_t e = 0; e < dp.n_epochs; ++e) {
            const size_t s = (size_t)e * n_pixels + p;
            const uint32_t hdr = dp.sp_hdr[s];
            const uint32_t frames = hdr & 0xffu;
            cost += frames + 1u;
            V3 value = v3(0.0f, 0.0f, 0.0f);
            if ((hdr >> 8) & 1u) {
                const float4 t = dp.sp_shade[(size_t)frames * n_samples + s];
                value = v3(t.x, t.y, t.z);
            }
            for (uint32_t k = frames; k-- > 0u;) {
                const float4 f = dp.sp_frame[(size_t)k * n_samples + s];
                const float4 sh4 = dp.sp_shade[(size_t)k * n_samples + s];
                const V3 shade = v3(sh4.x, sh4.y, sh4.z);
                if (dfu(f.w) == 2u) {
                    value = (value + shade) * f.x; /* main.rs:605 */
                } else {
                    const V3 sc_ = value * v3(f.x, f.y, f.z);  /* main.rs:566, 585 */
                    value = shade + (sc_ - shade) * 0.5f;       /* palette Mix::mix(&s, 0.5), main.rs:571, 590 */
                }
            }
            const bool ok = rtdm::is_normal(value.x) && rtdm::is_normal(value.y) && rtdm::is_normal(value.z);
            const size_t o = ((size_t)(dp.epoch0 + e)) * call_pixels_stride + p;
            if (dp.samples != nullptr) {
                dp.samples[o * 3u] = value.x;
                dp.samples[o * 3u + 1u] = value.y;
                dp.samples[o * 3u + 2u] = value.z;
            }
            if (dp.valid != nullptr) dp.valid[o] = ok ? 1 : 0;
            if (ok) accum = accum + value;
        }
        if (dp.accum != nullptr) {
            dp.accum[p * 3u] = accum.x;
            dp.accum[p * 3u + 1u] = accum.y;
            dp.accum[p * 3u + 2u] = accum.z;
        }
        if (dp.pixel_cost != nullptr) dp.pixel_cost[p] = (cost * 8u + dp.n_epochs - 1u) / dp.n_epochs; /* eighths of a cast per sample */
    }
}

/* ---- the chain kernel's pixels grouped by cost (rt_kernels.h DistParams::pixel_order): a counting sort in three launches ----
 * scratch[0..255] the histogram, scratch[256..511] the buckets' cursors; dearest first; the order within a bucket is whatever the
 * atomics make it (it only orders work) */
#define DIST_ORDER_CHUNK 4096u
__global__ __launch_bounds__(256) void dist_order_hist_kernel(const uint32_t *cost, uint32_t n, uint32_t *scratch) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * DIST_ORDER_CHUNK;
    for (uint32_t i = first + threadIdx.x; i < first + DIST_ORDER_CHUNK && i < n; i += 256u) atomicAdd(&h[cost[i] < 255u ? cost[i] : 255u], 1u);
    __syncthreads();
    if (h[threadIdx.x] != 0u) atomicAdd(&scratch[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void dist_order_scan_kernel(uint32_t *scratch) {
    if (threadIdx.x == 0u) {
        uint32_t sum = 0u;
        for (int b = 255; b >= 0; --b) { scratch[256 + b] = sum; sum += scratch[b]; }
    }
}
__global__ __launch_bounds__(256) void dist_order_scatter_kernel(const uint32_t *cost, uint32_t n, uint32_t *scratch, uint32_t *order) {
    __shared__ uint32_t h[256], base[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * DIST_ORDER_CHUNK;
    for (uint32_t i = first + threadIdx.x; i < first + DIST_ORDER_CHUNK && i < n; i += 256u) atomicAdd(&h[cost[i] < 255u ? cost[i] : 255u], 1u);
    __syncthreads();
    base[threadIdx.x] = h[threadIdx.x] != 0u ? atomicAdd(&scratch[256u + threadIdx.x], h[threadIdx.x]) : 0u; /* this workgroup's stretch of the bucket */
    __syncthreads();
    for (uint32_t i = first + threadIdx.x; i < first + DIST_ORDER_CHUNK && i < n; i += 256u) {
        const uint32_t b = cost[i] < 255u ? cost[i] : 255u;
        order[atomicAdd(&base[b], 1u)] = i;
    }
}

hipError_t launch_dist_pixel_order(const uint32_t *cost, uint32_t *order, uint32_t n_pixels, uint32_t *scratch, hipStream_t stream) {
    if (n_pixels == 0u) return hipSuccess;
    hipError_t e = hipMemsetAsync(scratch, 0, 256u * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t groups = (n_pixels + DIST_ORDER_CHUNK - 1u) / DIST_ORDER_CHUNK;
    hipLaunchKernelGGL(dist_order_hist_kernel, dim3(groups), dim3(256), 0, stream, cost, n_pixels, scratch);
    hipLaunchKernelGGL(dist_order_scan_kernel, dim3(1), dim3(256), 0, stream, scratch);
    hipLaunchKernelGGL(dist_order_scatter_kernel, dim3(groups), dim3(256), 0, stream, cost, n_pixels, scratch, order);
    return hipGetLastError();
}

size_t distributed_split_bytes_per_sample(int32_t max_depth) {
    const size_t d = (size_t)(max_depth > 0 ? max_depth : 0);
    return sizeof(uint32_t) + (d + 1u) * (4u * sizeof(uint4) + sizeof(float4)) + d * sizeof(float4);
}

/* one batch of dp.n_epochs epochs (dp.epoch0 = its first epoch within the call) in two halves, so that a caller may put them on
 * different streams: the chain kernel (dp.work_queue zeroed) ... */
/* the waves the chain kernel's grid has at most: as many as are resident together */
uint32_t dist_chain_waves(uint32_t resident_waves) {
    uint32_t chain_waves = resident_waves / 3u * (uint32_t)RT_DIST_CHAIN_MIN_WAVES; /* resident_waves is sized for 3 per SIMD */
    const uint32_t per_simd = (uint32_t)option(OPT_DIST_CHAIN_WAVES, 0); /* A/B: waves per SIMD of the chain kernel's grid */
    if (per_simd >= 1u && per_simd <= (uint32_t)RT_DIST_CHAIN_MIN_WAVES) chain_waves = resident_waves / 3u * per_simd;
    return chain_waves;
}

hipError_t launch_dist_chain(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, uint32_t resident_waves, hipStream_t stream) {
    const uint32_t total = fr.cols * fr.rows;
    if (total == 0u || dp.n_epochs == 0u) return hipSuccess;
    uint32_t waves = (total + 63u) / 64u;
    const uint32_t chain_waves = dist_chain_waves(resident_waves);
    if (waves > chain_waves) waves = chain_waves;
    if (dp.pixel_order != nullptr) hipLaunchKernelGGL((dist_chain_kernel<1>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
    else hipLaunchKernelGGL((dist_chain_kernel<0>), dim3(waves), dim3(64), 0, stream, sc, fr, dp);
    return hipGetLastError();
}

/* ... and what only reads its records: the shade kernel and the unwind */
hipError_t launch_dist_shade_unwind(const KernelScene &sc, const KernelFrame &fr, const DistParams &dp, hipStream_t stream, const hipEvent_t *ev) {
    const uint32_t total = fr.cols * fr.rows;
    if (total == 0u || dp.n_epochs == 0u) return hipSuccess;
    const size_t n_samples = (size_t)total * dp.n_epochs;
    const uint32_t slots = (uint32_t)(fr.max_depth > 0 ? fr.max_depth : 0) + 1u;
    if (ev != nullptr) (void)hipEventRecord(ev[0], stream); /* profiling: [0, 1] around the shade kernel, [2, 3] around the unwind */
    {
        uint32_t tile = 256u; /* samples per workgroup (18 KB of LDS at depth 8: six workgroups per CU); the two lists must fit 48 KB */
        uint32_t sort = 1u;
        /* A/B knobs (profiles/README.md) */
        { const uint32_t t = (uint32_t)option(OPT_SHADE_TILE, 0); if (t >= 32u && t <= 4096u) tile = t; }
        sort = option(OPT_SHADE_SORT, 1) != 0 ? 1u : 0u;
        while (tile > 32u && (DIST_SHADE_HDR + 2u * (size_t)tile * slots) * sizeof(uint32_t) > 49152u) tile >>= 1;
        const size_t shade_tiles = (n_samples + tile - 1u) / tile;
        const uint32_t list_cap = tile * slots;
        const size_t shade_lds = (DIST_SHADE_HDR + 2u * (size_t)list_cap) * sizeof(uint32_t);
        hipLaunchKernelGGL(dist_shade_kernel, dim3((unsigned)shade_tiles), dim3(RT_DIST_SHADE_THREADS), shade_lds, stream, sc, dp, n_samples, tile, list_cap, sort);
    }
    if (ev != nullptr) { (void)hipEventRecord(ev[1], stream); (void)hipEventRecord(ev[2], stream); }
    size_t blocks = ((size_t)total + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(dist_unwind_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dp, (size_t)total, (size_t)total);
    if (ev != nullptr) (void)hipEventRecord(ev[3], stream);
    return hipGetLastError();
}

} /* namespace rt */
#ifdef RT_DIAG_NEED
RT_DIAG_NEED_READER(rt_diag_read_need_dist)
#endif
#ifdef RT_DIAG_PAIR_TIME
RT_DIAG_PAIR_TIME_READER(rt_diag_read_pair_time)
#endif
#ifdef RT_DIAG_PREPARE
/* diagnostic build (tools/diag_prepare.py): the look-ahead pass alone over n freshly seeded records, `reps` times (the
 * prepared flags cleared in between); ms_out[0] = average ms of scan + prepare, ms_out[1] = of the scan alone */
namespace rt {
__global__ void rng_diag_unprepare_kernel(uint32_t *states, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) states[(size_t)p * RNG_WORDS + RNG_FLAGS] &= 1u;
}
}
extern "C" int rt_diag_prepare_time(uint32_t n_records, uint32_t reps, uint32_t compute_units, float *ms_out) {
    uint32_t *states = nullptr, *list = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&states), (size_t)n_records * rt::RNG_WORDS * sizeof(uint32_t)) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void **>(&list), ((size_t)n_records + 1u) * sizeof(uint32_t)) != hipSuccess) return -1;
    rt::KernelFrame fr = {};
    fr.cols = n_records; fr.rows = 1u; fr.y_step = 1u;
    if (rt::launch_rng_seed(states, fr, nullptr) != hipSuccess) return -2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float total = 0.0f, scan = 0.0f;
    for (uint32_t r = 0; r < reps + 1u; ++r) { /* the first is a warm-up */
        hipLaunchKernelGGL(rt::rng_diag_unprepare_kernel, dim3((n_records + 255u) / 256u), dim3(256), 0, nullptr, states, n_records);
        hipEventRecord(e0, nullptr);
        if (rt::launch_rng_prepare(states, n_records, list, compute_units, nullptr) != hipSuccess) return -3;
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0u) total += ms;
        hipEventRecord(e0, nullptr);
        if (rt::launch_rng_prepare(states, n_records, list, compute_units, nullptr) != hipSuccess) return -3; /* nothing left to prepare */
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0u) scan += ms;
    }
    ms_out[0] = total / (float)reps;
    ms_out[1] = scan / (float)reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipFree(states); (void)hipFree(list);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
#endif
