// Microbenchmark: issue cost of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950, with SGPR-pair
// operands (the triangle record lives in SGPRs) and op_sel broadcast of one VGPR to both halves.
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32_rate pk_f32_rate.hip && ./pk_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int MODE>
__global__ void k(float *out, unsigned long long *ticks, int iters, float sa, float sb) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f;
    float x = 1.0001f + threadIdx.x * 1e-6f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {b0, b1}, p3 = {b2, b3}, px = {x, x};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) { // 8 scalar-operand v_mul_f32 (4 independent chains x 2)
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %9, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %9, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %9, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %9, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "s"(sa), "s"(sb));
            } else if (MODE == 1) { // 4 v_pk_mul_f32 with an SGPR pair and a VGPR pair: the same 8 multiplies
                asm volatile("s_mov_b32 s40, %8\n s_mov_b32 s41, %9\n"
                             "v_pk_mul_f32 %0, s[40:41], %0\n v_pk_mul_f32 %1, s[40:41], %1\n v_pk_mul_f32 %2, s[40:41], %2\n v_pk_mul_f32 %3, s[40:41], %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a1), "+v"(a3), "+v"(b1), "+v"(b3)
                             : "s"(sa), "s"(sb) : "s40", "s41");
            } else if (MODE == 2) { // 4 v_pk_mul_f32, SGPR pair x one VGPR broadcast to both halves (op_sel_hi:[1,0])
                asm volatile("s_mov_b32 s40, %8\n s_mov_b32 s41, %9\n"
                             "v_pk_mul_f32 %0, s[40:41], %10 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, s[40:41], %10 op_sel_hi:[1,0]\n"
                             "v_pk_mul_f32 %2, s[40:41], %10 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, s[40:41], %10 op_sel_hi:[1,0]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a1), "+v"(a3), "+v"(b1), "+v"(b3)
                             : "s"(sa), "s"(sb), "v"(px) : "s40", "s41");
            } else if (MODE == 3) { // 8 v_add_f32 VGPR-VGPR
                asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                             "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(x));
            } else if (MODE == 4) { // 4 v_pk_add_f32 VGPR pairs
                asm volatile("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int MODE>
static void run(const char *name, int waves_per_simd) {
    float *out; unsigned long long *ticks;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd; // one block per CU, 4 SIMDs
    (void)hipMalloc(&out, sizeof(float) * blocks * threads); (void)hipMalloc(&ticks, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, ticks, 10, 1.0000001f, 0.9999999f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, ticks, iters, 1.0000001f, 0.9999999f);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double flops = 8.0 * (REP / 8) * iters; // f32 operations per lane
    printf("%-44s %d waves/SIMD: %.3f ms, %.2f ticks per 8 lane-ops per wave, %.2f T lane-op/s\n", name, waves_per_simd, ms,
           (double)t / (REP / 8) / iters, flops * blocks * threads / (ms * 1e-3) / 1e12);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    for (int w : {1, 2, 6}) {
        run<0>("8 x v_mul_f32 (SGPR x VGPR)", w);
        run<1>("4 x v_pk_mul_f32 (SGPR pair x VGPR pair)", w);
        run<2>("4 x v_pk_mul_f32 (SGPR pair x VGPR bcast)", w);
        run<3>("8 x v_add_f32 (VGPR + VGPR)", w);
        run<4>("4 x v_pk_add_f32 (VGPR pairs)", w);
    }
    return 0;
}
