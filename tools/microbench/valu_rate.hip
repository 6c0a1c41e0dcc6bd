// Microbenchmark: VALU issue rate on gfx950 by operand form and by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every variant executes the same number of wave-instructions (64 per loop trip, 8 independent chains); the table gives
// wave-instructions per SIMD per 4 cycles at 2.4 GHz (1.00 = the 16-lane SIMD's nominal rate) for 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));
enum { ADD_VV, MUL_SV, MUL_VV, FMA_VVV, FMA_SVV, PK_MUL_VV, PK_MUL_SV, PK_ADD_VV, PK_FMA_VVV, MUL_LIT, CMP_SV, N_MODES };
static const char *names[N_MODES] = {"v_add_f32 v,v,v", "v_mul_f32 v,s,v", "v_mul_f32 v,v,v", "v_fma_f32 v,v,v,v", "v_fma_f32 v,s,v,v",
                                     "v_pk_mul_f32 v2,v2,v2", "v_pk_mul_f32 v2,s2,v2", "v_pk_add_f32 v2,v2,v2", "v_pk_fma_f32 v2,v2,v2,v2",
                                     "v_mul_f32 v,literal,v", "v_cmp_lt_f32 vcc,s,v"};

#define X8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sa, float sb) {
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float x = 1.0000001f + threadIdx.x * 1e-9f;
    const v2f px = {x, x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == ADD_VV) {
#define OP(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                X8(OP)
#undef OP
            } else if (MODE == MUL_SV) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sa));
                X8(OP)
#undef OP
            } else if (MODE == MUL_VV) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                X8(OP)
#undef OP
            } else if (MODE == FMA_VVV) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(x));
                X8(OP)
#undef OP
            } else if (MODE == FMA_SVV) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(sa), "v"(x));
                X8(OP)
#undef OP
            } else if (MODE == PK_MUL_VV) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(px));
                X8(OP)
#undef OP
            } else if (MODE == PK_MUL_SV) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, s[40:41], %0" : "+v"(p[i]) : : "s40", "s41");
                asm volatile("s_mov_b32 s40, %0\n s_mov_b32 s41, %1" : : "s"(sa), "s"(sb) : "s40", "s41");
                X8(OP)
#undef OP
            } else if (MODE == PK_ADD_VV) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(px));
                X8(OP)
#undef OP
            } else if (MODE == PK_FMA_VVV) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p[i]) : "v"(px));
                X8(OP)
#undef OP
            } else if (MODE == MUL_LIT) {
#define OP(i) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(a[i]));
                X8(OP)
#undef OP
            } else if (MODE == CMP_SV) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(a[i]), "s"(sa) : "vcc");
                X8(OP)
#undef OP
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int waves_per_simd, int cus) {
    float *out;
    const int blocks = cus * waves_per_simd, threads = 256; // a block = one wave per SIMD of a CU
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 10, 1.0000001f, 0.9999999f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.0000001f, 0.9999999f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    const double wave_instr_per_simd = 64.0 * iters * waves_per_simd;
    return wave_instr_per_simd / (ms * 1e-3 * 2.4e9 / 4.0); // per 4 cycles at 2.4 GHz
}

template <int MODE>
static void row(int cus) {
    printf("%-28s", names[MODE]);
    for (int w : {1, 2, 3, 4, 6, 8}) printf(" %5.2f", run<MODE>(w, cus));
    printf("\n");
}

int main() {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    printf("CUs %d, clock %d kHz; wave-instructions per SIMD per 4 cycles @2.4 GHz, by waves per SIMD:\n", cus, prop.clockRate);
    printf("%-28s %5d %5d %5d %5d %5d %5d\n", "", 1, 2, 3, 4, 6, 8);
    row<ADD_VV>(cus); row<MUL_SV>(cus); row<MUL_VV>(cus); row<FMA_VVV>(cus); row<FMA_SVV>(cus); row<PK_MUL_VV>(cus); row<PK_MUL_SV>(cus);
    row<PK_ADD_VV>(cus); row<PK_FMA_VVV>(cus); row<MUL_LIT>(cus); row<CMP_SV>(cus);
    return 0;
}
