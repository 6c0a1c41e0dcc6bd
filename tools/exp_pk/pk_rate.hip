// Experiment: do packed FP32 operations (v_pk_mul_f32 / v_pk_add_f32, one operand an SGPR pair) issue at the rate of the plain ones?
// Whole-device timing: 8 waves per SIMD, every wave a loop of 8 independent instructions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void packed(float *out, unsigned long long sp, int iters) {
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_pk_mul_f32 %0, %8, %0\n\tv_pk_add_f32 %1, %8, %1\n\tv_pk_mul_f32 %2, %8, %2\n\tv_pk_add_f32 %3, %8, %3\n\t"
                     "v_pk_mul_f32 %4, %8, %4\n\tv_pk_add_f32 %5, %8, %5\n\tv_pk_mul_f32 %6, %8, %6\n\tv_pk_add_f32 %7, %8, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sp));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
}
__global__ void plain(float *out, unsigned long long sp, int iters) {
    float b0 = 1.0f + threadIdx.x, b1 = b0, b2 = b0, b3 = b0, b4 = b0, b5 = b0, b6 = b0, b7 = b0;
    const float s = __uint_as_float((uint32_t)sp);
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mul_f32 %0, %8, %0\n\tv_add_f32 %1, %8, %1\n\tv_mul_f32 %2, %8, %2\n\tv_add_f32 %3, %8, %3\n\t"
                     "v_mul_f32 %4, %8, %4\n\tv_add_f32 %5, %8, %5\n\tv_mul_f32 %6, %8, %6\n\tv_add_f32 %7, %8, %7"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "s"(s));
    out[blockIdx.x * blockDim.x + threadIdx.x] = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    const int iters = 1 << 16;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(packed, dim3(256 * 4 * 8), dim3(64), 0, 0, out, 0x3f8000003f800000ull, iters);
            else hipLaunchKernelGGL(plain, dim3(256 * 4 * 8), dim3(64), 0, 0, out, 0x3f8000003f800000ull, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr = 256.0 * 4 * 8 * iters * 8;  // wave-instructions
        printf("%s: %.3f ms, %.2f G wave-instructions/s per SIMD-cycle-equivalent: %.3f instructions per SIMD per ns (%.1f T lane-results/s)\n", which == 0 ? "packed" : "plain ", ms,
               instr / ms / 1e6, instr / 1024.0 / (ms * 1e6), instr * 64 * (which == 0 ? 2 : 1) / ms / 1e9);
    }
    return 0;
}
