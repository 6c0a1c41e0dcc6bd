#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float *out, unsigned long long sp) {
    const float x = in[threadIdx.x];
    f2 xx; xx.x = x; xx.y = 123.0f;
    f2 r, r2, r3;
    // (s.lo * x, s.hi * x): src1 low half for both
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(sp), "v"(xx));
    // r2 = r - (s.lo, s.hi)
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(r), "s"(sp));
    // r3 = r2 * r2
    asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(r3) : "v"(r2));
    out[threadIdx.x * 2] = r3.x; out[threadIdx.x * 2 + 1] = r3.y;
}
int main() {
    float h[64], *di, *dout, ho[128];
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 0.37f;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof ho); hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    float a = 3.25f, b = -0.7f; unsigned long long sp; uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); sp = (unsigned long long)ub << 32 | ua;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, sp);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        float e0 = a * h[i] - a; e0 = e0 * e0; float e1 = b * h[i] - b; e1 = e1 * e1;
        if (memcmp(&e0, &ho[2*i], 4) || memcmp(&e1, &ho[2*i+1], 4)) ++bad;
    }
    printf("packed f32 with an SGPR pair: %d of 64 lanes differ from the scalar arithmetic\n", bad);
    return bad != 0;
}
