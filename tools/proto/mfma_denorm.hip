// Probe: does v_mfma_f32_32x32x16_f16 keep f16 denormal inputs, and does the f32 -> f16 conversion produce them?
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_denorm mfma_denorm.hip ; prints the three values (expect 2^-16 each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float tiny) {
    f16x8 a, b, c;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)1.0f; c[i] = (_Float16)(tiny * 0.5f); }
    f32x16 z = {0};
    f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);      // denormal A
    f32x16 r2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, z, 0, 0, 0);      // denormal B
    f32x16 r3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, z, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = r1[0]; out[1] = r2[0]; out[2] = r3[0]; out[3] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    k<<<1, 64>>>(d, 9.5367431640625e-07f);   // 2^-20: f16 denormal (16 ulp)
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("A-denorm %g  B-denorm %g  half %g  cvt %g  (expect %g %g %g %g)\n", h[0], h[1], h[2], h[3], 16 * 9.5367431640625e-07, 16 * 9.5367431640625e-07,
           8 * 9.5367431640625e-07, 9.5367431640625e-07);
    return 0;
}
