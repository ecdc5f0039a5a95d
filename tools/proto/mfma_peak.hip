// What the f16 matrix pipe sustains on this part: a register-only v_mfma_f32_32x32x16_f16 loop (no memory, no LDS),
// operands all zero vs random, 1..3 waves per SIMD.  Context for every "fraction of the 2.5 PFLOP/s peak" in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, int random) {
    f16x8 a[4], b[4];
    unsigned s = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const float v = random ? ((float)(s >> 8) / 8388608.f - 1.f) : 0.f;
            a[i][j] = (_Float16)v;
            s = s * 1664525u + 1013904223u;
            b[i][j] = (_Float16)(random ? ((float)(s >> 8) / 8388608.f - 1.f) * 0.05f : 0.f);
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int random = 0; random < 2; ++random)
        for (int wps = 1; wps <= 3; ++wps) {            // waves per SIMD: blocks of 4 waves, wps blocks per CU
            const int blocks = 256 * wps;
            mfma_loop<<<blocks, 256>>>(out, 100, random);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            mfma_loop<<<blocks, 256>>>(out, iters, random);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
            printf("%s operands, %d wave(s)/SIMD: %.1f ms -> %.2f PFLOP/s (%.0f%% of 2.5)\n", random ? "random" : "zero  ", wps, ms, fl / ms / 1e12,
                   fl / ms / 1e12 / 2.5 * 100);
        }
    return 0;
}
