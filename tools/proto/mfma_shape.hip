// 32x32x16 vs 16x16x32 f16 MFMA at the SAME wave tile (64 x 32), operands re-read from LDS every k-step, 3-term split
// products, random data: which shape does the chip clock higher / run faster (MI355X_MICROARCH 'DVFS give-back' item 7)?
// Usage: hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void loop_kernel(float* out, int iters, int zero, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 256 * 144 / 2; i += 256) {
        unsigned s = (i + 1) * 2654435761u + blockIdx.x * 40503u;
        s = s * 1664525u + 1013904223u;
        reinterpret_cast<_Float16*>(smem)[i] = zero ? (_Float16)0.f : (_Float16)(((float)(s >> 8) / 8388608.f - 1.f) * 0.5f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    if (SHAPE == 32) {
        const char* base = smem + (lane & 31) * 144 + (lane >> 5) * 16;
        f32x16 acc[2];
        for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {        // one iteration = one 32-k step = two 16-k slices
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const char* pa = base + ((it + wave) & 3) * 32 * 144 + sl * 32;
                const char* pb = base + ((it * 3 + wave) & 3) * 32 * 144 + sl * 32 + 4608 * 2;
                f16x8 a0[2], a1[2], b0, b1;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a0[t] = *reinterpret_cast<const f16x8*>(pa + t * 4608);
                    a1[t] = *reinterpret_cast<const f16x8*>(pa + t * 4608 + 64);
                }
                b0 = *reinterpret_cast<const f16x8*>(pb);
                b1 = *reinterpret_cast<const f16x8*>(pb + 64);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[t], b0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[t], b1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b0, acc[t], 0, 0, 0);
                }
            }
        }
        for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {
        // 16x16x32: lane = row (lane & 15), k-group (lane >> 4) * 8 halfs; a 32-k step of one term is one 64-byte row piece
        const char* base = smem + (lane & 15) * 144 + (lane >> 4) * 16;
        f32x4 acc[4][2];
        for (int t = 0; t < 4; ++t) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc[t][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* pa = base + ((it + wave) & 3) * 32 * 144;
            const char* pb = base + ((it * 3 + wave) & 3) * 32 * 144 + 4608 * 2;
            f16x8 a0[4], a1[4], b0[2], b1[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a0[t] = *reinterpret_cast<const f16x8*>(pa + t * 2304);
                a1[t] = *reinterpret_cast<const f16x8*>(pa + t * 2304 + 64);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b0[j] = *reinterpret_cast<const f16x8*>(pb + j * 2304);
                b1[j] = *reinterpret_cast<const f16x8*>(pb + j * 2304 + 64);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[t], b0[j], acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[t], b1[j], acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[t], b0[j], acc[t][j], 0, 0, 0);
                }
        }
        for (int t = 0; t < 4; ++t) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[t][j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int SHAPE, int OCC>
void run(float* out, int zero, long long* clk) {
    const int lds = 160 * 1024 / OCC - 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<SHAPE, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * OCC, iters = 3000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        loop_kernel<SHAPE, OCC><<<blocks, 256, lds>>>(out, 64, zero, clk);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        loop_kernel<SHAPE, OCC><<<blocks, 256, lds>>>(out, iters, zero, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = (double)blocks * 4 * iters * 6.0 * 2.0 * 32768.0;      // 64 x 32 x 32 x 2 x 3 terms per wave and step
        long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        printf("%s shape %dx%d occ=%d: %.2f ms -> %.2f PFLOP/s issued, shader clock %.2f GHz\n", zero ? "zero  " : "random", SHAPE, SHAPE, OCC, ms, fl / ms / 1e12, ghz);
    }
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 4096 * 4));
    long long* clk; CK(hipMalloc(&clk, 16));
    for (int zero = 0; zero < 2; ++zero) {
        run<32, 4>(out, zero, clk); run<16, 4>(out, zero, clk);
        run<32, 2>(out, zero, clk); run<16, 2>(out, zero, clk);
    }
    return 0;
}
