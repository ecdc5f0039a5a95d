// What the f16 matrix pipe sustains when its operands come out of LDS: the conv kernels' inner loop without anything
// else (no global loads, no barriers, no stores) - 3-term split products, TM x TN register tile per wave,
// 2 (TM + TN) ds_read_b128 for 3 TM TN MFMAs per 16-k slice, random (dense) data, conflict-free 144-byte rows.
//   PIPE = 0: fragments read and consumed in the same slice (what hipcc makes of the straightforward loop)
//   PIPE = 1: fragments of slice s+1 are read while slice s multiplies (register double buffer, counted lgkmcnt)
// Usage: hipcc -O3 --offload-arch=gfx950 mfma_lds_ratio.hip -o /tmp/mlr && /tmp/mlr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TM, int TN>
struct Frags { f16x8 a0[TM], a1[TM], b0[TN], b1[TN]; };

template <int TM, int TN>
__device__ __forceinline__ void read_frags(Frags<TM, TN>& f, const char* pa, const char* pb) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        f.a0[t] = *reinterpret_cast<const f16x8*>(pa + t * 4608);
        f.a1[t] = *reinterpret_cast<const f16x8*>(pa + t * 4608 + 64);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f.b0[j] = *reinterpret_cast<const f16x8*>(pb + j * 4608);
        f.b1[j] = *reinterpret_cast<const f16x8*>(pb + j * 4608 + 64);
    }
}

template <int TM, int TN>
__device__ __forceinline__ void mul_frags(const Frags<TM, TN>& f, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0[t], f.b0[j], acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a0[t], f.b1[j], acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1[t], f.b0[j], acc[t][j], 0, 0, 0);
        }
}

template <int TM, int TN, int PIPE, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void loop_kernel(float* out, int iters, int zero, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();      // shader clock vs the constant 100 MHz counter
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 36 KB image: 256 rows x 144 B of random halfs
    for (int i = threadIdx.x; i < 256 * 144 / 2; i += 256) {
        unsigned s = (i + 1) * 2654435761u + blockIdx.x * 40503u;
        s = s * 1664525u + 1013904223u;
        reinterpret_cast<_Float16*>(smem)[i] = zero ? (_Float16)0.f : (_Float16)(((float)(s >> 8) / 8388608.f - 1.f) * 0.5f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = smem + (lane & 31) * 144 + (lane >> 5) * 16;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    auto addr_a = [&](int it) { return base + ((it + wave) & 3) * 32 * 144 + ((it >> 2) & 1) * 32; };   // <= rows 0..127 (+ t*4608)
    auto addr_b = [&](int it) { return base + ((it * 3 + wave) & 3) * 32 * 144 + ((it >> 3) & 1) * 32; };
    if (PIPE == 0) {
        for (int it = 0; it < iters; ++it) {
            Frags<TM, TN> f;
            read_frags<TM, TN>(f, addr_a(it), addr_b(it));
            mul_frags<TM, TN>(f, acc);
        }
    } else {
        Frags<TM, TN> f0, f1;
        read_frags<TM, TN>(f0, addr_a(0), addr_b(0));
        constexpr int NR = 2 * (TM + TN), NM = 3 * TM * TN;
        auto interleave = [&]() {
            if (PIPE == 2) {            // one read after each of the first NR MFMAs, the rest of the MFMAs behind them
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM > NR ? NM - NR : 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int it = 0; it < iters; it += 2) {
            read_frags<TM, TN>(f1, addr_a(it + 1), addr_b(it + 1));
            mul_frags<TM, TN>(f0, acc);
            interleave();
            read_frags<TM, TN>(f0, addr_a(it + 2), addr_b(it + 2));
            mul_frags<TM, TN>(f1, acc);
            interleave();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int TM, int TN, int PIPE, int OCC>
void run(float* out, int zero, long long* clk) {
    // TM, TN <= 2: rows up to 128 + 32*... keep the image inside 72 KB: t*4608 with t < 4 -> rows < 256
    const int lds = 160 * 1024 / OCC - 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<TM, TN, PIPE, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * OCC, iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    loop_kernel<TM, TN, PIPE, OCC><<<blocks, 256, lds>>>(out, 64, zero, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    loop_kernel<TM, TN, PIPE, OCC><<<blocks, 256, lds>>>(out, iters, zero, clk);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)blocks * 4 * iters * (3.0 * TM * TN) * 32768.0;
    long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);       // cycles per 10 ns tick
    const double busy = fl / 32768.0 * 32.0 / 1024.0 / ((double)h[0]);     // MFMA-busy share of the shader cycles (32 cycles each, 1024 SIMDs)
    printf("%s TM=%d TN=%d pipe=%d occ=%d: reads/MFMA %.2f  %.2f ms -> %.2f PFLOP/s (%.0f%% of 2.5)  shader clock %.2f GHz, pipe busy %.0f%%\n", zero ? "zero  " : "random", TM, TN, PIPE, OCC,
           2.0 * (TM + TN) / (3.0 * TM * TN), ms, fl / ms / 1e12, fl / ms / 1e12 / 2.5 * 100, ghz, busy * 100);
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 4096 * 4));
    long long* clk; CK(hipMalloc(&clk, 16));
    for (int zero = 0; zero < 2; ++zero) {
#define RUN3(TM_, TN_, OCC_) run<TM_, TN_, 0, OCC_>(out, zero, clk); run<TM_, TN_, 1, OCC_>(out, zero, clk); run<TM_, TN_, 2, OCC_>(out, zero, clk);
        RUN3(2, 1, 4) RUN3(2, 2, 3) RUN3(4, 2, 2)
    }
    return 0;
}
