// Shared helpers for libfocusflow_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "focusflow_hip.h"

namespace ff {

char* err_buf();  // thread-local, defined in api.hip

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define FF_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ff::fail(FF_EINVAL, __VA_ARGS__);   \
    } while (0)

// ff_launch_timing_begin / _end (api.hip): while a class of launches is being timed, its launch site asks for an event pair
// and hands it to hipExtLaunchKernelGGL, which binds both events to the dispatch itself - their distance is the kernel's
// execution time as the command processor stamps it (what rocprofv3 --kernel-trace reports), without the dispatch gaps
// an event pair recorded around the launch adds.  Both stay null when timing is off: an ordinary launch.
void launch_timing_events(int which, hipEvent_t* start, hipEvent_t* stop);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FF_EHIP, "%s: %s", what, hipGetErrorString(e));
    return FF_OK;
}

int conv2d_fwd_split(const FFConvParams& p, int M, int cin, hipStream_t s);  // conv_split.hip
// Split operands (precision f16x3).  An fp32 value v travels as two halfs at ONE power-of-two scale s:
//   h0 = f16(s v),  h1 = f16(s v - h0)            (activations: s = 2^2 applied by the loaders; packed rows: s = 2^4)
// so x w ~ (x0 w0 + x0 w1 + x1 w0) / (sx sw) with all three products on the same scale: ONE fp32 accumulator per
// tile (the earlier format scaled h1 by 2^11 and needed a second accumulator for the two cross terms).  h1 is
// a denormal half when |s v| < 2^-3; v_mfma_f32_32x32x16_f16 keeps denormal inputs (tools/proto/mfma_denorm.hip), so
// the representation error is max(2^-22 |v|, 2^-25 / s): fp32-like for every value that matters in a dot product.
// Limits: |x| < 16376 (activations), |w| < 4094 (packed rows).
constexpr float XSPLIT = 4.f, WSPLIT = 16.f, SPLIT_INV = 1.f / 64.f;
// power-of-two input scale of the split formats (FFConvParams.x_amax): xs puts max|x| at 2^10, xinv undoes it
__device__ __forceinline__ void input_scale(const unsigned int* x_amax, float& xs, float& xinv) {
    xs = 1.f; xinv = 1.f;
    if (x_amax) {
        const int e = (int)((*x_amax >> 23) & 0xffu);          // biased exponent of max|x|
        if (e > 0 && e < 255) {
            int k = 137 - e;                                       // 2^(127+10-e)
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            xs = __uint_as_float((unsigned)(127 + k) << 23);
            xinv = __uint_as_float((unsigned)(127 - k) << 23);
        }
    }
}
// FF_FMT_SPLIT activations (focusflow_hip.h): the pair a convolution loader makes of an fp32 value, written by a producer.
// 4 consecutive channels n4 .. n4 + 3 (n4 % 4 == 0) of one pixel -> two 8-byte stores into the pixel's 128-byte chunk.
typedef _Float16 ff_f16x4 __attribute__((ext_vector_type(4)));
typedef float ff_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_pair4(const ff_f32x4 v, ff_f16x4& h0, ff_f16x4& h1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sv = v[j] * XSPLIT;
        const _Float16 a = (_Float16)sv;
        h0[j] = a;
        h1[j] = (_Float16)(sv - (float)a);
    }
}
// pixel_row = base of the pixel's channels (fp32 addressing: y + pixel * ld); n4 = first of the four channels; nvalid = how
// many of them exist (partial groups at Cout's edge store half by half)
__device__ __forceinline__ void store_split4(float* pixel_row, int n4, const ff_f32x4 v, int nvalid = 4) {
    ff_f16x4 h0, h1;
    split_pair4(v, h0, h1);
    char* c = reinterpret_cast<char*>(pixel_row + (n4 & ~31)) + (n4 & 31) * 2;
    if (nvalid >= 4) {
        *reinterpret_cast<ff_f16x4*>(c) = h0;
        *reinterpret_cast<ff_f16x4*>(c + 64) = h1;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) {
                reinterpret_cast<_Float16*>(c)[j] = h0[j];
                reinterpret_cast<_Float16*>(c + 64)[j] = h1[j];
            }
    }
}
// one channel n of one pixel (kernels whose lanes own single channels: two 2-byte stores)
__device__ __forceinline__ void store_split1(float* pixel_row, int n, float v) {
    const float sv = v * XSPLIT;
    const _Float16 h0 = (_Float16)sv, h1 = (_Float16)(sv - (float)h0);
    char* c = reinterpret_cast<char*>(pixel_row + (n & ~31)) + (n & 31) * 2;
    *reinterpret_cast<_Float16*>(c) = h0;
    *reinterpret_cast<_Float16*>(c + 64) = h1;
}
int conv2d_dma_stats_parts(const FFConvParams& p, int cin);                   // conv_dma.hip (fp32-input route); entries per (image, channel), 0 = not this route
int conv2d_fwd_dma(const FFConvParams& p, int cin, hipStream_t s);           // conv_dma.hip (split-pair inputs by LDS-DMA); 1 = not eligible
int conv2d_wgrad_split(const FFConvParams& p, float* dw, float* db, int M, int cin, hipStream_t s);   // conv_wgrad_split.hip
int conv2d_wgrad_patch(const FFConvParams& p, float* dw, float* db, int cin, hipStream_t s);          // conv_wgrad_patch.hip; 1 = not eligible
int conv2d_fwd_small(const FFConvParams& p, int cin, hipStream_t s);         // conv_small.hip (Cout <= 2, 3x3); 1 = not eligible
int conv2d_fwd_patch(const FFConvParams& p, int cin, hipStream_t s);         // conv_patch.hip; 1 = not eligible
int conv2d_fwd_stem(const FFConvParams& p, int cin, hipStream_t s);          // conv_stem.hip (7x7 stride-2 stems over NHWC4); 1 = not eligible
int conv2d_stem_stats_parts(const FFConvParams& p, int cin);                  // conv_stem.hip; entries per (image, channel), 0 = not this route
int conv2d_stats_parts(const FFConvParams& p, int cin);                       // conv_patch.hip; entries per (image, channel), 0 = cannot
int conv2d_splitk_hint(const FFConvParams& p, int cin);                       // conv_patch.hip; K splits worth using, 0 = none
// corr_lookup_dma.hip: the LDS-DMA lookup; 1 = not eligible (levels more than 4 GB apart)
int lookup_dma_fwd(const void* const* levels, int half, const float* coords, long long queries, int h0, int w0, float* out,
                   int out_ld, int* taps_dbg, hipStream_t s);

// Tuning overrides (tile shapes, blocks per CU, launch caps ...): measurements behind every default are in docs/history.md.  They
// exist in the LAB build only (tools/build_lab.sh: -DFF_LAB); the product library reads none of them, so a stray variable in a
// shell cannot change what a benchmark measures.  The product's own switches - A/B of shipped routes - use getenv directly.
#ifdef FF_LAB
inline const char* tune_env(const char* name) { return getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set once per (call site, device), also from several
// threads (a process-wide `static bool once` left the second device without the opt-in).
#define FF_ALLOW_DYNAMIC_LDS(kernel_, bytes_)                                                                                   \
    do {                                                                                                                         \
        static std::atomic<unsigned long long> ff_done_{0};                                                                      \
        int ff_dev_ = 0;                                                                                                         \
        (void)hipGetDevice(&ff_dev_);                                                                                            \
        const unsigned long long ff_bit_ = 1ull << (ff_dev_ & 63);                                                               \
        if (!(ff_done_.load(std::memory_order_relaxed) & ff_bit_)) {                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes_)); \
            ff_done_.fetch_or(ff_bit_, std::memory_order_relaxed);                                                               \
        }                                                                                                                        \
    } while (0)

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// sigmoid / tanh of the GRU gates (update.py:47-49) on the hardware's exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each):
// absolute error < 2e-7 on values in (-1, 1) - below the 2^-22 of the split format the results are stored in - for 6 / 9
// vector instructions instead of the ~35 / ~50 of expf + IEEE division / tanhf.  (A z|r block's epilogue was a quarter
// of its main loop: 32 sigmoids per thread.)  exp2 overflows to inf for v < -88: rcp(inf) = 0, the right limit.
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504f)); }
__device__ __forceinline__ float fast_tanh(float v) {
    const float t = __builtin_amdgcn_exp2f(fabsf(v) * -2.88539008f);          // exp(-2 |v|) in (0, 1]
    return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), v);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case FF_ACT_RELU: return v < 0.f ? 0.f : v;      // torch.relu: a NaN stays a NaN (v > 0 ? v : 0 would turn it into 0 and hide an overflow of the split formats from the range guard)
#ifdef FF_EXACT_ACT      // lab build: libm's expf / tanhf and an IEEE division (what the fast forms were measured against)
        case FF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case FF_ACT_TANH: return tanhf(v);
#else
        case FF_ACT_SIGMOID: return fast_sigmoid(v);
        case FF_ACT_TANH: return fast_tanh(v);
#endif
        case FF_ACT_LEAKY: return v > 0.f ? v : 0.1f * v;
        default: return v;
    }
}

}  // namespace ff
