// InstanceNorm2d / BatchNorm2d on NHWC activations (HBM-bound, 16-byte accesses).
//
//   ff_norm_stats : one pass; a thread owns 4 channels (one float4 per pixel),
//                   a block walks a slab of pixels, partial sums are kept in fp64
//                   and merged with fp64 atomics into stats[s][c] = {sum, sumsq}.
//   ff_norm_apply : y = act((x-mean)*rstd*gamma+beta), optional residual+relu.
#include "ff_common.h"
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SLAB = 256;   // pixels per block (>= 1500 blocks at 192x256xB8: fills 256 CUs)

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, int ld, int HW, int C,
                                                         int per_sample, double* __restrict__ stats) {
    __shared__ double red[256 * 8];
    const int cg = C >> 2;               // float4 groups per pixel
    const int lanes_pix = 256 / cg;      // pixels handled per step (cg divides 256)
    const int t = threadIdx.x;
    const int g = t % cg, pl = t / cg;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * SLAB;
    const int p1 = min(p0 + SLAB, HW);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (pl < lanes_pix) {
        const float* base = x + ((long long)b * HW) * ld + g * 4;
        for (int p = p0 + pl; p < p1; p += lanes_pix) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * ld);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (double)v[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[t * 8 + j] = s[j];
        red[t * 8 + 4 + j] = q[j];
    }
    __syncthreads();
    // threads 0..C-1 each finish one channel
    if (t < C) {
        const int gg = t >> 2, j = t & 3;
        double ss = 0, qq = 0;
        for (int k = 0; k < lanes_pix; ++k) {
            ss += red[(k * cg + gg) * 8 + j];
            qq += red[(k * cg + gg) * 8 + 4 + j];
        }
        double* dst = stats + ((long long)(per_sample ? b : 0) * C + t) * 2;
        atomicAdd(dst, ss);
        atomicAdd(dst + 1, qq);
    }
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, int ld, float* __restrict__ y,
                                                         int y_ld, int HW, int C, const double* __restrict__ stats,
                                                         int per_sample, double inv_count, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, const float* __restrict__ res, int res_ld) {
    __shared__ float s_mul[256], s_add[256];
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    if (t < C) {
        const double* st = stats + ((long long)(per_sample ? b : 0) * C + t) * 2;
        const double mean = st[0] * inv_count;
        double var = st[1] * inv_count - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[t] : 1.f;
        const float bt = beta ? beta[t] : 0.f;
        s_mul[t] = rstd * gm;
        s_add[t] = bt - (float)mean * rstd * gm;
    }
    __syncthreads();
    const int cg = C >> 2;
    const int total = HW * cg;
    const float* xb = x + (long long)b * HW * ld;
    float* yb = y + (long long)b * HW * y_ld;
    const float* rb = res ? res + (long long)b * HW * res_ld : nullptr;
    for (int i = blockIdx.x * 256 + t; i < total; i += gridDim.x * 256) {
        const int g = i % cg;
        const long long p = i / cg;
        f32x4 v = *reinterpret_cast<const f32x4*>(xb + p * ld + g * 4);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (rb) r = *reinterpret_cast<const f32x4*>(rb + p * res_ld + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o = fmaf(v[j], s_mul[g * 4 + j], s_add[g * 4 + j]);       // (one fused operation: the normalise-on-load loaders of conv_patch.hip / conv_dma.hip do the same)
            o = ff::apply_act(o, act);
            if (rb) { o += r[j]; o = o < 0.f ? 0.f : o; }
            v[j] = o;
        }
        *reinterpret_cast<f32x4*>(yb + p * y_ld + g * 4) = v;
    }
}

// the coefficients norm_apply_kernel computes in its prologue, as tables (same arithmetic, so a consumer that applies
// them while loading sees bit-identical normalised values)
__global__ void norm_coeffs_kernel(const double* __restrict__ stats, int total, int C, double inv_count, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                                   float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    const double mean = stats[i * 2] * inv_count;
    double var = stats[i * 2 + 1] * inv_count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    scale[i] = rstd * gm;
    shift[i] = bt - (float)mean * rstd * gm;
}

// Partial statistics left by a convolution's epilogue (FFConvParams.stats_part: {pivot p, s1 = sum(v - p), s2 = sum((v - p)^2),
// count n} per entry) -> added into the {sum, sum of squares} table of ff_norm_stats, in double:
//   sum v = n p + s1,   sum v^2 = s2 + 2 p s1 + n p^2.
// One block per (image, 64 channels, slice of the parts): 4 sub-slices per channel merged through LDS, then one fp64 atomic
// pair per channel and block (as norm_stats_kernel ends).
__global__ __launch_bounds__(256) void norm_stats_finish_kernel(const float* __restrict__ parts, int nparts, int per_block, int C,
                                                               double* __restrict__ stats) {
    __shared__ double sh[2][4][64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int lo = blockIdx.z * per_block, hi = min(nparts, lo + per_block);
    double a1 = 0, a2 = 0;
    if (c < C) {
        const f32x4* e = reinterpret_cast<const f32x4*>(parts) + (long long)b * nparts * C + c;
#pragma unroll 4
        for (int i = lo + sl; i < hi; i += 4) {
            const f32x4 v = e[(long long)i * C];
            const double p = v[0], s1 = v[1], s2 = v[2], n = v[3];
            a1 += n * p + s1;
            a2 += s2 + 2.0 * p * s1 + n * p * p;
        }
    }
    sh[0][sl][threadIdx.x & 63] = a1;
    sh[1][sl][threadIdx.x & 63] = a2;
    __syncthreads();
    if (sl == 0 && c < C) {
        const int l = threadIdx.x;
        const long long i = (long long)b * C + c;
        atomicAdd(stats + i * 2, (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]));
        atomicAdd(stats + i * 2 + 1, (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]));
    }
}

__global__ void bn_fold_kernel(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                               float* sc, float* sh, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f;
    sc[c] = inv * g;
    sh[c] = (beta ? beta[c] : 0.f) - rm[c] * inv * g;
}

__global__ void bn_update_kernel(const double* stats, double count, float momentum, float* rm, float* rv, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = stats[c * 2] / count;
    double var = stats[c * 2 + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
}

}  // namespace

extern "C" int ff_norm_stats(const float* x, int ld, int B, int HW, int C, int per_sample, double* stats, void* stream) {
    FF_REQUIRE(x && stats, "ff_norm_stats: null pointer");
    FF_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 256, "ff_norm_stats: C=%d must be a multiple of 4, <= 256", C);
    FF_REQUIRE(ld >= C && ld % 4 == 0 && ff::aligned16(x), "ff_norm_stats: ld/alignment");
    dim3 grid((HW + SLAB - 1) / SLAB, B);
    norm_stats_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(x, ld, HW, C, per_sample, stats);
    return ff::check_launch("ff_norm_stats");
}

extern "C" int ff_norm_stats_finish(const float* parts, int B, int nparts, int C, double* stats, void* stream) {
    FF_REQUIRE(parts && stats && B > 0 && nparts > 0 && C > 0, "ff_norm_stats_finish: bad arguments");
    FF_REQUIRE(ff::aligned16(parts), "ff_norm_stats_finish: parts must be 16-byte aligned");
    const int slices = std::max(1, std::min(64, nparts / 32));        // >= 8 entries per thread
    const int per_block = (nparts + slices - 1) / slices;
    norm_stats_finish_kernel<<<dim3(B, (C + 63) / 64, (nparts + per_block - 1) / per_block), 256, 0, (hipStream_t)stream>>>(parts, nparts, per_block, C, stats);
    return ff::check_launch("ff_norm_stats_finish");
}

extern "C" int ff_norm_apply(const float* x, int ld, float* y, int y_ld, int B, int HW, int C, const double* stats,
                             int per_sample, float eps, const float* gamma, const float* beta, int act,
                             const float* res, int res_ld, void* stream) {
    FF_REQUIRE(x && y && stats, "ff_norm_apply: null pointer");
    FF_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 256 && (long long)HW * C < (1ll << 32),
               "ff_norm_apply: C=%d unsupported", C);
    FF_REQUIRE(ld >= C && ld % 4 == 0 && y_ld >= C && y_ld % 4 == 0 && ff::aligned16(x) && ff::aligned16(y),
               "ff_norm_apply: ld/alignment");
    FF_REQUIRE(!res || (res_ld >= C && res_ld % 4 == 0 && ff::aligned16(res)), "ff_norm_apply: residual ld/alignment");
    const double inv_count = 1.0 / ((double)HW * (per_sample ? 1 : B));
    const long long total = (long long)HW * (C / 4);
    int gx = (int)((total + 255) / 256);
    if (gx > 1024) gx = 1024;
    dim3 grid(gx, B);
    norm_apply_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(x, ld, y, y_ld, HW, C, stats, per_sample,
                                                                          inv_count, eps, gamma, beta, act, res, res_ld);
    return ff::check_launch("ff_norm_apply");
}

extern "C" int ff_norm_coeffs(const double* stats, int S, int C, long long count, float eps, const float* gamma,
                              const float* beta, float* scale, float* shift, void* stream) {
    FF_REQUIRE(stats && scale && shift && S > 0 && C > 0 && count > 0, "ff_norm_coeffs: bad argument");
    const int total = S * C;
    norm_coeffs_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(stats, total, C, 1.0 / (double)count, eps,
                                                                                          gamma, beta, scale, shift);
    return ff::check_launch("ff_norm_coeffs");
}

extern "C" int ff_bn_fold(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                          float* sc, float* sh, int C, void* stream) {
    FF_REQUIRE(rm && rv && sc && sh && C > 0, "ff_bn_fold: bad argument");
    bn_fold_kernel<<<(C + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(rm, rv, gamma, beta, eps, sc, sh, C);
    return ff::check_launch("ff_bn_fold");
}

extern "C" int ff_bn_update_running(const double* stats, long long count, float momentum, float* rm, float* rv, int C,
                                    void* stream) {
    FF_REQUIRE(stats && rm && rv && C > 0 && count > 0, "ff_bn_update_running: bad argument");
    bn_update_kernel<<<(C + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(stats, (double)count, momentum, rm, rv, C);
    return ff::check_launch("ff_bn_update_running");
}
