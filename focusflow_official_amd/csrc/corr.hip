// CorrBlock on ROW-MAJOR planes: pooling pass + generic radius-r bilinear window lookup (one thread per output).
// The product path is the tiled pyramid (corr_build.hip, corr_lookup_tiled.hip); these entry points serve the exact-fp32
// and plain-f16 conv precisions (volume by ff_conv2d_fwd, then ff_corr_retile) and the layout-equivalence tests.
//
// The lookup replays, op for op in separately rounded fp32 (no FMA contraction),
// the coordinate arithmetic of the reference: corr.py:41-43 (c = coord/2^i + d),
// utils.py:61-62 (g = 2*c/(n-1) - 1) and ATen's align_corners un-normalise
// (u = ((g+1)/2)*(n-1)), so that floor(u) — i.e. which taps are read — is
// bit-identical to the reference even where the round trip moves an integer
// coordinate across a pixel boundary.
#pragma clang fp contract(off)
#include <cstdlib>
#include "ff_common.h"

namespace {

__global__ __launch_bounds__(256) void pyramid_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                      float* __restrict__ l2, float* __restrict__ l3, int h0, int w0) {
    extern __shared__ float sm[];
    const int h1 = h0 >> 1, w1 = w0 >> 1, h2 = h1 >> 1, w2 = w1 >> 1, h3 = h2 >> 1, w3 = w2 >> 1;
    float* s1 = sm;
    float* s2 = sm + h1 * w1;
    const long long plane = blockIdx.x;
    const float* src = l0 + plane * h0 * w0;
    // ATen avg_pool2d: sum the window row-major, then divide by 4 (x0.25 is the same rounding)
    for (int i = threadIdx.x; i < h1 * w1; i += 256) {
        const int y = i / w1, x = i - y * w1;
        const float* p = src + (2 * y) * w0 + 2 * x;
        float2 a, b;
        if (w0 & 1) {   // odd plane width (W/8 odd): rows are not 8-byte aligned, the last column is dropped (floor)
            a = make_float2(p[0], p[1]);
            b = make_float2(p[w0], p[w0 + 1]);
        } else {
            a = *reinterpret_cast<const float2*>(p);
            b = *reinterpret_cast<const float2*>(p + w0);
        }
        const float v = (((a.x + a.y) + b.x) + b.y) * 0.25f;
        s1[i] = v;
        l1[plane * h1 * w1 + i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h2 * w2; i += 256) {
        const int y = i / w2, x = i - y * w2;
        const float* p = s1 + (2 * y) * w1 + 2 * x;
        const float v = (((p[0] + p[1]) + p[w1]) + p[w1 + 1]) * 0.25f;
        s2[i] = v;
        l2[plane * h2 * w2 + i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h3 * w3; i += 256) {
        const int y = i / w3, x = i - y * w3;
        const float* p = s2 + (2 * y) * w2 + 2 * x;
        l3[plane * h3 * w3 + i] = (((p[0] + p[1]) + p[w2]) + p[w2 + 1]) * 0.25f;
    }
}

struct LookupArgs {
    const float* lvl[4];
    int h[4], w[4];
    const float* coords;
    float* out;
    int* taps;
    long long queries;
    int out_ld, radius, num_levels;
};

// One separately-rounded replay of the sampler's coordinate chain.
__device__ __forceinline__ void tap_1d(float c, float inv_scale, int off, int n, int& i0, float& w1) {
    const float cl = __fmul_rn(c, inv_scale);                 // coords / 2**i   (exact: power of two)
    const float x = __fadd_rn(cl, (float)off);                // + delta
    const float nm1 = (float)(n - 1);
    const float g = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, x), nm1), 1.f);   // 2*x/(n-1) - 1
    const float u = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.f), 0.5f), nm1);  // ((g+1)/2)*(n-1)
    const float f = floorf(u);
    i0 = (int)f;
    w1 = __fsub_rn(u, f);
}

__global__ __launch_bounds__(256) void lookup_kernel(const LookupArgs a) {
    const int win = 2 * a.radius + 1, per_lvl = win * win, nk = a.num_levels * per_lvl;
    const long long total = a.queries * nk;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long q = idx / nk;
        const int k = (int)(idx - q * nk);
        const int lv = k / per_lvl, rem = k - lv * per_lvl;
        const int ia = rem / win, ib = rem - ia * win;  // ia: x-offset index (slow), ib: y-offset index (fast)
        const float cx = a.coords[q * 2], cy = a.coords[q * 2 + 1];
        const float inv = 1.f / (float)(1 << lv);
        const int hl = a.h[lv], wl = a.w[lv];
        int x0, y0;
        float wx, wy;
        tap_1d(cx, inv, ia - a.radius, wl, x0, wx);
        tap_1d(cy, inv, ib - a.radius, hl, y0, wy);
        if (a.taps) {
            int* t = a.taps + (q * a.num_levels + lv) * 2 * win;
            if (ib == 0) t[ia] = x0;
            if (ia == 0) t[win + ib] = y0;
        }
        const float* pl = a.lvl[lv] + q * hl * wl;
        const bool xin0 = (unsigned)x0 < (unsigned)wl, xin1 = (unsigned)(x0 + 1) < (unsigned)wl;
        const bool yin0 = (unsigned)y0 < (unsigned)hl, yin1 = (unsigned)(y0 + 1) < (unsigned)hl;
        const float v00 = (xin0 && yin0) ? pl[y0 * wl + x0] : 0.f;
        const float v01 = (xin1 && yin0) ? pl[y0 * wl + x0 + 1] : 0.f;
        const float v10 = (xin0 && yin1) ? pl[(y0 + 1) * wl + x0] : 0.f;
        const float v11 = (xin1 && yin1) ? pl[(y0 + 1) * wl + x0 + 1] : 0.f;
        const float ex = __fsub_rn(1.f, wx), sy = __fsub_rn(1.f, wy);
        // nw*s*e + ne*s*w + sw*n*e + se*n*w  (ATen's weight naming)
        float o = __fmul_rn(v00, __fmul_rn(sy, ex));
        o = __fadd_rn(o, __fmul_rn(v01, __fmul_rn(sy, wx)));
        o = __fadd_rn(o, __fmul_rn(v10, __fmul_rn(wy, ex)));
        o = __fadd_rn(o, __fmul_rn(v11, __fmul_rn(wy, wx)));
        a.out[q * a.out_ld + k] = o;
    }
}

}  // namespace

extern "C" int ff_corr_pyramid(const float* l0, float* l1, float* l2, float* l3, long long planes, int h0, int w0,
                               void* stream) {
    FF_REQUIRE(l0 && l1 && l2 && l3, "ff_corr_pyramid: null pointer");
    FF_REQUIRE(planes > 0 && planes < (1ll << 31) && h0 >= 8 && w0 >= 8, "ff_corr_pyramid: plane %dx%d too small (need >= 8x8)", h0, w0);
    FF_REQUIRE(((uintptr_t)l0 & 7) == 0, "ff_corr_pyramid: level 0 must be 8-byte aligned");
    const int h1 = h0 / 2, w1 = w0 / 2, h2 = h1 / 2, w2 = w1 / 2;
    const size_t lds = (size_t)(h1 * w1 + h2 * w2) * sizeof(float);
    FF_REQUIRE(lds <= 64 * 1024, "ff_corr_pyramid: plane too large for LDS staging");
    pyramid_kernel<<<(unsigned)planes, 256, lds, static_cast<hipStream_t>(stream)>>>(l0, l1, l2, l3, h0, w0);
    return ff::check_launch("ff_corr_pyramid");
}

extern "C" int ff_corr_lookup_fwd(const float* const* levels, int num_levels, int radius, const float* coords,
                                  long long queries, int h0, int w0, float* out, int out_ld, int* taps_dbg,
                                  void* stream) {
    FF_REQUIRE(levels && coords && out, "ff_corr_lookup_fwd: null pointer");
    FF_REQUIRE(num_levels >= 1 && num_levels <= 4 && radius >= 1 && radius <= 8, "ff_corr_lookup_fwd: levels/radius");
    const int nk = num_levels * (2 * radius + 1) * (2 * radius + 1);
    FF_REQUIRE(queries > 0 && out_ld >= nk, "ff_corr_lookup_fwd: out_ld %d < %d", out_ld, nk);
    LookupArgs a;
    int h = h0, w = w0;
    for (int i = 0; i < 4; ++i) {
        a.lvl[i] = i < num_levels ? levels[i] : nullptr;
        a.h[i] = h;
        a.w[i] = w;
        if (i < num_levels) {
            FF_REQUIRE(levels[i] != nullptr, "ff_corr_lookup_fwd: level %d null", i);
            FF_REQUIRE(h >= 2 && w >= 2, "ff_corr_lookup_fwd: level %d is %dx%d; the sampler divides by (n-1)", i, h, w);
        }
        h /= 2;
        w /= 2;
    }
    a.coords = coords;
    a.out = out;
    a.taps = taps_dbg;
    a.queries = queries;
    a.out_ld = out_ld;
    a.radius = radius;
    a.num_levels = num_levels;
    const long long total = queries * nk;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    lookup_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_corr_lookup_fwd");
}
