// CorrBlock: correlation pyramid + radius-r bilinear window lookup.
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

// ---------------------------------------------------------------------------
// Fast path (4 levels, radius 4): ONE WAVE PER QUERY, software-pipelined.
//   taps    lanes 0..35 = (level, offset) replay the x- and y- tap chains once
//           (72 chains per query instead of 648) and publish floor indices
//           (relative to the staged window) + fractional weights in LDS.
//   stage   the four source windows go to LDS as [11 rows][16 floats]: rows of
//           planes whose width is a multiple of 4 are fetched as aligned
//           16-byte loads (44 lanes x float4 per level), zero padding applied
//           per load; other widths use dword loads.  11 rows/cols because the
//           fp32 round trip can move a floor by one.
//   blend   324 outputs = 4 LDS reads + blend each, stored as coalesced rows.
// Pipeline: the window loads of query n+1 are issued into registers BEFORE the
// blend of query n and land in LDS after it; its coords are prefetched one step
// earlier still.  A block IS one wave, so barriers are wave-local.
// ---------------------------------------------------------------------------
constexpr int WROWS = 11, WCOLS = 16, NWIN = WROWS * WCOLS;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ALLVEC: every level's width is a multiple of 4 (W/8 a multiple of 32): only the 16-byte staging path is compiled.
// No load sits under a branch: out-of-range lanes read a valid dummy address and the zero padding is applied by a
// select when the window goes to LDS.  (With the loads inside `if (in range)` the compiler serialised the four
// levels' loads with s_waitcnt vmcnt(0) between them: four DRAM round trips per query instead of one.)
template <bool ALLVEC>
__global__ __launch_bounds__(64) void lookup_wave_kernel(const LookupArgs a) {
    __shared__ __attribute__((aligned(16))) float win[4 * NWIN];
    __shared__ int tab_i[2][4][2][9];
    __shared__ float tab_w[2][4][2][9];
    __shared__ int org[2][4][2];     // [buf][level][x base (aligned), y base]
    const int lane = threadIdx.x;
    const int h0 = a.h[0], w0 = a.w[0];

    const int t_lv = min(lane / 9, 3), t_o = lane - (lane / 9) * 9;   // tap role
    const float t_inv = 1.f / (float)(1 << t_lv);
    const int t_h = h0 >> t_lv, t_w = w0 >> t_lv;
    const bool t_vec = ALLVEC || (t_w & 3) == 0;
    const int s_r = lane >> 2, s_g = lane & 3;                        // vector staging role (lane < 44)
    const int d_r0 = lane / 11, d_c0 = lane - d_r0 * 11;              // dword staging role
    const int d_r1 = (lane + 64) / 11, d_c1 = (lane + 64) - d_r1 * 11;

    auto publish_taps = [&](long long q, float cx, float cy, int buf) {
        int x0, y0;
        float wx, wy;
        tap_1d(cx, t_inv, t_o - 4, t_w, x0, wx);
        tap_1d(cy, t_inv, t_o - 4, t_h, y0, wy);
        const int fx0 = __shfl(x0, t_lv * 9), oy = __shfl(y0, t_lv * 9);   // taps of offset -4 = window origin
        const int ox = t_vec ? (fx0 & ~3) : fx0;                          // aligned down for 16-byte loads
        if (lane < 36) {
            tab_i[buf][t_lv][0][t_o] = min(x0 - ox, WCOLS - 2);
            tab_i[buf][t_lv][1][t_o] = min(y0 - oy, WROWS - 2);
            tab_w[buf][t_lv][0][t_o] = wx;
            tab_w[buf][t_lv][1][t_o] = wy;
            if (t_o == 0) {
                org[buf][t_lv][0] = ox;
                org[buf][t_lv][1] = oy;
            }
            if (a.taps) {
                int* t = a.taps + (q * 4 + t_lv) * 18;
                t[t_o] = x0;
                t[9 + t_o] = y0;
            }
        }
    };

    f32x4 rv[4];        // vector path: one float4 per level (lanes < 44)
    float rd[4][2];     // dword path: two floats per level
    unsigned okm = 0;   // bit lv (vector) / bits 4+2lv, 5+2lv (dword): the load was inside the plane
    auto issue_loads = [&](long long q, int buf) {
        okm = 0;
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {   // compile-time level: plane pointer and sizes stay scalar
            const int hl = h0 >> lv, wl = w0 >> lv;
            const float* pl = a.lvl[lv] + q * (long long)(hl * wl);
            const int gx0 = org[buf][lv][0], gy0 = org[buf][lv][1];
            if (ALLVEC || (wl & 3) == 0) {
                const int gy = gy0 + s_r, gx = gx0 + 4 * s_g;
                const bool ok = lane < 44 && (unsigned)gy < (unsigned)hl && (unsigned)gx < (unsigned)wl;
                // lanes without a load of their own re-read a line of the window (clamped row / column): no extra traffic
                const int gyc = min(max(gy0 + min(s_r, WROWS - 1), 0), hl - 1), gxc = min(max(gx, 0), wl - 4);
                rv[lv] = *reinterpret_cast<const f32x4*>(pl + gyc * wl + gxc);
                okm |= ok ? 1u << lv : 0u;
            } else {
                int gy = gy0 + d_r0, gx = gx0 + d_c0;
                bool ok = (unsigned)gy < (unsigned)hl && (unsigned)gx < (unsigned)wl;
                rd[lv][0] = pl[ok ? gy * wl + gx : 0];
                okm |= ok ? 1u << (4 + 2 * lv) : 0u;
                gy = gy0 + d_r1, gx = gx0 + d_c1;
                ok = lane < 121 - 64 && (unsigned)gy < (unsigned)hl && (unsigned)gx < (unsigned)wl;
                rd[lv][1] = pl[ok ? gy * wl + gx : 0];
                okm |= ok ? 1u << (5 + 2 * lv) : 0u;
            }
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
            if (ALLVEC || ((w0 >> lv) & 3) == 0) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (lane < 44) *reinterpret_cast<f32x4*>(&win[lv * NWIN + s_r * WCOLS + 4 * s_g]) = (okm >> lv & 1u) ? rv[lv] : z;
            } else {
                win[lv * NWIN + d_r0 * WCOLS + d_c0] = (okm >> (4 + 2 * lv) & 1u) ? rd[lv][0] : 0.f;
                if (lane < 121 - 64) win[lv * NWIN + d_r1 * WCOLS + d_c1] = (okm >> (5 + 2 * lv) & 1u) ? rd[lv][1] : 0.f;
            }
        }
    };

    long long q = blockIdx.x;
    if (q >= a.queries) return;
    int cur = 0;
    publish_taps(q, a.coords[q * 2], a.coords[q * 2 + 1], 0);
    __syncthreads();
    issue_loads(q, 0);
    for (;;) {
        const long long qn = q + gridDim.x;
        const bool has_next = qn < a.queries;
        const long long qs = has_next ? qn : q;          // the last round re-stages its own query: nothing under a branch
        const float cxn = a.coords[qs * 2], cyn = a.coords[qs * 2 + 1];
        store_window();                                  // waits for this query's window loads
        publish_taps(qs, cxn, cyn, cur ^ 1);
        __syncthreads();                                 // win + both table sets visible
        issue_loads(qs, cur ^ 1);                        // in flight during the blend below
        float* orow = a.out + q * a.out_ld;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = lane + 64 * j;
            if (k < 324) {
                const int lv = k / 81, rem = k - lv * 81;
                const int ia = rem / 9, ib = rem - ia * 9;
                const int xi = tab_i[cur][lv][0][ia], yi = tab_i[cur][lv][1][ib];
                const float fx = tab_w[cur][lv][0][ia], fy = tab_w[cur][lv][1][ib];
                const float* p = &win[lv * NWIN + yi * WCOLS + xi];
                const float v00 = p[0], v01 = p[1], v10 = p[WCOLS], v11 = p[WCOLS + 1];
                const float ex = __fsub_rn(1.f, fx), sy = __fsub_rn(1.f, fy);
                float o = __fmul_rn(v00, __fmul_rn(sy, ex));
                o = __fadd_rn(o, __fmul_rn(v01, __fmul_rn(sy, fx)));
                o = __fadd_rn(o, __fmul_rn(v10, __fmul_rn(fy, ex)));
                o = __fadd_rn(o, __fmul_rn(v11, __fmul_rn(fy, fx)));
                orow[k] = o;
            }
        }
        if (!has_next) break;
        __syncthreads();                                 // everyone done reading win before it is overwritten
        q = qn;
        cur ^= 1;
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
    static const int variant = getenv("FF_LOOKUP_GENERIC") ? 0 : 1;   // A/B switch for profiling only
    if (variant == 1 && num_levels == 4 && radius == 4) {
        // one wave per block; 78 VGPRs = 6 waves per SIMD = 24 per CU resident: a grid of 256 x 24 runs as ONE round of
        // waves (4 queries each at B = 8), 256 x 32 as one round plus a third of a second one
        static const int wpc = getenv("FF_LOOKUP_WAVES_PER_CU") ? atoi(getenv("FF_LOOKUP_WAVES_PER_CU")) : 24;
        long long blocks = queries < 256ll * wpc ? queries : 256ll * wpc;
        const bool allvec = ((a.w[0] | a.w[1] | a.w[2] | a.w[3]) & 3) == 0;
        if (allvec) lookup_wave_kernel<true><<<(unsigned)blocks, 64, 0, static_cast<hipStream_t>(stream)>>>(a);
        else lookup_wave_kernel<false><<<(unsigned)blocks, 64, 0, static_cast<hipStream_t>(stream)>>>(a);
        return ff::check_launch("ff_corr_lookup_fwd");
    }
    const long long total = queries * nk;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    lookup_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_corr_lookup_fwd");
}
