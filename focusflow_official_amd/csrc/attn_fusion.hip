// SA / CA fusion units (parallel_fusion.py:14-73): the pieces that are not convolutions.
//
//   SA: q1 = conv3x3(cat[q, v]); v' = conv3x3(v); s = sigmoid(conv3x3([mean_c q1, max_c q1])); out = s * v' + q
//   CA: q1 = conv3x3(cat[q, v]); v' = conv3x3(v); c = mlp(avg_hw q1) + mlp(max_hw q1);          out = c * v' + q
//
// Everything is NHWC fp32.  All reductions are deterministic (fixed slab order, no float atomics).
//   ff_chan_stats_*     per-pixel mean / max over channels (+ arg max for the backward)
//   ff_spatial_stats_*  per-(sample, channel) mean / max over pixels, two stages over FF_SPATIAL_SLABS slabs
//   ff_scale_add_*      out = s * v + q with s per pixel (mode 0) or per (sample, channel) (mode 1)
#include "ff_common.h"

namespace {

constexpr int SLABS = FF_SPATIAL_SLABS;

__global__ __launch_bounds__(256) void chan_stats_fwd_kernel(const float* __restrict__ x, int x_ld, int C, long long npix,
                                                             float* __restrict__ st, int st_ld, int* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long p = wave0; p < npix; p += nwaves) {
        const float* row = x + p * x_ld;
        float sum = 0.f, mx = -INFINITY;
        int idx = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = row[c];
            sum += v;
            if (v > mx) { mx = v; idx = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            const float om = __shfl_xor(mx, o);
            const int oi = __shfl_xor(idx, o);
            if (om > mx || (om == mx && oi < idx)) { mx = om; idx = oi; }     // first maximum wins, as torch.max
        }
        if (lane < st_ld) st[p * st_ld + lane] = lane == 0 ? sum / (float)C : (lane == 1 ? mx : 0.f);
        if (lane == 0) amax[p] = idx;
    }
}

__global__ __launch_bounds__(256) void chan_stats_bwd_kernel(const float* __restrict__ g, int g_ld, const int* __restrict__ amax,
                                                             int C, long long npix, float* __restrict__ gx, int gx_ld) {
    const long long total = npix * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / C;
        const int c = (int)(i - p * C);
        gx[p * gx_ld + c] = g[p * g_ld] / (float)C + (c == amax[p] ? g[p * g_ld + 1] : 0.f);
    }
}

// stage 1: grid (SLABS, B, ceil(C/64)); 256 threads = 4 pixel lanes x 64 channels
__global__ __launch_bounds__(256) void spatial_partial_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ y,
                                                              int y_ld, int C, int HW, float* __restrict__ part, int want_max) {
    __shared__ float s_sum[4][64], s_max[4][64];
    __shared__ int s_idx[4][64];
    const int tc = threadIdx.x & 63, tp = threadIdx.x >> 6;
    const int slab = blockIdx.x, b = blockIdx.y, c = blockIdx.z * 64 + tc, B = gridDim.y;
    const int per = (HW + SLABS - 1) / SLABS, p0 = slab * per, p1 = min(p0 + per, HW);
    float sum = 0.f, mx = -INFINITY;
    int idx = 0x7fffffff;
    if (c < C) {
        for (int p = p0 + tp; p < p1; p += 4) {
            const long long pix = (long long)b * HW + p;
            float v = x[pix * x_ld + c];
            if (y) v *= y[pix * y_ld + c];                 // backward of the scaling: sum of v * gout
            sum += v;
            if (want_max && v > mx) { mx = v; idx = p; }
        }
    }
    s_sum[tp][tc] = sum; s_max[tp][tc] = mx; s_idx[tp][tc] = idx;
    __syncthreads();
    if (tp == 0 && c < C) {
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            sum += s_sum[j][tc];
            if (s_max[j][tc] > mx || (s_max[j][tc] == mx && s_idx[j][tc] < idx)) { mx = s_max[j][tc]; idx = s_idx[j][tc]; }
        }
        float* o = part + (((long long)slab * B + b) * C + c) * 3;
        o[0] = sum; o[1] = mx; o[2] = __int_as_float(idx);
    }
}

// stage 2: one thread per (b, c), slabs in order
__global__ __launch_bounds__(256) void spatial_final_kernel(const float* __restrict__ part, int BC, float scale, float* __restrict__ avg,
                                                            float* __restrict__ mx_out, int* __restrict__ amax, float* __restrict__ dup) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BC) return;
    float sum = 0.f, mx = -INFINITY;
    int idx = 0x7fffffff;
    for (int s = 0; s < SLABS; ++s) {
        const float* o = part + ((long long)s * BC + i) * 3;
        sum += o[0];
        if (o[1] > mx) { mx = o[1]; idx = __float_as_int(o[2]); }          // earlier slab = smaller pixel index wins ties
    }
    avg[i] = sum * scale;
    if (dup) dup[i] = sum * scale;
    if (mx_out) mx_out[i] = mx;
    if (amax) amax[i] = idx;
}

__global__ __launch_bounds__(256) void spatial_stats_bwd_kernel(const float* __restrict__ gavg, const float* __restrict__ gmax,
                                                                const int* __restrict__ amax, int C, int HW, long long total,
                                                                float* __restrict__ gx, int gx_ld) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / C;
        const int c = (int)(i - pix * C);
        const int b = (int)(pix / HW), p = (int)(pix - (long long)b * HW);
        const int bc = b * C + c;
        gx[pix * gx_ld + c] = gavg[bc] / (float)HW + (p == amax[bc] ? gmax[bc] : 0.f);
    }
}

__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const float* __restrict__ v, int v_ld, const float* __restrict__ s, int s_ld,
                                                            const float* __restrict__ s2, const float* __restrict__ q, int q_ld,
                                                            float* __restrict__ out, int out_ld, int C, int HW, long long total, int mode) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / C;
        const int c = (int)(i - pix * C);
        float sv;
        if (mode == 0) sv = s[pix * s_ld];
        else {
            const int bc = (int)(pix / HW) * C + c;
            sv = s[bc] + (s2 ? s2[bc] : 0.f);
        }
        out[pix * out_ld + c] = sv * v[pix * v_ld + c] + (q ? q[pix * q_ld + c] : 0.f);
    }
}

// mode 0 backward: one wave per pixel: gv = s * gout, gs = sum_c v * gout
__global__ __launch_bounds__(256) void scale_bwd_pixel_kernel(const float* __restrict__ gout, int g_ld, const float* __restrict__ v, int v_ld,
                                                              const float* __restrict__ s, int s_ld, float* __restrict__ gv, int gv_ld,
                                                              float* __restrict__ gs, int C, long long npix) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long p = wave0; p < npix; p += nwaves) {
        const float sv = s[p * s_ld];
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float g = gout[p * g_ld + c];
            acc += v[p * v_ld + c] * g;
            gv[p * gv_ld + c] = sv * g;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) gs[p] = acc;
    }
}

inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 * 64 ? 256 * 64 : b));
}

}  // namespace

extern "C" int ff_chan_stats_fwd(const float* x, int x_ld, int C, long long npix, float* st, int st_ld, int* argmax, void* stream) {
    FF_REQUIRE(x && st && argmax, "ff_chan_stats_fwd: null pointer");
    FF_REQUIRE(C >= 1 && x_ld >= C && npix > 0 && st_ld >= 2 && st_ld <= 64, "ff_chan_stats_fwd: C %d ld %d st_ld %d", C, x_ld, st_ld);
    chan_stats_fwd_kernel<<<grid_for(npix * 64), 256, 0, static_cast<hipStream_t>(stream)>>>(x, x_ld, C, npix, st, st_ld, argmax);
    return ff::check_launch("ff_chan_stats_fwd");
}

extern "C" int ff_chan_stats_bwd(const float* g, int g_ld, const int* argmax, int C, long long npix, float* gx, int gx_ld, void* stream) {
    FF_REQUIRE(g && argmax && gx, "ff_chan_stats_bwd: null pointer");
    FF_REQUIRE(C >= 1 && g_ld >= 2 && gx_ld >= C && npix > 0, "ff_chan_stats_bwd: C %d g_ld %d gx_ld %d", C, g_ld, gx_ld);
    chan_stats_bwd_kernel<<<grid_for(npix * C), 256, 0, static_cast<hipStream_t>(stream)>>>(g, g_ld, argmax, C, npix, gx, gx_ld);
    return ff::check_launch("ff_chan_stats_bwd");
}

extern "C" int ff_spatial_stats_fwd(const float* x, int x_ld, int C, int B, int HW, float* avg, float* mx, int* argmax,
                                    float* scratch, void* stream) {
    FF_REQUIRE(x && avg && mx && argmax && scratch, "ff_spatial_stats_fwd: null pointer");
    FF_REQUIRE(C >= 1 && x_ld >= C && B >= 1 && HW >= 1, "ff_spatial_stats_fwd: C %d ld %d B %d HW %d", C, x_ld, B, HW);
    hipStream_t s = static_cast<hipStream_t>(stream);
    spatial_partial_kernel<<<dim3(SLABS, B, (C + 63) / 64), 256, 0, s>>>(x, x_ld, nullptr, 0, C, HW, scratch, 1);
    if (int rc = ff::check_launch("ff_spatial_stats_fwd")) return rc;
    spatial_final_kernel<<<(B * C + 255) / 256, 256, 0, s>>>(scratch, B * C, 1.f / (float)HW, avg, mx, argmax, nullptr);
    return ff::check_launch("ff_spatial_stats_fwd");
}

extern "C" int ff_spatial_stats_bwd(const float* gavg, const float* gmax, const int* argmax, int C, int B, int HW, float* gx,
                                    int gx_ld, void* stream) {
    FF_REQUIRE(gavg && gmax && argmax && gx, "ff_spatial_stats_bwd: null pointer");
    FF_REQUIRE(C >= 1 && gx_ld >= C && B >= 1 && HW >= 1, "ff_spatial_stats_bwd: C %d ld %d", C, gx_ld);
    const long long total = (long long)B * HW * C;
    spatial_stats_bwd_kernel<<<grid_for(total), 256, 0, static_cast<hipStream_t>(stream)>>>(gavg, gmax, argmax, C, HW, total, gx, gx_ld);
    return ff::check_launch("ff_spatial_stats_bwd");
}

extern "C" int ff_scale_add_fwd(const float* v, int v_ld, const float* s, int s_ld, const float* s2, const float* q, int q_ld,
                                float* out, int out_ld, int C, int B, int HW, int mode, void* stream) {
    FF_REQUIRE(v && s && out, "ff_scale_add_fwd: null pointer");
    FF_REQUIRE(C >= 1 && v_ld >= C && out_ld >= C && (!q || q_ld >= C) && (mode == 0 || mode == 1) && (mode == 1 || (s_ld >= 1 && !s2)),
               "ff_scale_add_fwd: C %d mode %d", C, mode);
    const long long total = (long long)B * HW * C;
    scale_add_fwd_kernel<<<grid_for(total), 256, 0, static_cast<hipStream_t>(stream)>>>(v, v_ld, s, s_ld, s2, q, q_ld, out, out_ld, C, HW,
                                                                                          total, mode);
    return ff::check_launch("ff_scale_add_fwd");
}

extern "C" int ff_scale_add_bwd(const float* gout, int g_ld, const float* v, int v_ld, const float* s, int s_ld, const float* s2,
                                float* gv, int gv_ld, float* gs, float* scratch, int C, int B, int HW, int mode, void* stream) {
    FF_REQUIRE(gout && v && s && gv && gs, "ff_scale_add_bwd: null pointer");
    FF_REQUIRE(C >= 1 && g_ld >= C && v_ld >= C && gv_ld >= C && (mode == 0 || (mode == 1 && scratch)), "ff_scale_add_bwd: C %d mode %d", C, mode);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long npix = (long long)B * HW;
    if (mode == 0) {
        scale_bwd_pixel_kernel<<<grid_for(npix * 64), 256, 0, st>>>(gout, g_ld, v, v_ld, s, s_ld, gv, gv_ld, gs, C, npix);
        return ff::check_launch("ff_scale_add_bwd");
    }
    // gv = (s + s2) * gout: the forward kernel with q = null; gs[b][c] = sum_p v * gout, two deterministic stages
    scale_add_fwd_kernel<<<grid_for(npix * C), 256, 0, st>>>(gout, g_ld, s, 0, s2, nullptr, 0, gv, gv_ld, C, HW, npix * C, 1);
    if (int rc = ff::check_launch("ff_scale_add_bwd")) return rc;
    spatial_partial_kernel<<<dim3(SLABS, B, (C + 63) / 64), 256, 0, st>>>(v, v_ld, gout, g_ld, C, HW, scratch, 0);
    if (int rc = ff::check_launch("ff_scale_add_bwd")) return rc;
    spatial_final_kernel<<<(B * C + 255) / 256, 256, 0, st>>>(scratch, B * C, 1.f, gs, nullptr, nullptr, s2 ? gs + B * C : nullptr);
    return ff::check_launch("ff_scale_add_bwd");
}
