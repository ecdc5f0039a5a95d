// Backward kernels of the FF-RAFT hot path other than the convolution GEMMs
// (SURVEY §3.3): activation masks, Instance/BatchNorm backward, GRU gates,
// convex-upsampling backward.  All HBM-bound.  (Lookup / pooling backward: corr_lookup_tiled.hip.)
#include "ff_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_grad_from_output(float y, int act) {
    switch (act) {
        case FF_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case FF_ACT_SIGMOID: return y * (1.f - y);
        case FF_ACT_TANH: return 1.f - y * y;
        case FF_ACT_LEAKY: return y > 0.f ? 1.f : 0.1f;
        default: return 1.f;
    }
}

// g[c] = dy[c] * act'(y[c]) * scale for c < C, 0 for C <= c < Cpad
// amax (nullable, caller zeroes): bits of max|g| - the power-of-two scale of the f16 dgrad (FFConvParams.x_amax)
__global__ void act_bwd_kernel(const float* dy, int dy_ld, const float* __restrict__ y, int y_ld,
                               float* g, int g_ld, long long npix, int C, int Cpad, int act, float scale,
                               unsigned int* __restrict__ amax) {
    const long long total = npix * Cpad;
    float mx = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / Cpad;
        const int c = (int)(i - p * Cpad);
        float v = 0.f;
        if (c < C) {
            v = dy[p * dy_ld + c] * scale;
            if (act != FF_ACT_NONE) v *= act_grad_from_output(y[p * y_ld + c], act);
        }
        if (g != dy) g[p * g_ld + c] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (amax) {     // non-negative floats order like their bit patterns; max is order-independent (deterministic).
        __shared__ float wmax[4];   // ONE atomic per block: same-address atomics serialise at the memory side
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            // the running maximum only grows: a (possibly stale) plain read skips almost every atomic - same-address
            // atomics serialise at ~10 ns each, 4096 of them would cost more than the kernel itself
            if (mx > 0.f && mx < INFINITY && __float_as_uint(mx) > *reinterpret_cast<volatile unsigned int*>(amax))
                atomicMax(amax, __float_as_uint(mx));
        }
    }
}

// The same on groups of 4 channels (C, Cpad and every ld multiples of 4, 16-byte aligned pointers): 16-byte accesses
// and 32-bit index arithmetic instead of one float and a 64-bit division per thread and step.
template <bool HAS_ACT>      // compile-time, and no load under a branch: both loads of a step are in flight together
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(const float* dy, int dy_ld, const float* __restrict__ y, int y_ld,
                                                           float* g, int g_ld, unsigned npix, int C, int Cpad, int act,
                                                           float scale, unsigned int* __restrict__ amax) {
    const unsigned cg = (unsigned)Cpad >> 2, cgin = (unsigned)C >> 2;
    const unsigned total = npix * cg;
    float mx = 0.f;
    // four items per trip, all eight loads first: one item per trip left a thread with two loads in flight and the kernel
    // at 2 TB/s (35 us for the 70 MB of a 22 816 x 256 gradient)
    const unsigned stride = gridDim.x * 256u;
    for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; i0 < total; i0 += 4u * stride) {
        f32x4 v[4], yv[4];
        size_t off_g[4];
        bool live[4], pad[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * stride;
            live[u] = i < total;
            const unsigned ic = live[u] ? i : total - 1;             // (clamped: unconditional loads)
            const unsigned p = ic / cg, c4 = ic - p * cg;
            const unsigned cc = min(c4, cgin - 1);                    // padding groups re-read the last real one and store zeros
            pad[u] = c4 >= cgin;
            v[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)p * dy_ld + cc * 4);
            if (HAS_ACT) yv[u] = *reinterpret_cast<const f32x4*>(y + (size_t)p * y_ld + cc * 4);
            off_g[u] = (size_t)p * g_ld + c4 * 4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 w = v[u] * scale;
            if (HAS_ACT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] *= act_grad_from_output(yv[u][j], act);
            }
            if (pad[u]) w = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (live[u]) {
                if (g != dy) *reinterpret_cast<f32x4*>(g + off_g[u]) = w;     // g == dy: only max|dy| is wanted
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
            }
        }
    }
    if (amax) {
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (mx > 0.f && mx < INFINITY && __float_as_uint(mx) > *reinterpret_cast<volatile unsigned int*>(amax))
                atomicMax(amax, __float_as_uint(mx));
        }
    }
}

// zero-dilation by 2 (stride-2 dgrad): dst[b][2y][2x] = src[b][y][x], zeros elsewhere
__global__ void dilate2_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int B, int Ho,
                               int Wo, int Hd, int Wd, int C) {
    const int cg = C >> 2;
    const long long total = (long long)B * Hd * Wd * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % cg);
        long long t = i / cg;
        const int x = (int)(t % Wd); t /= Wd;
        const int y = (int)(t % Hd);
        const long long b = t / Hd;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(x & 1) && !(y & 1) && (x >> 1) < Wo && (y >> 1) < Ho)
            v = *reinterpret_cast<const f32x4*>(src + ((b * Ho + (y >> 1)) * Wo + (x >> 1)) * src_ld + g * 4);
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

// ---------------------------------------------------------------------------
// Norm backward.  xhat = (x-mean)*rstd ; y0 = gamma*xhat+beta ; y1 = relu?(y0) ; y = relu(y1+res)?
//   g = dy * [y>0 if res] * [y0>0 if relu]
//   pass 1: S1 = sum g, S2 = sum g*xhat   (fp64, per (sample|batch, channel))
//   pass 2: dx = rstd*gamma*(g - S1/N - xhat*S2/N)   (fixed_stats: dx = rstd*gamma*g)
//           dres = dy*[y>0]
// ---------------------------------------------------------------------------
struct NormBwdArgs {
    const float* x; int x_ld;
    const float* dy; int dy_ld;
    const float* y; int y_ld;          // forward output (needed only when has_res)
    const double* fstats;              // forward {sum, sumsq}
    const float* gamma; const float* beta;
    double* bstats;                    // {S1, S2}
    float* dx; int dx_ld;
    float* dres; int dres_ld;
    int HW, C, per_sample, relu, has_res, fixed_stats;
    double inv_count;
    float eps;
    unsigned int* dx_amax;             // nullable: bits of max|dx| (atomicMax; the consumer conv's gradient scale)
};

__device__ __forceinline__ void norm_coeffs(const NormBwdArgs& a, int b, int c, float& mean, float& rstd, float& gm, float& bt) {
    const double* st = a.fstats + ((long long)(a.per_sample ? b : 0) * a.C + c) * 2;
    const double m = st[0] * a.inv_count;
    double var = st[1] * a.inv_count - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    gm = a.gamma ? a.gamma[c] : 1.f;
    bt = a.beta ? a.beta[c] : 0.f;
}

constexpr int NSLAB = 256;

__global__ __launch_bounds__(256) void norm_bwd_stats_kernel(const NormBwdArgs a) {
    __shared__ double red[256 * 8];
    __shared__ float s_mean[256], s_rstd[256], s_g[256], s_b[256];
    const int b = blockIdx.y, t = threadIdx.x;
    if (t < a.C) norm_coeffs(a, b, t, s_mean[t], s_rstd[t], s_g[t], s_b[t]);
    __syncthreads();
    const int cg = a.C >> 2, lanes_pix = 256 / cg;
    const int g4 = t % cg, pl = t / cg;
    const int p0 = blockIdx.x * NSLAB, p1 = min(p0 + NSLAB, a.HW);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < lanes_pix) {
        for (int p = p0 + pl; p < p1; p += lanes_pix) {
            const long long pix = (long long)b * a.HW + p;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + pix * a.x_ld + g4 * 4);
            f32x4 gv = *reinterpret_cast<const f32x4*>(a.dy + pix * a.dy_ld + g4 * 4);
            f32x4 yv = {1.f, 1.f, 1.f, 1.f};
            if (a.has_res) yv = *reinterpret_cast<const f32x4*>(a.y + pix * a.y_ld + g4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = g4 * 4 + j;
                const float xhat = (xv[j] - s_mean[c]) * s_rstd[c];
                float g = gv[j];
                if (a.has_res && !(yv[j] > 0.f)) g = 0.f;
                if (a.relu && !(xhat * s_g[c] + s_b[c] > 0.f)) g = 0.f;
                s1[j] += (double)g;
                s2[j] += (double)g * (double)xhat;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[t * 8 + j] = s1[j];
        red[t * 8 + 4 + j] = s2[j];
    }
    __syncthreads();
    if (t < a.C) {
        const int gg = t >> 2, j = t & 3;
        double ss = 0, qq = 0;
        for (int k = 0; k < lanes_pix; ++k) {
            ss += red[(k * cg + gg) * 8 + j];
            qq += red[(k * cg + gg) * 8 + 4 + j];
        }
        double* dst = a.bstats + ((long long)(a.per_sample ? b : 0) * a.C + t) * 2;
        atomicAdd(dst, ss);
        atomicAdd(dst + 1, qq);
    }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const NormBwdArgs a) {
    __shared__ float s_mean[256], s_rstd[256], s_g[256], s_b[256], s_m1[256], s_m2[256];
    const int b = blockIdx.y, t = threadIdx.x;
    if (t < a.C) {
        norm_coeffs(a, b, t, s_mean[t], s_rstd[t], s_g[t], s_b[t]);
        const double* bs = a.bstats + ((long long)(a.per_sample ? b : 0) * a.C + t) * 2;
        s_m1[t] = a.fixed_stats ? 0.f : (float)(bs[0] * a.inv_count);
        s_m2[t] = a.fixed_stats ? 0.f : (float)(bs[1] * a.inv_count);
    }
    __syncthreads();
    const int cg = a.C >> 2;
    const int total = a.HW * cg;
    float mx = 0.f;
    for (int i = blockIdx.x * 256 + t; i < total; i += gridDim.x * 256) {
        const int g4 = i % cg;
        const long long pix = (long long)b * a.HW + i / cg;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + pix * a.x_ld + g4 * 4);
        f32x4 gv = *reinterpret_cast<const f32x4*>(a.dy + pix * a.dy_ld + g4 * 4);
        f32x4 yv = {1.f, 1.f, 1.f, 1.f};
        if (a.has_res) yv = *reinterpret_cast<const f32x4*>(a.y + pix * a.y_ld + g4 * 4);
        f32x4 dxv, drv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = g4 * 4 + j;
            const float xhat = (xv[j] - s_mean[c]) * s_rstd[c];
            float g = gv[j];
            if (a.has_res && !(yv[j] > 0.f)) g = 0.f;
            drv[j] = g;
            if (a.relu && !(xhat * s_g[c] + s_b[c] > 0.f)) g = 0.f;
            dxv[j] = s_rstd[c] * s_g[c] * (g - s_m1[c] - xhat * s_m2[c]);
        }
        *reinterpret_cast<f32x4*>(a.dx + pix * a.dx_ld + g4 * 4) = dxv;
        if (a.dres) *reinterpret_cast<f32x4*>(a.dres + pix * a.dres_ld + g4 * 4) = drv;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(dxv[0]), fabsf(dxv[1]))), fmaxf(fabsf(dxv[2]), fabsf(dxv[3])));
    }
    if (a.dx_amax) {                   // what ff_act_bwd would measure in a pass of its own over dx
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((t & 63) == 0) wmax[t >> 6] = mx;
        __syncthreads();
        if (t == 0) {
            mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (mx > 0.f && mx < INFINITY && __float_as_uint(mx) > *reinterpret_cast<volatile unsigned int*>(a.dx_amax))
                atomicMax(a.dx_amax, __float_as_uint(mx));
        }
    }
}

// ---------------------------------------------------------------------------
__global__ void gru_rh_bwd_kernel(const float* __restrict__ drh, int drh_ld, const float* __restrict__ r, int r_ld,
                                  const float* __restrict__ h, int h_ld, float* __restrict__ dr, int dr_ld,
                                  float* __restrict__ dh, int dh_ld, long long npix, int C) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int g = (int)(i - p * cg) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(drh + p * drh_ld + g);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(r + p * r_ld + g);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + p * h_ld + g);
        *reinterpret_cast<f32x4*>(dr + p * dr_ld + g) = d * hv;
        *reinterpret_cast<f32x4*>(dh + p * dh_ld + g) = d * rv;
    }
}

// h' = (1-z)*h + z*q : dz = dh'*(q-h), dq = dh'*z, dh = dh'*(1-z)
__global__ void gru_blend_bwd_kernel(const float* __restrict__ dhn, int dhn_ld, const float* __restrict__ z, int z_ld,
                                     const float* __restrict__ q, int q_ld, const float* __restrict__ h, int h_ld,
                                     float* __restrict__ dz, int dz_ld, float* __restrict__ dq, int dq_ld,
                                     float* __restrict__ dh, int dh_ld, long long npix, int C) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int g = (int)(i - p * cg) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dhn + p * dhn_ld + g);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + p * z_ld + g);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + p * q_ld + g);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + p * h_ld + g);
        *reinterpret_cast<f32x4*>(dz + p * dz_ld + g) = d * (qv - hv);
        *reinterpret_cast<f32x4*>(dq + p * dq_ld + g) = d * zv;
        *reinterpret_cast<f32x4*>(dh + p * dh_ld + g) = d * (1.f - zv);
    }
}

// Convex upsampling backward.  One block per coarse row (b,h) and half of its pixels; a wave is one coarse pixel (its
// 64 lanes = the 8x8 sub-pixels), so the flow gradient of each of the 9 taps is first summed over the wave and then
// added to the row's LDS accumulator by ONE lane (it was 64 same-address LDS atomics per tap and component: the
// kernel ran at 0.25 TB/s); rows h-1..h+1 go to dflow with global atomics at the end.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ flow,
                                                           int flow_ld, const float* __restrict__ mask, int mask_ld,
                                                           float* __restrict__ dflow, int dflow_ld, float* __restrict__ dmask,
                                                           float mask_scale, unsigned int* __restrict__ dmask_amax, int H,
                                                           int W) {
    extern __shared__ float acc[];   // [3][W][2]
    const int b = blockIdx.y, h = blockIdx.x;
    const int wbeg = (int)((long long)W * blockIdx.z / gridDim.z), wend = (int)((long long)W * (blockIdx.z + 1) / gridDim.z);
    for (int i = threadIdx.x; i < 3 * W * 2; i += 256) acc[i] = 0.f;
    __syncthreads();
    const long long rowpix = ((long long)b * H + h) * W;
    const int HW8 = 64 * H * W;
    const int ij = threadIdx.x & 63, i = ij >> 3, j = ij & 7;
    float amx = 0.f;
    for (int w = wbeg + (threadIdx.x >> 6); w < wend; w += 4) {
        const float* m = mask + (rowpix + w) * mask_ld + ij;
        float pk[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            pk[k] = m[k * 64];
            mx = fmaxf(mx, pk[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            pk[k] = expf(pk[k] - mx);
            den += pk[k];
        }
        const long long o = (long long)b * 2 * HW8 + (long long)(8 * h + i) * (8 * W) + 8 * w + j;
        const float gx = dout[o], gy = dout[o + HW8];
        float s[9], dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            pk[k] /= den;
            const int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
            float fx = 0.f, fy = 0.f;
            const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;      // wave-uniform
            if (in) {
                const float* f = flow + (((long long)b * H + yy) * W + xx) * flow_ld;
                fx = 8.f * f[0];
                fy = 8.f * f[1];
                float ax = 8.f * pk[k] * gx, ay = 8.f * pk[k] * gy;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    ax += __shfl_xor(ax, d);
                    ay += __shfl_xor(ay, d);
                }
                if (ij == 0) {
                    atomicAdd(&acc[((k / 3) * W + xx) * 2], ax);
                    atomicAdd(&acc[((k / 3) * W + xx) * 2 + 1], ay);
                }
            }
            s[k] = gx * fx + gy * fy;
            dot += pk[k] * s[k];
        }
        float* dm = dmask + (rowpix + w) * 576 + ij;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float gm = pk[k] * (s[k] - dot) * mask_scale;       // mask_scale: the ".25 *" of update.py:133 on the way
            dm[k * 64] = gm;
            amx = fmaxf(amx, fabsf(gm));
        }
    }
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < 3 * W * 2; i2 += 256) {
        const int rr = i2 / (W * 2), rem = i2 - rr * W * 2;
        const int yy = h + rr - 1;
        if ((unsigned)yy < (unsigned)H && acc[i2] != 0.f)
            atomicAdd(dflow + (((long long)b * H + yy) * W + (rem >> 1)) * dflow_ld + (rem & 1), acc[i2]);
    }
    if (dmask_amax) {                  // bits of max|d mask| for the convolution gradients that read it (FFConvParams.x_amax)
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amx;
        __syncthreads();
        if (threadIdx.x == 0) {
            amx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (amx > 0.f && amx < INFINITY && __float_as_uint(amx) > *reinterpret_cast<volatile unsigned int*>(dmask_amax))
                atomicMax(dmask_amax, __float_as_uint(amx));
        }
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ff_act_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float* g, int g_ld, long long npix,
                          int C, int Cpad, int act, float scale, unsigned int* amax, void* stream) {
    FF_REQUIRE(dy && g && npix > 0 && C > 0 && Cpad >= C && g_ld >= Cpad && dy_ld >= C, "ff_act_bwd: bad argument");
    FF_REQUIRE(act == FF_ACT_NONE || (y && y_ld >= C), "ff_act_bwd: activation needs the forward output");
    const bool vec = C % 4 == 0 && Cpad % 4 == 0 && dy_ld % 4 == 0 && g_ld % 4 == 0 && ff::aligned16(dy) && ff::aligned16(g) &&
                     (act == FF_ACT_NONE || (y_ld % 4 == 0 && ff::aligned16(y))) && npix * (Cpad / 4) < (1ll << 31);
    // (four items per thread and trip; at most FF_ACT_BWD_BLOCKS blocks: every block ends in an atomicMax on ONE word)
    static const int cap = ff::tune_env("FF_ACT_BWD_BLOCKS") ? atoi(ff::tune_env("FF_ACT_BWD_BLOCKS")) : 512;
    const int gv = (int)std::min<long long>(std::max<long long>((npix * (Cpad / 4) + 1023) / 1024, 1), cap);
    if (vec && act == FF_ACT_NONE)
        act_bwd_vec_kernel<false><<<gv, 256, 0, static_cast<hipStream_t>(stream)>>>(dy, dy_ld, y, y_ld, g, g_ld, (unsigned)npix, C, Cpad, act, scale, amax);
    else if (vec)
        act_bwd_vec_kernel<true><<<gv, 256, 0, static_cast<hipStream_t>(stream)>>>(dy, dy_ld, y, y_ld, g, g_ld, (unsigned)npix, C, Cpad, act, scale, amax);
    else
        act_bwd_kernel<<<grid_for(npix * Cpad), 256, 0, static_cast<hipStream_t>(stream)>>>(dy, dy_ld, y, y_ld, g, g_ld, npix, C, Cpad, act, scale, amax);
    return ff::check_launch("ff_act_bwd");
}

extern "C" int ff_dilate2(const float* src, int src_ld, float* dst, int B, int Ho, int Wo, int Hd, int Wd, int C,
                          void* stream) {
    FF_REQUIRE(src && dst && B > 0 && Ho > 0 && Wo > 0 && Hd >= 2 * Ho - 1 && Wd >= 2 * Wo - 1 && C % 4 == 0 && src_ld % 4 == 0 &&
               ff::aligned16(src) && ff::aligned16(dst), "ff_dilate2: bad argument");
    dilate2_kernel<<<grid_for((long long)B * Hd * Wd * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(src, src_ld, dst, B, Ho, Wo, Hd, Wd, C);
    return ff::check_launch("ff_dilate2");
}

extern "C" int ff_norm_bwd(const float* x, int x_ld, const float* dy, int dy_ld, const float* y, int y_ld,
                           const double* fstats, double* bstats, int per_sample, int fixed_stats, float eps,
                           const float* gamma, const float* beta, int relu, float* dx, int dx_ld, float* dres,
                           int dres_ld, int B, int HW, int C, unsigned int* dx_amax, void* stream) {
    FF_REQUIRE(x && dy && fstats && bstats && dx, "ff_norm_bwd: null pointer");
    FF_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 256, "ff_norm_bwd: C=%d unsupported", C);
    FF_REQUIRE(x_ld % 4 == 0 && dy_ld % 4 == 0 && dx_ld % 4 == 0 && ff::aligned16(x) && ff::aligned16(dy) && ff::aligned16(dx), "ff_norm_bwd: alignment");
    FF_REQUIRE(!dres || (y && y_ld % 4 == 0 && dres_ld % 4 == 0 && ff::aligned16(y) && ff::aligned16(dres)), "ff_norm_bwd: residual needs y and aligned dres");
    NormBwdArgs a;
    a.x = x; a.x_ld = x_ld; a.dy = dy; a.dy_ld = dy_ld; a.y = y; a.y_ld = y_ld;
    a.fstats = fstats; a.gamma = gamma; a.beta = beta; a.bstats = bstats;
    a.dx = dx; a.dx_ld = dx_ld; a.dres = dres; a.dres_ld = dres_ld;
    a.HW = HW; a.C = C; a.per_sample = per_sample; a.relu = relu; a.has_res = dres != nullptr; a.fixed_stats = fixed_stats;
    a.inv_count = 1.0 / ((double)HW * (per_sample ? 1 : B));
    a.eps = eps;
    a.dx_amax = dx_amax;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 g1((HW + NSLAB - 1) / NSLAB, B);
    norm_bwd_stats_kernel<<<g1, 256, 0, s>>>(a);
    int gx = (int)(((long long)HW * (C / 4) + 255) / 256);
    // at most ~FF_NORM_BWD_BLOCKS blocks in all when max|dx| is wanted: every block then ends in an atomicMax on ONE word
    static const int cap = ff::tune_env("FF_NORM_BWD_BLOCKS") ? atoi(ff::tune_env("FF_NORM_BWD_BLOCKS")) : 1024;
    const int gmax = dx_amax ? std::max(1, cap / B) : 1024;
    if (gx > gmax) gx = gmax;
    norm_bwd_apply_kernel<<<dim3(gx, B), 256, 0, s>>>(a);
    return ff::check_launch("ff_norm_bwd");
}

extern "C" int ff_gru_rh_bwd(const float* drh, int drh_ld, const float* r, int r_ld, const float* h, int h_ld, float* dr,
                             int dr_ld, float* dh, int dh_ld, long long npix, int C, void* stream) {
    FF_REQUIRE(drh && r && h && dr && dh && npix > 0 && C % 4 == 0, "ff_gru_rh_bwd: bad argument");
    FF_REQUIRE((drh_ld | r_ld | h_ld | dr_ld | dh_ld) % 4 == 0 && ff::aligned16(drh) && ff::aligned16(r) && ff::aligned16(h) && ff::aligned16(dr) && ff::aligned16(dh), "ff_gru_rh_bwd: alignment");
    gru_rh_bwd_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(drh, drh_ld, r, r_ld, h, h_ld, dr, dr_ld, dh, dh_ld, npix, C);
    return ff::check_launch("ff_gru_rh_bwd");
}

extern "C" int ff_gru_blend_bwd(const float* dhn, int dhn_ld, const float* z, int z_ld, const float* q, int q_ld,
                                const float* h, int h_ld, float* dz, int dz_ld, float* dq, int dq_ld, float* dh,
                                int dh_ld, long long npix, int C, void* stream) {
    FF_REQUIRE(dhn && z && q && h && dz && dq && dh && npix > 0 && C % 4 == 0, "ff_gru_blend_bwd: bad argument");
    FF_REQUIRE((dhn_ld | z_ld | q_ld | h_ld | dz_ld | dq_ld | dh_ld) % 4 == 0 && ff::aligned16(dhn) && ff::aligned16(z) && ff::aligned16(q) && ff::aligned16(h) && ff::aligned16(dz) && ff::aligned16(dq) && ff::aligned16(dh), "ff_gru_blend_bwd: alignment");
    gru_blend_bwd_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(dhn, dhn_ld, z, z_ld, q, q_ld, h, h_ld, dz, dz_ld, dq, dq_ld, dh, dh_ld, npix, C);
    return ff::check_launch("ff_gru_blend_bwd");
}

extern "C" int ff_upsample_flow_bwd(const float* dout_nchw, const float* flow, int flow_ld, const float* mask,
                                    int mask_ld, float* dflow, float* dmask, int B, int H, int W, void* stream) {
    FF_REQUIRE(dout_nchw && flow && mask && dflow && dmask && B > 0 && H > 0 && W > 0 && flow_ld >= 2 && mask_ld >= 576,
               "ff_upsample_flow_bwd: bad argument");
    const size_t lds = (size_t)3 * W * 2 * sizeof(float);
    upsample_bwd_kernel<<<dim3(H, B, W >= 16 ? 2 : 1), 256, lds, static_cast<hipStream_t>(stream)>>>(dout_nchw, flow, flow_ld, mask, mask_ld, dflow, 2, dmask, 1.f, nullptr, H, W);
    return ff::check_launch("ff_upsample_flow_bwd");
}

extern "C" int ff_upsample_flow_bwd_ex(const float* dout_nchw, const float* flow, int flow_ld, const float* mask, int mask_ld,
                                       float* dflow, int dflow_ld, float* dmask, float mask_scale, unsigned int* dmask_amax,
                                       int B, int H, int W, void* stream) {
    FF_REQUIRE(dout_nchw && flow && mask && dflow && dmask && B > 0 && H > 0 && W > 0 && flow_ld >= 2 && mask_ld >= 576 && dflow_ld >= 2,
               "ff_upsample_flow_bwd_ex: bad argument");
    const size_t lds = (size_t)3 * W * 2 * sizeof(float);
    upsample_bwd_kernel<<<dim3(H, B, W >= 16 ? 2 : 1), 256, lds, static_cast<hipStream_t>(stream)>>>(dout_nchw, flow, flow_ld, mask, mask_ld, dflow, dflow_ld, dmask, mask_scale, dmask_amax, H, W);
    return ff::check_launch("ff_upsample_flow_bwd_ex");
}
