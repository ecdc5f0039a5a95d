// FF-PWC native component (core/models/ff-pwcnet/PWCNet_Core/correlation.py:7-232, the
// reference's four CuPy CUDA kernels) re-designed for CDNA4, plus backwarp
// (ff_pwcnet.py:27-47).
//
//   cost volume   out[b,y,x,(p+4)*9+(o+4)] = 1/C * sum_c one[b,y,x,c] * two[b,y+p,x+o,c]
//                 p,o in [-4,4], zero padding   (correlation.py:34-102; s2o = ch%9-4 is the
//                 x displacement, s2p = ch/9-4 the y displacement, :71-72)
//   grad one      gOne[b,y,x,c] = 1/C * sum_{p,o} gOut[b,y,x,(p,o)] * two[b,y+p,x+o,c]      (:104-166)
//   grad two      gTwo[b,y,x,c] = 1/C * sum_{p,o} gOut[b,y-p,x-o,(p,o)] * one[b,y-p,x-o,c]  (:168-232)
//                 = the grad-one kernel applied to the displacement-transposed gradient
//                   G'[q,(p',o')] = gOut[q+(p',o'), (-p',-o')]   (ff_pwc_gout_transpose)
//
// The reference launches one 32-thread block per output pixel and loops over the 81
// displacements serially (and re-arranges NCHW->padded NHWC in a separate pass).  Here the
// activations are NHWC already; a 256-thread block owns an 8x16 pixel tile, stages the
// tile of `one` and the (8+8)x(16+8) halo of `two` in LDS 16 channels at a time (rows padded
// to 20 floats: 16 lanes x 16 B land on 16 distinct bank slots), and every thread keeps up
// to 45 displacement accumulators in registers.
#include <algorithm>
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16, R = 4, HH = TH + 2 * R, HW = TW + 2 * R;   // tile + halo
constexpr int CK = 16, LDP = 20;                                         // channels per chunk, padded LDS pitch

struct CvArgs {
    const float* one; int one_ld;
    const float* two; int two_ld;
    float* out; int out_ld;
    int B, H, W, C;
    float inv_c;
    int act;            // forward: activation applied to the volume (leaky_relu follows it everywhere, ff_pwcnet.py:289)
    float* ws;          // forward, split mode: partial volumes [split][pixel][81]
    int c_per_split;    // channels per blockIdx.z (a multiple of CK)
};

__device__ __forceinline__ void stage_halo(float* s, const float* src, int ld, int b, int y0, int x0, int c0, int H,
                                           int W, int C) {
    // (HH x HW) pixels x CK channels, zero outside the image / beyond C
    for (int e = threadIdx.x; e < HH * HW * (CK / 4); e += 256) {
        const int g = e % (CK / 4), px = e / (CK / 4);
        const int yy = y0 - R + px / HW, xx = x0 - R + px % HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && c0 + g * 4 < C)
            v = *reinterpret_cast<const f32x4*>(src + (((long long)b * H + yy) * W + xx) * ld + c0 + g * 4);
        *reinterpret_cast<f32x4*>(s + px * LDP + g * 4) = v;
    }
}

// SPLIT: blockIdx.z owns a range of the channels and writes a partial volume (cv_finish_kernel adds the ranges in a fixed
// order): the coarse pyramid levels are 1-16 tiles with 96-196 channels - one to sixteen blocks walking the channel
// chunks one after the other on an empty chip (194 us for 7 x 16 pixels).
template <bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void costvolume_fwd_kernel(const CvArgs a) {      // (41 KB of LDS: three blocks per CU)
    // (the two operand tiles, 40 KB; afterwards the same bytes hold the tile's finished volume [128 pixels][81], 40.5 KB)
    constexpr int N_ONE = TH * TW * LDP, N_TWO = HH * HW * LDP, N_OUT = TH * TW * 81;
    __shared__ __attribute__((aligned(16))) float s_all[N_ONE + N_TWO > N_OUT ? N_ONE + N_TWO : N_OUT];
    float* const s_one = s_all;
    float* const s_two = s_all + N_ONE;
    const int tiles_x = (a.W + TW - 1) / TW;
    const int b = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int px = threadIdx.x & 127, half = threadIdx.x >> 7;     // half 0: p in [-4,0] (45 d), half 1: p in [1,4] (36 d)
    const int ly = px / TW, lx = px - ly * TW;
    const int p_lo = half ? 1 : -4, np = half ? 4 : 5;
    float acc[45];
#pragma unroll
    for (int i = 0; i < 45; ++i) acc[i] = 0.f;
    const int c_beg = SPLIT ? (int)blockIdx.z * a.c_per_split : 0, c_end = SPLIT ? min(a.C, c_beg + a.c_per_split) : a.C;
    // staging items of this thread (fixed for the launch): 2 float4 of the `one` tile, 6 of the `two` halo; element offset of the
    // pixel's channel 0, or -1 outside the image
    const int sg = threadIdx.x & 3;                      // 4-channel group inside a chunk
    int off1[2], off2[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (threadIdx.x + 256 * i) >> 2;
        const int yy = y0 + q / TW, xx = x0 + q % TW;
        off1[i] = (yy < a.H && xx < a.W) ? ((b * a.H + yy) * a.W + xx) * a.one_ld : -1;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = (threadIdx.x + 256 * i) >> 2;
        const int yy = y0 - R + q / HW, xx = x0 - R + q % HW;
        off2[i] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? ((b * a.H + yy) * a.W + xx) * a.two_ld : -1;
    }
    for (int c0 = c_beg; c0 < c_end; c0 += CK) {
        const bool cok = c0 + sg * 4 < a.C;
        f32x4 v1[2], v2[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (off1[i] >= 0 && cok) v1[i] = *reinterpret_cast<const f32x4*>(a.one + off1[i] + c0 + sg * 4);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            v2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (off2[i] >= 0 && cok) v2[i] = *reinterpret_cast<const f32x4*>(a.two + off2[i] + c0 + sg * 4);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(s_one + ((threadIdx.x + 256 * i) >> 2) * LDP + sg * 4) = v1[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(s_two + ((threadIdx.x + 256 * i) >> 2) * LDP + sg * 4) = v2[i];
        __syncthreads();
        // (one displacement row at a time: left to itself the compiler hoists all 180 fragment reads of a chunk above the
        // arithmetic - 256 registers, 700 bytes of scratch per lane, one wave per SIMD: the kernel ran 7 x slower than its
        // LDS traffic allows; the asm statements keep the reads of a row behind the arithmetic of the row before)
#pragma unroll 1
        for (int g = 0; g < CK / 4; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(s_one + px * LDP + g * 4);
#pragma unroll
            for (int ip = 0; ip < 5; ++ip) {
                if (ip < np) {
                    const float* row = s_two + ((ly + R + p_lo + ip) * HW + lx) * LDP + g * 4;
                    f32x4 bv[9];
#pragma unroll
                    for (int io = 0; io < 9; ++io) bv[io] = *reinterpret_cast<const f32x4*>(row + io * LDP);
#pragma unroll
                    for (int io = 0; io < 9; ++io)
                        acc[ip * 9 + io] += av[0] * bv[io][0] + av[1] * bv[io][1] + av[2] * bv[io][2] + av[3] * bv[io][3];
                }
                // (ordered after this row's arithmetic through the accumulators, and no memory access crosses it)
                asm volatile("" : "+v"(acc[ip * 9]), "+v"(acc[ip * 9 + 1]), "+v"(acc[ip * 9 + 2]), "+v"(acc[ip * 9 + 3]), "+v"(acc[ip * 9 + 4]),
                             "+v"(acc[ip * 9 + 5]), "+v"(acc[ip * 9 + 6]), "+v"(acc[ip * 9 + 7]), "+v"(acc[ip * 9 + 8]) :: "memory");
            }
        }
    }
    // A lane owns a pixel, and a pixel's 81 values are 324 consecutive bytes: stored from the accumulators, every store
    // instruction wrote 64 single words into 64 different rows (the kernel spent most of its time there: 357 us for the 133 MB of
    // the finest level).  The tile's volume goes through LDS instead (pitch 81 words: odd, the 64 pixels of a wave hit 64 banks)
    // and leaves in element order - a wave instruction writes 256 consecutive bytes of one or two pixel rows.
    __syncthreads();
    {
        float* so = s_all + px * 81 + (p_lo + 4) * 9;
#pragma unroll
        for (int i = 0; i < 36; ++i) so[i] = acc[i] * a.inv_c;
        if (np == 5) {
#pragma unroll
            for (int i = 36; i < 45; ++i) so[i] = acc[i] * a.inv_c;
        }
    }
    __syncthreads();
    float* const dst = SPLIT ? a.ws + (long long)blockIdx.z * a.B * a.H * a.W * 81 : a.out;
    const int dld = SPLIT ? 81 : a.out_ld;
    const int act = SPLIT ? FF_ACT_NONE : a.act;
#pragma unroll 2
    for (int e = threadIdx.x; e < N_OUT; e += 256) {
        const int q = e / 81, d = e - q * 81;
        const int y = y0 + q / TW, x = x0 + q % TW;
        if (y < a.H && x < a.W) dst[(((long long)b * a.H + y) * a.W + x) * dld + d] = ff::apply_act(s_all[e], act);
    }
}

__global__ __launch_bounds__(256) void cv_finish_kernel(const CvArgs a, int splits) {
    const long long npix = (long long)a.B * a.H * a.W, total = npix * 81;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / 81;
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += a.ws[(long long)s * total + i];
        a.out[pix * a.out_ld + (int)(i - pix * 81)] = ff::apply_act(v, a.act);
    }
}

// grad[b,y,x,c] = 1/C * sum_d g[b,y,x,d] * other[b,(y,x)+d,c]
__global__ __launch_bounds__(256) void costvolume_bwd_kernel(const CvArgs a) {   // a.one = g (81 ch), a.two = other, a.out = grad
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_two = s_dyn;                       // [HH*HW][LDP]
    float* s_g = s_dyn + HH * HW * LDP;         // [TH*TW][84]
    const int tiles_x = (a.W + TW - 1) / TW;
    const int b = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    for (int e = threadIdx.x; e < TH * TW * 81; e += 256) {
        const int d = e % 81, q = e / 81;
        const int yy = y0 + q / TW, xx = x0 + q % TW;
        s_g[q * 84 + d] = (yy < a.H && xx < a.W) ? a.one[(((long long)b * a.H + yy) * a.W + xx) * a.one_ld + d] : 0.f;
    }
    // thread = (pixel, 8-channel half of the chunk)
    const int px = threadIdx.x & 127, hc = threadIdx.x >> 7;
    const int ly = px / TW, lx = px - ly * TW;
    const int y = y0 + ly, x = x0 + lx;
    for (int c0 = 0; c0 < a.C; c0 += CK) {
        __syncthreads();
        stage_halo(s_two, a.two, a.two_ld, b, y0, x0, c0, a.H, a.W, a.C);
        __syncthreads();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1      // (unrolled, the 243 LDS reads of a chunk are hoisted above the arithmetic: 256 registers + 364 bytes of scratch per lane)
        for (int ip = 0; ip < 9; ++ip) {
            const float* row = s_two + ((ly + ip) * HW + lx) * LDP + hc * 8;
#pragma unroll
            for (int io = 0; io < 9; ++io) {
                const float gv = s_g[px * 84 + ip * 9 + io];
                acc0 += gv * *reinterpret_cast<const f32x4*>(row + io * LDP);
                acc1 += gv * *reinterpret_cast<const f32x4*>(row + io * LDP + 4);
            }
        }
        const int c = c0 + hc * 8;
        if (y < a.H && x < a.W) {
            float* o = a.out + (((long long)b * a.H + y) * a.W + x) * a.out_ld + c;
            if (c < a.C) *reinterpret_cast<f32x4*>(o) = acc0 * a.inv_c;
            if (c + 4 < a.C) *reinterpret_cast<f32x4*>(o + 4) = acc1 * a.inv_c;
        }
    }
}

// G'[b,y,x,(p',o')] = g[b, y+p', x+o', (-p',-o')]  (zero outside)
__global__ void gout_transpose_kernel(const float* __restrict__ g, int g_ld, float* __restrict__ gt, int gt_ld, int B,
                                      int H, int W) {
    const long long total = (long long)B * H * W * 81;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % 81);
        const long long pix = i / 81;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const int p = d / 9 - 4, o = d % 9 - 4;
        const int yy = y + p, xx = x + o;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            v = g[((b * H + yy) * W + xx) * g_ld + (4 - p) * 9 + (4 - o)];
        gt[pix * gt_ld + d] = v;
    }
}

// backwarp (ff_pwcnet.py:27-47): bilinear, zeros padding, align_corners=False, validity mask
// from the warped ones-channel (> 0.999 -> 1 else 0).
__global__ void backwarp_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ flow, int flow_ld,
                                float fscale, float* __restrict__ out, int out_ld, int B, int H, int W, int C) {
    const int cg = C >> 2;
    const long long total = (long long)B * H * W * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % cg);
        const long long pix = i / cg;
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const float fx = flow[pix * flow_ld] * fscale, fy = flow[pix * flow_ld + 1] * fscale;   // tenFlow * fltBackwarp
        // grid = linspace(-1+1/W, 1-1/W, W)[x] + flow / ((W-1)/2) ; unnormalise: ((g+1)*W-1)/2
        const float gx = (-1.f + 1.f / W) + x * ((2.f - 2.f / W) / (W - 1)) + fx / ((W - 1.f) / 2.f);
        const float gy = (-1.f + 1.f / H) + y * ((2.f - 2.f / H) / (H - 1)) + fy / ((H - 1.f) / 2.f);
        const float ux = ((gx + 1.f) * W - 1.f) / 2.f, uy = ((gy + 1.f) * H - 1.f) / 2.f;
        const float x0f = floorf(ux), y0f = floorf(uy);
        const int ix = (int)x0f, iy = (int)y0f;
        const float wx = ux - x0f, wy = uy - y0f;
        const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
        const bool i00 = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
        const bool i01 = (unsigned)(ix + 1) < (unsigned)W && (unsigned)iy < (unsigned)H;
        const bool i10 = (unsigned)ix < (unsigned)W && (unsigned)(iy + 1) < (unsigned)H;
        const bool i11 = (unsigned)(ix + 1) < (unsigned)W && (unsigned)(iy + 1) < (unsigned)H;
        const float ones = (i00 ? w00 : 0.f) + (i01 ? w01 : 0.f) + (i10 ? w10 : 0.f) + (i11 ? w11 : 0.f);
        const float m = ones > 0.999f ? 1.f : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* base = in + (b * H * W) * in_ld + g * 4;
        if (i00) acc += w00 * *reinterpret_cast<const f32x4*>(base + ((long long)iy * W + ix) * in_ld);
        if (i01) acc += w01 * *reinterpret_cast<const f32x4*>(base + ((long long)iy * W + ix + 1) * in_ld);
        if (i10) acc += w10 * *reinterpret_cast<const f32x4*>(base + ((long long)(iy + 1) * W + ix) * in_ld);
        if (i11) acc += w11 * *reinterpret_cast<const f32x4*>(base + ((long long)(iy + 1) * W + ix + 1) * in_ld);
        *reinterpret_cast<f32x4*>(out + pix * out_ld + g * 4) = acc * m;
    }
}

// Backward of backwarp (ff_pwcnet.py:27-47 = grid_sample(bilinear, zeros, align_corners=False) times a thresholded
// validity mask, whose own gradient is zero: the mask is overwritten in place with constants).  One wave per
// output pixel, lane = 4-channel group:
//   d_in  += w_k * m * gout            scattered to the four source pixels (fp32 atomics; d_in zeroed by the caller)
//   d_flow = m * sum_c gout_c * d(sample_c)/d(u)  *  W/(W-1) * fscale     (u = ((g+1)*W-1)/2, g = grid + flow/((W-1)/2))
__global__ __launch_bounds__(256) void backwarp_bwd_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ flow,
                                                           int flow_ld, float fscale, const float* __restrict__ gout, int g_ld,
                                                           float* __restrict__ din, int din_ld, float* __restrict__ dflow,
                                                           int dflow_ld, int B, int H, int W, int C) {
    const int lane = threadIdx.x & 63, cg = C >> 2;
    const long long npix = (long long)B * H * W;
    for (long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const float fx = flow[pix * flow_ld] * fscale, fy = flow[pix * flow_ld + 1] * fscale;
        const float gx = (-1.f + 1.f / W) + x * ((2.f - 2.f / W) / (W - 1)) + fx / ((W - 1.f) / 2.f);
        const float gy = (-1.f + 1.f / H) + y * ((2.f - 2.f / H) / (H - 1)) + fy / ((H - 1.f) / 2.f);
        const float ux = ((gx + 1.f) * W - 1.f) / 2.f, uy = ((gy + 1.f) * H - 1.f) / 2.f;
        const float x0f = floorf(ux), y0f = floorf(uy);
        const int ix = (int)x0f, iy = (int)y0f;
        const float wx = ux - x0f, wy = uy - y0f;
        const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
        const bool i00 = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
        const bool i01 = (unsigned)(ix + 1) < (unsigned)W && (unsigned)iy < (unsigned)H;
        const bool i10 = (unsigned)ix < (unsigned)W && (unsigned)(iy + 1) < (unsigned)H;
        const bool i11 = (unsigned)(ix + 1) < (unsigned)W && (unsigned)(iy + 1) < (unsigned)H;
        const float ones = (i00 ? w00 : 0.f) + (i01 ? w01 : 0.f) + (i10 ? w10 : 0.f) + (i11 ? w11 : 0.f);
        const bool on = ones > 0.999f;                          // wave-uniform
        float dux = 0.f, duy = 0.f;
        if (on) {
            const long long base = b * H * W;
            for (int g = lane; g < cg; g += 64) {
                const f32x4 go = *reinterpret_cast<const f32x4*>(gout + pix * g_ld + g * 4);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v00 = i00 ? *reinterpret_cast<const f32x4*>(in + (base + (long long)iy * W + ix) * in_ld + g * 4) : z;
                const f32x4 v01 = i01 ? *reinterpret_cast<const f32x4*>(in + (base + (long long)iy * W + ix + 1) * in_ld + g * 4) : z;
                const f32x4 v10 = i10 ? *reinterpret_cast<const f32x4*>(in + (base + (long long)(iy + 1) * W + ix) * in_ld + g * 4) : z;
                const f32x4 v11 = i11 ? *reinterpret_cast<const f32x4*>(in + (base + (long long)(iy + 1) * W + ix + 1) * in_ld + g * 4) : z;
                const f32x4 ddx = (v01 - v00) * (1.f - wy) + (v11 - v10) * wy;
                const f32x4 ddy = (v10 - v00) * (1.f - wx) + (v11 - v01) * wx;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dux += go[j] * ddx[j];
                    duy += go[j] * ddy[j];
                    if (din) {
                        if (i00) atomicAdd(din + (base + (long long)iy * W + ix) * din_ld + g * 4 + j, w00 * go[j]);
                        if (i01) atomicAdd(din + (base + (long long)iy * W + ix + 1) * din_ld + g * 4 + j, w01 * go[j]);
                        if (i10) atomicAdd(din + (base + (long long)(iy + 1) * W + ix) * din_ld + g * 4 + j, w10 * go[j]);
                        if (i11) atomicAdd(din + (base + (long long)(iy + 1) * W + ix + 1) * din_ld + g * 4 + j, w11 * go[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dux += __shfl_xor(dux, o); duy += __shfl_xor(duy, o); }
        if (lane == 0 && dflow) {
            dflow[pix * dflow_ld] = dux * ((float)W / (W - 1.f)) * fscale;
            dflow[pix * dflow_ld + 1] = duy * ((float)H / (H - 1.f)) * fscale;
        }
    }
}

int check_cv(const char* who, const float* a, int a_ld, const float* b, int b_ld, const float* o, int o_ld, int B, int H,
             int W, int C, int a_c, int o_c) {
    FF_REQUIRE(a && b && o && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape (C must be a multiple of 4)", who);
    FF_REQUIRE(a_ld >= a_c && b_ld >= C && o_ld >= o_c, "%s: ld too small", who);
    return FF_OK;
}

}  // namespace

extern "C" int ff_pwc_costvolume_fwd_ex(const float* one, int one_ld, const float* two, int two_ld, float* out, int out_ld,
                                        int B, int H, int W, int C, int act, float* ws, int splits, void* stream) {
    if (int rc = check_cv("ff_pwc_costvolume_fwd", one, one_ld, two, two_ld, out, out_ld, B, H, W, C, C, 81)) return rc;
    FF_REQUIRE(one_ld % 4 == 0 && two_ld % 4 == 0 && ff::aligned16(one) && ff::aligned16(two), "ff_pwc_costvolume_fwd: alignment");
    FF_REQUIRE(splits <= 1 || ws, "ff_pwc_costvolume_fwd_ex: splits need a workspace");
    FF_REQUIRE((long long)B * H * W * std::max(one_ld, two_ld) < (1ll << 31), "ff_pwc_costvolume_fwd: tensor beyond 2^31 elements");
    CvArgs a{one, one_ld, two, two_ld, out, out_ld, B, H, W, C, 1.f / (float)C, act, ws, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (splits > 1) {
        const int chunks = (C + CK - 1) / CK;
        a.c_per_split = (chunks + splits - 1) / splits * CK;
        splits = (C + a.c_per_split - 1) / a.c_per_split;
        costvolume_fwd_kernel<true><<<dim3(tiles, B, splits), 256, 0, s>>>(a);
        const long long total = (long long)B * H * W * 81;
        cv_finish_kernel<<<(unsigned)std::min<long long>((total + 255) / 256, 1024), 256, 0, s>>>(a, splits);
    } else {
        costvolume_fwd_kernel<false><<<dim3(tiles, B), 256, 0, s>>>(a);
    }
    return ff::check_launch("ff_pwc_costvolume_fwd");
}

extern "C" int ff_pwc_costvolume_fwd(const float* one, int one_ld, const float* two, int two_ld, float* out, int out_ld,
                                     int B, int H, int W, int C, void* stream) {
    return ff_pwc_costvolume_fwd_ex(one, one_ld, two, two_ld, out, out_ld, B, H, W, C, FF_ACT_NONE, nullptr, 0, stream);
}

extern "C" int ff_pwc_costvolume_bwd(const float* g, int g_ld, const float* other, int other_ld, float* grad, int grad_ld,
                                     int B, int H, int W, int C, void* stream) {
    if (int rc = check_cv("ff_pwc_costvolume_bwd", g, g_ld, other, other_ld, grad, grad_ld, B, H, W, C, 81, C)) return rc;
    FF_REQUIRE(other_ld % 4 == 0 && grad_ld % 4 == 0 && ff::aligned16(other) && ff::aligned16(grad), "ff_pwc_costvolume_bwd: alignment");
    CvArgs a{g, g_ld, other, other_ld, grad, grad_ld, B, H, W, C, 1.f / (float)C, FF_ACT_NONE, nullptr, 0};
    dim3 grid(((H + TH - 1) / TH) * ((W + TW - 1) / TW), B);
    constexpr size_t lds = (HH * HW * LDP + TH * TW * 84) * sizeof(float);
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&costvolume_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    costvolume_bwd_kernel<<<grid, 256, lds, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_pwc_costvolume_bwd");
}

extern "C" int ff_pwc_gout_transpose(const float* g, int g_ld, float* gt, int gt_ld, int B, int H, int W, void* stream) {
    FF_REQUIRE(g && gt && B > 0 && H > 0 && W > 0 && g_ld >= 81 && gt_ld >= 81, "ff_pwc_gout_transpose: bad argument");
    long long n = ((long long)B * H * W * 81 + 255) / 256;
    if (n > 4096) n = 4096;
    gout_transpose_kernel<<<(unsigned)n, 256, 0, static_cast<hipStream_t>(stream)>>>(g, g_ld, gt, gt_ld, B, H, W);
    return ff::check_launch("ff_pwc_gout_transpose");
}

extern "C" int ff_pwc_backwarp(const float* in, int in_ld, const float* flow, int flow_ld, float flow_scale, float* out,
                               int out_ld, int B, int H, int W, int C, void* stream) {
    FF_REQUIRE(in && flow && out && B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "ff_pwc_backwarp: bad shape");
    FF_REQUIRE(in_ld % 4 == 0 && out_ld % 4 == 0 && in_ld >= C && out_ld >= C && flow_ld >= 2 && ff::aligned16(in) && ff::aligned16(out),
               "ff_pwc_backwarp: ld/alignment");
    long long n = ((long long)B * H * W * (C / 4) + 255) / 256;
    if (n > 4096) n = 4096;
    backwarp_kernel<<<(unsigned)n, 256, 0, static_cast<hipStream_t>(stream)>>>(in, in_ld, flow, flow_ld, flow_scale, out, out_ld, B, H, W, C);
    return ff::check_launch("ff_pwc_backwarp");
}

extern "C" int ff_pwc_backwarp_bwd(const float* in, int in_ld, const float* flow, int flow_ld, float flow_scale, const float* gout,
                                   int gout_ld, float* din, int din_ld, float* dflow, int dflow_ld, int B, int H, int W, int C,
                                   void* stream) {
    FF_REQUIRE(in && flow && gout && (din || dflow), "ff_pwc_backwarp_bwd: null pointer");
    FF_REQUIRE(B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0 && in_ld >= C && gout_ld >= C && flow_ld >= 2 && (!din || din_ld >= C) &&
               (!dflow || dflow_ld >= 2), "ff_pwc_backwarp_bwd: bad shape");
    FF_REQUIRE(in_ld % 4 == 0 && gout_ld % 4 == 0 && ff::aligned16(in) && ff::aligned16(gout), "ff_pwc_backwarp_bwd: alignment");
    const long long npix = (long long)B * H * W;
    long long blocks = (npix + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    backwarp_bwd_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(in, in_ld, flow, flow_ld, flow_scale, gout, gout_ld,
                                                                                      din, din_ld, dflow, dflow_ld, B, H, W, C);
    return ff::check_launch("ff_pwc_backwarp_bwd");
}
