// Small HBM-bound kernels around the update block: input scaling, coordinate
// bookkeeping, GRU gate arithmetic and the convex upsampler.
#pragma clang fp contract(off)
#include <algorithm>
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ff_raft.py:142-145: x -> 2*(x/255) - 1, in that rounding order.
__device__ __forceinline__ float scale255(float v) { return __fsub_rn(__fmul_rn(2.f, __fdiv_rn(v, 255.0f)), 1.0f); }

__global__ void prep_input_kernel(const float* __restrict__ src, int src_c, float fill, float* __restrict__ dst,
                                  int B, int HW) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW, p = i - b * HW;
        f32x4 o;
        if (!src) {
            const float v = scale255(fill);
            o = (f32x4){v, v, v, 0.f};
        } else if (src_c == 1) {
            const float v = scale255(src[b * HW + p]);
            o = (f32x4){v, v, v, 0.f};
        } else {
            const float* s = src + b * 3 * HW + p;
            o = (f32x4){scale255(s[0]), scale255(s[HW]), scale255(s[2ll * HW]), 0.f};
        }
        *reinterpret_cast<f32x4*>(dst + i * 4) = o;
    }
}

__global__ void act_copy_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld,
                                long long npix, int C, int act) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int g = (int)(i - p * cg);
        f32x4 v = *reinterpret_cast<const f32x4*>(src + p * src_ld + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ff::apply_act(v[j], act);
        *reinterpret_cast<f32x4*>(dst + p * dst_ld + g * 4) = v;
    }
}

// fp32 <-> FF_FMT_SPLIT (focusflow_hip.h), one thread = 4 channels of a pixel
__global__ void split_copy_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld,
                                  long long npix, int C, int act, int to_split) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int n4 = (int)(i - p * cg) * 4;
        if (to_split) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + p * src_ld + n4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ff::apply_act(v[j], act);
            ff::store_split4(dst + p * dst_ld, n4, v);
        } else {
            const char* c = reinterpret_cast<const char*>(src + p * src_ld + (n4 & ~31)) + (n4 & 31) * 2;
            const ff::ff_f16x4 h0 = *reinterpret_cast<const ff::ff_f16x4*>(c), h1 = *reinterpret_cast<const ff::ff_f16x4*>(c + 64);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)h0[j] + (float)h1[j]) * 0.25f;
            *reinterpret_cast<f32x4*>(dst + p * dst_ld + n4) = v;
        }
    }
}

// max|x| of an NHWC tensor as float bits into *word (atomicMax; non-negative floats order like their bit patterns); a NaN
// or an infinity anywhere leaves +inf there.  <= 512 blocks, one atomic each.
__global__ void range_probe_kernel(const float* __restrict__ x, int ld, long long npix, int C, unsigned int* __restrict__ word) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    float mx = 0.f;
    bool bad = false;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + (i - p * cg) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = fabsf(v[j]);
            bad |= !(a <= 3.402823466e38f);        // inf or NaN (fmaxf would drop a NaN)
            mx = fmaxf(mx, a);
        }
    }
    if (bad) mx = INFINITY;
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (mx > 0.f && __float_as_uint(mx) > *reinterpret_cast<volatile unsigned int*>(word)) atomicMax(word, __float_as_uint(mx));
    }
}

__global__ void coords_init_kernel(float* __restrict__ coords, const float* __restrict__ finit, int B, int H, int W) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW;
        const int p = (int)(i - b * HW);
        float x = (float)(p % W), y = (float)(p / W);
        if (finit) {  // raft.py:211-212, flow_init is NCHW (B,2,H,W)
            x = __fadd_rn(x, finit[(b * 2) * HW + p]);
            y = __fadd_rn(y, finit[(b * 2 + 1) * HW + p]);
        }
        coords[i * 2] = x;
        coords[i * 2 + 1] = y;
    }
}

__global__ void coords_step_kernel(float* __restrict__ coords1, const float* __restrict__ delta, int delta_ld,
                                   float* __restrict__ flow4, float* __restrict__ slot, int slot_ld, int B, int H,
                                   int W) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        float x = coords1[i * 2], y = coords1[i * 2 + 1];
        if (delta) {  // raft.py:223
            x = __fadd_rn(x, delta[i * delta_ld]);
            y = __fadd_rn(y, delta[i * delta_ld + 1]);
            coords1[i * 2] = x;
            coords1[i * 2 + 1] = y;
        }
        const float fx = __fsub_rn(x, (float)(p % W)), fy = __fsub_rn(y, (float)(p / W));  // raft.py:219
        if (flow4) *reinterpret_cast<f32x4*>(flow4 + i * 4) = (f32x4){fx, fy, 0.f, 0.f};
        if (slot) {
            slot[i * slot_ld] = fx;
            slot[i * slot_ld + 1] = fy;
        }
    }
}

__global__ void gru_rh_kernel(const float* __restrict__ r, int r_ld, const float* __restrict__ h, int h_ld,
                              float* __restrict__ rh, int rh_ld, long long npix, int C) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int g = (int)(i - p * cg);
        const f32x4 a = *reinterpret_cast<const f32x4*>(r + p * r_ld + g * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(h + p * h_ld + g * 4);
        *reinterpret_cast<f32x4*>(rh + p * rh_ld + g * 4) = a * b;
    }
}

__global__ void gru_blend_kernel(const float* __restrict__ z, int z_ld, const float* __restrict__ q, int q_ld,
                                 const float* __restrict__ h, int h_ld, float* __restrict__ hn, int hn_ld,
                                 long long npix, int C) {
    const int cg = C >> 2;
    const long long total = npix * cg;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cg;
        const int g = (int)(i - p * cg);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(z + p * z_ld + g * 4);
        const f32x4 qq = *reinterpret_cast<const f32x4*>(q + p * q_ld + g * 4);
        const f32x4 hh = *reinterpret_cast<const f32x4*>(h + p * h_ld + g * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)  // (1-z)*h + z*q, update.py:49
            o[j] = __fadd_rn(__fmul_rn(__fsub_rn(1.f, zz[j]), hh[j]), __fmul_rn(zz[j], qq[j]));
        *reinterpret_cast<f32x4*>(hn + p * hn_ld + g * 4) = o;
    }
}

// raft.py:159-170.  One block per 16 coarse pixels of a coarse row (b, h, w0..w0+15); thread = (w, sub-pixel ij).
// mask channel = k*64 + i*8 + j, k = ky*3+kx (F.unfold order); output row 8h+i, col 8w+j.
// The mask is read as it lies (a wave = the 64 sub-pixels of one coarse pixel: 256 contiguous bytes per k); the
// results go through an 8 x 128 LDS tile so that the full-resolution rows are written as 512 contiguous bytes.
constexpr int UPW = 16;
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ flow, int flow_ld,
                                                       const float* __restrict__ mask, int mask_ld,
                                                       float* __restrict__ out, int H, int W) {
    __shared__ float tile[2][8][UPW * 8 + 4];
    const int b = blockIdx.z, h = blockIdx.y, w0 = blockIdx.x * UPW;
    const long long rowpix = ((long long)b * H + h) * W;
    const int HW8 = 64 * H * W;
    for (int t = threadIdx.x; t < UPW * 64; t += 256) {
        const int wl = t >> 6, w = w0 + wl, ij = t & 63, i = ij >> 3, j = ij & 7;
        if (w >= W) continue;
        const float* m = mask + (rowpix + w) * mask_ld + ij;
        float mv[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mv[k] = m[k * 64];
            mx = fmaxf(mx, mv[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mv[k] = expf(mv[k] - mx);
            den += mv[k];
        }
        float ox = 0.f, oy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            // no load under a branch (nine dependent L2 round trips otherwise): clamp, load, select
            const int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
            const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const float* f = flow + (((long long)b * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1)) * flow_ld;
            const float f0 = f[0], f1 = f[1];
            const float fx = in ? 8.f * f0 : 0.f, fy = in ? 8.f * f1 : 0.f;
            const float wgt = mv[k] / den;
            ox += wgt * fx;
            oy += wgt * fy;
        }
        tile[0][i][wl * 8 + j] = ox;
        tile[1][i][wl * 8 + j] = oy;
    }
    __syncthreads();
    const int ncol = min(UPW, W - w0) * 8;
    for (int t = threadIdx.x; t < 2 * 8 * UPW * 8; t += 256) {
        const int c = t & (UPW * 8 - 1), i = (t >> 7) & 7, ch = t >> 10;
        if (c < ncol)
            out[(long long)b * 2 * HW8 + (long long)ch * HW8 + (long long)(8 * h + i) * (8 * W) + 8 * w0 + c] = tile[ch][i][c];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int B, int HW,
                                    int C) {
    const long long total = (long long)B * C * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const long long bc = i / HW;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        dst[i] = src[(b * HW + p) * ld + c];
    }
}

// raw NCHW (1 or 3 channels, or a constant) -> NHWC4, no scaling: FF-PWC consumes [0,255] (ff_pwcnet.py:405-410)
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ src, int src_c, float fill, float* __restrict__ dst, int B,
                                     int HW) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW, p = i - b * HW;
        f32x4 o;
        if (!src) o = (f32x4){fill, fill, fill, 0.f};
        else if (src_c == 1) { const float v = src[b * HW + p]; o = (f32x4){v, v, v, 0.f}; }
        else { const float* q = src + b * 3 * HW + p; o = (f32x4){q[0], q[HW], q[2ll * HW], 0.f}; }
        *reinterpret_cast<f32x4*>(dst + i * 4) = o;
    }
}

// F.interpolate(mode='bilinear', align_corners=False) from NHWC (ld) to NCHW, channel c scaled by mul[c]
__global__ void resize_bilinear_kernel(const float* __restrict__ src, int ld, int C, int Hi, int Wi, float* __restrict__ dst,
                                       int B, int Ho, int Wo, float mul0, float mul1) {
    const long long total = (long long)B * C * Ho * Wo;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        long long t = i / Wo;
        const int y = (int)(t % Ho); t /= Ho;
        const int c = (int)(t % C);
        const long long b = t / C;
        float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* base = src + (b * Hi * Wi) * ld + c;
        const float v00 = base[((long long)y0 * Wi + x0) * ld], v01 = base[((long long)y0 * Wi + x1) * ld];
        const float v10 = base[((long long)y1 * Wi + x0) * ld], v11 = base[((long long)y1 * Wi + x1) * ld];
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        dst[i] = v * (c == 0 ? mul0 : (c == 1 ? mul1 : 1.f));
    }
}

// F.interpolate(bilinear, align_corners=False) of a 1- or 3-channel NCHW image into NHWC4 (1 channel is repeated
// to 3, channel 3 = 0): FF_PWCNET.preprocess (ff_pwcnet.py:391-403) fused with the layout change
__global__ void resize_to_nhwc4_kernel(const float* __restrict__ src, int src_c, int Hi, int Wi, float* __restrict__ dst, int B,
                                       int Ho, int Wo) {
    const long long total = (long long)B * Ho * Wo;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        const long long t = i / Wo;
        const int y = (int)(t % Ho);
        const long long b = t / Ho;
        float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        float o[3];
        for (int c = 0; c < src_c; ++c) {
            const float* pl = src + (b * src_c + c) * (long long)Hi * Wi;
            const float v00 = pl[(long long)y0 * Wi + x0], v01 = pl[(long long)y0 * Wi + x1];
            const float v10 = pl[(long long)y1 * Wi + x0], v11 = pl[(long long)y1 * Wi + x1];
            o[c] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        if (src_c == 1) o[1] = o[2] = o[0];
        *reinterpret_cast<f32x4*>(dst + i * 4) = (f32x4){o[0], o[1], o[2], 0.f};
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ff_prep_input(const float* src, int src_c, float fill, float* dst, int B, int H, int W, void* stream) {
    FF_REQUIRE(dst && B > 0 && H > 0 && W > 0, "ff_prep_input: bad argument");
    FF_REQUIRE(!src || src_c == 1 || src_c == 3, "ff_prep_input: src_c must be 1 or 3 (got %d)", src_c);
    FF_REQUIRE(ff::aligned16(dst), "ff_prep_input: dst not 16-byte aligned");
    prep_input_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(src, src_c, fill, dst, B, H * W);
    return ff::check_launch("ff_prep_input");
}

extern "C" int ff_act_copy(const float* src, int src_ld, float* dst, int dst_ld, long long npix, int C, int act,
                           void* stream) {
    FF_REQUIRE(src && dst && npix > 0 && C > 0 && C % 4 == 0, "ff_act_copy: bad argument");
    FF_REQUIRE(src_ld % 4 == 0 && dst_ld % 4 == 0 && ff::aligned16(src) && ff::aligned16(dst), "ff_act_copy: alignment");
    act_copy_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(src, src_ld, dst, dst_ld, npix, C, act);
    return ff::check_launch("ff_act_copy");
}

extern "C" int ff_split_copy(const float* src, int src_ld, float* dst, int dst_ld, long long npix, int C, int act, int to_split, void* stream) {
    FF_REQUIRE(src && dst && npix > 0 && C > 0 && C % 32 == 0, "ff_split_copy: bad argument (C must be a multiple of 32)");
    FF_REQUIRE(src_ld % 4 == 0 && dst_ld % 4 == 0 && src_ld >= C && dst_ld >= C && ff::aligned16(src) && ff::aligned16(dst), "ff_split_copy: alignment / ld");
    FF_REQUIRE(to_split || act == FF_ACT_NONE, "ff_split_copy: no activation on the way back");
    split_copy_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(src, src_ld, dst, dst_ld, npix, C, act, to_split);
    return ff::check_launch("ff_split_copy");
}

extern "C" int ff_range_probe(const float* x, int ld, long long npix, int C, unsigned int* word, void* stream) {
    FF_REQUIRE(x && word && npix > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C && ff::aligned16(x), "ff_range_probe: bad argument");
    const long long groups = npix * (C / 4);
    range_probe_kernel<<<(unsigned)std::min<long long>((groups + 1023) / 1024, 512), 256, 0, static_cast<hipStream_t>(stream)>>>(x, ld, npix, C, word);
    return ff::check_launch("ff_range_probe");
}

extern "C" int ff_coords_init(float* coords, const float* flow_init, int B, int H, int W, void* stream) {
    FF_REQUIRE(coords && B > 0 && H > 0 && W > 0, "ff_coords_init: bad argument");
    coords_init_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(coords, flow_init, B, H, W);
    return ff::check_launch("ff_coords_init");
}

extern "C" int ff_coords_step(float* coords1, const float* delta, int delta_ld, float* flow4, float* slot, int slot_ld,
                              int B, int H, int W, void* stream) {
    FF_REQUIRE(coords1 && B > 0 && H > 0 && W > 0, "ff_coords_step: bad argument");
    FF_REQUIRE(!delta || delta_ld >= 2, "ff_coords_step: delta_ld");
    FF_REQUIRE(!flow4 || ff::aligned16(flow4), "ff_coords_step: flow4 alignment");
    FF_REQUIRE(!slot || slot_ld >= 2, "ff_coords_step: slot_ld");
    coords_step_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(coords1, delta, delta_ld, flow4, slot, slot_ld, B, H, W);
    return ff::check_launch("ff_coords_step");
}

extern "C" int ff_gru_rh(const float* r, int r_ld, const float* h, int h_ld, float* rh, int rh_ld, long long npix,
                         int C, void* stream) {
    FF_REQUIRE(r && h && rh && npix > 0 && C > 0 && C % 4 == 0, "ff_gru_rh: bad argument");
    FF_REQUIRE(r_ld % 4 == 0 && h_ld % 4 == 0 && rh_ld % 4 == 0 && ff::aligned16(r) && ff::aligned16(h) && ff::aligned16(rh), "ff_gru_rh: alignment");
    gru_rh_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(r, r_ld, h, h_ld, rh, rh_ld, npix, C);
    return ff::check_launch("ff_gru_rh");
}

extern "C" int ff_gru_blend(const float* z, int z_ld, const float* q, int q_ld, const float* h, int h_ld, float* hn,
                            int hn_ld, long long npix, int C, void* stream) {
    FF_REQUIRE(z && q && h && hn && npix > 0 && C > 0 && C % 4 == 0, "ff_gru_blend: bad argument");
    FF_REQUIRE(z_ld % 4 == 0 && q_ld % 4 == 0 && h_ld % 4 == 0 && hn_ld % 4 == 0 && ff::aligned16(z) && ff::aligned16(q) && ff::aligned16(h) && ff::aligned16(hn), "ff_gru_blend: alignment");
    gru_blend_kernel<<<grid_for(npix * (C / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(z, z_ld, q, q_ld, h, h_ld, hn, hn_ld, npix, C);
    return ff::check_launch("ff_gru_blend");
}

extern "C" int ff_upsample_flow(const float* flow, int flow_ld, const float* mask, int mask_ld, float* out, int B,
                                int H, int W, void* stream) {
    FF_REQUIRE(flow && mask && out && B > 0 && H > 0 && W > 0, "ff_upsample_flow: bad argument");
    FF_REQUIRE(flow_ld >= 2 && mask_ld >= 576, "ff_upsample_flow: flow_ld/mask_ld too small");
    dim3 grid((W + UPW - 1) / UPW, H, B);
    upsample_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(flow, flow_ld, mask, mask_ld, out, H, W);
    return ff::check_launch("ff_upsample_flow");
}

extern "C" int ff_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int H, int W, int C, void* stream) {
    FF_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && ld >= C, "ff_nhwc_to_nchw: bad argument");
    nhwc_to_nchw_kernel<<<grid_for((long long)B * H * W * C), 256, 0, static_cast<hipStream_t>(stream)>>>(src, ld, dst, B, H * W, C);
    return ff::check_launch("ff_nhwc_to_nchw");
}

extern "C" int ff_nchw_to_nhwc4(const float* src, int src_c, float fill, float* dst, int B, int H, int W, void* stream) {
    FF_REQUIRE(dst && B > 0 && H > 0 && W > 0 && (!src || src_c == 1 || src_c == 3) && ff::aligned16(dst), "ff_nchw_to_nhwc4: bad argument");
    nchw_to_nhwc4_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(src, src_c, fill, dst, B, H * W);
    return ff::check_launch("ff_nchw_to_nhwc4");
}

extern "C" int ff_resize_bilinear(const float* src_nhwc, int ld, int C, int Hi, int Wi, float* dst_nchw, int B, int Ho, int Wo,
                                  float mul0, float mul1, void* stream) {
    FF_REQUIRE(src_nhwc && dst_nchw && B > 0 && C > 0 && ld >= C && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "ff_resize_bilinear: bad argument");
    resize_bilinear_kernel<<<grid_for((long long)B * C * Ho * Wo), 256, 0, static_cast<hipStream_t>(stream)>>>(src_nhwc, ld, C, Hi, Wi, dst_nchw, B, Ho, Wo, mul0, mul1);
    return ff::check_launch("ff_resize_bilinear");
}

extern "C" int ff_resize_to_nhwc4(const float* src_nchw, int src_c, int Hi, int Wi, float* dst_nhwc4, int B, int Ho, int Wo,
                                  void* stream) {
    FF_REQUIRE(src_nchw && dst_nhwc4 && B > 0 && (src_c == 1 || src_c == 3) && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && ff::aligned16(dst_nhwc4),
               "ff_resize_to_nhwc4: bad argument");
    resize_to_nhwc4_kernel<<<grid_for((long long)B * Ho * Wo), 256, 0, static_cast<hipStream_t>(stream)>>>(src_nchw, src_c, Hi, Wi, dst_nhwc4, B, Ho, Wo);
    return ff::check_launch("ff_resize_to_nhwc4");
}
