// Implicit-GEMM convolution on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   C[m][n] = sum_k A[m][k] * Wt[n][k]
//     m : output pixel (b, ho, wo) flattened       n : output channel
//     k : (kh, kw, ci) with ci fastest — NHWC activations make every 4-float
//         k-group one 16-byte load, and the packed weights [Cout][KH*KW*Cin]
//         are K-contiguous as well, so A and Wt tiles share one LDS image.
//
// Block = 256 threads = 4 waves laid out WM x WN; each wave owns TM x TN tiles
// of 32x32 fp32 accumulators.  K advances in 32-float chunks, register-staged
// (issue global loads for chunk t+1, MFMA on chunk t, then ds_write + ONE
// barrier), double-buffered in LDS.
//
// LDS image: tile[row][32 floats] (128 B rows), the 16-B chunk index XOR-ed
// with (row>>1)&7.  Readers (ds_read_b128, lane = row) then hit 16 distinct
// 16-B slots per 16-lane group; writers (8 lanes = one row) fill one 128-B
// bank row — both conflict-free.
//
// Fragment trick: lane (i = l&31, h = l>>5) reads the 4 consecutive k's of
// chunk 2j+h with ONE ds_read_b128 and spends them on 4 successive MFMAs;
// A and B use the same lane->k map, so the k permutation cancels.
//
// The fp32 MFMA is an exact k-ordered fmaf chain (no reduced precision), which
// is what lets the path hold the reference's fp32 results to ~1e-6.
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

struct KernArgs {
    FFConvParams p;
    int M;        // rows per group = B*Ho*Wo
    int K;        // KH*KW*Cin
    int Cin;
    int m_tiles, n_tiles;
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const KernArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LA = BM / 32, LB = BN / 32;  // float4 loads per thread per chunk
    static_assert(WM * WN == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sA = reinterpret_cast<float*>(smem_raw);  // [2][BM][BK]
    float* sB = sA + 2 * BM * BK;                    // [2][BN][BK]

    const FFConvParams& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2), so give
    // each XCD a contiguous run of tiles; n-tiles of one m-tile stay neighbours.
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int grp = blockIdx.y;
    const int mt = bid / a.n_tiles, nt = bid - mt * a.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int H = p.H, W = p.W, Wo = p.Wo, HoWo = p.Ho * p.Wo;
    const int kq = tid & 7;       // which 16-B k-group of the chunk this thread stages
    const int rbase = tid >> 3;   // 0..31

    // per-thread A rows: decode once
    int hi0[LA], wi0[LA], img[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = m0 + rbase + 32 * i;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int ho = rem / Wo, wo = rem - ho * Wo;
            hi0[i] = ho * p.stride - p.pad_h;
            wi0[i] = wo * p.stride - p.pad_w;
            img[i] = b * H * W;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = 0;
            img[i] = 0;
        }
    }
    const float* wbase = p.w + (long long)grp * p.w_gstride;
    const float* xs0 = p.x[0] + (long long)grp * p.x_gstride[0];
    const float* xs1 = p.x[1] ? p.x[1] + (long long)grp * p.x_gstride[1] : nullptr;
    const float* xs2 = p.x[2] ? p.x[2] + (long long)grp * p.x_gstride[2] : nullptr;
    const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;

    f32x4 ra[LA], rb[LB];
    auto stage_load = [&](int kc) {
        const int k = kc * BK + kq * 4;
        const bool kok = k < a.K;
        const int tap = kok ? k / a.Cin : 0;
        int ci = k - tap * a.Cin;
        const int dy = tap / p.KW, dx = tap - dy * p.KW;
        const float* xp;
        int ld;
        if (ci < c0) { xp = xs0; ld = p.x_ld[0]; }
        else if (ci < c01) { xp = xs1; ld = p.x_ld[1]; ci -= c0; }
        else { xp = xs2; ld = p.x_ld[2]; ci -= c01; }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int hi = hi0[i] + dy * dlh, wi = wi0[i] + dx * dlw;
            const bool ok = kok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(xp + (long long)(img[i] + hi * W + wi) * ld + ci);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int n = n0 + rbase + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kok && n < p.Cout) v = *reinterpret_cast<const f32x4*>(wbase + (long long)n * a.K + k);
            rb[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        float* dA = sA + buf * BM * BK;
        float* dB = sB + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = rbase + 32 * i;
            *reinterpret_cast<f32x4*>(dA + row * BK + swz(row, kq) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int row = rbase + 32 * i;
            *reinterpret_cast<f32x4*>(dB + row * BK + swz(row, kq) * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (a.K + BK - 1) / BK;
    stage_load(0);
    stage_store(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    int cur = 0;
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) stage_load(kc + 1);
        const float* cA = sA + cur * BM * BK;
        const float* cB = sB + cur * BN * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + li;
                fa[i] = *reinterpret_cast<const f32x4*>(cA + row * BK + swz(row, 2 * j + lh) * 4);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = (wn * TN + i) * 32 + li;
                fb[i] = *reinterpret_cast<const f32x4*>(cB + row * BK + swz(row, 2 * j + lh) * 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[jj][t], acc[i][jj], 0, 0, 0);
        }
        if (kc + 1 < nk) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds column n = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* yb = p.y + (long long)grp * p.y_gstride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + li;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float cs = p.ch_scale ? p.ch_scale[n] : 1.f;
        const float ct = p.ch_scale ? p.ch_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * TM + i) * 32 + row;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bias;
                v *= p.out_scale;
                if (p.ch_scale) v = v * cs + ct;
                v = ff::apply_act(v, p.act);
                if (p.res) v = ff::apply_act(v + p.res[(long long)m * p.res_ld + n], p.act_res);
                yb[(long long)m * p.y_ld + n] = v;
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
int launch(const KernArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t lds = 2 * (BM + BN) * BK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<WM, WN, TM, TN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    KernArgs k = a;
    k.m_tiles = (a.M + BM - 1) / BM;
    k.n_tiles = (a.p.Cout + BN - 1) / BN;
    dim3 grid(k.m_tiles * k.n_tiles, a.p.groups);
    conv_fwd_kernel<WM, WN, TM, TN><<<grid, 256, lds, s>>>(k);
    return ff::check_launch("ff_conv2d_fwd");
}

__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin,
                                   int KH, int KW, int cin_pad, int cout_offset) {
    const long long total = (long long)Cout * KH * KW * cin_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % cin_pad;
        long long t = i / cin_pad;
        const int kw = t % KW; t /= KW;
        const int kh = t % KH; t /= KH;
        const int co = (int)t;
        float v = 0.f;
        if (ci < Cin) v = src[(((long long)co * Cin + ci) * KH + kh) * KW + kw];
        dst[((long long)(co + cout_offset) * KH * KW + kh * KW + kw) * cin_pad + ci] = v;
    }
}

}  // namespace

extern "C" int ff_conv2d_splitk_hint(const FFConvParams* pp) {
    if (!pp || pp->groups != 1 || pp->w_format == FF_W_F32) return 0;
    int cin = 0;
    for (int s = 0; s < FF_MAX_SEG && pp->x_c[s]; ++s) cin += pp->x_c[s];
    return cin > 0 ? ff::conv2d_splitk_hint(*pp, cin) : 0;
}

extern "C" int ff_conv2d_stats_parts(const FFConvParams* pp) {
    if (!pp || pp->groups != 1 || (pp->w_format != FF_W_F16X3 && pp->w_format != FF_W_F16) || !ff::aligned16(pp->y) || pp->y_ld % 4) return 0;
    int cin = 0;
    for (int s = 0; s < FF_MAX_SEG && pp->x_c[s]; ++s) cin += pp->x_c[s];
    if (cin <= 0) return 0;
    const int stem = ff::conv2d_stem_stats_parts(*pp, cin);
    if (stem) return stem;
    const int dma = ff::conv2d_dma_stats_parts(*pp, cin);       // the fp32-input route of conv_dma.hip takes the layer before conv_patch.hip
    return dma ? dma : ff::conv2d_stats_parts(*pp, cin);
}

extern "C" int ff_conv2d_fwd(const FFConvParams* pp, void* stream) {
    FF_REQUIRE(pp != nullptr, "ff_conv2d_fwd: null params");
    const FFConvParams& p = *pp;
    FF_REQUIRE(p.x[0] && p.w && p.y, "ff_conv2d_fwd: null x/w/y");
    FF_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && p.Cout > 0 && p.groups > 0, "ff_conv2d_fwd: empty shape");
    FF_REQUIRE(p.KH > 0 && p.KW > 0 && p.stride > 0, "ff_conv2d_fwd: bad kernel/stride");
    int cin = 0;
    for (int s = 0; s < FF_MAX_SEG; ++s) {
        if (p.x_c[s] == 0) {
            for (int t = s; t < FF_MAX_SEG; ++t) FF_REQUIRE(p.x_c[t] == 0, "ff_conv2d_fwd: segments must be packed");
            break;
        }
        FF_REQUIRE(p.x[s] != nullptr, "ff_conv2d_fwd: segment %d null", s);
        FF_REQUIRE(p.x_c[s] > 0 && p.x_c[s] % 4 == 0, "ff_conv2d_fwd: segment %d channels %d not a multiple of 4", s, p.x_c[s]);
        FF_REQUIRE(p.x_ld[s] >= p.x_c[s] && p.x_ld[s] % 4 == 0, "ff_conv2d_fwd: segment %d ld %d invalid", s, p.x_ld[s]);
        FF_REQUIRE(ff::aligned16(p.x[s]) && p.x_gstride[s] % 4 == 0, "ff_conv2d_fwd: segment %d not 16-byte aligned", s);
        cin += p.x_c[s];
    }
    FF_REQUIRE(cin > 0, "ff_conv2d_fwd: no input channels");
    FF_REQUIRE(ff::aligned16(p.w) && p.w_gstride % 4 == 0, "ff_conv2d_fwd: weights not 16-byte aligned");
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    FF_REQUIRE(dlh >= 1 && dlw >= 1, "ff_conv2d_fwd: bad dilation");
    const int Ho = (p.H + 2 * p.pad_h - dlh * (p.KH - 1) - 1) / p.stride + 1, Wo = (p.W + 2 * p.pad_w - dlw * (p.KW - 1) - 1) / p.stride + 1;
    FF_REQUIRE(Ho == p.Ho && Wo == p.Wo, "ff_conv2d_fwd: output %dx%d does not match conv arithmetic %dx%d", p.Ho, p.Wo, Ho, Wo);
    FF_REQUIRE(p.y_ld >= p.Cout, "ff_conv2d_fwd: y_ld %d < Cout %d", p.y_ld, p.Cout);
    FF_REQUIRE(!p.res || p.res_ld >= (p.res2 ? p.res_split : p.Cout), "ff_conv2d_fwd: res_ld too small");
    FF_REQUIRE((p.ch_scale == nullptr) == (p.ch_shift == nullptr), "ff_conv2d_fwd: ch_scale/ch_shift must come together");
    FF_REQUIRE(p.act >= FF_ACT_NONE && p.act <= FF_ACT_LEAKY, "ff_conv2d_fwd: bad act %d", p.act);
    FF_REQUIRE(p.act_res >= FF_ACT_NONE && p.act_res <= FF_ACT_LEAKY, "ff_conv2d_fwd: bad act_res %d", p.act_res);
    const long long M = (long long)p.B * Ho * Wo;
    FF_REQUIRE(M < (1ll << 30) && (long long)p.B * p.H * p.W < (1ll << 30), "ff_conv2d_fwd: too many pixels");

    FF_REQUIRE(p.w_format >= FF_W_F32 && p.w_format <= FF_W_F16, "ff_conv2d_fwd: bad w_format %d", p.w_format);
    FF_REQUIRE((p.in_scale == nullptr) == (p.in_shift == nullptr), "ff_conv2d_fwd: in_scale/in_shift must come together");
    FF_REQUIRE(!p.in_scale || ((p.w_format == FF_W_F16X3 || p.w_format == FF_W_F16) && p.groups == 1 && p.x_c[1] == 0 && !p.x_amax && p.KH == 3 && p.KW == 3 &&
                               p.stride == 1 && cin % 32 == 0 && (p.in_act == FF_ACT_NONE || p.in_act == FF_ACT_RELU) &&
                               ff::aligned16(p.in_scale) && ff::aligned16(p.in_shift)),
               "ff_conv2d_fwd: in_scale needs the patch kernel (a split weight format, one segment, 3x3, stride 1, Cin %% 32 == 0)");
    FF_REQUIRE(!p.res2 || (p.res && p.w_format != FF_W_F32 && p.KH == 1 && p.KW == 1 && p.groups == 1 && p.res_split > 0 &&
                           p.res_split < p.Cout && p.res2_ld >= p.Cout - p.res_split),
               "ff_conv2d_fwd: res2 needs res, a split weight format and a 1x1 kernel (0 < res_split < Cout)");
    FF_REQUIRE(!p.stats_part || (ff::aligned16(p.stats_part) && ff_conv2d_stats_parts(pp) > 0),
               "ff_conv2d_fwd: stats_part: this convolution cannot produce statistics (ff_conv2d_stats_parts returned 0) or the buffer is misaligned");
    FF_REQUIRE(p.ep_mode >= FF_EP_NONE && p.ep_mode <= FF_EP_MOTION_TAIL, "ff_conv2d_fwd: bad ep_mode %d", p.ep_mode);
    bool split_in = false;
    for (int s = 0; s < FF_MAX_SEG && p.x_c[s]; ++s) {
        FF_REQUIRE(p.x_fmt[s] == FF_FMT_F32 || p.x_fmt[s] == FF_FMT_SPLIT, "ff_conv2d_fwd: bad x_fmt[%d] = %d", s, p.x_fmt[s]);
        split_in |= p.x_fmt[s] == FF_FMT_SPLIT;
    }
    FF_REQUIRE(p.y_fmt == FF_FMT_F32 || p.y_fmt == FF_FMT_SPLIT, "ff_conv2d_fwd: bad y_fmt %d", p.y_fmt);
    FF_REQUIRE(!(split_in || p.y_fmt || p.y2) || (p.w_format != FF_W_F32 && p.groups == 1 && !p.splitk && !p.stats_part),
               "ff_conv2d_fwd: the split-pair activation format needs a split weight format, groups == 1, no splitk / stats_part");
    FF_REQUIRE(!p.y2 || split_in || (p.stride == 1 && cin % 32 == 0 && p.w_format == FF_W_F16X3),
               "ff_conv2d_fwd: y2 (second output) belongs to conv_dma.hip: split-pair inputs, or fp32 inputs of a stride-1 3x3 / 1x5 / 5x1 f16x3 layer");
    FF_REQUIRE(p.ep_mode != FF_EP_MOTION_TAIL || split_in, "ff_conv2d_fwd: FF_EP_MOTION_TAIL belongs to convolutions over split-pair inputs");
    FF_REQUIRE(p.y_fmt != FF_FMT_SPLIT || (p.y_fmt_from >= 0 && p.y_fmt_from % 32 == 0 && p.y_ld % 4 == 0 && ff::aligned16(p.y) && !p.res2),
               "ff_conv2d_fwd: split-pair output: y_fmt_from %% 32 == 0, y_ld %% 4 == 0, 16-byte aligned y, no res2");
    // a pixel's split-pair chunk is 128 bytes = 32 channels: the x1 half of the LAST chunk lies at channels (Cout & ~31) + 16 .. + 31
    // of the row, so a split output needs room for Cout rounded up to 32 (FF_EP_MOTION_TAIL: Cout + 2) floats per pixel
    {
        const int cout_sp = (p.Cout + (p.ep_mode == FF_EP_MOTION_TAIL ? 2 : 0) + 31) / 32 * 32;
        FF_REQUIRE(p.y_fmt != FF_FMT_SPLIT || p.y_ld >= cout_sp, "ff_conv2d_fwd: split-pair output: y_ld %d < Cout rounded up to 32 channels (%d)", p.y_ld, cout_sp);
        FF_REQUIRE(!p.y2 || p.y2_ld >= cout_sp, "ff_conv2d_fwd: split-pair second output: y2_ld %d < Cout rounded up to 32 channels (%d)", p.y2_ld, cout_sp);
    }
    if (p.ep_mode == FF_EP_COORDS) {
        FF_REQUIRE(p.w_format == FF_W_F32 && p.Cout == 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1 && p.groups == 1 &&
                   dlh == 1 && dlw == 1 && p.act == FF_ACT_NONE && !p.res && p.ep_a && p.ep_b && ff::aligned16(p.ep_b) && (reinterpret_cast<uintptr_t>(p.ep_a) & 7) == 0,
                   "ff_conv2d_fwd: FF_EP_COORDS belongs to the 2-channel 3x3 flow head in fp32 rows (ep_a = coords1, ep_b = flow4)");
    } else if (p.ep_mode == FF_EP_MOTION_TAIL) {
        FF_REQUIRE(p.ep_a && (reinterpret_cast<uintptr_t>(p.ep_a) & 7) == 0, "ff_conv2d_fwd: FF_EP_MOTION_TAIL: ep_a = coords1");
    } else if (p.ep_mode) {
        FF_REQUIRE((p.w_format == FF_W_F16X3 || p.w_format == FF_W_F16) && p.groups == 1 && p.stride == 1 && cin % 32 == 0 && p.KH * p.KW >= 3 && !p.res2 && !p.in_scale &&
                   p.Cout % 4 == 0 && p.y_ld % 4 == 0 && ff::aligned16(p.y),
                   "ff_conv2d_fwd: ep_mode needs the f16x3 patch kernel (stride 1, Cin %% 32 == 0, 3x3 / 1x5 / 5x1) and a 16-byte aligned output");
        FF_REQUIRE(p.ep_a && ff::aligned16(p.ep_a) && p.ep_a_ld % 4 == 0, "ff_conv2d_fwd: ep_a null or misaligned");
        if (p.ep_mode == FF_EP_GRU_RH)
            FF_REQUIRE(p.ep_split > 0 && p.ep_split < p.Cout && p.ep_split % 4 == 0 && p.ep_a_ld >= p.Cout - p.ep_split, "ff_conv2d_fwd: FF_EP_GRU_RH: bad ep_split %d / ep_a_ld %d", p.ep_split, p.ep_a_ld);
        else
            FF_REQUIRE(p.ep_b && ff::aligned16(p.ep_b) && p.ep_b_ld % 4 == 0 && p.ep_a_ld >= p.Cout && p.ep_b_ld >= p.Cout, "ff_conv2d_fwd: FF_EP_GRU_BLEND: ep_b null / misaligned or ld < Cout");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p.w_format != FF_W_F32) return ff::conv2d_fwd_split(p, (int)M, cin, s);
    if (const int rc = ff::conv2d_fwd_small(p, cin, s); rc != 1) return rc;    // 1- and 2-channel 3x3 heads: vector ALU
    // the exact-fp32 MFMA kernels below have no epilogue modes: never drop one silently
    FF_REQUIRE(p.ep_mode == FF_EP_NONE && !p.stats_part && !p.in_scale && !p.res2, "ff_conv2d_fwd: ep_mode / stats_part / in_scale / res2 are not available in the exact-fp32 format for this shape");
    KernArgs a;
    a.p = p;
    a.M = (int)M;
    a.Cin = cin;
    a.K = p.KH * p.KW * cin;
    a.m_tiles = a.n_tiles = 0;

    // Tile choice: widest N tile that Cout fills, then shrink M while the grid
    // would leave most of the 256 CUs idle.
    const long long g = p.groups;
    auto blocks = [&](int bm, int bn) { return g * ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    if (p.Cout > 96 && (p.Cout % 128 == 0 || p.Cout > 192) && blocks(128, 128) >= 200) return launch<2, 2, 2, 2>(a, s);
    if (p.Cout > 64 && p.Cout <= 96 && blocks(128, 96) >= 200) return launch<4, 1, 1, 3>(a, s);
    if (blocks(128, 64) >= 400) return launch<2, 2, 2, 1>(a, s);
    return launch<2, 2, 1, 1>(a, s);
}

extern "C" int ff_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, float* dst, int cin_pad,
                                   int cout_offset, void* stream) {
    FF_REQUIRE(w && dst, "ff_pack_conv_weight: null pointer");
    FF_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && cin_pad >= Cin && cin_pad % 4 == 0 && cout_offset >= 0,
               "ff_pack_conv_weight: bad shape");
    const long long total = (long long)Cout * KH * KW * cin_pad;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    pack_weight_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w, dst, Cout, Cin, KH, KW, cin_pad, cout_offset);
    return ff::check_launch("ff_pack_conv_weight");
}
