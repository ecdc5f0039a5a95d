// Weight gradient of the implicit-GEMM convolution on the fp32 matrix pipe.
//
//   dW[co][k] = sum_m dY[m][co] * Xcol[m][k]      m = output pixel, k = (kh,kw,ci)
//
// Both operands have the REDUCTION index (pixel) as their slow, strided
// dimension in NHWC memory, so tiles are staged exactly as they lie —
// sA[32 pixels][128 co], sB[32 pixels][128 k], 16-byte coalesced loads, 16-byte
// conflict-free LDS stores — and the MFMA fragments are picked out of LDS with
// ds_read_b32 at a row stride (lanes i=0..31 read 32 consecutive floats of one
// pixel row: conflict-free; the two half-waves read different rows, which the
// hardware services separately).  fp32 MFMA spends 64 cycles per instruction, so
// 4 narrow LDS reads per fragment are noise.
//
// The pixel range is split across blockIdx.y; every block adds its 128x128 (or
// 128x64) tile into dW with fp32 atomics (128-byte row segments per wave
// instruction = full atomic rate).  dW must be zeroed by the caller.
// With groups = B and a 1x1 kernel this is also d(corr volume)/d(fmap2).
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RK = 32;   // pixels per reduction chunk

struct WgArgs {
    FFConvParams p;   // x segments / geometry as in the forward; p.y = dY (NHWC, y_ld), p.w unused
    float* dw;        // [Cout][K] (+ group stride)
    long long dw_gstride;
    int M, K, Cin;
    int n1_tiles, n2_tiles, splits, chunks_per_split;
};

template <int TN2>   // BN2 = 64*TN2 columns of k per block
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
    constexpr int BN1 = 128, BN2 = 64 * TN2;
    constexpr int LB = BN2 / 32;                 // float4 loads per thread per chunk for the X tile
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sA = reinterpret_cast<float*>(smem_raw);      // [2][RK][BN1]
    float* sB = sA + 2 * RK * BN1;                       // [2][RK][BN2]
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;             // 2x2 waves: 64 co x (BN2/2) k each
    const int t1 = blockIdx.x / a.n2_tiles, t2 = blockIdx.x - t1 * a.n2_tiles;
    const int co0 = t1 * BN1, k0 = t2 * BN2;
    const int grp = blockIdx.z;
    const int H = p.H, W = p.W, Wo = p.Wo, HoWo = p.Ho * p.Wo;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;     // dilation (FF-PWC refiner)

    const float* dy = p.y + (long long)grp * p.y_gstride;
    const float* xs0 = p.x[0] + (long long)grp * p.x_gstride[0];
    const float* xs1 = p.x[1] ? p.x[1] + (long long)grp * p.x_gstride[1] : nullptr;
    const float* xs2 = p.x[2] ? p.x[2] + (long long)grp * p.x_gstride[2] : nullptr;

    // A tile (dY): thread stages float4 group ga of rows ra + 8*i
    const int ga = tid & 31, ra = tid >> 5;
    const int coa = co0 + ga * 4;
    // B tile (Xcol): thread stages k-group gb of rows rb + (256/ (BN2/4))*i ; decode k once
    constexpr int GB = BN2 / 4;                  // float4 groups per row
    constexpr int RSTEP = 256 / GB;              // rows covered per pass
    const int gb = tid % GB, rb = tid / GB;
    const int kk = k0 + gb * 4;
    const bool kok = kk < a.K;
    int dyk = 0, dxk = 0, cik = 0, ldk = 0;
    const float* xpk = nullptr;
    if (kok) {
        const int tap = kk / a.Cin;
        cik = kk - tap * a.Cin;
        dyk = tap / p.KW;
        dxk = tap - dyk * p.KW;
        const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];
        if (cik < c0) { xpk = xs0; ldk = p.x_ld[0]; }
        else if (cik < c01) { xpk = xs1; ldk = p.x_ld[1]; cik -= c0; }
        else { xpk = xs2; ldk = p.x_ld[2]; cik -= c01; }
    }

    f32x4 rav[4], rbv[LB];
    auto stage_load = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + ra + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < a.M && coa < p.Cout) {   // Cout padded to a multiple of 4 by the caller's buffer (y_ld)
                v = *reinterpret_cast<const f32x4*>(dy + (long long)m * p.y_ld + coa);
            }
            rav[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int m = mbase + rb + RSTEP * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kok && m < a.M) {
                const int b = m / HoWo, rem = m - b * HoWo;
                const int ho = rem / Wo, wo = rem - ho * Wo;
                const int hi = ho * p.stride - p.pad_h + dyk * dlh, wi = wo * p.stride - p.pad_w + dxk * dlw;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(xpk + (long long)(b * H * W + hi * W + wi) * ldk + cik);
            }
            rbv[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        float* dA = sA + buf * RK * BN1;
        float* dB = sB + buf * RK * BN2;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dA + (ra + 8 * i) * BN1 + ga * 4) = rav[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(dB + (rb + RSTEP * i) * BN2 + gb * 4) = rbv[i];
    };

    f32x16 acc[2][TN2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunk0 = blockIdx.y * a.chunks_per_split;
    const int nchunks_total = (a.M + RK - 1) / RK;
    const int nch = min(a.chunks_per_split, nchunks_total - chunk0);
    if (nch <= 0) return;
    stage_load(chunk0 * RK);
    stage_store(0);
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    int cur = 0;
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) stage_load((chunk0 + c + 1) * RK);
        const float* cA = sA + cur * RK * BN1 + wm * 64 + li;
        const float* cB = sB + cur * RK * BN2 + wn * (BN2 / 2) + li;
#pragma unroll
        for (int s = 0; s < RK / 2; ++s) {       // one MFMA step consumes 2 pixels: row 2s+lh
            const int row = 2 * s + lh;
            float fa[2], fb[TN2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = cA[row * BN1 + i * 32];
#pragma unroll
            for (int j = 0; j < TN2; ++j) fb[j] = cB[row * BN2 + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nch) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float* dw = a.dw + (long long)grp * a.dw_gstride;
#pragma unroll
    for (int j = 0; j < TN2; ++j) {
        const int k = k0 + wn * (BN2 / 2) + j * 32 + li;
        if (k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) atomicAdd(dw + (long long)co * a.K + k, acc[i][j][r] * p.out_scale);
            }
        }
    }
}

// packed [rows][KH][KW][cin_pad] (rows cout_offset..cout_offset+Cout) -> OIHW gradient
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH,
                                    int KW, int cin_pad, int cout_offset) {
    const long long total = (long long)Cout * Cin * KH * KW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kw = i % KW;
        long long t = i / KW;
        const int kh = t % KH; t /= KH;
        const int ci = t % Cin; t /= Cin;
        const int co = (int)t;
        dst[i] = src[((long long)(co + cout_offset) * KH * KW + kh * KW + kw) * cin_pad + ci];
    }
}

// dgrad weights: dst[ci][KH-1-kh][KW-1-kw][co_offset + co] = w[co][ci][kh][kw]; dst rows = cin_rows,
// row length KH*KW*cout_pad.  Entries never written stay as the caller zeroed them.
__global__ void pack_dgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH,
                                  int KW, int cout_pad, int cout_offset) {
    const long long total = (long long)Cout * Cin * KH * KW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kw = i % KW;
        long long t = i / KW;
        const int kh = t % KH; t /= KH;
        const int ci = t % Cin; t /= Cin;
        const int co = (int)t;
        dst[(((long long)ci * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * cout_pad + cout_offset + co] = src[i];
    }
}

}  // namespace

extern "C" int ff_conv2d_wgrad(const FFConvParams* pp, float* dw, long long dw_gstride, float* db, void* stream) {
    FF_REQUIRE(pp && dw, "ff_conv2d_wgrad: null pointer");
    const FFConvParams& p = *pp;
    FF_REQUIRE(p.x[0] && p.y, "ff_conv2d_wgrad: null x/dy");
    FF_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && p.Cout > 0 && p.groups > 0 && p.KH > 0 && p.KW > 0 && p.stride > 0,
               "ff_conv2d_wgrad: bad shape");
    int cin = 0;
    for (int s = 0; s < FF_MAX_SEG; ++s) {
        if (p.x_c[s] == 0) break;
        FF_REQUIRE(p.x[s] && p.x_c[s] % 4 == 0 && p.x_ld[s] >= p.x_c[s] && p.x_ld[s] % 4 == 0 && ff::aligned16(p.x[s]) &&
                   p.x_gstride[s] % 4 == 0, "ff_conv2d_wgrad: segment %d invalid", s);
        cin += p.x_c[s];
    }
    FF_REQUIRE(cin > 0, "ff_conv2d_wgrad: no input channels");
    FF_REQUIRE(ff::aligned16(p.y) && p.y_ld % 4 == 0 && p.y_ld >= (p.Cout + 3) / 4 * 4 && p.y_gstride % 4 == 0,
               "ff_conv2d_wgrad: dY must be 16-byte aligned with y_ld a multiple of 4 covering Cout rounded up to 4");
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    const int Ho = (p.H + 2 * p.pad_h - dlh * (p.KH - 1) - 1) / p.stride + 1, Wo = (p.W + 2 * p.pad_w - dlw * (p.KW - 1) - 1) / p.stride + 1;
    FF_REQUIRE(Ho == p.Ho && Wo == p.Wo, "ff_conv2d_wgrad: output size mismatch");
    const long long M = (long long)p.B * Ho * Wo;
    FF_REQUIRE(M < (1ll << 30), "ff_conv2d_wgrad: too many pixels");
    const bool split = p.w_format != FF_W_F32 && p.groups == 1;
    FF_REQUIRE(!db || split, "ff_conv2d_wgrad: the bias gradient is produced by the split-format kernel only (groups == 1)");
    if (split) {
        if (dw_gstride == 0 && ff::conv2d_wgrad_patch(p, dw, db, cin, static_cast<hipStream_t>(stream)) == FF_OK) return FF_OK;
        return ff::conv2d_wgrad_split(p, dw, db, (int)M, cin, static_cast<hipStream_t>(stream));
    }
    WgArgs a;
    a.p = p;
    a.dw = dw;
    a.dw_gstride = dw_gstride;
    a.M = (int)M;
    a.Cin = cin;
    a.K = p.KH * p.KW * cin;
    const bool wide = a.K > 64;
    const int bn2 = wide ? 128 : 64;
    a.n1_tiles = (p.Cout + 127) / 128;
    a.n2_tiles = (a.K + bn2 - 1) / bn2;
    const int nchunks = (a.M + RK - 1) / RK;
    const long long tiles = (long long)a.n1_tiles * a.n2_tiles * p.groups;
    int splits = (int)((1024 + tiles - 1) / tiles);       // aim at ~1024 blocks
    if (splits > nchunks) splits = nchunks;
    if (splits < 1) splits = 1;
    a.chunks_per_split = (nchunks + splits - 1) / splits;
    a.splits = (nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    dim3 grid(a.n1_tiles * a.n2_tiles, a.splits, p.groups);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (wide) {
        const size_t lds = 2 * RK * (128 + 128) * sizeof(float);
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
        conv_wgrad_kernel<2><<<grid, 256, lds, s>>>(a);
    } else {
        const size_t lds = 2 * RK * (128 + 64) * sizeof(float);
        conv_wgrad_kernel<1><<<grid, 256, lds, s>>>(a);
    }
    return ff::check_launch("ff_conv2d_wgrad");
}

extern "C" int ff_unpack_conv_wgrad(const float* packed, int Cout, int Cin, int KH, int KW, int cin_pad,
                                    int cout_offset, float* dw_oihw, void* stream) {
    FF_REQUIRE(packed && dw_oihw && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && cin_pad >= Cin && cout_offset >= 0,
               "ff_unpack_conv_wgrad: bad argument");
    const long long total = (long long)Cout * Cin * KH * KW;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    unpack_wgrad_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(packed, dw_oihw, Cout, Cin, KH, KW, cin_pad, cout_offset);
    return ff::check_launch("ff_unpack_conv_wgrad");
}

extern "C" int ff_pack_conv_weight_dgrad(const float* w_oihw, int Cout, int Cin, int KH, int KW, float* dst,
                                         int cout_pad, int cout_offset, void* stream) {
    FF_REQUIRE(w_oihw && dst && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && cout_pad % 4 == 0 &&
               cout_offset >= 0 && cout_offset + Cout <= cout_pad, "ff_pack_conv_weight_dgrad: bad argument");
    const long long total = (long long)Cout * Cin * KH * KW;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    pack_dgrad_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(w_oihw, dst, Cout, Cin, KH, KW, cout_pad, cout_offset);
    return ff::check_launch("ff_pack_conv_weight_dgrad");
}
