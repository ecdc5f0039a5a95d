// Weight gradient on the f16 matrix pipe with fp16x3 split operands (fp32-level accuracy, see conv_split.hip).
//
//   dW[co][k] = sum_m dY[m][co] * Xcol[m][k]      m = output pixel (the reduction), k = (kh, kw, ci)
//
// v_mfma_f32_32x32x16_f16 wants 8 consecutive reduction elements per lane, but in NHWC memory the pixel is the
// slow dimension of both operands.  The transpose happens in registers while staging: a thread loads the same
// 4 channels of 4 (dY) or 2 (Xcol) consecutive pixels, splits them to (x0, x1) fp16 and writes, per channel, the
// run of pixels as one ds_write_b64 / b32 into a [channel][pixel] LDS image (x0 run | x1 run per row, 144-byte
// pitch).  Channel c lives in LDS row (c & 3) * (rows / 4) + (c >> 2), so that the 16 lanes of a store hit 16
// consecutive rows (conflict-free) and a 32-row MFMA tile is 32 consecutive rows read with ds_read_b128.
//
// dY is far below fp16's range: it is scaled by the power of two that puts max|dY| (FFConvParams.x_amax, from
// ff_act_bwd) at 2^10, undone in the epilogue.  Xcol holds activations (|x| < 65504 as in the forward).
// The bias gradient (column sums of dY) rides along in the blocks of the first k-tile.
// Block tile 128 co x 64 k, 32-pixel chunks double-buffered in LDS, pixel range split over blockIdx.y; the
// 128 x 64 result goes through LDS so that every atomic wave-instruction adds 256 contiguous bytes of dW.
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int RK = 32;            // pixels per chunk
constexpr int BN1 = 128, BN2 = 64;
constexpr int PITCH = 144;        // bytes per LDS row: 32 x0 + 32 x1 halfs = 128 B, + 16 B pad
constexpr int A_BYTES = BN1 * PITCH, B_BYTES = BN2 * PITCH;

struct WsArgs {
    FFConvParams p;   // forward geometry; p.y = dY, p.x_amax = bits of max|dY| (nullable)
    float* dw;
    float* db;        // nullable: bias gradient [Cout] (fp32 atomics, caller zeroes)
    int M, K, Cin;
    int n1_tiles, n2_tiles, chunks_per_split;
};

template <int TERMS>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A_BYTES + B_BYTES]; reused by the epilogue
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves: 64 co x 32 k each
    const int t1 = blockIdx.x / a.n2_tiles, t2 = blockIdx.x - t1 * a.n2_tiles;
    const int co0 = t1 * BN1, k0 = t2 * BN2;
    const int H = p.H, W = p.W, Wo = p.Wo, HoWo = p.Ho * p.Wo;
    float xs, xinv;
    ff::input_scale(p.x_amax, xs, xinv);

    // dY staging: channel quad ga, pixels 4*ra .. 4*ra+3 of the chunk
    const int ga = tid & 31, ra = tid >> 5;
    const int coa = co0 + ga * 4;
    const bool aok = coa < p.Cout;                            // dY buffer is channel-padded to a multiple of 4
    // Xcol staging: k quad gb, pixels 2*rb, 2*rb+1
    const int gb = tid & 15, rb = tid >> 4;
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
        if (cik < c0) { xpk = p.x[0]; ldk = p.x_ld[0]; }
        else if (cik < c01) { xpk = p.x[1]; ldk = p.x_ld[1]; cik -= c0; }
        else { xpk = p.x[2]; ldk = p.x_ld[2]; cik -= c01; }
    }

    f32x4 rav[4], rbv[2];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto stage_load = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + ra * 4 + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (aok && m < a.M) v = *reinterpret_cast<const f32x4*>(p.y + (long long)m * p.y_ld + coa);
            rav[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mbase + rb * 2 + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kok && m < a.M) {
                const int b = m / HoWo, rem = m - b * HoWo;
                const int ho = rem / Wo, wo = rem - ho * Wo;
                const int hi = ho * p.stride - p.pad_h + dyk, wi = wo * p.stride - p.pad_w + dxk;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(xpk + (long long)(b * H * W + hi * W + wi) * ldk + cik);
            }
            rbv[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        char* dA = smem + buf * (A_BYTES + B_BYTES);
        char* dB = dA + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // channel 4*ga + j -> LDS row j*32 + ga
            f16x4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = rav[i][j];
                bsum[j] += v;
                const float sv = v * xs;
                const _Float16 x0 = (_Float16)sv;
                h0[i] = x0;
                h1[i] = (_Float16)((sv - (float)x0) * 2048.f);
            }
            char* row = dA + (j * 32 + ga) * PITCH + ra * 8;   // pixels 4*ra.. -> byte 8*ra of the x0 run
            *reinterpret_cast<f16x4*>(row) = h0;
            if (TERMS == 3) *reinterpret_cast<f16x4*>(row + 64) = h1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // k 4*gb + j -> LDS row j*16 + gb
            f16x2 h0, h1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float v = rbv[i][j];
                const _Float16 x0 = (_Float16)v;
                h0[i] = x0;
                h1[i] = (_Float16)((v - (float)x0) * 2048.f);
            }
            char* row = dB + (j * 16 + gb) * PITCH + rb * 4;
            *reinterpret_cast<f16x2*>(row) = h0;
            if (TERMS == 3) *reinterpret_cast<f16x2*>(row + 64) = h1;
        }
    };

    f32x16 acc[2], accx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }

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
        const char* cA = smem + cur * (A_BYTES + B_BYTES) + (wm * 64 + li) * PITCH + lh * 16;
        const char* cB = smem + cur * (A_BYTES + B_BYTES) + A_BYTES + (wn * 32 + li) * PITCH + lh * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // 16 pixels per MFMA step
            f16x8 y0[2], y1[2], x0, x1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y0[i] = *reinterpret_cast<const f16x8*>(cA + i * 32 * PITCH + s * 32);
                if (TERMS == 3) y1[i] = *reinterpret_cast<const f16x8*>(cA + i * 32 * PITCH + s * 32 + 64);
            }
            x0 = *reinterpret_cast<const f16x8*>(cB + s * 32);
            if (TERMS == 3) x1 = *reinterpret_cast<const f16x8*>(cB + s * 32 + 64);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y0[i], x0, acc[i], 0, 0, 0);
                if (TERMS == 3) {
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y0[i], x1, accx[i], 0, 0, 0);
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y1[i], x0, accx[i], 0, 0, 0);
                }
            }
        }
        if (c + 1 < nch) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: un-permute through LDS ([co 128][k 64] fp32, pitch 65), then coalesced atomics
    float* so = reinterpret_cast<float*>(smem);
    const float osc = p.out_scale * xinv;
    const int kcol = wn * 32 + li, kreal = (kcol & 15) * 4 + (kcol >> 4);      // LDS row j*16+gb holds k 4*gb+j
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int arow = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // LDS row j*32+ga holds co 4*ga+j
            const int coreal = (arow & 31) * 4 + (arow >> 5);
            float v = acc[i][r];
            if (TERMS == 3) v += accx[i][r] * (1.f / 2048.f);
            so[coreal * 65 + kreal] = v * osc;
        }
    }
    float* sb = so + 128 * 65;                                  // [8][128] bias partials
    if (a.db && t2 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[ra * 128 + ga * 4 + j] = bsum[j];
    }
    __syncthreads();
    for (int e = tid; e < BN1 * BN2; e += 256) {
        const int co = e >> 6, k = e & 63;
        if (co0 + co < p.Cout && k0 + k < a.K) atomicAdd(a.dw + (long long)(co0 + co) * a.K + k0 + k, so[co * 65 + k]);
    }
    if (a.db && t2 == 0 && tid < 128 && co0 + tid < p.Cout) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += sb[r * 128 + tid];
        atomicAdd(a.db + co0 + tid, s * p.out_scale);
    }
}

}  // namespace

namespace ff {
// called from ff_conv2d_wgrad (groups == 1, split formats) after argument validation
int conv2d_wgrad_split(const FFConvParams& p, float* dw, float* db, int M, int cin, hipStream_t s) {
    WsArgs a;
    a.p = p;
    a.dw = dw;
    a.db = db;
    a.M = M;
    a.Cin = cin;
    a.K = p.KH * p.KW * cin;
    a.n1_tiles = (p.Cout + BN1 - 1) / BN1;
    a.n2_tiles = (a.K + BN2 - 1) / BN2;
    const int nchunks = (M + RK - 1) / RK;
    const long long tiles = (long long)a.n1_tiles * a.n2_tiles;
    static const int target = getenv("FF_WGRAD_BLOCKS") ? atoi(getenv("FF_WGRAD_BLOCKS")) : 1536;   // tuning knob
    int splits = (int)((target + tiles - 1) / tiles);
    if (splits > nchunks) splits = nchunks;
    if (splits < 1) splits = 1;
    a.chunks_per_split = (nchunks + splits - 1) / splits;
    splits = (nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    dim3 grid(a.n1_tiles * a.n2_tiles, splits, 1);
    const size_t lds = std::max<size_t>(2 * (A_BYTES + B_BYTES), (128 * 65 + 8 * 128) * sizeof(float));
    if (p.w_format == FF_W_F16) conv_wgrad_split_kernel<1><<<grid, 256, lds, s>>>(a);
    else conv_wgrad_split_kernel<3><<<grid, 256, lds, s>>>(a);
    return check_launch("ff_conv2d_wgrad(split)");
}
}  // namespace ff
