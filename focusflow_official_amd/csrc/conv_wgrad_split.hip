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
// ff_act_bwd) at 2^10, undone in the epilogue.  Both operands use the same-scale split of ff_common.h (both halves
// on the operand's scale, x 4), so the three product terms share ONE accumulator per MFMA tile; Xcol holds
// activations (|x| < 16376 as in the forward).
// The bias gradient (column sums of dY) rides along in the blocks of the first k-tile.
// Block tile 128 co x 128 k (64 x 128 when Cout <= 64), 32-pixel chunks double-buffered in LDS, pixel range split over
// blockIdx.y; the result goes through LDS so that every atomic wave-instruction adds 256 contiguous bytes of dW.
// The kernel is bound by LDS stores (the transposing 8-byte runs move 85 B/clk per CU against 256 B/clk for reads) and
// by im2col re-reads out of L2, both per tile ROW + COLUMN, while the MFMA work grows with rows x columns: with one
// accumulator per tile the 128 x 128 tile costs the registers the 128 x 64 tile did with two, at a third less LDS
// and L2 traffic per MFMA.
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
constexpr int PITCH = 144;        // bytes per LDS row: 32 x0 + 32 x1 halfs = 128 B, + 16 B pad

struct WsArgs {
    FFConvParams p;   // forward geometry; p.y = dY, p.x_amax = bits of max|dY| (nullable)
    float* dw;
    float* db;        // nullable: bias gradient [Cout] (fp32 atomics, caller zeroes)
    int M, K, Cin;
    int n1_tiles, n2_tiles, chunks_per_split;
};

// One operand tile of N channels x 32 pixels: N/4 channel quads x (1024/N) pixel groups over the 256 threads, each
// thread moving PX = N/32 consecutive pixels of its quad (4 for a 128-wide tile: ds_write_b64 runs; 2 for a 64-wide
// one: ds_write_b32 runs).  Channel 4g + j lives in LDS row j * (N/4) + g.
template <int N>
struct Stage {
    static constexpr int Q = N / 4, PX = N / 32;
};

template <int TERMS, int BN1, int BN2>   // block tile: BN1 output channels x BN2 im2col columns, (128,64) or (64,128)
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const WsArgs a) {
    constexpr int A_BYTES = BN1 * PITCH, B_BYTES = BN2 * PITCH;
    constexpr int QA = Stage<BN1>::Q, PA = Stage<BN1>::PX, QB = Stage<BN2>::Q, PB = Stage<BN2>::PX;
    constexpr int TA = BN1 / 64, TB = BN2 / 64;               // 32-row MFMA tiles per wave (2 x 2 waves)
    static_assert(TA * TB == 2 || TA * TB == 4, "tile must be 128x128, 128x64 or 64x128");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A_BYTES + B_BYTES]; reused by the epilogue
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int t1 = blockIdx.x / a.n2_tiles, t2 = blockIdx.x - t1 * a.n2_tiles;
    const int co0 = t1 * BN1, k0 = t2 * BN2;
    const int H = p.H, W = p.W, Wo = p.Wo, HoWo = p.Ho * p.Wo;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;     // dilation (FF-PWC refiner)
    float xs, xinv;
    ff::input_scale(p.x_amax, xs, xinv);
    xs *= ff::XSPLIT; xinv *= 1.f / (ff::XSPLIT * ff::XSPLIT);      // both operands split at scale 4 (ff_common.h)

    // dY staging: channel quad ga, pixels PA*ra .. of the chunk
    const int ga = tid % QA, ra = tid / QA;
    const int coa = co0 + ga * 4;
    const bool aok = coa < p.Cout;                            // dY buffer is channel-padded to a multiple of 4
    // Xcol staging: k quad gb, pixels PB*rb ..
    const int gb = tid % QB, rb = tid / QB;
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

    f32x4 rav[PA], rbv[PB];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto stage_load = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = mbase + ra * PA + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (aok && m < a.M) v = *reinterpret_cast<const f32x4*>(p.y + (long long)m * p.y_ld + coa);
            rav[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int m = mbase + rb * PB + i;
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
    // split PX consecutive pixels of one channel and write the run: x0 at `row`, x1 64 bytes further
    auto put = [&](char* row, const float (&v)[4], int npx, float scale) {
        _Float16 h0[4], h1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sv = v[i] * scale;
            h0[i] = (_Float16)sv;
            h1[i] = (_Float16)(sv - (float)h0[i]);
        }
        if (npx == 4) {
            *reinterpret_cast<f16x4*>(row) = (f16x4){h0[0], h0[1], h0[2], h0[3]};
            if (TERMS == 3) *reinterpret_cast<f16x4*>(row + 64) = (f16x4){h1[0], h1[1], h1[2], h1[3]};
        } else {
            *reinterpret_cast<f16x2*>(row) = (f16x2){h0[0], h0[1]};
            if (TERMS == 3) *reinterpret_cast<f16x2*>(row + 64) = (f16x2){h1[0], h1[1]};
        }
    };
    auto stage_store = [&](int buf) {
        char* dA = smem + buf * (A_BYTES + B_BYTES);
        char* dB = dA + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // channel 4*ga + j -> LDS row j*QA + ga
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PA; ++i) { v[i] = rav[i][j]; bsum[j] += v[i]; }
            put(dA + (j * QA + ga) * PITCH + ra * PA * 2, v, PA, xs);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // k 4*gb + j -> LDS row j*QB + gb
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PB; ++i) v[i] = rbv[i][j];
            put(dB + (j * QB + gb) * PITCH + rb * PB * 2, v, PB, ff::XSPLIT);
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
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
        const char* cA = smem + cur * (A_BYTES + B_BYTES) + (wm * TA * 32 + li) * PITCH + lh * 16;
        const char* cB = smem + cur * (A_BYTES + B_BYTES) + A_BYTES + (wn * TB * 32 + li) * PITCH + lh * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // 16 pixels per MFMA step
            f16x8 y0[TA], y1[TA], x0[TB], x1[TB];
#pragma unroll
            for (int i = 0; i < TA; ++i) {
                y0[i] = *reinterpret_cast<const f16x8*>(cA + i * 32 * PITCH + s * 32);
                if (TERMS == 3) y1[i] = *reinterpret_cast<const f16x8*>(cA + i * 32 * PITCH + s * 32 + 64);
            }
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                x0[j] = *reinterpret_cast<const f16x8*>(cB + j * 32 * PITCH + s * 32);
                if (TERMS == 3) x1[j] = *reinterpret_cast<const f16x8*>(cB + j * 32 * PITCH + s * 32 + 64);
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y0[i], x0[j], acc[i][j], 0, 0, 0);
                    if (TERMS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y0[i], x1[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(y1[i], x0[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (c + 1 < nch) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: un-permute through LDS ([co BN1][k BN2] fp32, pitch BN2+1), then coalesced atomics
    constexpr int SP = BN2 + 1;
    float* so = reinterpret_cast<float*>(smem);
    const float osc = p.out_scale * xinv;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int brow = (wn * TB + j) * 32 + li;          // LDS row jj*QB + g holds k 4*g + jj
            const int kreal = (brow % QB) * 4 + brow / QB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = (wm * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // row jj*QA + g holds co 4*g + jj
                const int coreal = (arow % QA) * 4 + arow / QA;
                so[coreal * SP + kreal] = acc[i][j][r] * osc;
            }
        }
    float* sb = so + BN1 * SP;                                  // [256/QA][BN1] bias partials
    if (a.db && t2 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[ra * BN1 + ga * 4 + j] = bsum[j];
    }
    __syncthreads();
    for (int e = tid; e < BN1 * BN2; e += 256) {
        const int co = e / BN2, k = e - co * BN2;
        if (co0 + co < p.Cout && k0 + k < a.K) atomicAdd(a.dw + (long long)(co0 + co) * a.K + k0 + k, so[co * SP + k]);
    }
    if (a.db && t2 == 0 && tid < BN1 && co0 + tid < p.Cout) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 256 / QA; ++r) s += sb[r * BN1 + tid];
        atomicAdd(a.db + co0 + tid, s * p.out_scale);
    }
}

template <int BN1, int BN2>
int launch(WsArgs& a, int M, hipStream_t s) {
    const FFConvParams& p = a.p;
    a.n1_tiles = (p.Cout + BN1 - 1) / BN1;
    a.n2_tiles = (a.K + BN2 - 1) / BN2;
    const int nchunks = (M + RK - 1) / RK;
    const long long tiles = (long long)a.n1_tiles * a.n2_tiles;
    // every block ends in BN1 x BN2 fp32 atomics: as many blocks as the chip holds at once (2 per CU at 72 KB of LDS), not
    // more - 1 536 blocks cost the 1x1 and 7x7 layers 30-50 % (tools/wgrad_table.py: 17.8 -> 16.3 ms per step in total)
    static const int target = ff::tune_env("FF_WGRAD_BLOCKS") ? atoi(ff::tune_env("FF_WGRAD_BLOCKS")) : 512;   // tuning knob
    int splits = (int)((target + tiles - 1) / tiles);
    if (splits > nchunks) splits = nchunks;
    if (splits < 1) splits = 1;
    a.chunks_per_split = (nchunks + splits - 1) / splits;
    splits = (nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    dim3 grid(a.n1_tiles * a.n2_tiles, splits, 1);
    const size_t lds = std::max<size_t>(2 * (BN1 + BN2) * PITCH, (BN1 * (BN2 + 1) + (256 / (BN1 / 4)) * BN1) * sizeof(float));
    static bool attr_set = false;       // 128 x 128: 72 KB of staging
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<1, BN1, BN2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<3, BN1, BN2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.w_format == FF_W_F16) conv_wgrad_split_kernel<1, BN1, BN2><<<grid, 256, lds, s>>>(a);
    else conv_wgrad_split_kernel<3, BN1, BN2><<<grid, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_wgrad(split)");
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
    // a 128-row tile of output channels would be half empty for Cout <= 64 (and 3/4 full for 96): go wide in k instead
    static const int tile = ff::tune_env("FF_WGRAD_TILE") ? atoi(ff::tune_env("FF_WGRAD_TILE")) : 0;   // tuning: 1 = 128x64, 2 = 128x128 always
    if (p.Cout <= 64) return launch<64, 128>(a, M, s);
    // 128 x 128 pays where the tile is full and the reduction long (tools/wgrad_table.py: 256 -> 192 3x3 181 -> 158 us,
    // 128 -> 512 3x3 196 -> 181, 96 -> 96 3x3 159 -> 140); narrow or short problems lose blocks and run 15-50 % slower
    const bool big = (p.Cout > 128 && a.K >= 1024) || (p.Cout == 96 && a.K >= 864);
    if (tile == 2 || (tile == 0 && big)) return launch<128, 128>(a, M, s);
    return launch<128, 64>(a, M, s);
}
}  // namespace ff
