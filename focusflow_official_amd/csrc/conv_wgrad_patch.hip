// Patch-stationary weight gradient for stride-1 "same" convolutions (3x3, 1x5, 5x1) on the f16 matrix pipe, fp16x3 split
// operands (fp32-level accuracy, see conv_split.hip / conv_wgrad_split.hip).
//
//   dW[tap][co][ci] = sum over pixels  dY[pix][co] * X[pix + tap][ci]
//
// conv_wgrad_split.hip treats this as one GEMM over an im2col matrix: every 128-column k-tile re-reads (and re-splits, and
// re-stores to LDS) dY, every tap re-reads X, and the transposing stores into its [channel][pixel] LDS image are what
// bounds it (the VGPR -> LDS path moves ~80 B/clk per CU whatever the store width).  Here a block owns 64 output channels x
// 32 input channels x ALL taps and walks 8 x 16 pixel tiles: per tile it stages dY (128 pixels x 64 channels) and the
// input patch with its halo ((8+KH-1) x (16+KW-1) pixels x 32 channels) ONCE, pixel-major as they lie in memory, and the
// taps only change which patch rows a lane reads - 7x fewer LDS bytes stored per MFMA.  The reduction index of the
// MFMA is the pixel, the slow dimension of both operands in NHWC memory: the fragments come out of the pixel-major
// image through gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 pixel rows x 16 channels
// and each lane receives its channel's 4 pixels), two per 8-pixel fragment.
//
// LDS rows: dY pixel = 64 co x (x0 | x1) halfs = 256 B at a 320-byte pitch, patch pixel = 32 ci x (x0 | x1) = 128 B at a
// 192-byte pitch: the 4 rows x 64 bytes a 32-lane half touches in one transposed read then lie in 64 different banks.
// Block = 4 waves = 2 (32-channel halves of co) x 2 (tap groups); a k-slice is one 16-pixel tile row.
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int TW = 16, TH = 8, NPIX = TW * TH;
constexpr int PY = 320, PX = 192;          // LDS row pitches (bytes)
constexpr int YQ = 16;                     // 4-channel groups per dY pixel (the patch has 8)
constexpr int NYI = NPIX * YQ / 256;       // dY items per thread (8)

struct WpArgs {
    FFConvParams p;   // forward geometry; p.y = dY, p.x_amax = bits of max|dY| (nullable)
    float* dw;
    float* db;        // nullable: bias gradient [Cout] (fp32 atomics, caller zeroes)
    int K, Cin, nci;  // K = KH*KW*Cin, nci = Cin / 32
    int tiles_x, tiles_y, tiles, tiles_per_block;
};

__device__ __forceinline__ void split4(const f32x4 v, f16x4& h0, f16x4& h1) {     // both halves on v's scale (ff_common.h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 a = (_Float16)v[j];
        h0[j] = a;
        h1[j] = (_Float16)(v[j] - (float)a);
    }
}

__device__ __forceinline__ f16x4 tr_read(const char* p) {
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(f16x4, r);
}

__device__ __forceinline__ f16x8 cat8(const f16x4 a, const f16x4 b) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
    return r;
}

// NT = accumulators of a wave = (taps + 1) / 2: its own taps and the shared middle one; NXI = patch items per thread
template <int NT, int NXI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_patch_kernel(const WpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FFConvParams& p = a.p;
    const int KH = p.KH, KW = p.KW, PW = TW + KW - 1, PH = TH + KH - 1, PPIX = PH * PW, ntaps = KH * KW;
    char* sY = smem;                         // [128][PY]
    char* sX = smem + NPIX * PY;             // [PPIX][PX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wt = wave >> 1;
    // Taps = 2m + 1 over two wave groups: group wt owns the m taps wt (m + 1) .. and the MIDDLE tap m on the tile rows
    // of its parity (both groups add their share of it to dW) - 4.5 (3x3) or 2.5 (1x5, 5x1) taps of MFMAs each.
    const int wts = __builtin_amdgcn_readfirstlane(wt);
    const int t1 = blockIdx.x / a.nci, cchunk = blockIdx.x - t1 * a.nci;
    const int co0 = t1 * 64, ci0 = cchunk * 32;
    const int H = p.H, W = p.W;
    float xs, xinv;
    ff::input_scale(p.x_amax, xs, xinv);
    xs *= ff::XSPLIT; xinv *= 1.f / (ff::XSPLIT * ff::XSPLIT);

    // the input segment of this block's 32-channel chunk (block-uniform)
    const float* xseg; int xld, xc;        // xc: first channel of the chunk inside its segment
    {
        const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];
        if (ci0 < c0) { xseg = p.x[0]; xld = p.x_ld[0]; xc = ci0; }
        else if (ci0 < c01) { xseg = p.x[1]; xld = p.x_ld[1]; xc = ci0 - c0; }
        else { xseg = p.x[2]; xld = p.x_ld[2]; xc = ci0 - c01; }
    }
    // staging roles: dY item i = (pixel tid/16 + 16 i, channel quad tid%16); patch item i = (patch pixel tid/8 + 32 i, quad tid%8)
    const int yq = tid & 15, ypx = tid >> 4, xq = tid & 7, xpx = tid >> 3;
    const bool yok = co0 + yq * 4 < p.Cout;                   // dY is channel-padded to a multiple of 4

    f32x4 ry[NYI], rx[NXI];
    // Range-checked buffer loads: a pixel outside the image (or a channel quad past Cout) gets an out-of-range offset and
    // reads as zeros - no branch around a load (a load under a branch costs a vmcnt(0) and its own exec juggling).
    const long long pixels = (long long)p.B * H * W;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.y), 0, (int)(pixels * p.y_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xseg), 0, (int)(pixels * xld * 4), 0x00020000);
    const int ycol = yok ? (co0 + yq * 4) * 4 : 0x7fffffff, xcol = (xc + xq * 4) * 4;
    auto load_tile = [&](int t) {
        const int b = t / (a.tiles_y * a.tiles_x), r = t - b * a.tiles_y * a.tiles_x;
        const int ty = r / a.tiles_x, y0 = ty * TH, x0 = (r - ty * a.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < NYI; ++i) {
            const int px = ypx + 16 * i, y = y0 + (px >> 4), x = x0 + (px & 15);
            const int off = (yok && y < H && x < W) ? ((b * H + y) * W + x) * (p.y_ld * 4) + ycol : 0x7fffffff;
            ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int px = xpx + 32 * i, py = px / PW, y = y0 - p.pad_h + py, x = x0 - p.pad_w + (px - py * PW);
            const int off = (px < PPIX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? ((b * H + y) * W + x) * (xld * 4) + xcol : 0x7fffffff;
            rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
        }
    };
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NYI; ++i) {
            f16x4 h0, h1;
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum[j] += ry[i][j];
            split4(ry[i] * xs, h0, h1);
            char* d = sY + (ypx + 16 * i) * PY + yq * 8;
            *reinterpret_cast<f16x4*>(d) = h0;
            *reinterpret_cast<f16x4*>(d + 128) = h1;
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int px = xpx + 32 * i;
            if (px < PPIX) {
                f16x4 h0, h1;
                split4(rx[i] * ff::XSPLIT, h0, h1);
                char* d = sX + px * PX + xq * 8;
                *reinterpret_cast<f16x4*>(d) = h0;
                *reinterpret_cast<f16x4*>(d + 64) = h1;
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read role: 16-lane group g: k-group g >> 1 (pixels 8 (g >> 1) ..), channels 16 (g & 1) ..; lane 4q + p of the
    // group addresses pixel row q, channels 4p .. 4p + 3 of the block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kpix = 8 * (g >> 1) + q, kch = 16 * (g & 1) + 4 * pp;
    const char* aY = sY + kpix * PY + (wm * 32 + kch) * 2;              // + y * 16 * PY, + 4 * PY (second half), + 128 (x1)
    const char* aX = sX + kpix * PX + kch * 2;                          // + ((y + dy) * PW + dx) * PX, + 4 * PX, + 64 (x1)

    int tapoff[NT];                          // patch-row offset of each of this wave's taps (wave-uniform)
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int tap = k == NT - 1 ? ntaps / 2 : wts * NT + k, dy = tap / KW, dx = tap - dy * KW;
        tapoff[k] = (dy * PW + dx) * PX;
    }
    const int tbeg = blockIdx.y * a.tiles_per_block, tend = min(tbeg + a.tiles_per_block, a.tiles);
    if (tbeg >= tend) return;
    load_tile(tbeg);
    for (int t = tbeg; t < tend; ++t) {
        __syncthreads();                     // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (t + 1 < tend) load_tile(t + 1);  // lands during this tile's MFMAs
#pragma unroll 1
        for (int y = 0; y < TH; ++y) {       // one k-slice = the 16 pixels of tile row y (not unrolled: 8 x 15 MFMAs of fragments in flight spill)
            const char* py = aY + y * 16 * PY;
            const f16x8 a0 = cat8(tr_read(py), tr_read(py + 4 * PY));
            const f16x8 a1 = cat8(tr_read(py + 128), tr_read(py + 4 * PY + 128));
#pragma unroll
            for (int k = 0; k < NT; ++k) {   // no branch around the own taps: tap k + 1's fragments are read under tap k's MFMAs
                if (k == NT - 1 && (y & 1) != wts) break;          // the shared middle tap: rows of this group's parity only
                const char* px = aX + (y * PW * PX + tapoff[k]);
                const f16x8 b0 = cat8(tr_read(px), tr_read(px + 4 * PX));
                const f16x8 b1 = cat8(tr_read(px + 64), tr_read(px + 4 * PX + 64));
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[k], 0, 0, 0);
            }
        }
    }

    // epilogue: acc[k][r]: row = co = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column = ci = lane & 31 -> 128 contiguous bytes of
    // dW per atomic wave-instruction
    const float osc = p.out_scale * xinv;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int tap = k == NT - 1 ? ntaps / 2 : wts * NT + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
            if (co < p.Cout) atomicAdd(a.dw + (long long)co * a.K + tap * a.Cin + ci0 + li, acc[k][r] * osc);
        }
    }
    if (a.db && cchunk == 0) {               // bias gradient: column sums of dY, once per co tile
        __syncthreads();
        float* sb = reinterpret_cast<float*>(smem);          // [16 pixel rows of threads][64 channels]
#pragma unroll
        for (int j = 0; j < 4; ++j) sb[ypx * 64 + yq * 4 + j] = bsum[j];
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cout) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += sb[r * 64 + tid];
            atomicAdd(a.db + co0 + tid, s * p.out_scale);
        }
    }
}

template <int NT, int NXI>
int launch(const WpArgs& a, size_t lds, int splits, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_patch_kernel<NT, NXI>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    dim3 grid(((a.p.Cout + 63) / 64) * a.nci, splits, 1);
    conv_wgrad_patch_kernel<NT, NXI><<<grid, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_wgrad(patch)");
}

}  // namespace

namespace ff {
// returns FF_OK if launched, 1 if the shape is not eligible (the caller falls back to conv_wgrad_split.hip)
int conv2d_wgrad_patch(const FFConvParams& p, float* dw, float* db, int cin, hipStream_t s) {
    static const bool enabled = !(ff::tune_env("FF_WGRAD_PATCH") && atoi(ff::tune_env("FF_WGRAD_PATCH")) == 0);
    if (!enabled || p.w_format != FF_W_F16X3) return 1;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1) return 1;
    const int taps = p.KH * p.KW;
    if (!((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 5) || (p.KH == 5 && p.KW == 1))) return 1;
    if (p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 || p.Ho != p.H || p.Wo != p.W) return 1;
    if (cin % 32) return 1;
    for (int i = 0; i < FF_MAX_SEG; ++i)
        if (p.x_c[i] % 32) return 1;
    if ((long long)p.B * p.H * p.W * std::max(p.y_ld, std::max(p.x_ld[0], std::max(p.x_ld[1], p.x_ld[2]))) * 4 >= (1ll << 31)) return 1;
    WpArgs a;
    a.p = p;
    a.dw = dw;
    a.db = db;
    a.Cin = cin;
    a.K = taps * cin;
    a.nci = cin / 32;
    a.tiles_x = (p.W + TW - 1) / TW;
    a.tiles_y = (p.H + TH - 1) / TH;
    a.tiles = p.B * a.tiles_y * a.tiles_x;
    const int combos = ((p.Cout + 63) / 64) * a.nci;
    // Pixel splits: at most one resident wave of blocks (2 per CU: every block ends in NT x 16 atomic wave-instructions and
    // restages from scratch, so more, shorter blocks lost 10-25 % in tools/wgrad_table.py), and the fewest blocks that
    // keep the longest block's tile count.
    static const int target = ff::tune_env("FF_WGRAD_PATCH_BLOCKS") ? atoi(ff::tune_env("FF_WGRAD_PATCH_BLOCKS")) : 512;   // tuning knob
    int splits = std::max(1, std::min(a.tiles, target / combos));
    a.tiles_per_block = (a.tiles + splits - 1) / splits;
    splits = (a.tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    const int ppix = (TH + p.KH - 1) * (TW + p.KW - 1);
    const size_t lds = (size_t)NPIX * PY + (size_t)ppix * PX;
    if (taps == 9) return launch<5, 6>(a, lds, splits, s);          // 180 patch pixels x 8 quads = 1440 items
    return p.KW == 5 ? launch<3, 5>(a, lds, splits, s)               // 8 x 20 = 160 pixels: 1280 items
                     : launch<3, 6>(a, lds, splits, s);              // 12 x 16 = 192 pixels: 1536 items
}
}  // namespace ff
