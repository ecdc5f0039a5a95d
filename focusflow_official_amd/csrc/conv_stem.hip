// The encoders' stem: Conv2d(3 -> 64, 7, stride 2, padding 3) over an NHWC4 image or mask (extractor.py:123, ff_raft's
// condition branch likewise), precision f16x3.
//
// On the generic im2col route (conv_split.hip) this layer ran 2.5x off the other layers: Cin = 4 makes every 32-wide K
// chunk a gather of eight 16-byte pieces per pixel, re-done for each of the 49 taps.  Here the layer is what it is - a
// small-K (7 x 7 x 4 = 196) convolution whose output (64 channels x 4 B per pixel at half resolution) is the traffic:
//   * a block owns 8 x 16 output pixels x 64 channels and stages the (2*8+5) x (2*16+6) input patch ONCE, already split
//     into its two half planes (x0, x1: 8 bytes per pixel each), double-buffered over the tiles a block walks;
//   * K is ordered (ky, kx', c) with kx' = 0..7 (kx' = 7 carries zero weights): one 32x32x16 MFMA step = one kernel row
//     ky x four taps, and the A fragment of a lane - eight consecutive k = two neighbouring input pixels x 4 channels - is
//     ONE aligned 16-byte LDS read straight out of the patch: no im2col image exists anywhere.  16 lanes read 256
//     contiguous bytes (stride-2 pixels of 8 bytes): conflict-free;
//   * the weights (32 channels x 224 k x two halves per wave = 112 registers) are loaded once per block and stay in
//     registers while the block walks its tiles (persistent blocks, grid = 3 per CU);
//   * epilogue as conv_patch.hip: lane = channel, 32 lanes store 128 contiguous bytes; bias, eval-BatchNorm scale/shift,
//     activation, and the InstanceNorm statistics of the output (FFConvParams.stats_part).
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;                 // output tile
constexpr int PR = 2 * TH + 5, PC = 2 * TW + 6;  // patch: 21 rows x 38 columns (column 37 only meets the zero tap kx' = 7)
constexpr int PITCH = 40 * 8;                  // bytes per patch row of one half plane (40 pixels x 4 halfs)
constexpr int PLANE = PR * PITCH;              // 6720
constexpr int BUF = 2 * PLANE;                 // x0 plane | x1 plane
constexpr int NSLOT = (PR * PC + 255) / 256;   // patch pixels per thread: 4

struct SArgs {
    FFConvParams p;
    int tiles_x, tiles_y, total_tiles;
    long long w_row_bytes;
};

// LATE: the next tile's patch loads are issued behind the matrix phase instead of in front of it (16 staging registers
// are then dead during it - the statistics variant spills otherwise) and land during the epilogue.
template <bool STATS, bool LATE>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const SArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
    const float xs = ff::XSPLIT, xinv = ff::SPLIT_INV;

    // ---- weights of this wave's 32 channels, all 14 steps x two halves, in registers.  Step s = (ky, h): taps
    // kx = 4h + 2lh and 4h + 2lh + 1 (the latter zero when it is 7) x 4 channels; packed row: k = (ky*7 + kx)*4 + c.
    const int co = wn * 32 + li;
    f16x8 w0[14], w1[14];
    {
        const char* wrow = reinterpret_cast<const char*>(p.w) + (long long)min(co, p.Cout - 1) * a.w_row_bytes;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int ky = s >> 1, kx = 4 * (s & 1) + 2 * lh;
            f16x4 t0[2], t1[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // the tap kx' = 7 reads k = 196: the zero padding behind the 196 real k of a packed row (no select: all 56
                // loads of the prologue go out back to back); rows of channels >= Cout are clamped, their results unused
                const int k = kx + e > 6 ? 196 : (ky * 7 + kx + e) * 4;
                const char* src = wrow + (k >> 5) * 128 + (k & 31) * 2;
                t0[e] = *reinterpret_cast<const f16x4*>(src);
                t1[e] = *reinterpret_cast<const f16x4*>(src + 64);
            }
            w0[s] = (f16x8){t0[0][0], t0[0][1], t0[0][2], t0[0][3], t0[1][0], t0[1][1], t0[1][2], t0[1][3]};
            w1[s] = (f16x8){t1[0][0], t1[0][1], t1[0][2], t1[0][3], t1[1][0], t1[1][1], t1[1][2], t1[1][3]};
        }
    }
    const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    const float cs = (p.ch_scale && co < p.Cout) ? p.ch_scale[co] : 1.f;
    const float ct = (p.ch_scale && co < p.Cout) ? p.ch_shift[co] : 0.f;

    // ---- patch staging: thread -> up to NSLOT patch pixels (row, column), one 16-byte load each
    int slot_off[NSLOT];       // LDS byte offset inside a half plane, -1: no pixel
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int e = tid + 256 * i;
        slot_off[i] = e < PR * PC ? (e / PC) * PITCH + (e % PC) * 8 : -1;
    }
    f32x4 stage[NSLOT];
    auto load_patch = [&](int tile) {
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
        const float* xb = p.x[0] + (long long)b * H * W * p.x_ld[0];
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int e = tid + 256 * i;
            const int iy = iy0 + e / PC, ix = ix0 + e % PC;
            const bool ok = e < PR * PC && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* src = xb + ((long long)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)) * p.x_ld[0];
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);        // unconditional load, select afterwards
            stage[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_patch = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            if (slot_off[i] < 0) continue;
            f16x4 h0, h1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sv = stage[i][j] * xs;
                const _Float16 t = (_Float16)sv;
                h0[j] = t;
                h1[j] = (_Float16)(sv - (float)t);
            }
            *reinterpret_cast<f16x4*>(base + slot_off[i]) = h0;
            *reinterpret_cast<f16x4*>(base + PLANE + slot_off[i]) = h1;
        }
    };

    // A fragment of MFMA tile t (output rows 4wm + 2t, + 1): lane = pixel (row li >> 4, column li & 15), lh = tap pair
    int aoff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) aoff[t] = (2 * (4 * wm + 2 * t + (li >> 4))) * PITCH + (2 * (li & 15) + 2 * lh) * 8;

    int tile = blockIdx.x;
    if (tile >= a.total_tiles) return;
    load_patch(tile);
    store_patch(0);
    __syncthreads();
    int buf = 0;
    for (; tile < a.total_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (!LATE && next < a.total_tiles) load_patch(next);          // in flight during this tile's matrix work

        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const char* pb = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int so = (s >> 1) * PITCH + (s & 1) * 32;      // kernel row ky, taps 4h..4h+3
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f16x8 x0 = *reinterpret_cast<const f16x8*>(pb + aoff[t] + so);
                const f16x8 x1 = *reinterpret_cast<const f16x8*>(pb + PLANE + aoff[t] + so);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, w0[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, w1[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, w0[s], acc[t], 0, 0, 0);
            }
        }

        // !LATE: the next tile's patch goes to the other buffer BEFORE this tile's output stores are issued: the wait for its
        // loads (vmcnt counts loads and stores in order) then covers only stores of the previous tile, a whole matrix phase
        // old - behind the epilogue it waited for this tile's 32 stores per lane to be acknowledged, 2-3 us per tile.
        // LATE: loads now, output stores behind them, then a counted wait for the loads alone.
        if (LATE) {
            if (next < a.total_tiles) load_patch(next);
        } else {
            if (next < a.total_tiles) store_patch(buf ^ 1);
            __syncthreads();
        }

        // ---- epilogue: lane = channel co; register r of tile t = pixel (r & 3) + 8 (r >> 2) + 4 lh of the tile's 32
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        float st_p = 0.f, st_s1 = 0.f, st_s2 = 0.f, st_n = 0.f;
        const bool full = y0 + TH <= Ho && x0 + TW <= Wo;       // block-uniform: the whole tile is inside the plane
        if (co < p.Cout) {
            // pixel (t, r) of this lane sits at row 4wm + 2t + (r >> 3), column 8 ((r >> 2) & 1) + 4 lh + (r & 3) of the tile
            float* yb = p.y + (((long long)b * Ho + y0 + 4 * wm) * Wo + x0 + 4 * lh) * p.y_ld + co;
            const int rowp = Wo * p.y_ld;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float vv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t][r] * xinv + bias;
                    v *= p.out_scale;
                    vv[r] = v;
                }
                if (p.ch_scale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[r] = vv[r] * cs + ct;
                }
                if (p.act == FF_ACT_RELU) {         // the activation is a launch constant: one branch per tile, not per value
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[r] = vv[r] < 0.f ? 0.f : vv[r];      // (NaN-propagating, as ff::apply_act)
                }
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        yb[(2 * t + (r >> 3)) * rowp + (8 * ((r >> 2) & 1) + (r & 3)) * p.y_ld] = vv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int y = y0 + 4 * wm + 2 * t + (r >> 3), x = x0 + 8 * ((r >> 2) & 1) + 4 * lh + (r & 3);
                        if (y < Ho && x < Wo) yb[(2 * t + (r >> 3)) * rowp + (8 * ((r >> 2) & 1) + (r & 3)) * p.y_ld] = vv[r];
                    }
                }
                if constexpr (STATS) {
                    // pivot: the lane's first value, inside the plane or not (outside it is the response to the clamped /
                    // zero-padded input: as good a centre as any); pixels outside the plane count for nothing
                    if (t == 0) st_p = vv[0];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int y = y0 + 4 * wm + 2 * t + (r >> 3), x = x0 + 8 * ((r >> 2) & 1) + 4 * lh + (r & 3);
                        const bool in = full || (y < Ho && x < Wo);
                        const float d = in ? vv[r] - st_p : 0.f;
                        st_s1 += d;
                        st_s2 = fmaf(d, d, st_s2);
                        st_n += in ? 1.f : 0.f;
                    }
                }
            }
        }
        if constexpr (STATS) {       // as conv_patch.hip: the upper half-wave folded into the lower one, re-centred on its pivot
            const float p2 = __shfl_xor(st_p, 32), s2 = __shfl_xor(st_s1, 32), q2 = __shfl_xor(st_s2, 32), n2 = __shfl_xor(st_n, 32);
            if (lh == 0 && co < p.Cout) {
                const float pv = st_p, d = p2 - pv;
                const float s1 = st_s1 + s2 + n2 * d;
                const float q1 = st_s2 + q2 + 2.f * d * s2 + n2 * d * d;
                const int part = (ty * a.tiles_x + tx) * 2 + wm, nparts = a.tiles_y * a.tiles_x * 2;
                *reinterpret_cast<f32x4*>(p.stats_part + (((long long)b * nparts + part) * p.Cout + co) * 4) = (f32x4){pv, s1, q1, st_n + n2};
            }
        }

        if (LATE) {
            if (next < a.total_tiles) store_patch(buf ^ 1);
            __syncthreads();
        }
        buf ^= 1;
    }
}

}  // namespace

namespace ff {

// Which convolutions take this route: 7x7, stride 2, padding 3, one NHWC4 segment, f16x3 rows, Cout <= 64, no activation or
// relu, no residual / normalise-on-load / gradient scale / GRU epilogue / K split.
static bool stem_eligible(const FFConvParams& p, int cin) {
    static const bool enabled = !(ff::tune_env("FF_STEM_CONV") && atoi(ff::tune_env("FF_STEM_CONV")) == 0);
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    return enabled && p.w_format == FF_W_F16X3 && p.groups == 1 && p.KH == 7 && p.KW == 7 && p.stride == 2 && p.pad_h == 3 && p.pad_w == 3 &&
           dlh == 1 && dlw == 1 && cin == 4 && p.x_c[0] == 4 && p.x_c[1] == 0 && p.x_ld[0] % 4 == 0 && p.Cout <= 64 && !p.res && !p.res2 &&
           !p.in_scale && !p.x_amax && !p.ep_mode && !(p.splitk > 1 && p.splitk_ws) && (p.act == FF_ACT_NONE || p.act == FF_ACT_RELU) &&
           p.Ho == (p.H + 6 - 7) / 2 + 1 && p.Wo == (p.W + 6 - 7) / 2 + 1;
}

int conv2d_stem_stats_parts(const FFConvParams& p, int cin) {
    if (!stem_eligible(p, cin)) return 0;
    return ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW) * 2;
}

int conv2d_fwd_stem(const FFConvParams& p, int cin, hipStream_t s) {
    if (!stem_eligible(p, cin)) return 1;
    SArgs a;
    a.p = p;
    a.tiles_x = (p.Wo + TW - 1) / TW;
    a.tiles_y = (p.Ho + TH - 1) / TH;
    const long long total = (long long)p.B * a.tiles_x * a.tiles_y;
    if (total >= (1ll << 31)) return 1;
    a.total_tiles = (int)total;
    a.w_row_bytes = (long long)((7 * 7 * 4 + 31) / 32) * 128;
    static const int per_cu = ff::tune_env("FF_STEM_BLOCKS_PER_CU") ? atoi(ff::tune_env("FF_STEM_BLOCKS_PER_CU")) : 2;
    const int blocks = (int)std::min<long long>(total, 256ll * std::max(1, per_cu));
    static const int late = ff::tune_env("FF_STEM_LATE") ? atoi(ff::tune_env("FF_STEM_LATE")) : 3;      // bit 0: statistics variant, bit 1: plain
    if (p.stats_part) {
        if (late & 1) conv_stem_kernel<true, true><<<blocks, 256, 0, s>>>(a);
        else conv_stem_kernel<true, false><<<blocks, 256, 0, s>>>(a);
    } else {
        if (late & 2) conv_stem_kernel<false, true><<<blocks, 256, 0, s>>>(a);
        else conv_stem_kernel<false, false><<<blocks, 256, 0, s>>>(a);
    }
    return check_launch("ff_conv2d_fwd(stem)");
}

}  // namespace ff
