// Patch-stationary fp16x3 convolution for stride-1 "same" convolutions (3x3, 1x5, 5x1, 7x7, 1x1).
//
// conv_split.hip re-reads and re-splits every activation once per kernel tap (9x for 3x3) and pays a
// global-load round trip per 32-k chunk.  Here a block owns an 8x16 tile of output pixels and, per
// 32-channel chunk of the input, stages the tile PLUS ITS HALO ((8+KH-1) x (16+KW-1) pixels) in LDS once —
// already converted to the split format [x0: 32 fp16 | x1: 32 fp16] — and then walks the KH*KW taps over
// that stationary patch: a tap only changes which LDS rows a lane reads.  Per tap the only new bytes are
// the 64x32 weight chunk (8 KB, L2-resident; double-buffered, or single-buffered in the high-occupancy variants
// conv_patch_kernel_occ: 35 KB of LDS, four blocks per CU).  Activation traffic and conversion work
// drop by KH*KW, address arithmetic is precomputed once per block (buffer loads, hardware zero padding).
//
// Block = 256 threads = 4 waves (2 x 2): wave (wm, wn) computes output rows 4wm..4wm+3 of the tile
// (two 32-pixel MFMA tiles: rows 2t, 2t+1 x 16 columns) for 32 of the block's 64 output channels.
// The epilogue's arithmetic is written as separately rounded operations and must stay that: conv_patch.hip and conv_dma.hip
// promise the same bits for the same layer (tests/test_hip_split.py), and a multiply-add that one of them contracts into an
// fma - after the optimiser specialised a path on the activation code - breaks that.  build.py reads the next line.
// hipcc-flags: -ffp-contract=off
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 16, ROWB = 128;   // tile height TH = 4*TM (template): 16, 8 or 4 rows; BN = 64*TN output channels
// LDS rows (one patch pixel / one weight row: 32 x0 + 32 x1 halfs = 128 B) sit at a 144-byte pitch instead of being
// XOR-swizzled: 16 lanes reading 16 B from 16 consecutive rows hit 64 distinct banks, the 8-byte stores of the patch
// and the 16-byte stores of the weights are conflict-free as well, and - no XOR - every k-slice / term / B tile is an
// immediate offset from one base register.  PMC on the swizzled version: 35 % of the LDS-active cycles were
// bank-conflict cycles and the kernel issued 10 VALU instructions per MFMA, mostly swizzle arithmetic.
constexpr int ROWP = 144;
// MF16: row i of a 16-row MFMA tile <-> position in the tile (pixel column / LDS row of the weight tile)
__device__ __forceinline__ int PI16(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * (i - 12) + 9); }
__device__ __forceinline__ int KG16(int g) { return ((g & 1) << 1) | (g >> 1); }     // place of k-group g inside a term's 64 bytes


struct PArgs {
    FFConvParams p;
    int Cin, nci;            // channels, 32-channel chunks
    int tiles_x, tiles_y, n_tiles;
    long long w_row_bytes;
    int nci_split;           // SPLITK: input chunks per K split (blockIdx.y), partial sums to p.splitk_ws
};

__device__ __forceinline__ void split4(const f32x4 v, f16x4& h0, f16x4& h1) {     // both halves on v's scale (ff_common.h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 a = (_Float16)v[j];
        h0[j] = a;
        h1[j] = (_Float16)(v[j] - (float)a);
    }
}

// NITEM = patch items per thread; TM x TN = 32-pixel x 32-channel MFMA tiles per wave (block: 2 x 2 waves =
// 64 TM pixels x 64 TN channels); ABL = timing-only ablation
// SPLITK: blockIdx.y owns a range of the 32-channel input chunks and writes raw partial sums (splitk_finish_kernel adds
// them in a fixed order and applies the epilogue): small planes with long reductions - FF-PWC's DenseNet decoders at
// 7x16 .. 56x128 - are otherwise 8-60 blocks walking 100-190 taps one after the other on an empty chip.
// MF16: the matrix work as v_mfma_f32_16x16x32_f16 instead of 32x32x16 - on dense data the chip holds a higher clock on
// the small shape (tools/proto/mfma_shape.hip: the LDS-fed three-term loop of this kernel's wave tile runs 1.59 vs
// 1.28-1.37 PFLOP/s).  Channels are the MFMA rows (A operand: a lane's four accumulator registers of a tile are four
// CONSECUTIVE output channels of one pixel: 16-byte stores and residual loads), pixels the columns; one MFMA covers the
// whole 32-channel chunk of a term.  LDS image for conflict-free ds_read_b128 in this lane layout (lane = (row i = lane &
// 15, k-group g = lane >> 4); the instruction serves lanes {0-3,12-15} of one k-group together with {4-11} of the next):
// the 16-byte k-groups of a term sit in the order 0, 2, 1, 3, and MFMA rows 4..11 read the even rows of a tile,
// rows 0..3 / 12..15 the odd ones (a weight row's place in LDS, a pixel's column in the tile: PI16).
// EPI (MF16 only): FFConvParams.ep_mode - a GRU step applied to the finished value (FF_EP_GRU_RH / FF_EP_GRU_BLEND)
// STATS: FFConvParams.stats_part - every lane also reduces the values it stores to {pivot, sum(v - pivot), sum((v - pivot)^2),
// count} and writes them as one 16-byte entry [image][part][channel]; ff_norm_stats_finish adds the parts up in double.
// The pivot (the lane's first value) keeps the fp32 sums of a nearly constant plane - the condition branch of frame 2 -
// free of cancellation; the InstanceNorm statistics pass that re-read every convolution output is gone.
template <int TERMS, int NITEM, int TM, int TN, int ABL, int WB, bool PIN = false, bool INORM = false, bool SPLITK = false, bool MF16 = false, int EPI = 0, bool STATS = false>   // WB = weight buffers in LDS
__device__ __forceinline__ void conv_patch_body(const PArgs& a) {
    constexpr int TH = 4 * TM, BN = 64 * TN, NW = 2 * TN;       // NW = 16-byte weight pieces per thread and tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FFConvParams& p = a.p;
    float xs, xinv;
    ff::input_scale(p.x_amax, xs, xinv);       // 1, 1 unless the caller passed max|x| (gradients: dgrad on the f16 pipe)
    xs *= ff::XSPLIT; xinv *= ff::SPLIT_INV;   // operand scales of the split format
    const int KH = p.KH, KW = p.KW, PH = TH + KH - 1, PW = TW + KW - 1, NPIX = PH * PW;
    char* sP = smem;                          // [NPIX][128 B] patch, split format
    char* sW = smem + ((NPIX * ROWP + 255) & ~255);   // [2][BN] weight rows of one (tap, ci-chunk)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int H = p.H, W = p.W;
    int bid = blockIdx.x;
    const int nt = bid % a.n_tiles; bid /= a.n_tiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int bimg = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;

    // buffer resources
    const long long pix_total = (long long)p.B * H * W;
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[0]), 0, (int)(pix_total * p.x_ld[0] * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[1] ? p.x[1] : p.x[0]), 0, p.x[1] ? (int)(pix_total * p.x_ld[1] * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[2] ? p.x[2] : p.x[0]), 0, p.x[2] ? (int)(pix_total * p.x_ld[2] * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long long)p.Cout * a.w_row_bytes), 0x00020000);
    const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];

    // patch items of this thread: item = tid + 256*i -> (patch pixel, 16-byte channel group kq)
    int ppix[NITEM];        // image pixel index of the patch pixel, or -1 (outside the image / no item)
    int prow[NITEM];        // LDS row (patch pixel)
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
        const int item = tid + 256 * i;
        const int px = item >> 3;
        ppix[i] = -1;
        prow[i] = px;
        if (px < NPIX) {
            const int yy = y0 - p.pad_h + px / PW, xx = x0 - p.pad_w + px % PW;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ppix[i] = (bimg * H + yy) * W + xx;
        }
    }
    const int kq = tid & 7;
    // weight pieces of this thread: BN rows x 8 pieces -> NW per thread
    int woff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        woff[i] = n < p.Cout ? (int)(n * a.w_row_bytes) + kq * 16 : 0x7fffffff;
    }

    f32x4 rp[NITEM], rw[NW];
    f32x4 nmul = {1.f, 1.f, 1.f, 1.f}, nadd = {0.f, 0.f, 0.f, 0.f};   // INORM: the chunk's (scale, shift) of this thread's 4 channels
    auto load_patch = [&](int c) {          // 32-channel chunk c of the concatenated input (block-uniform)
        int ci0 = c * 32;
        if (INORM) {                        // FFConvParams.in_scale / in_shift: one segment, tables [B][Cin]
            const long long t = (long long)bimg * a.Cin + ci0 + kq * 4;
            nmul = *reinterpret_cast<const f32x4*>(p.in_scale + t);
            nadd = *reinterpret_cast<const f32x4*>(p.in_shift + t);
        }
        __amdgpu_buffer_rsrc_t rs;
        int ldb;
        if (ci0 < c0) { rs = rs0; ldb = p.x_ld[0] * 4; }
        else if (ci0 < c01) { rs = rs1; ldb = p.x_ld[1] * 4; ci0 -= c0; }
        else { rs = rs2; ldb = p.x_ld[2] * 4; ci0 -= c01; }
        const int cib = (ci0 + kq * 4) * 4;
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
            const int off = ppix[i] >= 0 ? ppix[i] * ldb + cib : 0x7fffffff;     // out of range -> zeros
            rp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };
    auto store_patch = [&]() {
        const int pc = MF16 ? KG16(kq >> 1) : kq >> 1, half = (kq & 1) * 8;
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
            const int row = prow[i];
            if (row < NPIX) {
                f16x4 h0, h1;
                f32x4 v = rp[i];
                if (INORM) {                // the producer's normalisation (+ ReLU) on the way in; padding is zero AFTER it
                    v = __builtin_elementwise_fma(v, nmul, nadd);      // (one fused operation, as norm_apply_kernel's: the two give the same bits)
                    if (p.in_act == FF_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];      // (not fmaxf: a NaN must stay one - ff::apply_act)
                    }
                    if (ppix[i] < 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                split4(v * xs, h0, h1);
                *reinterpret_cast<f16x4*>(sP + row * ROWP + pc * 16 + half) = h0;
                if (TERMS == 3) *reinterpret_cast<f16x4*>(sP + row * ROWP + 64 + pc * 16 + half) = h1;
            }
        }
    };
    auto load_w = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NW; ++i)
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[i], kc * ROWB, 0));
    };
    auto store_w = [&](int buf) {
        char* d = sW + buf * BN * ROWP;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int row = (tid >> 3) + 32 * i;
            if (MF16) *reinterpret_cast<f32x4*>(d + ((row & ~15) + PI16(row & 15)) * ROWP + ((kq & 4) | KG16(kq & 3)) * 16) = rw[i];
            else *reinterpret_cast<f32x4*>(d + row * ROWP + kq * 16) = rw[i];
        }
    };

    if constexpr (MF16) {
        constexpr int NU = 2 * TM, NV = 2 * TN;          // pixel rows (16 columns each) and 16-channel tiles of this wave
        f32x4 acc[NV][NU];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[v][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int i16 = lane & 15, g16 = lane >> 4, pcol = PI16(i16);
        const int ntaps = KH * KW;
        const char* xbase = sP + ((wm * NU) * PW + pcol) * ROWP + KG16(g16) * 16;            // pixel row u: + u * PW * ROWP
        const int wrow0 = (wn * TN * 32 + PI16(i16)) * ROWP + KG16(g16) * 16;                 // channel tile v: + 16 v * ROWP
        const int c_beg = SPLITK ? (int)blockIdx.y * a.nci_split : 0;
        const int c_end = SPLITK ? min(c_beg + a.nci_split, a.nci) : a.nci;
        load_patch(c_beg);
        load_w(c_beg);
        int wbuf = 0;
        for (int c = c_beg; c < c_end; ++c) {
            __syncthreads();                     // previous chunk's taps are done with sP
            store_patch();
            store_w(wbuf);
            __syncthreads();
            if (c + 1 < c_end) load_patch(c + 1);        // lands during this chunk's taps
            for (int tap = 0; tap < ntaps; ++tap) {
                const bool last = tap + 1 == ntaps;
                const int next_kc = last ? (c + 1) : (tap + 1) * a.nci + c;
                if (!(last && c + 1 == c_end)) load_w(last ? c + 1 : next_kc);
                if (PIN) __builtin_amdgcn_sched_barrier(0);
                const int dy = tap / KW, dx = tap - dy * KW;
                const char* pa = xbase + (dy * PW + dx) * ROWP;
                const char* cW = sW + wbuf * BN * ROWP + wrow0;
                f16x8 w0[NV], w1[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    w0[v] = *reinterpret_cast<const f16x8*>(cW + v * 16 * ROWP);
                    if (TERMS == 3) w1[v] = *reinterpret_cast<const f16x8*>(cW + v * 16 * ROWP + 64);
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const f16x8 xa = *reinterpret_cast<const f16x8*>(pa + u * PW * ROWP);
                    f16x8 xb;
                    if (TERMS == 3) xb = *reinterpret_cast<const f16x8*>(pa + u * PW * ROWP + 64);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[v], xa, acc[v][u], 0, 0, 0);
                        if (TERMS == 3) {
                            acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[v], xa, acc[v][u], 0, 0, 0);
                            acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[v], xb, acc[v][u], 0, 0, 0);
                        }
                    }
                }
                if (!last) {                  // next tap's weights go to the other buffer
                    if (WB == 1) __syncthreads();     // ... or, with one buffer, wait until everybody has read this tap's
                    store_w(WB == 1 ? 0 : wbuf ^ 1);
                    __syncthreads();
                    if (WB == 2) wbuf ^= 1;
                }
            }
            if (WB == 2) wbuf ^= 1;
        }
        // epilogue: acc[v][u][r] = channel n4 + r (n4 = tile base + 4 g16) of pixel (row u, column pcol)
        const int x = x0 + pcol;
        const bool vec_y = (p.y_ld & 3) == 0 && ff::aligned16(p.y);
        const bool vec_r = !p.res || ((p.res_ld & 3) == 0 && ff::aligned16(p.res));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int n4 = n0 + wn * TN * 32 + v * 16 + g16 * 4;
            if (n4 >= p.Cout) continue;
            const bool full = n4 + 3 < p.Cout;
            if constexpr (SPLITK) {       // raw partial sums [split][pixel][Cout]
                float* ws = p.splitk_ws + (long long)blockIdx.y * pix_total * p.Cout;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int y = y0 + wm * NU + u;
                    if (y >= H || x >= W) continue;
                    float* d = ws + (((long long)bimg * H + y) * W + x) * p.Cout + n4;
                    if (full && (p.Cout & 3) == 0) *reinterpret_cast<f32x4*>(d) = acc[v][u] * xinv;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n4 + r < p.Cout) d[r] = acc[v][u][r] * xinv;
                    }
                }
                continue;
            }
            f32x4 bias = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f}, ct = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = min(n4 + r, p.Cout - 1);
                if (p.bias) bias[r] = p.bias[n];
                if (p.ch_scale) { cs[r] = p.ch_scale[n]; ct[r] = p.ch_shift[n]; }
            }
            f32x4 vv[NU], rr[NU];
            long long po[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int y = y0 + wm * NU + u;
                po[u] = (y < H && x < W) ? ((long long)bimg * H + y) * W + x : -1;
                f32x4 t = acc[v][u] * xinv + bias;
                t *= p.out_scale;
                if (p.ch_scale) t = t * cs + ct;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r], p.act);
                vv[u] = t;
            }
            if (p.res) {                  // all residual loads of the tile column together, then add + store
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    rr[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (po[u] < 0) continue;
                    const float* rp2 = p.res + po[u] * p.res_ld + n4;
                    if (full && vec_r) rr[u] = *reinterpret_cast<const f32x4*>(rp2);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n4 + r < p.Cout) rr[u][r] = rp2[r];
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[u][r] = ff::apply_act(vv[u][r] + rr[u][r], p.act_res);
            }
            if constexpr (EPI == FF_EP_GRU_RH) {          // [z | r] -> [z | r * h]: channels >= ep_split times ep_a (update.py:47-48)
                if (n4 >= p.ep_split) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) rr[u] = po[u] >= 0 ? *reinterpret_cast<const f32x4*>(p.ep_a + po[u] * p.ep_a_ld + (n4 - p.ep_split)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < NU; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[u][r] = __fmul_rn(vv[u][r], rr[u][r]);
                }
            }
            if constexpr (EPI == FF_EP_GRU_BLEND) {       // v = tanh(q) -> (1 - z) h + z v (update.py:49), z = ep_a, h = ep_b
                f32x4 zz[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    zz[u] = po[u] >= 0 ? *reinterpret_cast<const f32x4*>(p.ep_a + po[u] * p.ep_a_ld + n4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    rr[u] = po[u] >= 0 ? *reinterpret_cast<const f32x4*>(p.ep_b + po[u] * p.ep_b_ld + n4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        vv[u][r] = __fadd_rn(__fmul_rn(__fsub_rn(1.f, zz[u][r]), rr[u][r]), __fmul_rn(zz[u][r], vv[u][r]));
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (po[u] < 0) continue;
                float* d = p.y + po[u] * p.y_ld + n4;
                if (full && vec_y) *reinterpret_cast<f32x4*>(d) = vv[u];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n4 + r < p.Cout) d[r] = vv[u][r];
                }
            }
        }
        return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane -> output pixel of m-tile t: rows 2t (lanes 0-15) and 2t+1 (lanes 16-31), columns 0..15
    // A ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): a group mixes columns of
    // both pixel rows of the m-tile, and at the 144-byte pitch two patch rows share banks iff their indices agree
    // mod 16.  The second pixel row sits PW = 16 + KW-1 rows further on, so its lanes take their columns rotated by
    // KW-1: every group then reads 16 rows that differ mod 16 (it was a 2-way conflict in every group: 35 % of the
    // LDS cycles).  Tap and m-tile offsets shift all lanes alike and keep that.
    const int li = lane & 31, lh = lane >> 5;
    const int rot = (KW - 1) & 15;
    const int lrow = li >> 4, lcol = ((li & 15) - lrow * rot) & 15;
    const int ntaps = KH * KW;
    const int brow = wn * TN * 32 + li;      // weight LDS row of this lane's first output channel (n-tile j: + 32 j)
    const char* abase[TM];                    // this lane's patch row for tap (0, 0), k-half lh, per m-tile
#pragma unroll
    for (int t = 0; t < TM; ++t) abase[t] = sP + (((wm * TM + t) * 2 + lrow) * PW + lcol) * ROWP + lh * 16;

    f16x8 abl_f;                               // ABL 5 only
#pragma unroll
    for (int i = 0; i < 8; ++i) abl_f[i] = (_Float16)(0.37f + 0.01f * (float)((lane * 7 + i * 3) & 31));
    const int c_beg = SPLITK ? (int)blockIdx.y * a.nci_split : 0;
    const int c_end = SPLITK ? min(c_beg + a.nci_split, a.nci) : a.nci;
    load_patch(c_beg);
    load_w(c_beg);
    int wbuf = 0;
    for (int c = c_beg; c < c_end; ++c) {
        __syncthreads();                     // previous chunk's taps are done with sP
        if (ABL != 6 || c == c_beg) store_patch();       // ABL 6: timing only, the patch is staged once
        store_w(wbuf);
        __syncthreads();
        if (c + 1 < c_end && ABL != 6) load_patch(c + 1);        // lands during this chunk's taps
        for (int tap = 0; tap < ntaps; ++tap) {
            const bool last = tap + 1 == ntaps;
            const int next_kc = last ? (c + 1) : (tap + 1) * a.nci + c;   // K order = (tap, ci): chunk index tap*nci + c
            if (!(last && c + 1 == c_end) && ABL != 1 && ABL != 2) load_w(last ? c + 1 : next_kc);
            // Keep the next tap's weight loads HERE: left alone, the scheduler sinks them below this tap's MFMAs
            // (8 fewer live registers) to two MFMAs before the store that waits for them - an L2 round trip per tap
            // with nothing to hide it.
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            const int dy = tap / KW, dx = tap - dy * KW;
            const int tapoff = (dy * PW + dx) * ROWP;              // block-uniform
            const char* cW = sW + wbuf * BN * ROWP + brow * ROWP + lh * 16;      // this lane's weight row, k-half lh
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 a0[TM], a1[TM], b0[TN], b1[TN];
                if constexpr (ABL == 5) {      // timing only: no fragment reads, the MFMAs run on dense garbage kept in a register
#pragma unroll
                    for (int t = 0; t < TM; ++t) { a0[t] = abl_f; a1[t] = abl_f; }
#pragma unroll
                    for (int j = 0; j < TN; ++j) { b0[j] = abl_f; b1[j] = abl_f; }
                    asm volatile("" : "+v"(abl_f));
                } else {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    const char* pa = abase[t] + tapoff;            // + k-slice s (32 B) and term (64 B) as immediates
                    a0[t] = *reinterpret_cast<const f16x8*>(pa + s * 32);
                    if (TERMS == 3) a1[t] = *reinterpret_cast<const f16x8*>(pa + 64 + s * 32);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    b0[j] = *reinterpret_cast<const f16x8*>(cW + j * 32 * ROWP + s * 32);
                    if (TERMS == 3) b1[j] = *reinterpret_cast<const f16x8*>(cW + j * 32 * ROWP + 64 + s * 32);
                }
                }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (ABL == 3) {      // no MFMA: keep operands alive
                            asm volatile("" :: "v"(a0[t]), "v"(b0[j]));
                            if (TERMS == 3) asm volatile("" :: "v"(a1[t]), "v"(b1[j]));
                            continue;
                        }
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[t], b0[j], acc[t][j], 0, 0, 0);
                        if (TERMS == 3) {
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[t], b1[j], acc[t][j], 0, 0, 0);
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b0[j], acc[t][j], 0, 0, 0);
                        }
                    }
            }
            if (!last && ABL != 2) {          // next tap's weights go to the other buffer
                if (WB == 1 && ABL != 4) __syncthreads(); // ... or, with one buffer, wait until everybody has read this tap's
                store_w(WB == 1 ? 0 : wbuf ^ 1);
                if (ABL != 4) __syncthreads();            // ABL 4: timing only, no per-tap barriers (WRONG results)
                if (WB == 2) wbuf ^= 1;
            }
        }
        if (WB == 2) wbuf ^= 1;               // the chunk-boundary store_w(wbuf) above targets the free buffer
    }

    // epilogue: acc[t][j][r]: column n = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*lh = pixel index within the m-tile
    if constexpr (SPLITK) {           // raw partial sums [split][pixel][Cout]; bias, scale, activation, residual: splitk_finish_kernel
        float* ws = p.splitk_ws + (long long)blockIdx.y * pix_total * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
            if (n >= p.Cout) continue;
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int y = y0 + (wm * TM + t) * 2 + (pi >> 4), x = x0 + ((pi - (pi >> 4) * rot) & 15);
                    if (y < H && x < W) ws[(((long long)bimg * H + y) * W + x) * p.Cout + n] = acc[t][j][r] * xinv;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + li;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float cs = p.ch_scale ? p.ch_scale[n] : 1.f;
        const float ct = p.ch_scale ? p.ch_shift[n] : 0.f;
        const bool split_out = p.y_fmt == FF_FMT_SPLIT && n >= p.y_fmt_from;
        float st_p = 0.f, st_s1 = 0.f, st_s2 = 0.f, st_n = 0.f;      // STATS: this lane's channel over its TM x 16 pixels
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            // Three passes per tile: values first, then ALL residual loads of the tile together (res may alias y as
            // far as the compiler knows: inside the store loop they become 16 serial load -> store round trips per
            // lane), then add + store.
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][j][r] * xinv + bias;
                v *= p.out_scale;
                if (p.ch_scale) v = v * cs + ct;
                vv[r] = ff::apply_act(v, p.act);
            }
            if (p.res) {
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int y = y0 + (wm * TM + t) * 2 + (pi >> 4), x = x0 + ((pi - (pi >> 4) * rot) & 15);
                    rr[r] = (y < H && x < W) ? p.res[(((long long)bimg * H + y) * W + x) * p.res_ld + n] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = ff::apply_act(vv[r] + rr[r], p.act_res);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;        // 0..31: row = pi>>4, col = pi&15 rotated as the loads
                const int y = y0 + (wm * TM + t) * 2 + (pi >> 4), x = x0 + ((pi - (pi >> 4) * rot) & 15);
                if (y >= H || x >= W) continue;
                const long long pix = ((long long)bimg * H + y) * W + x;
                if (split_out) ff::store_split1(p.y + pix * p.y_ld, n, vv[r]);      // FF_FMT_SPLIT output (the consumer is conv_dma.hip)
                else p.y[pix * p.y_ld + n] = vv[r];
            }
            if constexpr (STATS) {
                if (y0 + TH <= H && x0 + TW <= W) {      // block-uniform: every pixel of the tile is inside the image
                    if (t == 0) st_p = vv[0];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = vv[r] - st_p;
                        st_s1 += d;
                        st_s2 = fmaf(d, d, st_s2);
                    }
                    st_n += 16.f;
                } else {                                  // ragged tile: the pivot is the lane's first value inside the image
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int y = y0 + (wm * TM + t) * 2 + (pi >> 4), x = x0 + ((pi - (pi >> 4) * rot) & 15);
                        if (y >= H || x >= W) continue;
                        if (st_n == 0.f) st_p = vv[r];
                        const float d = vv[r] - st_p;
                        st_s1 += d;
                        st_s2 = fmaf(d, d, st_s2);
                        st_n += 1.f;
                    }
                }
            }
        }
        if constexpr (STATS) {
            // the upper half-wave holds the same channel over other pixels: fold it into the lower one, re-centred on the
            // lower pivot (d is a difference of two outputs - small against the values when it matters):
            //   sum(v - p) = s + n d,  sum((v - p)^2) = q + 2 d s + n d^2   for entries around p + d
            const float p2 = __shfl_xor(st_p, 32), s2 = __shfl_xor(st_s1, 32), q2 = __shfl_xor(st_s2, 32), n2 = __shfl_xor(st_n, 32);
            if (lh == 0) {
                const float pv = st_n > 0.f ? st_p : p2, d = p2 - pv;
                const float s1 = st_s1 + s2 + n2 * d;
                const float q1 = st_s2 + q2 + 2.f * d * s2 + n2 * d * d;
                // entry [image][part = (tile, wm)][channel]: 32 lanes write 512 contiguous bytes
                const int part = (ty * a.tiles_x + tx) * 2 + wm;
                const int nparts = a.tiles_y * a.tiles_x * 2;
                *reinterpret_cast<f32x4*>(p.stats_part + (((long long)bimg * nparts + part) * p.Cout + n) * 4) = (f32x4){pv, s1, q1, st_n + n2};
            }
        }
    }
}

template <int TERMS, int NITEM, int TM, int TN, int ABL = 0, int WB = 2>
__global__ __launch_bounds__(256) void conv_patch_kernel(const PArgs a) { conv_patch_body<TERMS, NITEM, TM, TN, ABL, WB>(a); }
// one weight buffer: 35 KB of LDS, so four blocks fit a CU if the registers allow four waves per SIMD
template <int TERMS, int NITEM, int TM, int TN, int OCC, bool PIN = true, int ABL = 0, bool INORM = false, bool SPLITK = false, bool MF16 = false, int EPI = 0, bool STATS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_patch_kernel_occ(const PArgs a) { conv_patch_body<TERMS, NITEM, TM, TN, ABL, 1, PIN, INORM, SPLITK, MF16, EPI, STATS>(a); }

// sum of the K splits in a fixed order, then the epilogue of conv_patch_body (same operations in the same order)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const FFConvParams p, int splits, long long npix) {
    const long long total = npix * p.Cout;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / p.Cout;
        const int n = (int)(i - m * p.Cout);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += p.splitk_ws[(long long)s * total + i];
        v += p.bias ? p.bias[n] : 0.f;
        v *= p.out_scale;
        if (p.ch_scale) v = v * p.ch_scale[n] + p.ch_shift[n];
        v = ff::apply_act(v, p.act);
        if (p.res) v = ff::apply_act(v + p.res[m * p.res_ld + n], p.act_res);
        p.y[m * p.y_ld + n] = v;
    }
}

template <int TERMS, int NITEM, int TM, int TN, int OCC>
int launch_occ(const PArgs& a, size_t lds, hipStream_t s) {
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    // FF_MFMA16=1: the 16x16x32 matrix instruction instead of 32x32x16.  Opt-in: parity-green, and the bare LDS-fed loop is
    // 16-24 % faster on the small shape (tools/proto/mfma_shape.hip), but the whole kernel is not bound by its matrix
    // loop: per-layer times are equal within 1 % (12.70 vs 12.73 ms of convolutions per step) and the normalise-on-load
    // variant spills (+13 % on its layers).
    static const bool mf16 = ff::tune_env("FF_MFMA16") && atoi(ff::tune_env("FF_MFMA16")) == 1;
    if (a.p.stats_part) {        // partial InstanceNorm statistics of the output from the epilogue (validated by the caller)
        if (a.p.in_scale) conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, true, false, false, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        else conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, false, false, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch, statistics)");
    }
    if (a.p.ep_mode) {           // GRU step in the epilogue: the 16x16x32 variants carry it (four channels per lane: 16-byte operand loads)
        if (a.p.ep_mode == FF_EP_GRU_RH) conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, false, true, FF_EP_GRU_RH><<<(unsigned)blocks, 256, lds, s>>>(a);
        else conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, false, true, FF_EP_GRU_BLEND><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch, GRU epilogue)");
    }
    if (mf16) {
        if (a.p.splitk > 1) {
            const int splits = (a.nci + a.nci_split - 1) / a.nci_split;
            conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, true, true><<<dim3((unsigned)blocks, splits), 256, lds, s>>>(a);
            const long long npix = (long long)a.p.B * a.p.H * a.p.W, total = npix * a.p.Cout;
            splitk_finish_kernel<<<(unsigned)std::min<long long>((total + 255) / 256, 2048), 256, 0, s>>>(a.p, splits, npix);
            return ff::check_launch("ff_conv2d_fwd(patch, split-K)");
        }
        if (a.p.in_scale) conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, true, false, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        else conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, false, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch)");
    }
    if (a.p.splitk > 1) {        // validated by the caller: no in_scale, no res2, workspace present
        const int splits = (a.nci + a.nci_split - 1) / a.nci_split;
        conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, false, true><<<dim3((unsigned)blocks, splits), 256, lds, s>>>(a);
        const long long npix = (long long)a.p.B * a.p.H * a.p.W, total = npix * a.p.Cout;
        splitk_finish_kernel<<<(unsigned)std::min<long long>((total + 255) / 256, 2048), 256, 0, s>>>(a.p, splits, npix);
        return ff::check_launch("ff_conv2d_fwd(patch, split-K)");
    }
    static const bool pin = !(ff::tune_env("FF_PATCH_PIN") && atoi(ff::tune_env("FF_PATCH_PIN")) == 0);      // A/B switch
    if (a.p.in_scale) {          // normalise-on-load variant
        conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch)");
    }
#ifdef FF_LAB
    // Timing-only ablations of the high-occupancy kernels (WRONG results; lab build only), FF_PATCH_ABLATE = 4 no per-tap
    // barriers, 11 no weight loads, 12 no weight loads / stores / per-tap barriers, 13 no MFMAs, 15 no LDS fragment reads,
    // 16 the patch is staged once per block instead of once per chunk.  Measured table: DESIGN.md section 4.
    static const int oabl = getenv("FF_PATCH_ABLATE") ? atoi(getenv("FF_PATCH_ABLATE")) : 0;
#define FF_OCC_ABL(ENV_, ABL_) \
    if (oabl == ENV_) { conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, true, ABL_><<<(unsigned)blocks, 256, lds, s>>>(a); return ff::check_launch("ff_conv2d_fwd(patch)"); }
    FF_OCC_ABL(4, 4) FF_OCC_ABL(11, 1) FF_OCC_ABL(12, 2) FF_OCC_ABL(13, 3) FF_OCC_ABL(15, 5) FF_OCC_ABL(16, 6)
#undef FF_OCC_ABL
#endif
    if (!pin) {
        conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC, false><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch)");
    }
    conv_patch_kernel_occ<TERMS, NITEM, TM, TN, OCC><<<(unsigned)blocks, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_fwd(patch)");
}

// The same high-occupancy variants for the one-term reduced-precision mode (FF_W_F16): plain, normalise-on-load, statistics,
// GRU epilogues, split-K - no ablations.
template <int NITEM, int TM, int TN, int OCC>
int launch_occ_f16(const PArgs& a, size_t lds, hipStream_t s) {
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    if (a.p.stats_part) {
        if (a.p.in_scale) conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, true, false, false, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        else conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, false, false, false, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch, f16, statistics)");
    }
    if (a.p.ep_mode) {
        if (a.p.ep_mode == FF_EP_GRU_RH) conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, false, false, true, FF_EP_GRU_RH><<<(unsigned)blocks, 256, lds, s>>>(a);
        else conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, false, false, true, FF_EP_GRU_BLEND><<<(unsigned)blocks, 256, lds, s>>>(a);
        return ff::check_launch("ff_conv2d_fwd(patch, f16, GRU epilogue)");
    }
    if (a.p.splitk > 1) {
        const int splits = (a.nci + a.nci_split - 1) / a.nci_split;
        conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, false, true><<<dim3((unsigned)blocks, splits), 256, lds, s>>>(a);
        const long long npix = (long long)a.p.B * a.p.H * a.p.W, total = npix * a.p.Cout;
        splitk_finish_kernel<<<(unsigned)std::min<long long>((total + 255) / 256, 2048), 256, 0, s>>>(a.p, splits, npix);
        return ff::check_launch("ff_conv2d_fwd(patch, f16, split-K)");
    }
    if (a.p.in_scale) conv_patch_kernel_occ<1, NITEM, TM, TN, OCC, true, 0, true><<<(unsigned)blocks, 256, lds, s>>>(a);
    else conv_patch_kernel_occ<1, NITEM, TM, TN, OCC><<<(unsigned)blocks, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_fwd(patch, f16)");
}

template <int TERMS, int NITEM, int TM, int TN, int ABL = 0, int WB = 2>
int launch(const PArgs& a, size_t lds, hipStream_t s) {
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<TERMS, NITEM, TM, TN, ABL, WB>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        once = true;
    }
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    conv_patch_kernel<TERMS, NITEM, TM, TN, ABL, WB><<<(unsigned)blocks, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_fwd(patch)");
}

}  // namespace

namespace ff {
// returns FF_OK if launched, 1 if the shape is not eligible (caller falls back to conv_split)
int conv2d_fwd_patch(const FFConvParams& p, int cin, hipStream_t s) {
    static const bool enabled = !ff::tune_env("FF_NO_PATCH_CONV");
    static const bool dbg = ff::tune_env("FF_DEBUG_DISPATCH") != nullptr;
    if (dbg) fprintf(stderr, "[ff] patch? en=%d stride=%d dil=%d,%d groups=%d k=%dx%d pad=%d,%d cin=%d xc=%d,%d,%d fmt=%d\n", (int)enabled, p.stride,
                     p.dil_h, p.dil_w, p.groups, p.KH, p.KW, p.pad_h, p.pad_w, cin, p.x_c[0], p.x_c[1], p.x_c[2], p.w_format);
    if (!enabled) return 1;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1) return 1;
    if (p.KH % 2 == 0 || p.KW % 2 == 0 || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 || p.KH > 7 || p.KW > 7) return 1;
    if (p.KH * p.KW < 3) return 1;          // 1x1: nothing to reuse, the plain kernel is as good
    if (cin % 32) return 1;
    long long max_bytes = 0;
    for (int i = 0; i < FF_MAX_SEG; ++i) {
        if (p.x_c[i] % 32) return 1;
        if (p.x_c[i]) max_bytes = std::max(max_bytes, (long long)p.B * p.H * p.W * p.x_ld[i] * 4);
    }
    PArgs a;
    a.p = p;
    a.Cin = cin;
    a.nci = cin / 32;
    a.tiles_x = (p.W + TW - 1) / TW;
    a.w_row_bytes = (long long)((p.KH * p.KW * cin + 31) / 32) * ROWB;
    max_bytes = std::max(max_bytes, (long long)p.Cout * a.w_row_bytes);
    if (max_bytes >= (1ll << 31)) return 1;
    // Tile choice: th = 8 rows x 16 columns x 64 channels per block unless that leaves the chip under-filled (update
    // block at 1/8 resolution: 4 rows).  The larger register tiles (FF_PATCH_TH=16: 256 pixels; FF_PATCH_TN=2: 128
    // channels) exist for measurement only - they halve weight or LDS traffic per MFMA but run at 2 blocks per CU and
    // lose 13-18 %.
    static const int force_th = ff::tune_env("FF_PATCH_TH") ? atoi(ff::tune_env("FF_PATCH_TH")) : 0;
    static const int force_tn = ff::tune_env("FF_PATCH_TN") ? atoi(ff::tune_env("FF_PATCH_TN")) : 0;
    auto nblocks = [&](int th, int tn) { return (long long)p.B * ((p.H + th - 1) / th) * a.tiles_x * ((p.Cout + 64 * tn - 1) / (64 * tn)); };
    int th = nblocks(8, 1) < 512 ? 4 : 8, tn = 1;
    if (force_th) th = force_th;
    if (force_tn) tn = force_tn;
    a.tiles_y = (p.H + th - 1) / th;
    a.n_tiles = (p.Cout + 64 * tn - 1) / (64 * tn);
    const int npix = (th + p.KH - 1) * (TW + p.KW - 1);
    static const int lds_pad = ff::tune_env("FF_PATCH_LDS_PAD") ? atoi(ff::tune_env("FF_PATCH_LDS_PAD")) : 0;   // occupancy experiments
    const int nitem = (npix * 8 + 255) / 256;
    const bool t3 = p.w_format == FF_W_F16X3;
    // Occupancy is what this kernel responds to (3 -> 2 blocks per CU: +15-25 % time; 3 -> 4: -5-10 %): with ONE weight
    // buffer (a second barrier per tap instead) the 8-row tile needs 35 KB of LDS and the 4-row tile 25 KB, so 4 / 5
    // blocks fit a CU once the registers are capped to match (amdgpu_waves_per_eu).
    static const int wb1 = ff::tune_env("FF_PATCH_WB1") ? atoi(ff::tune_env("FF_PATCH_WB1")) : 3;   // bit 0: 8-row tiles, bit 1: 4-row tiles
    // (the one-term reduced-precision mode runs the same high-occupancy variants: 496 -> 559 pairs/s end to end)
    const bool occ = (t3 || p.w_format == FF_W_F16) && tn == 1 && ((th == 8 && nitem <= 6 && (wb1 & 1)) || (th == 4 && nitem <= 4 && (wb1 & 2)));
    const size_t lds = ((npix * ROWP + 255) & ~255) + (occ ? 1 : 2) * 64 * tn * ROWP + lds_pad;
    if (lds > 96 * 1024) return 1;
#ifdef FF_LAB      // timing-only ablations (WRONG results): lab build only (tools/build_lab.sh), not in libfocusflow_hip.so
    static const int abl = getenv("FF_PATCH_ABLATE") ? atoi(getenv("FF_PATCH_ABLATE")) : 0;
    if (abl >= 1 && abl <= 3 && th == 8 && tn == 1 && nitem <= 6 && t3) {
        if (abl == 1) return launch<3, 6, 2, 1, 1>(a, lds, s);
        if (abl == 2) return launch<3, 6, 2, 1, 2>(a, lds, s);
        return launch<3, 6, 2, 1, 3>(a, lds, s);
    }
#endif
    // split-pair output (y_fmt; y2 belongs to split-pair inputs = conv_dma.hip): the 32x32x16 epilogue of the high-occupancy
    // variants writes it; everything else declines (the im2col kernel takes the convolution)
    if (p.y_fmt != FF_FMT_F32) {
        static const bool mf16 = ff::tune_env("FF_MFMA16") && atoi(ff::tune_env("FF_MFMA16")) == 1;
        if (!occ || p.ep_mode || p.splitk > 1 || mf16) return 1;
    }
    // split-K (FFConvParams.splitk, see conv2d_splitk_hint): the 4-row high-occupancy variant only
    if (!(occ && th == 4 && p.splitk > 1 && p.splitk_ws && !p.in_scale && !p.res2)) a.p.splitk = 0;
    a.nci_split = a.p.splitk > 1 ? (a.nci + a.p.splitk - 1) / a.p.splitk : a.nci;
    if (p.stats_part && (!occ || a.p.splitk > 1 || p.ep_mode))
        return ff::fail(FF_EINVAL, "ff_conv2d_fwd: stats_part: ask ff_conv2d_stats_parts first (this convolution cannot produce statistics)");
    if (p.ep_mode) {             // validated by ff_conv2d_fwd; the high-occupancy variants carry the GRU epilogues
        if (!occ || a.p.splitk > 1 || p.in_scale || p.Cout % 4) return ff::fail(FF_EINVAL, "ff_conv2d_fwd: ep_mode needs the f16x3 patch kernel's 8x16 / 4x16 tiles (got %dx%d kernel, Cin %d, Cout %d)", p.KH, p.KW, cin, p.Cout);
    }
    if (occ && !t3) return th == 8 ? launch_occ_f16<6, 2, 1, 4>(a, lds, s) : launch_occ_f16<4, 1, 1, 5>(a, lds, s);
    if (occ) return th == 8 ? launch_occ<3, 6, 2, 1, 4>(a, lds, s) : launch_occ<3, 4, 1, 1, 5>(a, lds, s);
    if (p.in_scale) return 1;            // only the two variants above normalise while loading (the caller fails loudly)
#define FF_PATCH_CASE(TH_, TN_, NI_) \
    if (th == TH_ && tn == TN_ && nitem <= NI_) return t3 ? launch<3, NI_, TH_ / 4, TN_>(a, lds, s) : launch<1, NI_, TH_ / 4, TN_>(a, lds, s);
    FF_PATCH_CASE(8, 1, 6) FF_PATCH_CASE(8, 1, 10)
    FF_PATCH_CASE(4, 1, 4) FF_PATCH_CASE(4, 1, 8)
    FF_PATCH_CASE(8, 2, 6) FF_PATCH_CASE(8, 2, 10)
    FF_PATCH_CASE(4, 2, 4) FF_PATCH_CASE(4, 2, 8)
    FF_PATCH_CASE(16, 1, 11) FF_PATCH_CASE(16, 1, 16)
#undef FF_PATCH_CASE
    return 1;
}

// Entries per image and channel of FFConvParams.stats_part for this convolution (0: it cannot produce statistics - the
// caller runs ff_norm_stats over the output instead).  Same eligibility and tile choice as conv2d_fwd_patch's
// high-occupancy variants: 2 entries (one per wave row) per 8x16 / 4x16 tile.
int conv2d_stats_parts(const FFConvParams& p, int cin) {
    static const bool enabled = !ff::tune_env("FF_NO_PATCH_CONV") && !(getenv("FF_CONV_STATS") && atoi(getenv("FF_CONV_STATS")) == 0);
    if (!enabled || (p.w_format != FF_W_F16X3 && p.w_format != FF_W_F16) || p.res2 || p.ep_mode || p.x_amax) return 0;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1) return 0;
    if (p.KH % 2 == 0 || p.KW % 2 == 0 || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 || p.KH > 7 || p.KW > 7 || p.KH * p.KW < 3) return 0;
    if (cin % 32) return 0;
    for (int i = 0; i < FF_MAX_SEG; ++i)
        if (p.x_c[i] % 32) return 0;
    if (ff::tune_env("FF_PATCH_TH") || ff::tune_env("FF_PATCH_TN") || ff::tune_env("FF_PATCH_WB1")) return 0;
    long long max_bytes = (long long)p.Cout * ((p.KH * p.KW * cin + 31) / 32) * ROWB;
    for (int i = 0; i < FF_MAX_SEG; ++i)
        if (p.x_c[i]) max_bytes = std::max(max_bytes, (long long)p.B * p.H * p.W * p.x_ld[i] * 4);
    if (max_bytes >= (1ll << 31)) return 0;
    const int tiles_x = (p.W + TW - 1) / TW, n_tiles = (p.Cout + 63) / 64;
    const int th = (long long)p.B * ((p.H + 7) / 8) * tiles_x * n_tiles < 512 ? 4 : 8;
    const int npix = (th + p.KH - 1) * (TW + p.KW - 1), nitem = (npix * 8 + 255) / 256;
    if (!((th == 8 && nitem <= 6) || (th == 4 && nitem <= 4))) return 0;
    if (th == 4 && p.splitk > 1 && p.splitk_ws) return 0;
    return ((p.H + th - 1) / th) * tiles_x * 2;
}

// How many K splits ff_conv2d_fwd would use for this convolution if given a workspace (0: none).  Same eligibility as
// conv2d_fwd_patch's 4-row variant; worth it when the plane is so small that the blocks do not even cover the CUs and
// the reduction is long: every split needs >= 2 chunks, the grid is brought to ~768 blocks at most.
int conv2d_splitk_hint(const FFConvParams& p, int cin) {
    static const bool enabled = !ff::tune_env("FF_NO_PATCH_CONV") && !(ff::tune_env("FF_SPLITK") && atoi(ff::tune_env("FF_SPLITK")) == 0);
    if (!enabled || p.w_format != FF_W_F16X3 || p.in_scale || p.res2) return 0;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1) return 0;
    if (p.KH % 2 == 0 || p.KW % 2 == 0 || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 || p.KH * p.KW < 3) return 0;
    if (cin % 32) return 0;
    for (int i = 0; i < FF_MAX_SEG; ++i)
        if (p.x_c[i] % 32) return 0;
    const int tiles_x = (p.W + TW - 1) / TW, n_tiles = (p.Cout + 63) / 64;
    if ((long long)p.B * ((p.H + 7) / 8) * tiles_x * n_tiles >= 512) return 0;            // the dispatcher takes 8-row tiles
    const int npix = (4 + p.KH - 1) * (TW + p.KW - 1);
    if ((npix * 8 + 255) / 256 > 4) return 0;
    const long long blocks = (long long)p.B * ((p.H + 3) / 4) * tiles_x * n_tiles;
    const int nci = cin / 32;
    // long reductions only (more than 72 tap steps per block): below that the finishing launch and the workspace cost a
    // host-bound caller more than the kernel gains (FF-RAFT's update block at one pair per step: 7.3 -> 8.0 ms eager)
    // p.splitk < 0 on input = the caller is being captured into a hipGraph (no host cost per launch): split from 36 steps
    if (blocks > 256 || nci < 6 || nci * p.KH * p.KW <= (p.splitk < 0 ? 36 : 72)) return 0;
    const int splits = (int)std::min<long long>(std::min(16, nci / 2), 768 / blocks);
    return splits >= 2 ? splits : 0;
}
}  // namespace ff
