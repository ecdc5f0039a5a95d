// One fusion unit of the Condition Control Encoder, type '1x1conv', both directions (parallel_fusion.py:98-150):
//     img' = img + conv1x1_{mask -> img}(mask) ,   mask' = mask + conv1x1_{img -> mask}(img)
// as ONE bandwidth-shaped launch.  The generic 1x1 route (conv_split.hip, both convolutions as one anti-diagonal 2C x 2C
// matrix) ran these at 3.3-3.6 TB/s: 12 288 short-lived blocks, each a chain of four K chunks = four dependent global
// round trips, half of their MFMAs on the zero blocks, the residual read a second time.  Here:
//   * persistent blocks walk pixel tiles; a thread loads its (pixel, 4 channel) items of BOTH tensors in one batch of
//     16-byte loads, KEEPS the fp32 values in registers - they are the residual - and writes their split pair (x0, x1) to
//     LDS as the matrix operand; the result goes back through LDS into the loader's layout, is added to the registers and
//     leaves as whole 16-byte stores: every input byte is read once, every output byte written once;
//   * only the two C x C blocks that are not zero are multiplied; the weights (fragment order, ff_pack_frag16) stay in
//     registers for the whole launch (waves split the 2C output channels);
//   * LAZY INPUTS: the tensor a fusion unit reads is the output of a memory-bound normalisation pass -
//     relu(InstanceNorm(stem conv)) after the stems, relu(x + relu(InstanceNorm(conv2))) at the end of a residual stage
//     (extractor.py:6-56 / parallel_fusion.py:40-95).  With scale / shift tables (ff_norm_coeffs) and the residual tensor
//     given, the loader computes that value itself - the same operations as ff_norm_apply: one fma, relu, add, relu - so
//     the pass and its write + re-read of the activation disappear (the normalised tensor has no other consumer).
// Matrix form and LDS image are conv_dma.hip's: v_mfma_f32_16x16x32_f16, A = weights, B = 16 pixels; an LDS row is one
// pixel's 32-channel chunk [x0: 32 fp16 | x1: 32 fp16], its eight 16-byte slots XOR-swizzled by (pixel >> 1) & 7, lanes
// mapped to pixels through PI16 (bank-conflict-free fragment reads).  Terms in conv_dma.hip's order (w0 x0, w1 x0, w0 x1),
// K in ascending chunks.  The epilogue is separately rounded operations: acc / 64 + bias, + residual.
// hipcc-flags: -ffp-contract=off
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int PI16(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * (i - 12) + 9); }

struct FArgs {
    FFFusionPair p;
    long long tiles;
};

// C channels per branch, G = C / 4 four-channel groups per pixel: the first PSTEP * G threads (PSTEP = 16 pixels per pass for
// C = 64: all 256; 8 for C = 96: three waves) each own one group of one pixel per pass - pixel PSTEP j + tid / G, group tid % G:
// the pixel step is a compile-time immediate and the group is the thread's own.  TP pixels per tile, TERMS 3 (f16x3) / 1
// (f16).  The tensors are CONTIGUOUS NHWC (ld == C): a tile is TP * C * 4 consecutive bytes and the item of pass j lies at
// byte PSTEP G 16 j + 16 tid: one vector offset per thread, everything else scalar.
template <int C, int TP, int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fusion_pair_kernel(const FArgs a) {
    constexpr int G = C / 4;              // 4-channel groups per pixel and branch
    constexpr int PSTEP = (256 / G) / 8 * 8;   // pixels per pass of the loader threads: 16 (C = 64), 8 (C = 96)
    constexpr int NACT = PSTEP * G;       // loader threads
    constexpr int NI = TP / PSTEP;        // items per thread and branch
    constexpr int NCH = C / 32;           // 32-channel chunks per branch = K chunks of a convolution
    constexpr int NPG = TP / 16;          // 16-pixel groups of a tile
    constexpr int NTW = C / 32;           // 16-channel output tiles per wave: 2C channels over 4 waves
    constexpr int PLANE = TP * 128;       // one (branch, chunk) plane of the operand image
    constexpr int SROW = 2 * C * 4 + 16;  // staging row: one pixel's 2C results, padded (16 pixels of a wave-store hit 16 different bank groups)
    constexpr int TBYTES = TP * C * 4;    // bytes of one tile of one tensor
    static_assert((PSTEP == 8 || PSTEP == 16) && NACT % 64 == 0 && TP % PSTEP == 0 && C % 32 == 0 && TP % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // max(2 NCH PLANE, TP SROW) bytes: operand image, then the result staging
    const FFFusionPair& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g16 = lane >> 4;
    const int grp = tid % G, pl = tid / G;          // this thread's channel group and its pixel inside a pass
    const bool loader = NACT == 256 || tid < NACT;  // whole waves

    // ---- this wave's output channels: NTW tiles of 16 from channel nb of the concatenation [img' | mask']
    const int nb = wave * (C / 2);
    const int ob = nb / C;                // output branch (0 = img', written from the mask input)
    const int ib = 1 - ob;                // input branch of that convolution
    const int lb = nb - ob * C;           // first channel inside the branch
    // The weights stay in registers for the whole launch (C = 64: 32 registers; C = 96: 72, which is why its tiles are 32 pixels:
    // with 64-pixel tiles the loader's values, the lazy residuals and the weights spilled); the non-resident form - re-read from
    // L2 once per tile, issued behind the operand writes - is kept for shapes that would need it
    constexpr bool WRES = NTW * NCH <= 4 || TP <= 32;
    f32x4 wr[NTW][NCH][TERMS == 3 ? 2 : 1];
    f32x4 bias[NTW];
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_frag[ob]), 0, (C / 16) * NCH * 2048, 0x00020000);
    auto load_w = [&]() {          // [C / 16][NCH][term][lane][16 B]
#pragma unroll
        for (int v = 0; v < NTW; ++v) {
            const int tile = lb / 16 + v;
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) {
                wr[v][kc][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (tile * NCH + kc) * 2048, 0));
                if (TERMS == 3) wr[v][kc][TERMS == 3 ? 1 : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (tile * NCH + kc) * 2048 + 1024, 0));
            }
        }
    };
    if (WRES) load_w();
    {
        const float* bp = p.bias[ob];
#pragma unroll
        for (int v = 0; v < NTW; ++v) {
            bias[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bp) bias[v] = *reinterpret_cast<const f32x4*>(bp + lb + v * 16 + g16 * 4);
        }
    }
    const float xinv = ff::SPLIT_INV;
    const int nbytes = (int)(a.tiles * TBYTES);       // < 2^31 (checked by the host)
    __amdgpu_buffer_rsrc_t rx[2], rr[2], ry[2];
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        rx[br] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[br]), 0, nbytes, 0x00020000);
        rr[br] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.xres[br] ? p.xres[br] : p.x[br]), 0, nbytes, 0x00020000);
        ry[br] = __builtin_amdgcn_make_buffer_rsrc(p.y[br], 0, nbytes, 0x00020000);
    }
    const int voff = tid * 16;
    // LDS addresses of this thread's items: pixel pl (+ PSTEP j: an immediate), channels 4 grp ..
    const int cc = (grp * 4) & 31, kcw = (grp * 4) >> 5;
    const int sw0 = (pl >> 1) & 7;                    // pass j: ((pl + PSTEP j) >> 1) & 7 = sw0 for PSTEP = 16, sw0 ^ 4 (j & 1) for PSTEP = 8 - i.e. bit 6 of the offset flips
    const int wofs0 = kcw * PLANE + pl * 128 + (((cc >> 3) ^ sw0) << 4) + (cc & 7) * 2;
    const int wofs1 = kcw * PLANE + pl * 128 + (((4 + (cc >> 3)) ^ sw0) << 4) + (cc & 7) * 2;
    const int sofs = pl * SROW + grp * 16;

    for (long long tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int tb = (int)(tile * TBYTES);
        const int bimg = (int)(tile * TP / p.HW);     // HW % TP == 0: a tile lies inside one image
        // ---- load phase: every item of both branches in flight together
        f32x4 val[2][NI], rsd[2][NI];
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (!loader) continue;
                val[br][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx[br], voff, tb + j * (NACT * 16), 0));
                if (p.xres[br]) rsd[br][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr[br], voff, tb + j * (NACT * 16), 0));
            }
        // ---- the value the unit reads: x, or in_act(fma(x, scale, shift)), or relu(xres + that)  (ff_norm_apply's operations)
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (!p.scale[br] || !loader) continue;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale[br] + (long long)bimg * C + grp * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift[br] + (long long)bimg * C + grp * 4);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o = fmaf(val[br][j][r], sc[r], sh[r]);
                    o = ff::apply_act(o, p.in_act);
                    if (p.xres[br]) { o += rsd[br][j][r]; o = o < 0.f ? 0.f : o; }
                    val[br][j][r] = o;
                }
            }
        }
        // ---- operand image: split pairs into LDS (row = pixel, slot swizzle by (pixel >> 1) & 7)
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (!loader) continue;
                ff::ff_f16x4 h0, h1;
                ff::split_pair4(val[br][j], h0, h1);
                const int jx = (PSTEP == 8 && (j & 1)) ? 64 : 0;
                *reinterpret_cast<ff::ff_f16x4*>(smem + br * NCH * PLANE + j * PSTEP * 128 + (wofs0 ^ jx)) = h0;
                if (TERMS == 3) *reinterpret_cast<ff::ff_f16x4*>(smem + br * NCH * PLANE + j * PSTEP * 128 + (wofs1 ^ jx)) = h1;
            }
        if (!WRES) load_w();
        __syncthreads();
        // ---- the two C x C convolutions: this wave's NTW channel tiles over all pixel groups
        f32x4 acc[NTW][NPG];
#pragma unroll
        for (int v = 0; v < NTW; ++v)
#pragma unroll
            for (int g = 0; g < NPG; ++g) acc[v][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int pxl = PI16(i16), swl = (pxl >> 1) & 7;             // ((16 g + pxl) >> 1) & 7 == (pxl >> 1) & 7
            const char* fa = smem + ib * NCH * PLANE + pxl * 128 + ((g16 ^ swl) << 4);
            const char* fb = smem + ib * NCH * PLANE + pxl * 128 + (((4 + g16) ^ swl) << 4);
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) {
#pragma unroll
                for (int g = 0; g < NPG; ++g) {
                    const f16x8 xa = *reinterpret_cast<const f16x8*>(fa + kc * PLANE + g * 2048);
                    f16x8 xb;
                    if (TERMS == 3) xb = *reinterpret_cast<const f16x8*>(fb + kc * PLANE + g * 2048);
#pragma unroll
                    for (int v = 0; v < NTW; ++v) {
                        const f16x8 w0 = __builtin_bit_cast(f16x8, wr[v][kc][0]);
                        acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa, acc[v][g], 0, 0, 0);
                        if (TERMS == 3) {
                            const f16x8 w1 = __builtin_bit_cast(f16x8, wr[v][kc][TERMS == 3 ? 1 : 0]);
                            acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa, acc[v][g], 0, 0, 0);
                            acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[v][g], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();          // everybody is done reading the operand image: the staging rows take its place
        {
            char* st = smem + PI16(i16) * SROW + (nb + g16 * 4) * 4;
#pragma unroll
            for (int v = 0; v < NTW; ++v)
#pragma unroll
                for (int g = 0; g < NPG; ++g) {
                    const f32x4 t = acc[v][g] * xinv + bias[v];
                    *reinterpret_cast<f32x4*>(st + g * 16 * SROW + v * 64) = t;
                }
        }
        __syncthreads();
        // ---- out = value + convolution of the other branch, in the loader's layout: whole 16-byte stores
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (!loader) continue;
                const f32x4 cv = *reinterpret_cast<const f32x4*>(smem + sofs + j * PSTEP * SROW + br * C * 4);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = cv[r] + val[br][j][r];
                // (vector offset only: a 16-byte buffer store with a scalar offset register followed at once by a write of its
                // data registers - the next pass's v_pk_add_f32 - lost the race on gfx950: the fourth dword of pass j left as
                // pass j + 1's.  No scalar offset, no hazard.)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ry[br], voff + tb + j * (NACT * 16), 0, 0);
            }
        __syncthreads();          // the staging rows are read: the next tile's operand image may overwrite them
    }
}

template <int C, int TP>
int launch(const FArgs& a, hipStream_t s) {
    constexpr int NCH = C / 32;
    constexpr size_t lds = std::max<size_t>((size_t)2 * NCH * TP * 128, (size_t)TP * (2 * C * 4 + 16));
    static const int per_cu = ff::tune_env("FF_FUSION_BLOCKS_PER_CU") ? std::max(1, atoi(ff::tune_env("FF_FUSION_BLOCKS_PER_CU"))) : 2;
    const unsigned blocks = (unsigned)std::min<long long>(a.tiles, 256ll * per_cu);
    if (a.p.w_format == FF_W_F16X3) {
        FF_ALLOW_DYNAMIC_LDS((&fusion_pair_kernel<C, TP, 3>), (int)lds);
        fusion_pair_kernel<C, TP, 3><<<blocks, 256, lds, s>>>(a);
    } else {
        FF_ALLOW_DYNAMIC_LDS((&fusion_pair_kernel<C, TP, 1>), (int)lds);
        fusion_pair_kernel<C, TP, 1><<<blocks, 256, lds, s>>>(a);
    }
    return ff::check_launch("ff_fusion_pair_fwd");
}

}  // namespace

// pixels per tile for C channels per branch (0: this channel count has no instance)
extern "C" int ff_fusion_pair_tile(int C) { return C == 64 ? 128 : (C == 96 ? 32 : 0); }

extern "C" int ff_fusion_pair_fwd(const FFFusionPair* pp, void* stream) {
    FF_REQUIRE(pp, "ff_fusion_pair_fwd: null argument");
    const FFFusionPair& p = *pp;
    const int tp = ff_fusion_pair_tile(p.C);
    FF_REQUIRE(tp > 0, "ff_fusion_pair_fwd: C = %d (64 or 96 channels per branch; other fusion units take ff_conv2d_fwd)", p.C);
    FF_REQUIRE(p.B > 0 && p.HW > 0 && p.HW % tp == 0, "ff_fusion_pair_fwd: B %d, HW %d: pixels per image must be a multiple of %d", p.B, p.HW, tp);
    FF_REQUIRE((long long)p.B * p.HW * p.C * 4 < (1ll << 31), "ff_fusion_pair_fwd: a tensor of 2 GiB or more");
    FF_REQUIRE(p.w_format == FF_W_F16X3 || p.w_format == FF_W_F16, "ff_fusion_pair_fwd: w_format %d (a split weight format)", p.w_format);
    FF_REQUIRE(p.in_act >= FF_ACT_NONE && p.in_act <= FF_ACT_TANH, "ff_fusion_pair_fwd: bad in_act %d", p.in_act);
    for (int i = 0; i < 2; ++i) {
        FF_REQUIRE(p.x[i] && p.y[i] && p.w_frag[i], "ff_fusion_pair_fwd: null tensor (branch %d)", i);
        FF_REQUIRE(p.x_ld[i] == p.C && p.y_ld[i] == p.C && ff::aligned16(p.x[i]) && ff::aligned16(p.y[i]) && ff::aligned16(p.w_frag[i]) && (!p.bias[i] || ff::aligned16(p.bias[i])),
                   "ff_fusion_pair_fwd: contiguous NHWC tensors (leading dimension == C) / 16-byte alignment (branch %d)", i);
        FF_REQUIRE(!p.xres[i] || (p.scale[i] && p.xres_ld[i] == p.C && ff::aligned16(p.xres[i])),
                   "ff_fusion_pair_fwd: xres needs scale / shift, leading dimension == C, 16-byte alignment (branch %d)", i);
        FF_REQUIRE(!p.scale[i] || (p.shift[i] && ff::aligned16(p.scale[i]) && ff::aligned16(p.shift[i])), "ff_fusion_pair_fwd: scale without shift / alignment (branch %d)", i);
        FF_REQUIRE(p.y[i] != p.x[1 - i] && (!p.xres[1 - i] || p.y[i] != p.xres[1 - i]), "ff_fusion_pair_fwd: an output aliases the other branch's input");
    }
    FArgs a;
    a.p = p;
    a.tiles = (long long)p.B * p.HW / tp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.C == 64 ? launch<64, 128>(a, s) : launch<96, 32>(a, s);
}
