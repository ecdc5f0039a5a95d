// A 1x1 convolution over a contiguous fp32 NHWC tensor that leaves in the split-pair format (FF_FMT_SPLIT): the motion encoder's
// convc1 (update.py:89-92: relu(conv1x1(corr)), 324 -> 256 channels; the lookup writes its 324 channels into 352-wide rows), twelve
// times per forward, between the lookup and convc2 (conv_dma.hip).  On the generic im2col route (conv_split.hip) it took 29 us at
// 8 pairs against an MFMA time of 11 us and 60 MB of traffic: 768 blocks walking eleven K chunks, each chunk a dependent global ->
// registers -> LDS round trip behind a barrier.  Here the shape of fusion_pair.hip: persistent blocks (two per CU) over 48-pixel
// tiles; a tile's whole input (48 x 352 floats) is loaded in ONE batch of 16-byte loads, split and written to LDS as the matrix
// operand (conv_dma.hip's image: pixel rows of 128 bytes per 32-channel chunk, slot swizzle, PI16); the eight waves split the 256
// output channels and stream their weights from L2 in fragment order (ff_pack_frag16), three chunks ahead; the result leaves straight
// from the accumulators' lanes (four consecutive channels of a pixel: the two 8-byte stores of a split pair).  While one block of a CU
// multiplies, the other loads.  Terms and K order as conv_dma.hip (w0 x0, w1 x0, w0 x1; ascending chunks); epilogue acc / 64 + bias,
// activation - separately rounded operations.
// PROTOTYPE, not in the library (round 4): correct (it passed its parity tests: fp64, the generic route, ragged tiles, one-term mode), 25 us
// against the generic route's 30 us in isolation, and no measurable gain end to end (613 / 616 -> 616 / 615 pairs/s).  What holds it at
// 25 us: every wave streams its own weight fragments from L2 for 48 pixels only - 512 tiles x 8 waves x 44 KB = 180 MB per launch -
// and the 512 one-tile blocks run their load / multiply / store phases in lockstep.  A 96-pixel tile in two halves (weights reused twice
// as long, the second half's loads under the first half's MFMAs) is the next step; not built.
// hipcc-flags: -ffp-contract=off
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int PI16(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * (i - 12) + 9); }

struct PArgs {
    const float* x;
    const void* w_frag;
    const float* bias;
    float* y;
    long long M;          // pixels
    long long tiles;
    int y_ld, act, Cout;
};

// NCH 32-channel chunks of input (Cin = 32 NCH, rows contiguous: ld == Cin), TP pixels per tile, NW waves, NTW 16-channel output
// tiles per wave (Cout <= 16 NW NTW), TERMS 3 / 1
template <int NCH, int TP, int NW, int NTW, int TERMS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void pointwise_split_kernel(const PArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int G = NCH * 8;                 // 4-channel groups per pixel
    constexpr int ITEMS = TP * G;              // 16-byte items per tile
    constexpr int NI = (ITEMS + NT - 1) / NT;  // passes of the block over a tile
    constexpr int NPG = TP / 16;
    constexpr int PLANE = TP * 128;
    constexpr int TBYTES = TP * G * 16;
    constexpr int NT2 = TERMS == 3 ? 2 : 1;
    static_assert(TP % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NCH planes of TP rows x 128 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g16 = lane >> 4;
    const int ntile16 = (a.Cout + 15) >> 4;
    const long long xbytes = a.M * (long long)(G * 16);               // < 2^31 (checked by the host)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_frag), 0, ntile16 * NCH * 2048, 0x00020000);
    // this wave's weights: tile t = wave NTW + v, fragment (t, kc, term) at ((t NCH + kc) 2 + term) KB; tiles past Cout read zeros
    int wofs[NTW];
    f32x4 bias[NTW];
#pragma unroll
    for (int v = 0; v < NTW; ++v) {
        const int t = wave * NTW + v;
        wofs[v] = t < ntile16 ? t * NCH * 2048 + lane * 16 : 0x7ffff000;
        bias[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = t * 16 + g16 * 4 + r;
            if (a.bias && n < a.Cout) bias[v][r] = a.bias[n];
        }
    }
    const float xinv = ff::SPLIT_INV;
    const int pxl = PI16(i16), swl = (pxl >> 1) & 7;
    const char* fa = smem + pxl * 128 + ((g16 ^ swl) << 4);
    const char* fb = smem + pxl * 128 + (((4 + g16) ^ swl) << 4);

    for (long long tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int tb = (int)(tile * TBYTES);
        // ---- the tile's input, one batch of loads (rows past M read zeros: the buffer's range check)
        f32x4 val[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int e = j * NT + tid;
            val[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((j + 1) * NT <= ITEMS || e < ITEMS) val[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, tid * 16, tb + j * (NT * 16), 0));
        }
        // weights of the first two chunks, in flight with the input
        constexpr int DEPTH = 4;      // register sets: the weights of chunk kc + DEPTH - 1 are requested while chunk kc multiplies
        f32x4 wr[DEPTH][NTW][NT2];
        auto load_w = [&](int set, int kc) {
#pragma unroll
            for (int v = 0; v < NTW; ++v) {
                wr[set][v][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wofs[v], kc * 2048, 0));
                if (TERMS == 3) wr[set][v][NT2 - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wofs[v], kc * 2048 + 1024, 0));
            }
        };
        load_w(0, 0);
        if (NCH > 1) load_w(1, 1);
        // ---- operand image
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int e = j * NT + tid;
            if ((j + 1) * NT > ITEMS && e >= ITEMS) continue;
            const int px = e / G, grp = e - px * G;
            const int kc = grp >> 3, cc = (grp & 7) * 4, sw = (px >> 1) & 7;
            ff::ff_f16x4 h0, h1;
            ff::split_pair4(val[j], h0, h1);
            char* row = smem + kc * PLANE + px * 128 + (cc & 7) * 2;
            *reinterpret_cast<ff::ff_f16x4*>(row + (((cc >> 3) ^ sw) << 4)) = h0;
            if (TERMS == 3) *reinterpret_cast<ff::ff_f16x4*>(row + (((4 + (cc >> 3)) ^ sw) << 4)) = h1;
        }
        __syncthreads();
        // ---- Cout x Cin on the matrix pipe: this wave's NTW channel tiles over the tile's pixel groups
        f32x4 acc[NTW][NPG];
#pragma unroll
        for (int v = 0; v < NTW; ++v)
#pragma unroll
            for (int g = 0; g < NPG; ++g) acc[v][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            if (kc == 0 && DEPTH > 3 && NCH > 2) load_w(2, 2);      // (behind the operand writes: the input's registers are free again)
            if (kc + DEPTH - 1 < NCH) load_w((kc + DEPTH - 1) % DEPTH, kc + DEPTH - 1);
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                const f16x8 xa = *reinterpret_cast<const f16x8*>(fa + kc * PLANE + g * 2048);
                f16x8 xb;
                if (TERMS == 3) xb = *reinterpret_cast<const f16x8*>(fb + kc * PLANE + g * 2048);
#pragma unroll
                for (int v = 0; v < NTW; ++v) {
                    const f16x8 w0 = __builtin_bit_cast(f16x8, wr[kc % DEPTH][v][0]);
                    acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa, acc[v][g], 0, 0, 0);
                    if (TERMS == 3) {
                        const f16x8 w1 = __builtin_bit_cast(f16x8, wr[kc % DEPTH][v][NT2 - 1]);
                        acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa, acc[v][g], 0, 0, 0);
                        acc[v][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[v][g], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue from the accumulators' lanes: channels n4 .. n4 + 3 of pixel 16 g + PI16(i): one split pair
#pragma unroll
        for (int v = 0; v < NTW; ++v) {
            const int n4 = (wave * NTW + v) * 16 + g16 * 4;
            if (n4 >= a.Cout) continue;
            const int nv = min(4, a.Cout - n4);
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                const long long pix = tile * TP + g * 16 + pxl;
                if (pix >= a.M) continue;
                f32x4 t = acc[v][g] * xinv + bias[v];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r], a.act);
                ff::store_split4(a.y + pix * a.y_ld, n4, t, nv);
            }
        }
        __syncthreads();          // everybody is done with the operand image before the next tile's is written
    }
}

}  // namespace

extern "C" int ff_pointwise_split_fwd(const float* x, int x_ld, long long npix, int Cin, const void* w_frag, const float* bias, int Cout, int act,
                                      int w_format, float* y_split, int y_ld, void* stream) {
    FF_REQUIRE(x && w_frag && y_split && npix > 0, "ff_pointwise_split_fwd: null pointer / empty tensor");
    FF_REQUIRE(Cin == 352 && x_ld == Cin, "ff_pointwise_split_fwd: Cin = %d, ld %d: the instance is the motion encoder's 352-channel rows, contiguous (other 1x1 layers take ff_conv2d_fwd)", Cin, x_ld);
    FF_REQUIRE(Cout > 0 && Cout <= 256 && Cout % 32 == 0 && y_ld >= Cout && y_ld % 32 == 0, "ff_pointwise_split_fwd: Cout %d (a multiple of 32, <= 256), y_ld %d (a multiple of 32)", Cout, y_ld);
    FF_REQUIRE(w_format == FF_W_F16X3 || w_format == FF_W_F16, "ff_pointwise_split_fwd: w_format %d (a split weight format)", w_format);
    FF_REQUIRE(act >= FF_ACT_NONE && act <= FF_ACT_TANH, "ff_pointwise_split_fwd: bad act %d", act);
    FF_REQUIRE(ff::aligned16(x) && ff::aligned16(w_frag) && ff::aligned16(y_split), "ff_pointwise_split_fwd: 16-byte alignment");
    FF_REQUIRE(npix * Cin * 4 < (1ll << 31), "ff_pointwise_split_fwd: an input of 2 GiB or more");
    constexpr int TP = 48, NCH = 11, NW = 8, NTW = 2;
    PArgs a{x, w_frag, bias, y_split, npix, (npix + TP - 1) / TP, y_ld, act, Cout};
    static const int per_cu = getenv("FF_POINTWISE_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("FF_POINTWISE_BLOCKS_PER_CU"))) : 2;
    const unsigned blocks = (unsigned)std::min<long long>(a.tiles, 256ll * per_cu);
    constexpr size_t lds = (size_t)NCH * TP * 128;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (w_format == FF_W_F16X3) pointwise_split_kernel<NCH, TP, NW, NTW, 3><<<blocks, 64 * NW, lds, s>>>(a);
    else pointwise_split_kernel<NCH, TP, NW, NTW, 1><<<blocks, 64 * NW, lds, s>>>(a);
    return ff::check_launch("ff_pointwise_split_fwd");
}
