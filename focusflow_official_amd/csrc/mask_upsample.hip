// The up-sampling mask's second convolution and the convex up-sampling as ONE kernel (update.py:121-124 mask[2] + the
// ".25 *" of update.py:133, raft.py:159-170): the (B, H, W, 576) mask tensor - 56.6 MB per iteration at 8 x 48 x 64, written by
// a 1x1 convolution only to be read once by the soft-max - never exists.
//
// mask channel = k * 64 + s, k = 3x3 neighbour (F.unfold order), s = 8 x 8 sub-pixel.  As a GEMM the weights are the
// A operand (M = channels) and the pixels the B operand (N = pixels), so an accumulator register of a lane is ONE
// (pixel, sub-pixel) and the nine accumulators of a wave - one 32 x 32 tile per k - hold, in the same lane and register,
// the nine values the soft-max runs over: max, exp, sum and the weighted sum of the nine neighbouring flow vectors are
// plain in-lane arithmetic (the arithmetic of upsample_kernel, update_ops.hip).  A lane's four registers of a group are
// four consecutive sub-pixel columns: the result leaves as 16-byte stores, 1 KB contiguous per wave instruction.
//
// Block = 96 pixels x 576 channels = 6 waves (at 8 x 48 x 64 pixels that is exactly one block per CU), K = 256 in sixteen
// 16-channel stages through a ring of THREE 42 KB LDS stages: the pre-split weight rows go L2 -> LDS by LDS-DMA two stages
// ahead (16-byte slots XOR-swizzled on the source side, slot ^ ((row >> 2) & 3): the 64-byte rows read conflict-free; a
// counted s_waitcnt leaves the newest stage in flight - with two stages and two blocks per CU the L2 round trip of every
// stage was exposed: 46 us), the activations are split on the way in.
// Wave (wp, ws): pixels 32 wp .. + 31, sub-pixels 32 ws .. + 31 of every k: 9 accumulators = 144 registers.
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int NCH = 576, KIN = 256, PXB = 96, NTHR = 384;
constexpr int WROWS = NCH * 64;                 // weight image of a stage: 576 rows x 64 B
constexpr int STAGE = WROWS + PXB * 64;         // + 96 pixel rows x 64 B = 43008
constexpr int NDMA = NCH * 4 / NTHR;            // 6 sixteen-byte pieces per thread and stage

struct MUArgs {
    const float* hid; int hid_ld;
    const char* w;                               // ff_mask_upsample_pack's image: [16 stages][576 rows][4 slots of 16 B, swizzled]
    const float* bias;
    const float* flow; int flow_ld;
    float* out;
    int B, H, W;
    long long P;                                 // B * H * W
    float out_scale;
};

template <int TERMS>
__global__ __launch_bounds__(NTHR) void mask_upsample_kernel(const MUArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % 3, ws = wave / 3;
    const int li = lane & 31, lh = lane >> 5;
    const long long p0 = (long long)blockIdx.x * PXB;

    // ---- LDS-DMA of the weights: piece idx = (i * 6 + wave) * 64 + lane -> row idx >> 2, physical slot idx & 3,
    // logical slot = physical ^ ((row >> 2) & 3) = term * 2 + k-slice
    unsigned woff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        woff[i] = (unsigned)((i * 6 + wave) * 64 + lane) * 16u;      // stage-major image (ff_mask_upsample_pack): a straight copy
    }
    // Issued as inline assembly (M0 = LDS base of the wave's 1 KB, destination lane-linear) so that the compiler's own
    // wait-count bookkeeping does not see them: it put a vmcnt(0) at the top of the loop otherwise (an LDS store behind an
    // outstanding LDS-DMA) and the ring degenerated into one stage in flight.  The waits are the counted ones below.
    auto issue_w = [&](int st, int buf) {
        const unsigned dst = (unsigned)(unsigned long)(lptr_t)(smem + buf * STAGE) + wave * 1024u;
        const char* src = a.w + st * WROWS;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off\n\ts_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, off\n\ts_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %4, off\n\ts_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %5, off\n\ts_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %6, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + woff[0]), "v"(src + woff[1]), "v"(src + woff[2]), "v"(src + woff[3]), "v"(src + woff[4]), "v"(src + woff[5]), "s"(dst)
                     : "memory", "scc");
    };
    // ---- activations: thread -> pixel tid >> 2, channel quad tid & 3 of the stage's 16 channels
    const int apx = tid >> 2, aq = tid & 3;          // 96 pixels x 4 quads = 384 threads
    const long long apix = p0 + apx < a.P ? p0 + apx : a.P - 1;        // clamped: rows past the end are computed and dropped
    const float* asrc = a.hid + apix * a.hid_ld + aq * 4;
    const int arow = apx * 64, asw = (apx >> 2) & 3;
    f32x4 areg;
    auto load_a = [&](int st) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg) : "v"(asrc + st * 16) : "memory"); };
    auto store_a = [&](int buf) {
        char* base = smem + buf * STAGE + WROWS + arow;
        f16x4 h0, h1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = areg[j] * ff::XSPLIT;
            const _Float16 t = (_Float16)sv;
            h0[j] = t;
            h1[j] = (_Float16)(sv - (float)t);
        }
        *reinterpret_cast<f16x4*>(base + (((aq >> 1)) ^ asw) * 16 + (aq & 1) * 8) = h0;
        if (TERMS == 3) *reinterpret_cast<f16x4*>(base + ((2 + (aq >> 1)) ^ asw) * 16 + (aq & 1) * 8) = h1;
    };

    // fragment addresses inside a stage: this lane's pixel row (B operand) and its channel row of k-group 0 (A operand)
    const int prow = wp * 32 + li, crow = ws * 32 + li;
    const int pb0 = WROWS + prow * 64 + ((lh) ^ ((prow >> 2) & 3)) * 16, pb1 = WROWS + prow * 64 + ((2 + lh) ^ ((prow >> 2) & 3)) * 16;
    const int ca0 = crow * 64 + ((lh) ^ ((crow >> 2) & 3)) * 16, ca1 = crow * 64 + ((2 + lh) ^ ((crow >> 2) & 3)) * 16;
    // (k-group kk adds kk * 64 rows = kk * 4096 bytes; (row >> 2) & 3 is unchanged by a multiple of 16 rows)

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // Ring of three stages.  Order of the memory operations of a wave: ..., DMA(st), load_a(st + 1), DMA(st + 1): at the top
    // of iteration st a wait that leaves the newest NDMA operations in flight has stage st's weights and stage st + 1's
    // activations; the activations of stage st were stored one iteration earlier and became visible with that barrier.
    static_assert(NDMA == 6, "issue_w is written for six pieces per thread");
    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg)::"memory");
    store_a(0);
    issue_w(0, 0);
    load_a(1);
    issue_w(1, 1);
    for (int st = 0; st < KIN / 16; ++st) {
        const int buf = st % 3;
        if (st + 1 < KIN / 16) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(areg) : "n"(NDMA) : "memory");      // (areg: its uses must stay behind the wait)
            store_a((st + 1) % 3);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                       // stage st is complete for every wave; everybody is done with stage st - 1's buffer
        if (st + 2 < KIN / 16) {
            load_a(st + 2);
            issue_w(st + 2, (st + 2) % 3);
        }
        const char* sb = smem + buf * STAGE;
        const f16x8 x0 = *reinterpret_cast<const f16x8*>(sb + pb0);
        f16x8 x1;
        if (TERMS == 3) x1 = *reinterpret_cast<const f16x8*>(sb + pb1);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const f16x8 w0 = *reinterpret_cast<const f16x8*>(sb + ca0 + k * 4096);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc[k], 0, 0, 0);
            if (TERMS == 3) {
                const f16x8 w1 = *reinterpret_cast<const f16x8*>(sb + ca1 + k * 4096);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, acc[k], 0, 0, 0);
            }
        }
    }

    // ---- soft-max over the nine neighbours + convex combination (upsample_kernel's arithmetic), lane = pixel
    const long long p = p0 + prow;
    if (p >= a.P) return;
    const int HW = a.H * a.W;
    const int b = (int)(p / HW), rem = (int)(p - (long long)b * HW), y = rem / a.W, x = rem - y * a.W;
    float fx[9], fy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {      // no load under a branch: clamp, load, select
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool in = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const float* f = a.flow + (((long long)b * a.H + min(max(yy, 0), a.H - 1)) * a.W + min(max(xx, 0), a.W - 1)) * a.flow_ld;
        const float f0 = f[0], f1 = f[1];
        fx[k] = in ? 8.f * f0 : 0.f;
        fy[k] = in ? 8.f * f1 : 0.f;
    }
    const float xinv = ff::SPLIT_INV;
    const long long HW8 = 64ll * HW;
    float* ob = a.out + (long long)b * 2 * HW8 + (long long)(8 * y) * (8 * a.W) + 8 * x;
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // registers 4g .. 4g + 3: sub-pixels s4 .. s4 + 3 (one row of the 8 x 8 block, columns 0-3 or 4-7)
        const int s4 = ws * 32 + 8 * g + 4 * lh;
        f32x4 ox, oy, bs[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) bs[k] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + k * 64 + s4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mv[9], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = acc[k][4 * g + e] * xinv + bs[k][e];
                v *= a.out_scale;
                mv[k] = v;
                mx = fmaxf(mx, v);
            }
            // (v_exp_f32 on x log2 e and ONE reciprocal per output: with expf and nine IEEE divisions the epilogue of a wave
            // was ~3000 vector instructions - as long as its 432 MFMAs; the inputs lie in [-20, 0], the difference is 1e-6 relative)
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                mv[k] = __expf(mv[k] - mx);
                den += mv[k];
            }
            const float rden = 1.f / den;
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float wgt = mv[k] * rden;
                sx += wgt * fx[k];
                sy += wgt * fy[k];
            }
            ox[e] = sx;
            oy[e] = sy;
        }
        float* o = ob + (long long)(s4 >> 3) * (8 * a.W) + (s4 & 7);
        *reinterpret_cast<f32x4*>(o) = ox;
        *reinterpret_cast<f32x4*>(o + HW8) = oy;
    }
}

// ---- twelve-wave variant: three waves per SIMD instead of 2 / 2 / 1 / 1 --------------------------------------------------
// Wave (wp, wq): pixels 32 wp .. + 31, sub-pixels 16 wq .. + 15 of every k.  A 32-row MFMA tile carries TWO neighbours k:
// rows 0-15 = (k = 2j, s), rows 16-31 = (k = 2j + 1, s) - five tiles for the nine k (the odd half of the last one is
// unused) - and because accumulator register r of a lane is row (r & 3) + 8 (r >> 2) + 4 lh, registers r and r + 8 are the
// same sub-pixel under k = 2j and k = 2j + 1: the soft-max stays in-lane.  80 accumulator registers, 12 waves per block.
constexpr int NTHR12 = 768, NDMA12 = NCH * 4 / NTHR12;      // 3 pieces per thread and stage

template <int TERMS>
__global__ __launch_bounds__(NTHR12) void mask_upsample12_kernel(const MUArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % 3, wq = wave / 3;
    const int li = lane & 31, lh = lane >> 5;
    const long long p0 = (long long)blockIdx.x * PXB;

    unsigned woff[NDMA12];
#pragma unroll
    for (int i = 0; i < NDMA12; ++i) {
        woff[i] = (unsigned)((i * 12 + wave) * 64 + lane) * 16u;     // stage-major image (ff_mask_upsample_pack): a straight copy
    }
    auto issue_w = [&](int st, int buf) {
        const unsigned dst = (unsigned)(unsigned long)(lptr_t)(smem + buf * STAGE) + wave * 1024u;
        const char* src = a.w + st * WROWS;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_add_u32 m0, m0, 0x3000\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off\n\ts_add_u32 m0, m0, 0x3000\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + woff[0]), "v"(src + woff[1]), "v"(src + woff[2]), "s"(dst)
                     : "memory", "scc");
    };
    // activations: the first six waves, thread -> pixel tid >> 2, channel quad tid & 3.  The other waves issue a load of
    // the same kind too (address of pixel 0) so that every wave's memory-operation count per stage is the same.
    const bool astage = wave < 6;
    const int apx = astage ? tid >> 2 : 0, aq = tid & 3;
    const long long apix = p0 + apx < a.P ? p0 + apx : a.P - 1;
    const float* asrc = a.hid + apix * a.hid_ld + aq * 4;
    const int arow = apx * 64, asw = (apx >> 2) & 3;
    f32x4 areg;
    auto load_a = [&](int st) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg) : "v"(asrc + st * 16) : "memory"); };
    auto store_a = [&](int buf) {
        if (!astage) return;
        char* base = smem + buf * STAGE + WROWS + arow;
        f16x4 h0, h1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = areg[j] * ff::XSPLIT;
            const _Float16 t = (_Float16)sv;
            h0[j] = t;
            h1[j] = (_Float16)(sv - (float)t);
        }
        *reinterpret_cast<f16x4*>(base + (((aq >> 1)) ^ asw) * 16 + (aq & 1) * 8) = h0;
        if (TERMS == 3) *reinterpret_cast<f16x4*>(base + ((2 + (aq >> 1)) ^ asw) * 16 + (aq & 1) * 8) = h1;
    };

    // B operand: this lane's pixel row; A operand: row (half = li >> 4 selects k = 2j / 2j + 1) * 64 + 16 wq + (li & 15) of tile j = 0
    const int prow = wp * 32 + li;
    const int pb0 = WROWS + prow * 64 + ((lh) ^ ((prow >> 2) & 3)) * 16, pb1 = WROWS + prow * 64 + ((2 + lh) ^ ((prow >> 2) & 3)) * 16;
    const int i16 = li & 15, crow = (li >> 4) * 64 + wq * 16 + i16, csw = (i16 >> 2) & 3;     // ((row >> 2) & 3) = (i16 >> 2) & 3 for every tile
    const int ca0 = crow * 64 + ((lh) ^ csw) * 16, ca1 = crow * 64 + ((2 + lh) ^ csw) * 16;
    // tile j adds 128 rows = 8192 bytes; the odd half of tile 4 (k = 9) reads the first activation rows behind the weight
    // image: finite halfs, results never used

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg)::"memory");
    store_a(0);
    issue_w(0, 0);
    load_a(1);
    issue_w(1, 1);
    for (int st = 0; st < KIN / 16; ++st) {
        const int buf = st % 3;
        if (st + 1 < KIN / 16) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(areg) : "n"(NDMA12) : "memory");
            store_a((st + 1) % 3);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (st + 2 < KIN / 16) {
            load_a(st + 2);
            issue_w(st + 2, (st + 2) % 3);
        }
        const char* sb = smem + buf * STAGE;
        const f16x8 x0 = *reinterpret_cast<const f16x8*>(sb + pb0);
        f16x8 x1;
        if (TERMS == 3) x1 = *reinterpret_cast<const f16x8*>(sb + pb1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const f16x8 w0 = *reinterpret_cast<const f16x8*>(sb + ca0 + j * 8192);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc[j], 0, 0, 0);
            if (TERMS == 3) {
                const f16x8 w1 = *reinterpret_cast<const f16x8*>(sb + ca1 + j * 8192);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, acc[j], 0, 0, 0);
            }
        }
    }

    const long long p = p0 + prow;
    if (p >= a.P) return;
    const int HW = a.H * a.W;
    const int b = (int)(p / HW), rem = (int)(p - (long long)b * HW), y = rem / a.W, x = rem - y * a.W;
    float fx[9], fy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool in = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const float* f = a.flow + (((long long)b * a.H + min(max(yy, 0), a.H - 1)) * a.W + min(max(xx, 0), a.W - 1)) * a.flow_ld;
        const float f0 = f[0], f1 = f[1];
        fx[k] = in ? 8.f * f0 : 0.f;
        fy[k] = in ? 8.f * f1 : 0.f;
    }
    const float xinv = ff::SPLIT_INV;
    const long long HW8 = 64ll * HW;
    float* ob = a.out + (long long)b * 2 * HW8 + (long long)(8 * y) * (8 * a.W) + 8 * x;
#pragma unroll
    for (int g = 0; g < 2; ++g) {          // registers 4g .. 4g + 3 (k even) and 8 + 4g .. (k odd): sub-pixels s4 .. s4 + 3
        const int s4 = wq * 16 + 8 * g + 4 * lh;
        f32x4 ox, oy, bs[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) bs[k] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + k * 64 + s4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mv[9], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = acc[k >> 1][(k & 1) * 8 + 4 * g + e] * xinv + bs[k][e];
                v *= a.out_scale;
                mv[k] = v;
                mx = fmaxf(mx, v);
            }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                mv[k] = __expf(mv[k] - mx);
                den += mv[k];
            }
            const float rden = 1.f / den;
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float wgt = mv[k] * rden;
                sx += wgt * fx[k];
                sy += wgt * fy[k];
            }
            ox[e] = sx;
            oy[e] = sy;
        }
        float* o = ob + (long long)(s4 >> 3) * (8 * a.W) + (s4 & 7);
        *reinterpret_cast<f32x4*>(o) = ox;
        *reinterpret_cast<f32x4*>(o + HW8) = oy;
    }
}

}  // namespace

namespace {
// Split rows [576][8 chunks][x0: 32 halfs | x1: 32 halfs] -> the stage-major LDS image [16 stages][576 rows][4 slots x 16 B]:
// stage st = channels 16 st .. + 15, slot (physical) p of row r holds logical slot p ^ ((r >> 2) & 3) = term * 2 + k-slice.
// With it a stage is 36 KB of CONTIGUOUS source for the LDS-DMA (64 lanes x 16 B = 1 KB per instruction); straight out of
// the split rows every instruction touched 32 half-used 128-byte lines (31.5 us per launch, L2-request bound).
__global__ void mask_upsample_pack_kernel(const char* __restrict__ w, char* __restrict__ out) {
    const int total = 16 * NCH * 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int st = i / (NCH * 4), rem = i - st * (NCH * 4), row = rem >> 2, slot = (rem & 3) ^ ((row >> 2) & 3);
        const char* src = w + (long long)row * 1024 + (st >> 1) * 128 + (st & 1) * 32 + (slot >> 1) * 64 + (slot & 1) * 16;
        *reinterpret_cast<f32x4*>(out + (long long)i * 16) = *reinterpret_cast<const f32x4*>(src);
    }
}
}  // namespace

extern "C" int ff_mask_upsample_pack(const void* w_split, void* w_stage, void* stream) {
    FF_REQUIRE(w_split && w_stage && ff::aligned16(w_split) && ff::aligned16(w_stage), "ff_mask_upsample_pack: null or misaligned pointer");
    mask_upsample_pack_kernel<<<144, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const char*>(w_split), static_cast<char*>(w_stage));
    return ff::check_launch("ff_mask_upsample_pack");
}

extern "C" int ff_mask_upsample_fwd(const float* hid, int hid_ld, const void* w_stage, int w_format, const float* bias, float out_scale,
                                    const float* flow, int flow_ld, float* out, int B, int H, int W, void* stream) {
    FF_REQUIRE(hid && w_stage && flow && out && B > 0 && H > 0 && W > 0, "ff_mask_upsample_fwd: bad argument");
    FF_REQUIRE(w_format == FF_W_F16X3 || w_format == FF_W_F16, "ff_mask_upsample_fwd: the weights must be split rows (FF_W_F16X3 / FF_W_F16)");
    FF_REQUIRE(hid_ld >= KIN && hid_ld % 4 == 0 && ff::aligned16(hid) && ff::aligned16(w_stage) && ff::aligned16(out) && (!bias || ff::aligned16(bias)) && flow_ld >= 2,
               "ff_mask_upsample_fwd: alignment / leading dimensions");
    MUArgs a;
    a.hid = hid; a.hid_ld = hid_ld;
    a.w = static_cast<const char*>(w_stage);
    a.bias = bias;
    a.flow = flow; a.flow_ld = flow_ld;
    a.out = out;
    a.B = B; a.H = H; a.W = W;
    a.P = (long long)B * H * W;
    a.out_scale = out_scale;
    const long long blocks = (a.P + PXB - 1) / PXB;
    FF_REQUIRE(blocks < (1ll << 31), "ff_mask_upsample_fwd: too many pixels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int waves = ff::tune_env("FF_MASK_UPSAMPLE_WAVES") ? atoi(ff::tune_env("FF_MASK_UPSAMPLE_WAVES")) : 12;     // 12 (default) or 6
    if (waves == 12) {
        if (w_format == FF_W_F16X3) FF_ALLOW_DYNAMIC_LDS((&mask_upsample12_kernel<3>), 3 * STAGE);
        else FF_ALLOW_DYNAMIC_LDS((&mask_upsample12_kernel<1>), 3 * STAGE);
        if (w_format == FF_W_F16X3) mask_upsample12_kernel<3><<<(unsigned)blocks, NTHR12, 3 * STAGE, s>>>(a);
        else mask_upsample12_kernel<1><<<(unsigned)blocks, NTHR12, 3 * STAGE, s>>>(a);
        return ff::check_launch("ff_mask_upsample_fwd");
    }
    if (w_format == FF_W_F16X3) FF_ALLOW_DYNAMIC_LDS((&mask_upsample_kernel<3>), 3 * STAGE);
    else FF_ALLOW_DYNAMIC_LDS((&mask_upsample_kernel<1>), 3 * STAGE);
    if (w_format == FF_W_F16X3) mask_upsample_kernel<3><<<(unsigned)blocks, NTHR, 3 * STAGE, s>>>(a);
    else mask_upsample_kernel<1><<<(unsigned)blocks, NTHR, 3 * STAGE, s>>>(a);
    return ff::check_launch("ff_mask_upsample_fwd");
}
