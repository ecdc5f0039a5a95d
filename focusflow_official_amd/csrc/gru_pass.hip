// One pass of SepConvGRU (update.py:45-60) as ONE kernel - inference, split-pair activations:
//     z | r = sigmoid(conv_zr([h, motion]) + pre_zr)      rh = r * h      q = tanh(conv_q([rh, motion]) + pre_q)
//     h' = (1 - z) h + z q
// with the (1,5) taps of pass 1 or the (5,1) taps of pass 2.  As two launches of conv_dma.hip a pass is: z|r kernel (ramp,
// 35 us main loop, 11 us of epilogue traffic - 62 MB at the HBM rate, every block of the launch in that phase at once - L2
// write-back, launch gap), then the same again for q: 120 us at 8 pairs, of which the two main loops are 57.  Here a block
// owns a TH x 16 pixel tile for ALL channels and runs the two convolutions back to back:
//   * r is computed on the tile plus a halo of 2 pixels in the tap direction (what q's taps reach), z on the tile only;
//   * r * h goes straight into LDS as the split-pair patch of the q convolution (four 32-channel planes) - it never
//     exists in HBM; z stays in the registers of the wave that will blend with it (a wave owns the same 16 of the 128
//     channels in z, r and q);
//   * the motion features are DMA'd again for the second convolution (L2 hits); one barrier per 32-channel chunk as in
//     conv_dma.hip, weights straight into registers in fragment order, no weights in LDS.
// Same arithmetic in the same order as the two-kernel route (chunk -> tap -> the three terms; the epilogue's separately
// rounded steps): the new state comes out bit for bit the same (tests/test_hip_split.py).
//
// 8 waves per block, one block per CU.  TH = 6 at the headline shape: 8 x (48 / 6) x (64 / 16) = 256 blocks = the chip.
// hipcc-flags: -ffp-contract=off
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_ptr_t;

__device__ __forceinline__ int PI16(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * (i - 12) + 9); }
__device__ __forceinline__ f16x8 lds_ld16(unsigned addr) { return *(__attribute__((address_space(3))) const f16x8*)(unsigned long)addr; }
constexpr unsigned OOB = 0x7fffffffu;

__device__ __forceinline__ void dma_piece(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(dst), "s"(soff)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct GArgs {
    const float* hs;   int hs_ld;      // state, split-pair [B][H][W][128]
    const float* mo;   int mo_ld;      // motion features, split-pair [..][128]
    const float* h;    int h_ld;       // state, fp32
    const float* zr_pre; int zr_pre_ld;   // the context features' share of z|r (+ nothing else), fp32 [..][256]
    const float* q_pre;  int q_pre_ld;    // ... of q, fp32 [..][128]
    const void* wzr;                   // fragment order (ff_pack_frag16): 16 tiles x 40 chunks x 2 KB
    const void* wq;                    // 8 tiles x 40 chunks x 2 KB
    const float* bzr;                  // [256]
    const float* bq;                   // [128]
    float* y;  int y_ld;               // new state fp32
    float* y2; int y2_ld;              // new state split-pair
    float* sv_z; float* sv_r; float* sv_q; int sv_ld;      // nullable (recorded passes): the gates on the tile's pixels, fp32 [..][128] each - what the backward differentiates through
    int B, H, W, tiles_x, tiles_y;
    unsigned long long* stamps;        // lab build only (FF_LAB): five s_memrealtime stamps per block
};
#ifdef FF_LAB
unsigned long long* g_stamps = nullptr;
#define GP_STAMP(i_) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 5 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GP_STAMP(i_) do { } while (0)
#endif

// DIR 0: taps along x (1x5, pass 1); DIR 1: taps along y (5x1, pass 2)
template <int DIR, int TH, int TERMS>
__global__ __launch_bounds__(512) void gru_pass_kernel(const GArgs a) {
    constexpr int C = 128, NCH = 4;                    // channels of h / motion / each gate; 32-channel chunks of one of them
    constexpr int NT = 5, NKC = 2 * NCH * NT;          // taps; chunks of a packed weight row (K = 5 x 256)
    constexpr int PW = DIR == 0 ? 24 : 16, PH = DIR == 0 ? TH : TH + 8;       // input patch: the tile + 4 pixels either way along the taps
    constexpr int RW = DIR == 0 ? 20 : 16, RH = DIR == 0 ? TH : TH + 4;       // r region: the tile + 2 pixels either way
    constexpr int NPIX = PH * PW, NPIECE = (NPIX + 7) / 8, NPP = (NPIECE + 7) / 8, PBYTES = NPIECE * 1024;
    constexpr int RPIX = RH * RW, RPLANE = ((RPIX + 7) / 8) * 1024;           // one 32-channel plane of r * h
    constexpr int NREG = DIR == 0 ? TH : RH;           // regular r tiles (a 16-pixel row piece each)
    constexpr int NHALO = DIR == 0 ? (TH * 4 + 15) / 16 : 0;                  // DIR 0: the 2 + 2 halo columns of the rows, 16 pixels per tile
    constexpr int NR = NREG + NHALO;
    constexpr int ZOFF = DIR == 0 ? 0 : 2;             // z tile u shares the fragments of r tile u + ZOFF
    constexpr int NWL1 = 2 * (TERMS == 3 ? 2 : 1), NWL2 = (TERMS == 3 ? 2 : 1);
    static_assert(PW % 2 == 0 && RW % 2 == 0, "even widths (bank argument of conv_dma.hip)");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 patch buffers][4 planes of r * h]
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
    const unsigned rh0 = lds0 + 2 * PBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..7: channels 16 wave .. 16 wave + 15 of z, r and q
    const int H = a.H, W = a.W;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int bimg = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;
    const int py0 = y0 - (DIR == 1 ? 4 : 0), px0 = x0 - (DIR == 0 ? 4 : 0);      // patch origin
    const int ry0 = y0 - (DIR == 1 ? 2 : 0), rx0 = x0 - (DIR == 0 ? 2 : 0);      // r-region origin
    const long long pix_total = (long long)a.B * H * W;
    const int i16 = lane & 15, g16 = lane >> 4, pcol = PI16(i16);

    // ---- DMA roles (8 waves): piece pc = wave + 8 j covers patch rows 8 pc .. 8 pc + 7
    int ppix[NPP];
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int r = (wave + 8 * j) * 8 + (lane >> 3);
        const int py = r / PW, px = r - py * PW;
        const int yy = py0 + py, xx = px0 + px;
        const bool in = r < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        ppix[j] = in ? ((((bimg * H + yy) * W + xx) << 7) | (((lane & 7) ^ ((px >> 1) & 7)) * 16)) : -1;
    }
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.hs), 0, (int)(pix_total * a.hs_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mo), 0, (int)(pix_total * a.mo_ld * 4), 0x00020000);
    auto issue_patch = [&](bool motion, int ci, int buf) {          // chunk ci (0..3) of h or of the motion features -> patch buffer buf
        const int ldb = (motion ? a.mo_ld : a.hs_ld) * 4;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ci * 128);
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            if ((wave + 8 * j) < NPIECE) {
                const unsigned voff = ppix[j] >= 0 ? __umul24((unsigned)(ppix[j] >> 7), (unsigned)ldb) + (unsigned)(ppix[j] & 127) : OOB;
                if (motion) dma_piece(voff, rs_m, lds0 + buf * PBYTES + (wave + 8 * j) * 1024, soff);
                else dma_piece(voff, rs_h, lds0 + buf * PBYTES + (wave + 8 * j) * 1024, soff);
            }
        }
    };

    // ---- weights in fragment order: tile T, chunk kc, term -> ((T * NKC + kc) * 2 + term) * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t rs_wzr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wzr), 0, 16 * NKC * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wq), 0, 8 * NKC * 2048, 0x00020000);
    const int woff_z = wave * NKC * 2048 + lane * 16, woff_r = (8 + wave) * NKC * 2048 + lane * 16, woff_q = woff_z;
    f32x4 wz[2][2], wr[2][2], wq_[2][2];          // [register set][term]
    auto issue_w1 = [&](auto set_tag, int kc) {
        constexpr int S = decltype(set_tag)::value;
        const int soff = kc * 2048;
        wz[S][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wzr, woff_z, soff, 0));
        if (TERMS == 3) wz[S][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wzr, woff_z + 1024, soff, 0));
        wr[S][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wzr, woff_r, soff, 0));
        if (TERMS == 3) wr[S][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wzr, woff_r + 1024, soff, 0));
    };
    auto issue_w2 = [&](auto set_tag, int kc) {
        constexpr int S = decltype(set_tag)::value;
        const int soff = kc * 2048;
        wq_[S][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wq, woff_q, soff, 0));
        if (TERMS == 3) wq_[S][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wq, woff_q + 1024, soff, 0));
    };

    // ---- this lane's pixel in the r tiles: regular tile j = r-region row j (DIR 1) / row j, columns 2 + .. (DIR 0); halo tile k
    // (DIR 0) = pixel 16 k + i of the list (row, halo column) with halo columns 0, 1, 18, 19 of the r region
    int hrow[NHALO ? NHALO : 1], hcol[NHALO ? NHALO : 1];
#pragma unroll
    for (int k = 0; k < NHALO; ++k) {
        const int pi = 16 * k + i16, hc = pi & 3;
        hrow[k] = pi >> 2;                                   // may run past TH - 1 in the last tile: masked where it matters
        hcol[k] = hc < 2 ? hc : 16 + hc;
    }
    const int rcol_reg = DIR == 0 ? 2 + pcol : pcol;         // r-region column of the lane's pixel in a regular tile

    f32x4 az[TH], ar[NR];
#pragma unroll
    for (int u = 0; u < TH; ++u) az[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NR; ++j) ar[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // =================== phase 1: z | r = conv([h, motion]) ===================
    GP_STAMP(0);
    issue_patch(false, 0, 0);
    issue_w1(std::integral_constant<int, 0>{}, 0);
    constexpr int NSTEP = 2 * NCH * NT;
    // B-fragment bases of this lane, per tap and term, without the patch buffer's base: pixel column pcol + k (k = 0..6) of
    // row 0.  DIR 1: the column never changes (k = 0 only), a tap is a row offset - an immediate.
    constexpr int NXK = DIR == 0 ? 7 : 1;
    unsigned xk[NXK][2];
#pragma unroll
    for (int k = 0; k < NXK; ++k) {
        const int px = pcol + k;
        const unsigned sw = (unsigned)((px >> 1) & 7);
        xk[k][0] = (unsigned)(px * 128) + ((g16 ^ sw) << 4);
        xk[k][1] = (unsigned)(px * 128) + (((4 + g16) ^ sw) << 4);
    }
    // One tap of one chunk, tap index and weight register set known at compile time (the taps are unrolled inside a run-time
    // loop over chunk pairs: with 2 waves per SIMD the scalar bookkeeping of a run-time tap index - 60 scalar instructions
    // beside 18-42 MFMAs - was what a wave's step took, not its MFMAs).
    auto tap1 = [&](int c, auto t_tag, auto set_tag) {
        constexpr int T = decltype(t_tag)::value, CUR = decltype(set_tag)::value, NXT = CUR ^ 1;
        const bool more = c + 1 < 2 * NCH;
        if constexpr (T == 0) {
            wait_vm<NWL1>();
            __builtin_amdgcn_s_barrier();
        }
        issue_w1(std::integral_constant<int, NXT>{}, T + 1 < NT ? (T + 1) * (2 * NCH) + c : (more ? c + 1 : 0));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned pb = lds0 + (unsigned)(c & 1) * PBYTES;
        // regular tiles: patch pixel (row j + T, column pcol) for DIR 1, (row j, column 2 + pcol + T) for DIR 0
        const unsigned xa0 = pb + (DIR == 0 ? xk[DIR == 0 ? T + 2 : 0][0] : xk[0][0] + T * PW * 128);
        const unsigned xa1 = pb + (DIR == 0 ? xk[DIR == 0 ? T + 2 : 0][1] : xk[0][1] + T * PW * 128);
        auto tiles = [&](auto lo_tag, auto hi_tag) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
            for (int j = LO; j < HI; ++j) {
                f16x8 xa, xb;
                if (j < NREG) {
                    xa = lds_ld16(xa0 + j * PW * 128);
                    if (TERMS == 3) xb = lds_ld16(xa1 + j * PW * 128);
                } else {
                    const int k = j - NREG, hp = hcol[k] + T;
                    const unsigned hs_ = (unsigned)((hp >> 1) & 7);
                    const unsigned hb = pb + (unsigned)((min(hrow[k], TH - 1) * PW + hp) * 128);
                    xa = lds_ld16(hb + ((g16 ^ hs_) << 4));
                    if (TERMS == 3) xb = lds_ld16(hb + (((4 + g16) ^ hs_) << 4));
                }
                const f16x8 r0 = __builtin_bit_cast(f16x8, wr[CUR][0]);
                ar[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0, xa, ar[j], 0, 0, 0);
                if (TERMS == 3) {
                    const f16x8 r1 = __builtin_bit_cast(f16x8, wr[CUR][1]);
                    ar[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r1, xa, ar[j], 0, 0, 0);
                    ar[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0, xb, ar[j], 0, 0, 0);
                }
                if (j >= ZOFF && j < ZOFF + TH) {        // the tile's own pixels: z from the same fragments
                    const f16x8 z0 = __builtin_bit_cast(f16x8, wz[CUR][0]);
                    az[j - ZOFF] = __builtin_amdgcn_mfma_f32_16x16x32_f16(z0, xa, az[j - ZOFF], 0, 0, 0);
                    if (TERMS == 3) {
                        const f16x8 z1 = __builtin_bit_cast(f16x8, wz[CUR][1]);
                        az[j - ZOFF] = __builtin_amdgcn_mfma_f32_16x16x32_f16(z1, xa, az[j - ZOFF], 0, 0, 0);
                        az[j - ZOFF] = __builtin_amdgcn_mfma_f32_16x16x32_f16(z0, xb, az[j - ZOFF], 0, 0, 0);
                    }
                }
            }
        };
        if constexpr (T == 0) {
            if (more) {
                tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                issue_patch(c + 1 >= NCH, (c + 1) & (NCH - 1), (c + 1) & 1);
                tiles(std::integral_constant<int, 2>{}, std::integral_constant<int, NR>{});
            } else {
                tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, NR>{});
            }
        } else {
            tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, NR>{});
        }
    };
#define FF_TAPS(FN_, C_, S0_, S1_) \
    FN_(C_, std::integral_constant<int, 0>{}, std::integral_constant<int, S0_>{}); FN_(C_, std::integral_constant<int, 1>{}, std::integral_constant<int, S1_>{}); \
    FN_(C_, std::integral_constant<int, 2>{}, std::integral_constant<int, S0_>{}); FN_(C_, std::integral_constant<int, 3>{}, std::integral_constant<int, S1_>{}); \
    FN_(C_, std::integral_constant<int, 4>{}, std::integral_constant<int, S0_>{});
    for (int c = 0; c < 2 * NCH; c += 2) {       // five taps per chunk: the register sets swap roles from one chunk to the next
        FF_TAPS(tap1, c, 0, 1)
        FF_TAPS(tap1, c + 1, 1, 0)
    }

    GP_STAMP(1);
    // =================== between the phases: z -> registers, r * h -> LDS (split pair) ===================
    // the second convolution's first weights and its first motion patch travel meanwhile (patch buffer 0: its last reader
    // was chunk 6 of phase 1, finished behind chunk 7's barrier)
    issue_w2(std::integral_constant<int, 0>{}, 0);
    issue_patch(true, 0, 0);
    const float xinv = ff::SPLIT_INV;
    const int cw = wave * 16 + g16 * 4;                         // this lane's first channel (of 128) in z, r, q
    // ALL operand loads of the rest of the block in flight together, unconditionally (pixels outside the image read pixel 0
    // and are zeroed by a select: a load under a branch is waited for before the next one is issued): the gates' context
    // shares on the tile (z, q) and on the r region (r), the state on the r region - the tile's own pixels of it are the
    // blend's h.  One memory round trip here, none in the epilogue.
    f32x4 pq[TH], hout[TH];
    {
        const f32x4 bz = *reinterpret_cast<const f32x4*>(a.bzr + cw), br = *reinterpret_cast<const f32x4*>(a.bzr + C + cw);
        f32x4 pz[TH], pr[NR], hh[NR];
        int rrow[NR], rcol[NR];
        bool live[NR], rin[NR], oin[TH];
#pragma unroll
        for (int u = 0; u < TH; ++u) {
            const int y = y0 + u, x = x0 + pcol;
            oin[u] = y < H && x < W;
            const long long po = oin[u] ? ((long long)bimg * H + y) * W + x : 0;
            pz[u] = *reinterpret_cast<const f32x4*>(a.zr_pre + po * a.zr_pre_ld + cw);
            pq[u] = *reinterpret_cast<const f32x4*>(a.q_pre + po * a.q_pre_ld + cw);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (j < NREG) { rrow[j] = j; rcol[j] = rcol_reg; live[j] = true; }
            else { rrow[j] = hrow[j < NREG ? 0 : j - NREG]; rcol[j] = hcol[j < NREG ? 0 : j - NREG]; live[j] = rrow[j] < RH; }      // (the last halo tile may be partly empty)
            const int y = ry0 + rrow[j], x = rx0 + rcol[j];
            rin[j] = live[j] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const long long po = rin[j] ? ((long long)bimg * H + y) * W + x : 0;
            pr[j] = *reinterpret_cast<const f32x4*>(a.zr_pre + po * a.zr_pre_ld + C + cw);
            hh[j] = *reinterpret_cast<const f32x4*>(a.h + po * a.h_ld + cw);
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // z on the tile's pixels
#pragma unroll
        for (int u = 0; u < TH; ++u) {
            f32x4 t = az[u] * xinv + bz;
            const f32x4 pv = oin[u] ? pz[u] : zero4;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = ff::fast_sigmoid(t[r] + pv[r]);
            az[u] = t;
            hout[u] = oin[u] ? hh[u + ZOFF] : zero4;
            pq[u] = oin[u] ? pq[u] : zero4;
        }
        // r * h on the r region -> LDS; outside the image h = 0: r * h = 0 there, the q convolution's zero padding
        const int chunk = wave >> 1, slot = 2 * (wave & 1) + (g16 >> 1), half8 = (g16 & 1) * 8;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            f32x4 t = ar[j] * xinv + br;
            const f32x4 pv = rin[j] ? pr[j] : zero4, hv = rin[j] ? hh[j] : zero4;
            f32x4 rv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { rv[r] = ff::fast_sigmoid(t[r] + pv[r]); t[r] = __fmul_rn(rv[r], hv[r]); }
            if (a.sv_r && j >= ZOFF && j - ZOFF < TH && j < NREG && rin[j])      // r tile u + ZOFF sits on the output tile's row u (the same pixels as z and q)
                *reinterpret_cast<f32x4*>(a.sv_r + (((long long)bimg * H + (ry0 + rrow[j])) * W + (rx0 + rcol[j])) * a.sv_ld + cw) = rv;
            if (!live[j]) continue;
            ff::ff_f16x4 h0, h1;
            ff::split_pair4(t, h0, h1);
            const unsigned adr = rh0 + chunk * RPLANE + (unsigned)((rrow[j] * RW + rcol[j]) * 128) + (((unsigned)slot ^ (unsigned)((rcol[j] >> 1) & 7)) << 4) + half8;
            *(__attribute__((address_space(3))) ff::ff_f16x4*)(unsigned long)adr = h0;
            if (TERMS == 3) *(__attribute__((address_space(3))) ff::ff_f16x4*)(unsigned long)(adr ^ 64) = h1;
        }
    }
    __syncthreads();            // r * h complete for every wave (lgkmcnt + barrier; the vmcnt(0) also lands the motion patch and q's first weights)

    GP_STAMP(2);
    // =================== phase 2: q = conv([r * h, motion]) ===================
    f32x4 aq[TH];
#pragma unroll
    for (int u = 0; u < TH; ++u) aq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tap2 = [&](int c, auto t_tag, auto set_tag) {
        constexpr int T = decltype(t_tag)::value, CUR = decltype(set_tag)::value, NXT = CUR ^ 1;
        const bool mo = c >= NCH;                       // chunks 0..3: r * h (resident planes), 4..7: motion (patch buffers (c & 1))
        if constexpr (T == 0) {
            if (mo && c > NCH) {                        // (chunk 4's patch was landed by the __syncthreads above)
                wait_vm<NWL2>();
                __builtin_amdgcn_s_barrier();
            }
        }
        issue_w2(std::integral_constant<int, NXT>{}, T + 1 < NT ? (T + 1) * (2 * NCH) + c : (c + 1 < 2 * NCH ? c + 1 : 0));
        __builtin_amdgcn_sched_barrier(0);
        // fragment of output row u: r-region (u + T, pcol) / (u, pcol + T); motion patch (u + T + 2, pcol) / (u, pcol + T + 2)
        const unsigned base = mo ? lds0 + (unsigned)(c & 1) * PBYTES : rh0 + (unsigned)c * RPLANE;
        const unsigned rstride = (unsigned)((mo ? PW : RW) * 128);
        unsigned xa0, xa1;
        if (DIR == 0) {
            xa0 = base + (mo ? xk[DIR == 0 ? T + 2 : 0][0] : xk[DIR == 0 ? T : 0][0]);
            xa1 = base + (mo ? xk[DIR == 0 ? T + 2 : 0][1] : xk[DIR == 0 ? T : 0][1]);
        } else {
            xa0 = base + xk[0][0] + (unsigned)(mo ? (T + 2) * PW * 128 : T * RW * 128);
            xa1 = base + xk[0][1] + (unsigned)(mo ? (T + 2) * PW * 128 : T * RW * 128);
        }
        auto tiles = [&](auto lo_tag, auto hi_tag) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
            for (int u = LO; u < HI; ++u) {
                const f16x8 xa = lds_ld16(xa0 + u * rstride);
                f16x8 xb;
                if (TERMS == 3) xb = lds_ld16(xa1 + u * rstride);
                const f16x8 q0 = __builtin_bit_cast(f16x8, wq_[CUR][0]);
                aq[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0, xa, aq[u], 0, 0, 0);
                if (TERMS == 3) {
                    const f16x8 q1 = __builtin_bit_cast(f16x8, wq_[CUR][1]);
                    aq[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1, xa, aq[u], 0, 0, 0);
                    aq[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0, xb, aq[u], 0, 0, 0);
                }
            }
        };
        if constexpr (T == 0) {
            if (mo && c + 1 < 2 * NCH) {
                tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                issue_patch(true, c + 1 - NCH, (c + 1) & 1);
                tiles(std::integral_constant<int, 2>{}, std::integral_constant<int, TH>{});
            } else {
                tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, TH>{});
            }
        } else {
            tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, TH>{});
        }
    };
    for (int c = 0; c < 2 * NCH; c += 2) {
        FF_TAPS(tap2, c, 0, 1)
        FF_TAPS(tap2, c + 1, 1, 0)
    }
#undef FF_TAPS

    GP_STAMP(3);
    // =================== epilogue: q = tanh(.), h' = (1 - z) h + z q (update.py:49-50) ===================
    {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bq + cw);
#pragma unroll
        for (int u = 0; u < TH; ++u) {
            const int y = y0 + u, x = x0 + pcol;
            if (!(y < H && x < W)) continue;
            const long long po = ((long long)bimg * H + y) * W + x;
            f32x4 t = aq[u] * xinv + bq, qv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float q = ff::fast_tanh(t[r] + pq[u][r]);
                qv[r] = q;
                t[r] = __fadd_rn(__fmul_rn(__fsub_rn(1.f, az[u][r]), hout[u][r]), __fmul_rn(az[u][r], q));
            }
            *reinterpret_cast<f32x4*>(a.y + po * a.y_ld + cw) = t;
            ff::store_split4(a.y2 + po * a.y2_ld, cw, t);
            if (a.sv_z) {
                *reinterpret_cast<f32x4*>(a.sv_z + po * a.sv_ld + cw) = az[u];
                *reinterpret_cast<f32x4*>(a.sv_q + po * a.sv_ld + cw) = qv;
            }
        }
    }
#ifdef FF_LAB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GP_STAMP(4);
}

template <int DIR, int TH, int TERMS>
int launch(const GArgs& a, hipStream_t s) {
    constexpr int PW = DIR == 0 ? 24 : 16, PH = DIR == 0 ? TH : TH + 8, RW = DIR == 0 ? 20 : 16, RH = DIR == 0 ? TH : TH + 4;
    constexpr int PBYTES = ((PH * PW + 7) / 8) * 1024, RPLANE = ((RH * RW + 7) / 8) * 1024;
    constexpr size_t lds = 2 * PBYTES + 4 * RPLANE;
    FF_ALLOW_DYNAMIC_LDS((&gru_pass_kernel<DIR, TH, TERMS>), (int)lds);
    const long long blocks = (long long)a.B * a.tiles_y * a.tiles_x;
    gru_pass_kernel<DIR, TH, TERMS><<<(unsigned)blocks, 512, lds, s>>>(a);
    return ff::check_launch("ff_gru_pass");
}

}  // namespace

#ifdef FF_LAB
extern "C" int ff_lab_gru_pass_stamps(void* buf) { g_stamps = static_cast<unsigned long long*>(buf); return 0; }      // lab build only
#endif

static int gru_pass_impl(int dir, const float* hs, int hs_ld, const float* motion, int mo_ld, const float* h, int h_ld,
                         const float* zr_pre, int zr_pre_ld, const float* q_pre, int q_pre_ld, const void* wzr_frag, const void* wq_frag,
                         const float* bzr, const float* bq, int w_format, float* y, int y_ld, float* y2, int y2_ld, float* sv_z, float* sv_r,
                         float* sv_q, int sv_ld, int B, int H, int W, void* stream) {
    FF_REQUIRE(hs && motion && h && zr_pre && q_pre && wzr_frag && wq_frag && bzr && bq && y && y2, "ff_gru_pass: null pointer");
    FF_REQUIRE(dir == 0 || dir == 1, "ff_gru_pass: dir 0 (1x5) or 1 (5x1)");
    FF_REQUIRE(w_format == FF_W_F16X3 || w_format == FF_W_F16, "ff_gru_pass: a split weight format");
    FF_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)B * H * W < (1ll << 24), "ff_gru_pass: shape");
    const int lds[] = {hs_ld, mo_ld, h_ld, zr_pre_ld, q_pre_ld, y_ld, y2_ld};
    for (int v : lds) FF_REQUIRE(v % 4 == 0 && v >= 128, "ff_gru_pass: every ld a multiple of 4 and >= 128");
    FF_REQUIRE(zr_pre_ld >= 256, "ff_gru_pass: zr_pre holds 256 channels");
    const void* ptrs[] = {hs, motion, h, zr_pre, q_pre, wzr_frag, wq_frag, bzr, bq, y, y2};
    for (const void* q : ptrs) FF_REQUIRE(ff::aligned16(q), "ff_gru_pass: 16-byte alignment");
    FF_REQUIRE((long long)B * H * W * std::max(hs_ld, mo_ld) * 4 < (1ll << 31), "ff_gru_pass: an input of 2 GiB or more");
    GArgs a;
    a.hs = hs; a.hs_ld = hs_ld; a.mo = motion; a.mo_ld = mo_ld; a.h = h; a.h_ld = h_ld;
    a.zr_pre = zr_pre; a.zr_pre_ld = zr_pre_ld; a.q_pre = q_pre; a.q_pre_ld = q_pre_ld;
    a.wzr = wzr_frag; a.wq = wq_frag; a.bzr = bzr; a.bq = bq; a.y = y; a.y_ld = y_ld; a.y2 = y2; a.y2_ld = y2_ld;
    FF_REQUIRE((sv_z != nullptr) == (sv_r != nullptr) && (sv_z != nullptr) == (sv_q != nullptr), "ff_gru_pass_rec: the three gate outputs come together");
    FF_REQUIRE(!sv_z || (sv_ld % 4 == 0 && sv_ld >= 128 && ff::aligned16(sv_z) && ff::aligned16(sv_r) && ff::aligned16(sv_q)), "ff_gru_pass_rec: gate outputs: ld % 4, >= 128, 16-byte aligned");
    a.sv_z = sv_z; a.sv_r = sv_r; a.sv_q = sv_q; a.sv_ld = sv_ld;
    a.B = B; a.H = H; a.W = W;
#ifdef FF_LAB
    a.stamps = g_stamps;
#else
    a.stamps = nullptr;
#endif
    a.tiles_x = (W + 15) / 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool t3 = w_format == FF_W_F16X3;
    // tile height 6 or 4 (8-row tiles need more than 256 registers in the vertical pass): one block per CU, so a launch costs
    // rounds of 256 blocks x (rows per tile + the fixed part of a block, ~2 rows' worth) - 48 / 6 x 64 / 16 x 8 pairs = 256 blocks
    // = one round; the training crop's 46 rows take 6-row tiles too (8 x 8 x 4 = 256 blocks, the last tile row ragged) rather
    // than 4-row ones (384 blocks: two rounds).  FF_GRU_PASS_TH overrides (tests walk both).
    auto cost = [&](int t) { return (((long long)B * ((H + t - 1) / t) * a.tiles_x + 255) / 256) * (t + 2); };
    int th = cost(6) <= cost(4) ? 6 : 4;
    if (const char* e = getenv("FF_GRU_PASS_TH")) th = atoi(e);
    a.tiles_y = (H + th - 1) / th;
#define FF_GP(D_, T_) if (dir == D_ && th == T_) return t3 ? launch<D_, T_, 3>(a, s) : launch<D_, T_, 1>(a, s);
    FF_GP(0, 6) FF_GP(1, 6) FF_GP(0, 4) FF_GP(1, 4)
#undef FF_GP
    return ff::fail(FF_EINVAL, "ff_gru_pass: tile height %d (4 or 6)", th);
}

extern "C" int ff_gru_pass(int dir, const float* hs, int hs_ld, const float* motion, int mo_ld, const float* h, int h_ld,
                           const float* zr_pre, int zr_pre_ld, const float* q_pre, int q_pre_ld, const void* wzr_frag, const void* wq_frag,
                           const float* bzr, const float* bq, int w_format, float* y, int y_ld, float* y2, int y2_ld, int B, int H, int W,
                           void* stream) {
    return gru_pass_impl(dir, hs, hs_ld, motion, mo_ld, h, h_ld, zr_pre, zr_pre_ld, q_pre, q_pre_ld, wzr_frag, wq_frag, bzr, bq, w_format, y, y_ld,
                         y2, y2_ld, nullptr, nullptr, nullptr, 0, B, H, W, stream);
}

extern "C" int ff_gru_pass_rec(int dir, const float* hs, int hs_ld, const float* motion, int mo_ld, const float* h, int h_ld,
                               const float* zr_pre, int zr_pre_ld, const float* q_pre, int q_pre_ld, const void* wzr_frag, const void* wq_frag,
                               const float* bzr, const float* bq, int w_format, float* y, int y_ld, float* y2, int y2_ld, float* z, float* r,
                               float* q, int gate_ld, int B, int H, int W, void* stream) {
    return gru_pass_impl(dir, hs, hs_ld, motion, mo_ld, h, h_ld, zr_pre, zr_pre_ld, q_pre, q_pre_ld, wzr_frag, wq_frag, bzr, bq, w_format, y, y_ld,
                         y2, y2_ld, z, r, q, gate_ld, B, H, W, stream);
}
