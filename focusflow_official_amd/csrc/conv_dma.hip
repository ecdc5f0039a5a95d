// Convolutions over SPLIT-PAIR activations (FF_FMT_SPLIT, focusflow_hip.h): stride-1 "same" 3x3 / 1x5 / 5x1 layers whose
// inputs were written by their producers as [x0: 32 fp16 | x1: 32 fp16] per 32-channel chunk - the update block's
// motion encoder, SepConvGRU and heads (update.py:45-60, 89-97, 121-135), twelve times per forward.
//
// What conv_patch.hip pays for per 32-channel chunk and block - load fp32 -> convert -> split -> ds_write of the input
// patch through registers (16 % of its time), an 8 KB weight chunk per tap through registers and LDS with two barriers
// (10 %), and one LDS fragment read per MFMA (12 %; DESIGN.md section 4, the ablation table) - is removed here:
//
//   * the patch with its halo travels L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR staging, no VALU, zero
//     padding by the buffer's range check; two patch buffers, the next chunk's patch in flight during this chunk's taps;
//   * the weights never touch LDS: the four waves of a block own DIFFERENT output channels (1 x 4 wave grid over N) and
//     every wave loads its own 16 x 32 A fragments straight from the packed rows (a lane's 16 bytes = one k-group of one
//     channel: the rows are already in MFMA operand order), one tap ahead, into two register sets;
//   * so the only barrier left is ONE per 32-channel chunk (patch buffers flip), and a wave reads
//     2 x TH fragments per 3 x NV x TH MFMAs: 0.33 LDS reads per MFMA at NV = 2 (conv_patch: 1.0 / 0.5).
//
// Matrix form: v_mfma_f32_16x16x32_f16, A = weights (rows = 16 output channels), B = 16 pixels of one patch row; a lane
// then owns FOUR CONSECUTIVE CHANNELS of a pixel (the epilogue's 16-byte loads / stores, and the two 8-byte stores of a
// split-pair output).  Arithmetic and summation order are those of conv_patch.hip's 16x16x32 variant: chunk -> tap ->
// three terms (w0 x0, w1 x0, w0 x1), so a layer computes the same bits whichever of the two kernels runs it.
//
// LDS image of a patch: row r = py * PW + px (PW = 16 + KW - 1, even), 128 bytes per row, the eight 16-byte slots of a
// row XOR-swizzled by (px >> 1) & 7 ON THE SOURCE SIDE (a DMA lane picks which slot of its pixel it fetches).  A B-fragment
// read (lane = (i, g): pixel column PI16(i) + dx, k-group g) is then free of bank conflicts for every tap: the hardware
// serves {i = 0-3, 12-15 of k-group g} together with {i = 4-11 of g + 1}; PI16 puts the first set on odd and the second
// on even columns, i.e. on different halves of the 64 banks (PW even), and eight same-parity consecutive columns differ in
// (px >> 1) & 7.  Because the key depends on px only, a tap's dy and the pixel row u are plain immediate offsets: the
// main loop has no address arithmetic at all.
// The epilogue's arithmetic is written as separately rounded operations and must stay that: conv_patch.hip and conv_dma.hip
// promise the same bits for the same layer (tests/test_hip_split.py), and a multiply-add that one of them contracts into an
// fma - after the optimiser specialised a path on the activation code - breaks that.  build.py reads the next line.
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

struct DArgs {
    FFConvParams p;
    int Cin, nci, nkc;        // nkc = 32-k chunks of a packed weight row = KH * KW * nci
    int tiles_x, tiles_y, n_tiles;
    long long w_row_bytes;
};

__device__ __forceinline__ f16x8 lds_ld16(unsigned addr) { return *(__attribute__((address_space(3))) const f16x8*)(unsigned long)addr; }

#if defined(FF_LAB) && !defined(FF_DMA_STAMPS)
#define FF_DMA_STAMPS 1       // the lab build stamps its blocks' phases (tools/dma_stamps.py); FF_DMA_ABL=<bits> adds timing-only ablations
#endif
#ifndef FF_DMA_NSET
#define FF_DMA_NSET 2      // weight register sets of the three-waves-per-SIMD instances.  3 (two taps of lead) was measured with in-kernel
#endif                     // stamps on the 3x3 layers: main loop 44.4 vs 42.2 us (256->192), 24.5 vs 23.4 us (128->512) - slower; kept as a lab switch
constexpr unsigned OOB = 0x7fffffffu;     // + any soffset < 2^31 stays below 2^32: the range check sees it out of range -> zeros

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KB at LDS address `dst` (M0), source = rsrc base + voff + soff
__device__ __forceinline__ void dma_piece(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
    unsigned keep;
    // (uniform values: when the scalar registers run short the allocator parks them in a vector register, which these operands do not take)
    dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
    soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(dst), "s"(soff)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KH x KW taps; TH pixel rows x 16 columns per block; NV 16-channel tiles per wave (block = 4 waves = 64 NV channels)
// EPI: FFConvParams.ep_mode (GRU steps, motion tail); TERMS 3 (f16x3) or 1 (f16)
// UG: pixel rows whose epilogue operands are loaded together
// (An fp32-input mode of this kernel - patch staged through registers - for the encoders' 3x3 layers was built and measured:
// 2.5 % slower end to end than conv_patch.hip at four blocks per CU; tools/proto/conv_dma_f32_inputs.hip.)
// NW: waves per block = 16 NV NW output channels per block.  4 in the general instances; Cout / (16 NV) in the "all channels"
// instances: ONE block per 6 x 16 pixel tile computes every output channel (a patch is DMA'd once per tile instead of once
// per 64 channels; 8 x 8 x 4 = 256 blocks at the headline shape = one per CU, evenly - the 64-channel blocks of a 192- or
// 126-channel layer are 576 / 384 blocks on 768 slots, and the CUs that hold three of them set the kernel's time).
// XMODE (round 5): 0 = split-pair inputs.  1 = FP32 inputs: the raw patch travels L2 -> LDS by the same LDS-DMA (no registers
// hold the next chunk's patch during the tap loop - what cost the register-staged prototype its fourth wave per SIMD), and at the
// top of a chunk the block converts it IN PLACE: a pixel's 32 fp32 channels and its split-pair chunk are the same 128 bytes,
// the four lanes that own a pixel sit in one wave and read all of it before they write any of it.  2 = 1 + the producer's
// normalisation applied on the way (FFConvParams in_scale / in_shift / in_act: InstanceNorm + ReLU of the encoders' residual
// blocks; zero padding AFTER it).  Everything else - no weights in LDS, one weight set per tap straight into registers, 0.33 LDS
// reads per MFMA - is the split-pair kernel's: this is what takes the stride-1 3x3 / 1x5 / 5x1 layers with fp32 inputs (the
// encoders, every recorded forward convolution and input gradient) from conv_patch.hip.
// STATS: FFConvParams.stats_part (partial InstanceNorm statistics of the output from the epilogue, as conv_patch.hip's)
template <int KH, int KW, int TH, int NV, int TERMS, int EPI, int UG, int NSET, int NW, int XMODE = 0, bool STATS = false>
__device__ __forceinline__ void conv_dma_body(const DArgs& a) {
    constexpr bool XF32 = XMODE != 0, INORM = XMODE == 2;
    constexpr int PW = 16 + KW - 1, PH = TH + KH - 1, NPIX = PH * PW, NPIECE = (NPIX + 7) / 8, NPP = (NPIECE + NW - 1) / NW;
    constexpr int PBYTES = NPIECE * 1024, NT = KH * KW, NWL = NV * (TERMS == 3 ? 2 : 1);
    static_assert(PW % 2 == 0, "the bank argument needs an even patch width");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // two patch buffers
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    // blocks of one XCD (blockIdx % 8) take a contiguous run of (pixel tile, channel tile): the channel tiles of a pixel
    // tile - which read the same patch - then meet in one L2
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, x = bid & 7;
        bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (bid >> 3);
    }
    const int nt = bid % a.n_tiles; bid /= a.n_tiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int bimg = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16, n0 = nt * (16 * NV * NW) + wave * (16 * NV);    // n0: this WAVE's first output channel

    const long long pix_total = (long long)p.B * H * W;
    const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];

    // ---- DMA roles: piece pc = wave + NW j covers LDS rows 8 pc .. 8 pc + 7; lane -> row (lane >> 3), slot (lane & 7)
    // per piece one word: (image pixel << 7) | byte offset of the source slot inside the pixel's 128-byte chunk, or -1
    // (outside the image / past the patch: zeros)
    int ppix[NPP];
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int r = (wave + NW * j) * 8 + (lane >> 3);
        const int py = r / PW, px = r - py * PW;
        const int yy = y0 - p.pad_h + py, xx = x0 - p.pad_w + px;
        const bool in = r < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        // (fp32 inputs land unswizzled - the in-place conversion applies the key when it writes the split-pair chunk)
        ppix[j] = in ? ((((bimg * H + yy) * W + xx) << 7) | ((XF32 ? (lane & 7) : ((lane & 7) ^ ((px >> 1) & 7))) * 16)) : -1;
    }
    // (segment bookkeeping by mask arithmetic on scalars: `?:` chains over kernel-argument arrays or over buffer resources
    // become a scratch-resident table whose loads - and their vmcnt(0) - would sit inside the pipelined loop)
    const int ld0 = p.x_ld[0] * 4, ld1 = p.x_ld[1] * 4, ld2 = p.x_ld[2] * 4;
    const unsigned long long xp0 = (unsigned long long)p.x[0], xp1 = (unsigned long long)p.x[1], xp2 = (unsigned long long)p.x[2];
    auto issue_patch = [&](int c, int buf) {          // 32-channel chunk c of the concatenated input -> patch buffer buf
        const int ci = c * 32;
        const int in0 = -(int)(ci < c0), in1 = -(int)(ci >= c0 && ci < c01), in2 = -(int)(ci >= c01);     // all ones / zero
        const int ldb = (ld0 & in0) | (ld1 & in1) | (ld2 & in2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((ci - (c0 & (in1 | in2)) - ((c01 - c0) & in2)) * 4);
        const unsigned long long xp = (xp0 & (unsigned long long)(long long)in0) | (xp1 & (unsigned long long)(long long)in1) | (xp2 & (unsigned long long)(long long)in2);
        // (readfirstlane: the asm below needs the descriptor in scalar registers whatever unit computed the select)
        const unsigned long long xpu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)xp);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(xpu), 0, __builtin_amdgcn_readfirstlane((int)(pix_total * ldb)), 0x00020000);
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            if ((wave + NW * j) < NPIECE) {       // wave-uniform
                const unsigned voff = ppix[j] >= 0 ? __umul24((unsigned)(ppix[j] >> 7), (unsigned)ldb) + (unsigned)(ppix[j] & 127) : OOB;
                dma_piece(voff, rs, lds0 + buf * PBYTES + (wave + NW * j) * 1024, soff);
            }
        }
    };
    // ---- fp32 inputs: in-place conversion of a landed patch.  item = tid + NTHR i -> patch row r = item >> 2, k-group s4 = item & 3
    // (= tid & 3): the lane reads the row's fp32 slots 2 s4, 2 s4 + 1 (channels 8 s4 .. 8 s4 + 7) and writes the x0 halfs to slot
    // s4 ^ key(px), the x1 halfs to slot (4 + s4) ^ key(px) of the same row.
    constexpr int NTHR = 64 * NW, NCV = XF32 ? (NPIX * 4 + NTHR - 1) / NTHR : 1;
    float xs = 1.f, xinv_in = 1.f;
    if constexpr (XF32) ff::input_scale(p.x_amax, xs, xinv_in);     // gradients (dgrad on the f16 pipe): the input times 2^k, undone in the epilogue
    const int s4 = tid & 3;
    unsigned cv_key[NCV];       // swizzle key of the item's pixel column | 8 if the pixel lies inside the image | 16 if the item exists
    if constexpr (XF32) {
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int r = (tid >> 2) + (NTHR / 4) * i;
            const int py = r / PW, px = r - py * PW;
            const int yy = y0 - p.pad_h + py, xx = x0 - p.pad_w + px;
            cv_key[i] = (unsigned)((px >> 1) & 7) | (((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? 8u : 0u) | (r < NPIX ? 16u : 0u);
        }
    }
    auto convert = [&](int c, int buf) {
        const unsigned base = lds0 + buf * PBYTES + (unsigned)(tid >> 2) * 128;
        f32x4 va[NCV], vb[NCV];
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            va[i] = vb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cv_key[i] & 16u) {
                va[i] = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + i * (NTHR * 32));
                vb[i] = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + 16 + i * (NTHR * 32));
            }
        }
        f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = m0, a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if constexpr (INORM) {              // FFConvParams.in_scale / in_shift: one segment, tables [B][Cin]
            const long long t = (long long)bimg * a.Cin + c * 32 + s4 * 8;
            m0 = *reinterpret_cast<const f32x4*>(p.in_scale + t);
            m1 = *reinterpret_cast<const f32x4*>(p.in_scale + t + 4);
            a0 = *reinterpret_cast<const f32x4*>(p.in_shift + t);
            a1 = *reinterpret_cast<const f32x4*>(p.in_shift + t + 4);
        }
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            if (!(cv_key[i] & 16u)) continue;
            f32x4 v0 = va[i], v1 = vb[i];
            if constexpr (INORM) {          // the producer's normalisation (+ ReLU) on the way in; padding is zero AFTER it
                v0 = __builtin_elementwise_fma(v0, m0, a0);      // (one fused operation, as norm_apply_kernel's: the two give the same bits)
                v1 = __builtin_elementwise_fma(v1, m1, a1);
                if (p.in_act == FF_ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v0[j] = v0[j] < 0.f ? 0.f : v0[j]; v1[j] = v1[j] < 0.f ? 0.f : v1[j]; }      // (a NaN stays one: ff::apply_act)
                }
                if (!(cv_key[i] & 8u)) v0 = v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            ff::ff_f16x4 h00, h01, h10, h11;
            ff::split_pair4(v0 * xs, h00, h10);
            ff::split_pair4(v1 * xs, h01, h11);
            f16x8 x0, x1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { x0[j] = h00[j]; x0[4 + j] = h01[j]; x1[j] = h10[j]; x1[4 + j] = h11[j]; }
            const unsigned key = cv_key[i] & 7u;
            const unsigned row = base + i * (NTHR * 32);
            *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)s4 ^ key) << 4)) = x0;
            if (TERMS == 3) *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)(4 + s4) ^ key) << 4)) = x1;
        }
    };
    // ---- weights: lane (i, g) holds k-group g of channel n0 + 16 v + i; term 1 = + 64 bytes
    const int i16 = lane & 15, g16 = lane >> 4;
    // Two sources: the packed rows [Cout][nkc][128 B] (a wave-load = 16 rows x 64 bytes in 16 different lines), or - FFConvParams
    // w_frag, ff_pack_frag16 - the same bytes in fragment order [tile][nkc][term][lane][16 B]: one contiguous KB per load
    // (whole lines, no second fetch of a line for its other term)
    const bool frag = p.w_frag != nullptr;
    const int ntile16 = (p.Cout + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rsw = frag ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_frag), 0, ntile16 * a.nkc * 2048, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long long)p.Cout * a.w_row_bytes), 0x00020000);
    const int kc_stride = frag ? 2048 : 128, term_off = frag ? 1024 : 64;
    int woff[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int n = n0 + 16 * v + i16;
        if (frag) woff[v] = (n >> 4) < ntile16 ? ((n >> 4) * a.nkc * 2048 + lane * 16) : (int)(OOB - 2048);
        else woff[v] = n < p.Cout ? (int)(n * a.w_row_bytes) + g16 * 16 : (int)(OOB - 2048);
    }
    // (ordinary buffer loads: the compiler keeps the registers and counts vmcnt for them - an asm load into a register the
    // compiler believes ready was copied around by the register allocator BEFORE its data had landed.  The LDS-DMA pieces are
    // asm and unknown to that count, which makes the compiler's waits conservative, never early: vmcnt(N) leaves the N
    // youngest operations in flight whatever they are.)
    // NSET register sets: step s computes from set s % NSET while the loads of step s + NSET - 1 are issued (NSET = 3: two taps
    // of lead - at three waves per SIMD one tap did not always cover the L2 round trip: 10 % of the main loop, by ablation)
    f32x4 wr[NSET][NV][2];     // [register set][channel tile][term]
    auto issue_w = [&](auto set_tag, int kc) {
        constexpr int SET = decltype(set_tag)::value;
        const int soff = kc * kc_stride;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            wr[SET][v][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[v], soff, 0));
            if (TERMS == 3) wr[SET][v][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[v] + term_off, soff, 0));
        }
    };

    // ---- B fragments: pixel column PI16(i) + dx of patch row u + dy
    const int pcol = PI16(i16);

    f32x4 acc[NV][TH];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < TH; ++u) acc[v][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nci = a.nci;

    issue_patch(0, 0);
    static_assert(NT >= 3, "the chunk-top wait assumes that a patch is older than the newest weight loads (1x1 kernels need vmcnt(0) there)");
    issue_w(std::integral_constant<int, 0>{}, 0);
    if constexpr (NSET == 3) issue_w(std::integral_constant<int, 1>{}, a.nci * NT > 1 ? (NT > 1 ? a.nci : 1) : 0);      // step 1 = (chunk 0, tap 1)
    // One step = one (chunk c, tap t).  A RUN-TIME loop over the steps, two per trip so that the weight register sets
    // alternate without copies (set of step s = s & 1): the accumulators are loop-carried values and stay in place - fully
    // unrolled, every MFMA chain ended in a fresh register and the kernel needed twice the accumulator registers.
    // Memory operations of a wave, in issue order (vmcnt counts them in order):
    //   step (c, 0):  W(c, 1)  P(c + 1)        step (c, 1):  W(c, 2)        ...        step (c, NT - 1):  W(c + 1, 0)
    // Step (c, t) uses W(c, t), loaded one step earlier.  At a chunk's top everything older than the NWL weight loads of
    // W(c, 0) must have landed - that includes the patch of chunk c, issued a whole chunk ago.
    const int nsteps = nci * NT;
#ifdef FF_DMA_STAMPS      // lab build (FF_HIPCC_EXTRA_conv_dma=-DFF_DMA_STAMPS): phase stamps of every block into FFConvParams.splitk_ws
    unsigned long long stamp[4];
    stamp[0] = __builtin_amdgcn_s_memrealtime();
#endif
    auto step = [&](int s, auto set_tag) {
        constexpr int CUR = decltype(set_tag)::value, NXT = (CUR + NSET - 1) % NSET;
        const int c = s / NT, t = s - c * NT, dy = t / KW, dx = t - dy * KW;       // (scalar unit; NT, KW are constants)
        const bool more = c + 1 < nci;
#ifdef FF_DMA_ABL      // lab build: timing-only ablations (WRONG results): 1 patch staged once, 2 weights loaded once, 4 one barrier only, 8 no LDS reads
        if (t == 0 && (!(FF_DMA_ABL & 4) || s == 0)) {
#else
        if (t == 0) {
#endif
            wait_vm<NWL>();
            __builtin_amdgcn_s_barrier();      // every wave's pieces of patch c have landed; everybody is done with the other buffer
            if constexpr (XF32) {              // fp32 patch -> split pairs, in place; nobody reads a fragment before everybody has converted
                convert(c, c & 1);
                __syncthreads();
            }
#ifdef FF_DMA_STAMPS
            if (s == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        // (unconditionally - the last step re-reads chunk 0 for nothing: a load under a condition would leave the compiler's
        // vmcnt bookkeeping with two histories to merge, and it then waits for vmcnt(0), i.e. for the loads just issued)
#ifdef FF_DMA_ABL
        if (!(FF_DMA_ABL & 2) || s == 0)
#endif
        {
            const int sd = s + NSET - 1, cd = sd / NT, td = sd - cd * NT;
            issue_w(std::integral_constant<int, NXT>{}, sd < nsteps ? td * nci + cd : 0);
        }
        __builtin_amdgcn_sched_barrier(0);     // the loads stay HERE: left alone the scheduler sinks them below this tap's MFMAs (fewer live registers) and the next tap starts with their round trip
        // this lane's fragment rows of the tap: column px = pcol + dx (swizzle key (px >> 1) & 7), patch row dy (+ u: immediates)
        const int px = pcol + dx;
        const unsigned sw = (unsigned)((px >> 1) & 7);
        const unsigned rowb = lds0 + (unsigned)(c & 1) * PBYTES + (unsigned)((dy * PW + px) * 128);
        const unsigned xa0 = rowb + ((g16 ^ sw) << 4), xa1 = rowb + (((4 + g16) ^ sw) << 4);
        auto rows = [&](auto lo_tag, auto hi_tag) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
            for (int u = LO; u < HI; ++u) {
                const f16x8 xa = lds_ld16(xa0 + u * PW * 128);
                f16x8 xb;
                if (TERMS == 3) xb = lds_ld16(xa1 + u * PW * 128);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const f16x8 w0 = __builtin_bit_cast(f16x8, wr[CUR][v][0]);
                    acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa, acc[v][u], 0, 0, 0);
                    if (TERMS == 3) {
                        const f16x8 w1 = __builtin_bit_cast(f16x8, wr[CUR][v][1]);
                        acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa, acc[v][u], 0, 0, 0);
                        acc[v][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[v][u], 0, 0, 0);
                    }
                }
            }
            // schedule: the fragment reads run two pixel rows ahead of the MFMAs (three rows of fragments live, not all)
            constexpr int RD = TERMS == 3 ? 2 : 1, N = HI - LO, LEAD = N < 2 ? N : 2;
            __builtin_amdgcn_sched_group_barrier(0x100, RD * LEAD, 0);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (u + LEAD < N) __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NV * TERMS, 0);
            }
        };
        // The next chunk's patch, once per chunk, is issued BEHIND the first rows of tap 0 - i.e. behind the compiler's vmcnt
        // wait for this tap's weights.  That count does not know the DMA pieces: every later weight wait (one per tap) also
        // waits for every piece older than the loads it leaves in flight, so the pieces must be as old as possible when
        // the next wait comes - here they have the rest of this tap.  (At the end of the tap they had nothing: stamps
        // showed 7 % of the main loop waiting for pieces issued a moment earlier.)
        constexpr int SPLIT = TH >= 4 ? 2 : TH;
        if (t == 0 && more) {
            rows(std::integral_constant<int, 0>{}, std::integral_constant<int, SPLIT>{});
#ifdef FF_DMA_ABL
            if (!(FF_DMA_ABL & 1))
#endif
            issue_patch(c + 1, (c + 1) & 1);
            rows(std::integral_constant<int, SPLIT>{}, std::integral_constant<int, TH>{});
        } else {
            rows(std::integral_constant<int, 0>{}, std::integral_constant<int, TH>{});
        }
    };
    int s = 0;
    if constexpr (NSET == 2) {
        for (; s + 1 < nsteps; s += 2) {
            step(s, std::integral_constant<int, 0>{});
            step(s + 1, std::integral_constant<int, 1>{});
        }
        if (s < nsteps) step(s, std::integral_constant<int, 0>{});
    } else {
        for (; s + 2 < nsteps; s += 3) {
            step(s, std::integral_constant<int, 0>{});
            step(s + 1, std::integral_constant<int, 1>{});
            step(s + 2, std::integral_constant<int, 2>{});
        }
        if (s < nsteps) step(s, std::integral_constant<int, 0>{});
        if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>{});
    }
#ifdef FF_DMA_STAMPS
    stamp[2] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue (conv_patch.hip's 16x16x32 form): acc[v][u][r] = channel n4 + r (n4 = n0 + 16 v + 4 g) of pixel (y0 + u, x0 + pcol)
    // Per channel tile: ALL operand loads of its UG pixel rows first (residual, GRU operands: up to 3 x UG 16-byte loads in
    // flight), then the arithmetic, then the stores.  In-kernel stamps (tools/dma_stamps.py) had shown the z|r and q blocks
    // spending 12 us of their 48 behind the main loop - every block of the launch in that phase at the same time, each
    // running load -> use -> load -> use chains of four memory round trips.
    const float xinv = ff::SPLIT_INV * xinv_in;
    const int x = x0 + pcol;
    const bool vec_y = (p.y_ld & 3) == 0 && ff::aligned16(p.y);
    const bool vec_r = !p.res || ((p.res_ld & 3) == 0 && ff::aligned16(p.res));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int n4 = n0 + v * 16 + g16 * 4;
        if (EPI == FF_EP_MOTION_TAIL ? n4 >= p.Cout + 2 : n4 >= p.Cout) continue;
        const bool full = n4 + 3 < p.Cout;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f}, ct = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = min(n4 + r, p.Cout - 1);
            if (p.bias) bias[r] = p.bias[n];
            if (p.ch_scale) { cs[r] = p.ch_scale[n]; ct[r] = p.ch_shift[n]; }
        }
        int nv = min(4, p.Cout - n4);
        const bool tail = EPI == FF_EP_MOTION_TAIL && n4 + 3 >= p.Cout;       // the group that holds channels Cout, Cout + 1 (Cout % 4 == 2)
        if (tail) nv = 4;
        const bool out_full = nv == 4;
        const bool split_out = p.y_fmt == FF_FMT_SPLIT && n4 >= p.y_fmt_from;
        const bool rh = EPI == FF_EP_GRU_RH && n4 >= p.ep_split;
        f32x4 st_p = {0.f, 0.f, 0.f, 0.f}, st_s1 = st_p, st_s2 = st_p;      // STATS: this lane's four channels over its pixels
        float st_n = 0.f;
#pragma unroll
        for (int ug = 0; ug < TH; ug += UG) {
            f32x4 rr[UG], aa[UG], bb[UG];
            long long po[UG];
            // -- loads
#pragma unroll
            for (int k = 0; k < UG; ++k) {
                const int y = y0 + ug + k;
                po[k] = (y < H && x < W) ? ((long long)bimg * H + y) * W + x : -1;
                rr[k] = aa[k] = bb[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (po[k] < 0) continue;
                if (p.res) {
                    const float* rp2 = p.res + po[k] * p.res_ld + n4;
                    if (full && vec_r) rr[k] = *reinterpret_cast<const f32x4*>(rp2);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n4 + r < p.Cout) rr[k][r] = rp2[r];
                    }
                }
                if (rh) aa[k] = *reinterpret_cast<const f32x4*>(p.ep_a + po[k] * p.ep_a_ld + (n4 - p.ep_split));     // h (update.py:47-48)
                if constexpr (EPI == FF_EP_GRU_BLEND) {
                    aa[k] = *reinterpret_cast<const f32x4*>(p.ep_a + po[k] * p.ep_a_ld + n4);      // z
                    bb[k] = *reinterpret_cast<const f32x4*>(p.ep_b + po[k] * p.ep_b_ld + n4);      // h
                }
                if constexpr (EPI == FF_EP_MOTION_TAIL) {
                    if (tail) { aa[k][0] = p.ep_a[po[k] * 2]; aa[k][1] = p.ep_a[po[k] * 2 + 1]; }      // coords1
                }
            }
            // -- arithmetic (the roundings of the separate kernels: ff_gru_rh / ff_gru_blend / ff_coords_step)
            f32x4 vv[UG];
#pragma unroll
            for (int k = 0; k < UG; ++k) {
                f32x4 t = acc[v][ug + k] * xinv + bias;
                t *= p.out_scale;
                if (p.ch_scale) t = t * cs + ct;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r], p.act);
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r] + rr[k][r], p.act_res);
                }
                if (rh) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = __fmul_rn(t[r], aa[k][r]);
                }
                if constexpr (EPI == FF_EP_GRU_BLEND) {       // v = tanh(q) -> (1 - z) h + z v (update.py:49)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = __fadd_rn(__fmul_rn(__fsub_rn(1.f, aa[k][r]), bb[k][r]), __fmul_rn(aa[k][r], t[r]));
                }
                if constexpr (EPI == FF_EP_MOTION_TAIL) {     // channels Cout, Cout + 1 of the padded buffer = flow = coords1 - grid (raft.py:219)
                    if (tail) {
                        t[2] = __fsub_rn(aa[k][0], (float)x);
                        t[3] = __fsub_rn(aa[k][1], (float)(y0 + ug + k));
                    }
                }
                vv[k] = t;
            }
            if constexpr (STATS) {       // around a pivot (the lane's first value): fp32 sums without cancellation on nearly constant planes
#pragma unroll
                for (int k = 0; k < UG; ++k) {
                    if (po[k] < 0) continue;
                    if (st_n == 0.f) st_p = vv[k];
                    const f32x4 d = vv[k] - st_p;
                    st_s1 += d;
                    st_s2 = __builtin_elementwise_fma(d, d, st_s2);
                    st_n += 1.f;
                }
            }
            // -- stores
#pragma unroll
            for (int k = 0; k < UG; ++k) {
                if (po[k] < 0) continue;
                if (split_out) ff::store_split4(p.y + po[k] * p.y_ld, n4, vv[k], nv);
                else {
                    float* d = p.y + po[k] * p.y_ld + n4;
                    if (out_full && vec_y) *reinterpret_cast<f32x4*>(d) = vv[k];
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < nv) d[r] = vv[k][r];
                    }
                }
                if (p.y2) ff::store_split4(p.y2 + po[k] * p.y2_ld, n4, vv[k], nv);
            }
        }
        if constexpr (STATS) {
            // the sixteen lanes i16 of a k-group hold the same four channels over other pixel columns: butterfly merge, every
            // partial re-centred on the receiving lane's pivot (sum(v - p) = s + n d, sum((v - p)^2) = q + 2 d s + n d^2 for
            // entries around p + d); lane i16 == 0 writes the entry [image][part = tile][channel] = {pivot, s1, s2, n}
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                f32x4 p2, t1, t2;
#pragma unroll
                for (int r = 0; r < 4; ++r) { p2[r] = __shfl_xor(st_p[r], off); t1[r] = __shfl_xor(st_s1[r], off); t2[r] = __shfl_xor(st_s2[r], off); }
                const float n2 = __shfl_xor(st_n, off);
                if (st_n == 0.f) { st_p = p2; st_s1 = t1; st_s2 = t2; st_n = n2; }
                else if (n2 > 0.f) {
                    const f32x4 d = p2 - st_p;
                    st_s2 += t2 + 2.f * d * t1 + n2 * d * d;
                    st_s1 += t1 + n2 * d;
                    st_n += n2;
                }
            }
            if (i16 == 0) {
                const int nparts = a.tiles_y * a.tiles_x, part = ty * a.tiles_x + tx;
                float* e = p.stats_part + (((long long)bimg * nparts + part) * p.Cout + n4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n4 + r < p.Cout) *reinterpret_cast<f32x4*>(e + 4 * r) = (f32x4){st_p[r], st_s1[r], st_s2[r], st_n};
            }
        }
    }
#ifdef FF_DMA_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    if (p.splitk_ws && tid == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.splitk_ws) + (size_t)blockIdx.x * 4;
        d[0] = stamp[0]; d[1] = stamp[1]; d[2] = stamp[2]; d[3] = stamp[3];
    }
#endif
}

template <int KH, int KW, int TH, int NV, int TERMS, int EPI, int OCC, int NW = 4>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_dma_kernel(const DArgs a) {
    // epilogue operands of all TH rows at once where the registers allow (one channel tile per wave at three waves per SIMD),
    // else in groups of 4 (or 3: the 6-row tiles)
    conv_dma_body<KH, KW, TH, NV, TERMS, EPI, (NV == 1 && OCC <= 3) ? TH : (TH < 4 ? TH : (TH % 4 == 0 ? 4 : 3)), (OCC <= 3 && NV == 1) ? FF_DMA_NSET : 2, NW>(a);
}

template <int KH, int KW, int TH, int NV, int TERMS, int OCC>
int launch_ep(const DArgs& a, hipStream_t s) {
    constexpr int PW = 16 + KW - 1, PH = TH + KH - 1, NPIECE = (PH * PW + 7) / 8;
    constexpr size_t lds = 2 * NPIECE * 1024;
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    switch (a.p.ep_mode) {
        case FF_EP_NONE: conv_dma_kernel<KH, KW, TH, NV, TERMS, FF_EP_NONE, OCC><<<(unsigned)blocks, 256, lds, s>>>(a); break;
        case FF_EP_GRU_RH: conv_dma_kernel<KH, KW, TH, NV, TERMS, FF_EP_GRU_RH, OCC><<<(unsigned)blocks, 256, lds, s>>>(a); break;
        case FF_EP_GRU_BLEND: conv_dma_kernel<KH, KW, TH, NV, TERMS, FF_EP_GRU_BLEND, OCC><<<(unsigned)blocks, 256, lds, s>>>(a); break;
        case FF_EP_MOTION_TAIL: conv_dma_kernel<KH, KW, TH, NV, TERMS, FF_EP_MOTION_TAIL, OCC><<<(unsigned)blocks, 256, lds, s>>>(a); break;
        default: return ff::fail(FF_EINVAL, "ff_conv2d_fwd(dma): ep_mode %d", a.p.ep_mode);
    }
    return ff::check_launch("ff_conv2d_fwd(dma)");
}

// "all channels" instances (3x3, 6-row tiles): NW waves x NV channel tiles = every output channel in one block
template <int NV, int NW, int TERMS, int OCC>
int launch_allch(DArgs& a, hipStream_t s) {
    constexpr size_t lds = 2 * ((8 * 18 + 7) / 8) * 1024;
    a.tiles_y = (a.p.H + 5) / 6;
    a.n_tiles = (a.p.Cout + (a.p.ep_mode == FF_EP_MOTION_TAIL ? 2 : 0) + 16 * NV * NW - 1) / (16 * NV * NW);
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    if (a.p.ep_mode == FF_EP_NONE) conv_dma_kernel<3, 3, 6, NV, TERMS, FF_EP_NONE, OCC, NW><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
    else if (a.p.ep_mode == FF_EP_MOTION_TAIL) conv_dma_kernel<3, 3, 6, NV, TERMS, FF_EP_MOTION_TAIL, OCC, NW><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
    else return ff::fail(FF_EINVAL, "ff_conv2d_fwd(dma, all channels): ep_mode %d", a.p.ep_mode);
    return ff::check_launch("ff_conv2d_fwd(dma, all channels)");
}

template <int KH, int KW, int TERMS>
int launch_tile(DArgs& a, int th, hipStream_t s) {
    // One 16-channel tile per wave (NV = 1: 64 channels per block).  Two tiles per wave - a third of the LDS reads per MFMA -
    // were built and measured: 10-40 % slower at 8 pairs (half the blocks on a chip that the 64-channel blocks just fill),
    // equal at 32 pairs; the instances are gone, the template parameter stays.
    a.tiles_y = (a.p.H + th - 1) / th;
    a.n_tiles = (a.p.Cout + (a.p.ep_mode == FF_EP_MOTION_TAIL ? 2 : 0) + 63) / 64;
    if (th == 8) {
        // 3x3 and 5x1: two 23-24 KB patch buffers, three blocks per CU; 1x5: two 20 KB buffers, four
        if constexpr (KH == 1) { static const bool occ4 = !(ff::tune_env("FF_DMA_OCC4") && atoi(ff::tune_env("FF_DMA_OCC4")) == 0); if (occ4) return launch_ep<KH, KW, 8, 1, TERMS, 4>(a, s); }
        return launch_ep<KH, KW, 8, 1, TERMS, 3>(a, s);
    }
    return launch_ep<KH, KW, 4, 1, TERMS, 4>(a, s);      // (five waves per SIMD would cap the registers at 96: the 5x1 instances spill)
}

// ---- fp32 inputs (XMODE 1 / 2): one 16-channel tile per wave, NW waves = 16 NW output channels per block
template <int KH, int KW, int TH, int OCC, int NW, int XMODE, bool STATS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_dma_f32_kernel(const DArgs a) {
    conv_dma_body<KH, KW, TH, 1, 3, FF_EP_NONE, (TH % 4 == 0 ? 4 : (TH % 3 == 0 ? 3 : TH)), 2, NW, XMODE, STATS>(a);
}

template <int KH, int KW, int TH, int OCC, int NW>
int launch_f32(DArgs& a, hipStream_t s) {
    constexpr int PW = 16 + KW - 1, PH = TH + KH - 1, NPIECE = (PH * PW + 7) / 8;
    constexpr size_t lds = 2 * NPIECE * 1024;
    a.tiles_y = (a.p.H + TH - 1) / TH;
    a.n_tiles = (a.p.Cout + 16 * NW - 1) / (16 * NW);
    const long long blocks = (long long)a.p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    const bool inorm = a.p.in_scale != nullptr, stats = a.p.stats_part != nullptr;
    if constexpr (KH == 3 && KW == 3) {
        if (inorm && stats) conv_dma_f32_kernel<KH, KW, TH, OCC, NW, 2, true><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
        else if (inorm) conv_dma_f32_kernel<KH, KW, TH, OCC, NW, 2, false><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
        else if (stats) conv_dma_f32_kernel<KH, KW, TH, OCC, NW, 1, true><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
        else conv_dma_f32_kernel<KH, KW, TH, OCC, NW, 1, false><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
    } else {
        if (inorm || stats) return ff::fail(FF_EINVAL, "ff_conv2d_fwd(dma, fp32 inputs): in_scale / stats_part belong to the 3x3 layers");
        conv_dma_f32_kernel<KH, KW, TH, OCC, NW, 1, false><<<(unsigned)blocks, 64 * NW, lds, s>>>(a);
    }
    return ff::check_launch("ff_conv2d_fwd(dma, fp32 inputs)");
}

// Which fp32-input convolutions this kernel takes from conv_patch.hip: stride-1 "same" 3x3 / 1x5 / 5x1 layers in the f16x3
// format with every segment a multiple of 32 channels.  -> tile height (0 = not this kernel) and waves per block.
struct F32Route { int th, nw; };
F32Route f32_route(const FFConvParams& p, int cin) {
    static const bool enabled = !(getenv("FF_DMA_F32") && atoi(getenv("FF_DMA_F32")) == 0);      // A/B switch: 0 = conv_patch.hip
    F32Route no{0, 0};
    if (!enabled) return no;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    const bool k33 = p.KH == 3 && p.KW == 3, k15 = p.KH == 1 && p.KW == 5, k51 = p.KH == 5 && p.KW == 1;
    if (!(k33 || k15 || k51) || p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1 || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2) return no;
    if (p.w_format != FF_W_F16X3 || cin % 32 || p.res2 || p.splitk > 1 || p.ep_mode) return no;
    if ((p.y_fmt == FF_FMT_SPLIT && (p.y_fmt_from % 32 || p.y_ld % 4 || !ff::aligned16(p.y))) || (p.y2 && (p.y2_ld % 4 || !ff::aligned16(p.y2)))) return no;
    if ((p.in_scale || p.stats_part) && !k33) return no;
    // Measured per layer (round 5, same box): with 128 input channels and more this route beats conv_patch.hip (128 -> 128 at
    // 16 x 48 x 64: 54.8 against 59.4 us; the recorded update block and its input gradients: training step 56.0 -> 55.1 ms with
    // every layer routed, the encoders' included), with two or three chunks per tile it loses (64 -> 64 at 16 x 192 x 256: 215
    // against 203 us, 96 -> 96: 144 against 136): a tile of two chunks is over before the first patch's round trip and the
    // epilogue's stores are paid off, and conv_patch.hip hides them with four blocks per CU.  FF_DMA_F32_MINCH overrides (A/B).
    static const int min_chunks = ff::tune_env("FF_DMA_F32_MINCH") ? atoi(ff::tune_env("FF_DMA_F32_MINCH")) : 4;
    if (cin / 32 < min_chunks) return no;
    long long max_bytes = (long long)(p.Cout + 15) * ((p.KH * p.KW * cin + 31) / 32) * 128;
    for (int i = 0; i < FF_MAX_SEG && p.x_c[i]; ++i) {
        if (p.x_c[i] % 32 || p.x_fmt[i] != FF_FMT_F32) return no;
        max_bytes = std::max(max_bytes, (long long)p.B * p.H * p.W * p.x_ld[i] * 4);
    }
    if (max_bytes >= (1ll << 31) || (long long)p.B * p.H * p.W >= (1ll << 24)) return no;
    if (p.in_scale && (p.x_amax || p.x_c[1] || !ff::aligned16(p.in_scale) || !ff::aligned16(p.in_shift))) return no;
    if (p.stats_part && (p.x_amax || p.Cout % 4 || !ff::aligned16(p.stats_part))) return no;
    const int tiles_x = (p.W + 15) / 16;
    // all output channels in one block where they fit eight waves (the encoders' 64 / 96 / 128-channel layers: the patch is
    // fetched and converted once per pixel tile), else 64 channels per block
    // (eight waves x 16 channels for the 128-channel layers spill at the 128 registers two such blocks per CU leave: 64 channels per block there)
    const int nw = k33 && p.Cout > 64 && p.Cout <= 96 ? 6 : 4;
    const long long blocks8 = (long long)p.B * ((p.H + 7) / 8) * tiles_x * ((p.Cout + 16 * nw - 1) / (16 * nw));
    // small planes with long reductions keep conv_patch.hip's K splits (FF-PWC's decoders; one-pair forwards under a hipGraph)
    if (blocks8 < 256 && p.splitk_ws) return no;
    if (blocks8 >= (nw == 4 ? 384 : 256)) return F32Route{8, nw};
    return F32Route{4, 4};
}

int launch_f32_route(DArgs& a, F32Route r, hipStream_t s) {
    const bool k33 = a.p.KH == 3, k15 = a.p.KH == 1;
    if (k33) {
        if (r.th == 8 && r.nw == 6) return launch_f32<3, 3, 8, 3, 6>(a, s);
        if (r.th == 8) return launch_f32<3, 3, 8, 3, 4>(a, s);
        return launch_f32<3, 3, 4, 4, 4>(a, s);
    }
    if (k15) return r.th == 8 ? launch_f32<1, 5, 8, 3, 4>(a, s) : launch_f32<1, 5, 4, 4, 4>(a, s);
    return r.th == 8 ? launch_f32<5, 1, 8, 3, 4>(a, s) : launch_f32<5, 1, 4, 4, 4>(a, s);
}

// rows -> fragment order: one thread per 16-byte piece of the destination
__global__ void pack_frag16_kernel(const char* __restrict__ src, char* __restrict__ dst, int rows, int nkc, long long pieces) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < pieces; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63), term = (int)((i >> 6) & 1);
        const long long tk = i >> 7;                      // tile * nkc + kc
        const int kc = (int)(tk % nkc), tile = (int)(tk / nkc);
        const int row = tile * 16 + (lane & 15), g = lane >> 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) v = *reinterpret_cast<const f32x4*>(src + ((long long)row * nkc + kc) * 128 + term * 64 + g * 16);
        *reinterpret_cast<f32x4*>(dst + i * 16) = v;
    }
}

}  // namespace

extern "C" int ff_pack_frag16(const void* split_rows, void* dst, int rows, int nkc, void* stream) {
    FF_REQUIRE(split_rows && dst && rows > 0 && nkc > 0 && ff::aligned16(split_rows) && ff::aligned16(dst), "ff_pack_frag16: bad argument");
    const long long pieces = (long long)((rows + 15) / 16) * nkc * 128;
    FF_REQUIRE(pieces * 16 < (1ll << 31), "ff_pack_frag16: 2 GiB or more");
    pack_frag16_kernel<<<(unsigned)std::min<long long>((pieces + 255) / 256, 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const char*>(split_rows), static_cast<char*>(dst), rows, nkc, pieces);
    return ff::check_launch("ff_pack_frag16");
}

namespace ff {
// returns FF_OK if launched, 1 if this kernel does not take the convolution (the caller goes on to the other kernels); a
// split-pair input that it cannot take is an error - nothing else can read it
// entries per (image, channel) of FFConvParams.stats_part if THIS kernel runs the convolution (0: it does not)
int conv2d_dma_stats_parts(const FFConvParams& p, int cin) {
    static const bool stats_on = !(getenv("FF_CONV_STATS") && atoi(getenv("FF_CONV_STATS")) == 0);
    for (int i = 0; i < FF_MAX_SEG && p.x_c[i]; ++i)
        if (p.x_fmt[i] != FF_FMT_F32) return 0;
    if (!stats_on || p.KH != 3 || p.KW != 3 || p.x_amax || p.Cout % 4) return 0;
    FFConvParams q = p;
    q.stats_part = nullptr;         // (the route is asked before the buffer exists)
    const F32Route r = f32_route(q, cin);
    return r.th ? ((p.H + r.th - 1) / r.th) * ((p.W + 15) / 16) : 0;
}

int conv2d_fwd_dma(const FFConvParams& p, int cin, hipStream_t s) {
    bool any = false, all = true;
    for (int i = 0; i < FF_MAX_SEG && p.x_c[i]; ++i) {
        any |= p.x_fmt[i] == FF_FMT_SPLIT;
        all &= p.x_fmt[i] == FF_FMT_SPLIT;
    }
    if (!any) {
        const F32Route r = f32_route(p, cin);
        if (!r.th) return 1;
        DArgs a;
        a.p = p;
        a.Cin = cin;
        a.nci = cin / 32;
        a.nkc = p.KH * p.KW * a.nci;
        a.tiles_x = (p.W + 15) / 16;
        a.w_row_bytes = (long long)a.nkc * 128;
        if (p.w_frag && !aligned16(p.w_frag)) return fail(FF_EINVAL, "ff_conv2d_fwd: w_frag not 16-byte aligned");
        return launch_f32_route(a, r, s);
    }
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    const bool k33 = p.KH == 3 && p.KW == 3, k15 = p.KH == 1 && p.KW == 5, k51 = p.KH == 5 && p.KW == 1;
    if (!all || p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1 || !(k33 || k15 || k51) || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 ||
        (p.w_format != FF_W_F16X3 && p.w_format != FF_W_F16) || p.x_amax || p.in_scale || p.res2 || p.splitk > 1 || p.stats_part || cin % 32)
        return fail(FF_EINVAL, "ff_conv2d_fwd: split-pair inputs (x_fmt) need a stride-1 3x3 / 1x5 / 5x1 convolution in a split weight format "
                               "with every segment split and none of x_amax / in_scale / res2 / splitk / stats_part (got %dx%d stride %d, format %d)",
                    p.KH, p.KW, p.stride, p.w_format);
    long long max_bytes = 0;
    for (int i = 0; i < FF_MAX_SEG && p.x_c[i]; ++i) {
        if (p.x_c[i] % 32) return fail(FF_EINVAL, "ff_conv2d_fwd: split-pair segment %d has %d channels (multiples of 32 only)", i, p.x_c[i]);
        max_bytes = std::max(max_bytes, (long long)p.B * p.H * p.W * p.x_ld[i] * 4);
    }
    DArgs a;
    a.p = p;
    a.Cin = cin;
    a.nci = cin / 32;
    a.nkc = p.KH * p.KW * a.nci;
    a.tiles_x = (p.W + 15) / 16;
    a.w_row_bytes = (long long)((p.KH * p.KW * cin + 31) / 32) * 128;
    max_bytes = std::max(max_bytes, (long long)(p.Cout + 15) * a.w_row_bytes);
    if (p.w_frag && !aligned16(p.w_frag)) return fail(FF_EINVAL, "ff_conv2d_fwd: w_frag not 16-byte aligned");
    if (max_bytes >= (1ll << 31) || (long long)p.B * p.H * p.W >= (1ll << 24))
        return fail(FF_EINVAL, "ff_conv2d_fwd: split-pair convolution: a buffer of 2 GiB or more, or 2^24 pixels or more");
    if (p.ep_mode == FF_EP_COORDS) return fail(FF_EINVAL, "ff_conv2d_fwd: FF_EP_COORDS belongs to the fp32 flow head");
    if (p.ep_mode == FF_EP_MOTION_TAIL && (p.Cout % 4 != 2 || p.y_ld < p.Cout + 2 || !p.ep_a || (p.y2 && p.y2_ld < p.Cout + 2)))
        return fail(FF_EINVAL, "ff_conv2d_fwd: FF_EP_MOTION_TAIL: Cout %% 4 == 2, room for two more channels and ep_a = coords1");
    if ((p.y_fmt == FF_FMT_SPLIT && (p.y_fmt_from % 32 || p.y_ld % 4 || !aligned16(p.y))) || (p.y2 && (p.y2_ld % 4 || !aligned16(p.y2))))
        return fail(FF_EINVAL, "ff_conv2d_fwd: split-pair output: y_fmt_from %% 32, ld %% 4 and 16-byte alignment");
    // Tile choice, measured per layer of the update block (tools/bench_dma_conv.py): 8 x 16 pixels x 64 channels per block
    // wins at 8 pairs and at 32; 4-row tiles double the weight traffic per pixel and only pay on tiny planes.
    // FF_DMA_TILE = 4 or 8 overrides (tuning, tests).
    const int couts = p.Cout + (p.ep_mode == FF_EP_MOTION_TAIL ? 2 : 0);
    auto nblocks = [&](int th) { return (long long)p.B * ((p.H + th - 1) / th) * a.tiles_x * ((couts + 63) / 64); };
    int th = nblocks(8) >= 384 ? 8 : 4;
    if (const char* e = getenv("FF_DMA_TILE")) {
        const int v = atoi(e);
        if (v > 0) th = v;
        if (th != 4 && th != 8) return fail(FF_EINVAL, "FF_DMA_TILE: 4 or 8 (pixel rows per block)");
    }
    const bool t3 = p.w_format == FF_W_F16X3;
    // "All channels" layout (launch_allch): where 6 x 16 tiles give about one block per CU - the update block's 3x3 layers at
    // 8 pairs: 256 blocks.  More tiles than CUs would run as rounds of one block per CU with nothing to overlap the prologue and
    // epilogue phases: the 64-channel blocks (three per CU, out of phase) are better there.  FF_DMA_ALLCH=0: A/B switch.
    static const bool allch_on = !(ff::tune_env("FF_DMA_ALLCH") && atoi(ff::tune_env("FF_DMA_ALLCH")) == 0);
    const long long tiles6 = (long long)p.B * ((p.H + 5) / 6) * a.tiles_x;
    if (allch_on && k33 && !getenv("FF_DMA_TILE") && tiles6 >= 192 && tiles6 <= 272 && (p.ep_mode == FF_EP_NONE || p.ep_mode == FF_EP_MOTION_TAIL)) {
        const int ct = (couts + 15) / 16;          // 16-channel tiles
        if (ct > 8 && ct <= 12) return t3 ? launch_allch<1, 12, 3, 3>(a, s) : launch_allch<1, 12, 1, 3>(a, s);
        if (ct > 4 && ct <= 8) return t3 ? launch_allch<1, 8, 3, 2>(a, s) : launch_allch<1, 8, 1, 2>(a, s);
        // 512 channels (the flow / mask heads): two 16-wave blocks of 256 channels per tile, one block per CU at a time at four
        // waves per SIMD (16 waves x TWO channel tiles each spills at the 128 registers that leaves)
        static const bool heads16 = !(ff::tune_env("FF_DMA_HEADS16") && atoi(ff::tune_env("FF_DMA_HEADS16")) == 0);
        if (heads16 && ct > 16 && ct <= 32) return t3 ? launch_allch<1, 16, 3, 4>(a, s) : launch_allch<1, 16, 1, 4>(a, s);
    }
    if (k33) return t3 ? launch_tile<3, 3, 3>(a, th, s) : launch_tile<3, 3, 1>(a, th, s);
    if (k15) return t3 ? launch_tile<1, 5, 3>(a, th, s) : launch_tile<1, 5, 1>(a, th, s);
    return t3 ? launch_tile<5, 1, 3>(a, th, s) : launch_tile<5, 1, 1>(a, th, s);
}
}  // namespace ff
