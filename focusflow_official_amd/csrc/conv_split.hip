// fp32-accurate convolution on the HALF-PRECISION matrix pipe (v_mfma_f32_32x32x16_f16).
//
// CDNA4 has no TF32/xf32 and its fp32 MFMA runs at 1/16 of the f16 rate.  Every fp32
// operand is therefore split into two fp16 numbers,
//        x = x0 + 2^-11 * x1 ,   x0 = fp16(x),  x1 = fp16((x - x0) * 2^11)
// (11+11 = 22 significand bits; the 2^11 pre-scale keeps x1 out of the fp16 subnormals), and
//        x*w = x0*w0 + 2^-11 * (x0*w1 + x1*w0) + O(2^-22)
// costs three f16 MFMAs (fp32 accumulate, products exact) instead of eight fp32 MFMAs of
// twice the latency: 5.3x the matrix throughput at a relative error of ~2^-22 per product —
// measured end to end it is as close to an fp64 evaluation as the fp32 reference itself.
// TERMS = 1 keeps only x0*w0 (plain fp16 operands: a reduced-precision throughput mode).
//
// Same implicit-GEMM skeleton as conv_mfma.hip (tiles, XOR-swizzled 128-byte LDS rows, one
// barrier per 32-k chunk).  A 128-byte LDS row now holds [x0: 32 fp16 | x1: 32 fp16] for the
// chunk, so one ds_read_b128 is exactly one MFMA operand (8 consecutive k).  Weights arrive
// pre-split in that row format (ff_pack_split_f16); activations stay fp32 in HBM and are split
// in registers on their way to LDS.  Operands must satisfy |x| < 65504.
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int ROWB = 128;   // bytes per LDS row = per 32-k chunk of one packed weight row

struct KernArgs {
    FFConvParams p;
    int M, K, Cin;
    int m_tiles, n_tiles;
    long long w_row_bytes;   // packed split weights: bytes per output channel
};

__device__ __forceinline__ int swz(int row, int piece) { return piece ^ ((row >> 1) & 7); }

__device__ __forceinline__ void split4(const f32x4 v, f16x4& h0, f16x4& h1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 a = (_Float16)v[j];
        h0[j] = a;
        h1[j] = (_Float16)(v[j] - (float)a);
    }
}

template <int WM, int WN, int TM, int TN, int TERMS, int NST, bool UNI, int NB>   // NB = LDS buffers per operand (2, or 1: see _occ)
__device__ __forceinline__ void conv_split_body(const KernArgs& a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LA = BM / 32, LB = BN / 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(NB == 2 || NST == 1, "the single-buffer variant has no register ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                        // [NB][BM][128 B]
    char* sB = smem + NB * BM * ROWB;       // [NB][BN][128 B]

    const FFConvParams& p = a.p;
    float xs, xinv;
    ff::input_scale(p.x_amax, xs, xinv);       // 1, 1 unless the caller passed max|x| (gradients: dgrad on the f16 pipe)
    xs *= ff::XSPLIT; xinv *= ff::SPLIT_INV;   // operand scales of the split format (ff_common.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int grp = blockIdx.y;
    const int mt = bid / a.n_tiles, nt = bid - mt * a.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int H = p.H, W = p.W, Wo = p.Wo, HoWo = p.Ho * p.Wo;
    const int kq = tid & 7, rbase = tid >> 3;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;

    int hi0[LA], wi0[LA], img[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = m0 + rbase + 32 * i;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int ho = rem / Wo, wo = rem - ho * Wo;
            hi0[i] = ho * p.stride - p.pad_h;
            wi0[i] = wo * p.stride - p.pad_w;
            img[i] = b * H * W;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = 0;
            img[i] = 0;
        }
    }
    // ---- UNI fast path (Cin and every segment a multiple of 32): a 32-k chunk lies in ONE tap of ONE
    // segment, so tap / segment / channel bookkeeping is block-uniform (scalar registers), each row
    // carries a precomputed bitmask of the taps that fall inside the image, and loads are buffer loads
    // whose hardware range check returns 0 for masked rows (offset forced out of range) — no
    // divisions, no 64-bit multiplies, no exec-masked branches in the K loop.
    unsigned long long vmask[LA];
    int pixoff[LA];
    if (UNI) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            unsigned long long mk = 0;
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int hi = hi0[i] + (t / p.KW) * dlh, wi = wi0[i] + (t % p.KW) * dlw;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) mk |= 1ull << t;
            }
            vmask[i] = mk;
            pixoff[i] = img[i] + hi0[i] * W + wi0[i];
        }
    }
    const char* wbase = reinterpret_cast<const char*>(p.w) + (long long)grp * p.w_gstride * 4;
    const float* xs0 = p.x[0] + (long long)grp * p.x_gstride[0];
    const float* xs1 = p.x[1] ? p.x[1] + (long long)grp * p.x_gstride[1] : nullptr;
    const float* xs2 = p.x[2] ? p.x[2] + (long long)grp * p.x_gstride[2] : nullptr;
    const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];

    // buffer resources (wave-uniform): one per input segment + the packed weights
    const long long pix_total = (long long)p.B * H * W;
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs0), 0, (int)(pix_total * p.x_ld[0] * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs1 ? xs1 : xs0), 0, xs1 ? (int)(pix_total * p.x_ld[1] * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs2 ? xs2 : xs0), 0, xs2 ? (int)(pix_total * p.x_ld[2] * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, (int)((long long)p.Cout * a.w_row_bytes), 0x00020000);
    int woff[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int n = n0 + rbase + 32 * i;
        woff[i] = n < p.Cout ? (int)(n * a.w_row_bytes) + kq * 16 : 0x7fffffff;   // out of range -> zeros
    }
    auto stage_load_uni = [&](int kc, f32x4 (&ra)[LA], f32x4 (&rb)[LB]) {
        // block-uniform decode of chunk kc (scalar unit): K order is (tap, ci)
        const int k0 = kc * BK;
        const int tap = k0 / a.Cin;                 // uniform: SALU
        int ci0 = k0 - tap * a.Cin;
        const int dy = tap / p.KW, dx = tap - dy * p.KW;
        __amdgpu_buffer_rsrc_t rs;
        int ld;
        if (ci0 < c0) { rs = rs0; ld = p.x_ld[0]; }
        else if (ci0 < c01) { rs = rs1; ld = p.x_ld[1]; ci0 -= c0; }
        else { rs = rs2; ld = p.x_ld[2]; ci0 -= c01; }
        const int dpix = dy * dlh * W + dx * dlw;
        const int cib = (ci0 + kq * 4) * 4;          // byte offset of this thread's 4 channels
        const int ldb = ld * 4;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool ok = (vmask[i] >> tap) & 1ull;
            const int off = ok ? (pixoff[i] + dpix) * ldb + cib : 0x7fffffff;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[i], kc * ROWB, 0));
    };
    // NST = chunks of global loads kept in flight per thread (register ring depth)
    f32x4 ra[NST][LA], rb[NST][LB];
    auto stage_load = [&](int kc, f32x4 (&ra)[LA], f32x4 (&rb)[LB]) {
        const int k = kc * BK + kq * 4;
        const bool kok = k < a.K;
        const int tap = kok ? k / a.Cin : 0;
        int ci = k - tap * a.Cin;
        const int dy = tap / p.KW, dx = tap - dy * p.KW;
        const float* xp;
        int ld;
        if (!kok) { xp = xs0; ld = p.x_ld[0]; ci = 0; }
        else if (ci < c0) { xp = xs0; ld = p.x_ld[0]; }
        else if (ci < c01) { xp = xs1; ld = p.x_ld[1]; ci -= c0; }
        else { xp = xs2; ld = p.x_ld[2]; ci -= c01; }
        // No load under a branch (each would be waited for before the next is issued: LA + LB dependent round trips per
        // chunk): masked rows read the segment's first pixel and are zeroed by a select.
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int hi = hi0[i] + dy * dlh, wi = wi0[i] + dx * dlw;
            const bool ok = kok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (ok ? (long long)(img[i] + hi * W + wi) * ld + ci : 0ll));
            ra[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {   // weights: 16-byte piece kq of this chunk's 128-byte row, already split
            const int n = n0 + rbase + 32 * i;
            const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + (long long)min(n, p.Cout - 1) * a.w_row_bytes + (long long)kc * ROWB + kq * 16);
            rb[i] = n < p.Cout ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](int buf, const f32x4 (&ra)[LA], const f32x4 (&rb)[LB]) {
        char* dA = sA + buf * BM * ROWB;
        char* dB = sB + buf * BN * ROWB;
        const int pc = kq >> 1, half = (kq & 1) * 8;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = rbase + 32 * i;
            f16x4 h0, h1;
            split4(ra[i] * xs, h0, h1);
            *reinterpret_cast<f16x4*>(dA + row * ROWB + swz(row, pc) * 16 + half) = h0;
            if (TERMS == 3) *reinterpret_cast<f16x4*>(dA + row * ROWB + swz(row, 4 + pc) * 16 + half) = h1;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int row = rbase + 32 * i;
            *reinterpret_cast<f32x4*>(dB + row * ROWB + swz(row, kq) * 16) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (a.K + BK - 1) / BK;
    const int li = lane & 31, lh = lane >> 5;
    auto compute = [&](int cur) {
        const char* cA = sA + cur * BM * ROWB;
        const char* cB = sB + cur * BN * ROWB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {    // two k16 steps per chunk: piece 2s+lh holds k = 16s + 8lh .. +7
            f16x8 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + li;
                a0[i] = *reinterpret_cast<const f16x8*>(cA + row * ROWB + swz(row, 2 * s + lh) * 16);
                if (TERMS == 3) a1[i] = *reinterpret_cast<const f16x8*>(cA + row * ROWB + swz(row, 4 + 2 * s + lh) * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 32 + li;
                b0[j] = *reinterpret_cast<const f16x8*>(cB + row * ROWB + swz(row, 2 * s + lh) * 16);
                if (TERMS == 3) b1[j] = *reinterpret_cast<const f16x8*>(cB + row * ROWB + swz(row, 4 + 2 * s + lh) * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
                    if (TERMS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b1[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b0[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
    // Register ring: chunk c lives in stage c % NST from its load until it is written to LDS.
    // Iteration kc: issue the loads of chunk kc+NST (its stage was freed when chunk kc went to LDS),
    // MFMA on chunk kc from LDS, then move chunk kc+1 (loaded NST-1 iterations ago) into the other
    // LDS buffer.  hipcc's counted vmcnt waits only for that oldest stage.
#pragma unroll
    for (int st = 0; st < NST; ++st)
        if (st < nk) { if (UNI) stage_load_uni(st, ra[st], rb[st]); else stage_load(st, ra[st], rb[st]); }
    stage_store(0, ra[0], rb[0]);
    __syncthreads();
    int cur = 0;
    for (int kc0 = 0; kc0 < nk; kc0 += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int kc = kc0 + st;
            if (kc < nk) {
                if (kc + NST < nk) { if (UNI) stage_load_uni(kc + NST, ra[st], rb[st]); else stage_load(kc + NST, ra[st], rb[st]); }
                compute(cur);
                if (NB == 1) {              // one buffer: everybody must be done reading before it is overwritten
                    if (kc + 1 < nk) { __syncthreads(); stage_store(0, ra[0], rb[0]); }
                    __syncthreads();
                } else {
                    if (kc + 1 < nk) stage_store(cur ^ 1, ra[(st + 1) % NST], rb[(st + 1) % NST]);
                    __syncthreads();
                    cur ^= 1;
                }
            }
        }
    }

    float* yb = p.y + (long long)grp * p.y_gstride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + li;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float cs = p.ch_scale ? p.ch_scale[n] : 1.f;
        const float ct = p.ch_scale ? p.ch_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // Three passes per tile: values first, then ALL residual loads of the tile
            // together (the compiler must assume res aliases y: inside the store loop they become 16 serial
            // load -> store round trips per lane), then add + store.
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                v = v * xinv + bias;
                v *= p.out_scale;
                if (p.ch_scale) v = v * cs + ct;
                vv[r] = ff::apply_act(v, p.act);
            }
            if (p.res) {
                // FFConvParams.res2: output channels >= res_split add a second tensor (paired fusion convs)
                const bool second = p.res2 && n >= p.res_split;
                const float* rb = second ? p.res2 + (n - p.res_split) : p.res + n;
                const int rld = second ? p.res2_ld : p.res_ld;
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    rr[r] = m < a.M ? rb[(long long)m * rld] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = ff::apply_act(vv[r] + rr[r], p.act_res);
            }
            if (p.y_fmt == FF_FMT_SPLIT && n >= p.y_fmt_from) {       // FF_FMT_SPLIT: the lane's channel as one half in x0 and one in x1 of its chunk
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= a.M) continue;
                    const float sv = vv[r] * ff::XSPLIT;
                    const _Float16 h0 = (_Float16)sv, h1 = (_Float16)(sv - (float)h0);
                    char* c = reinterpret_cast<char*>(yb + (long long)m * p.y_ld + (n & ~31)) + (n & 31) * 2;
                    *reinterpret_cast<_Float16*>(c) = h0;
                    *reinterpret_cast<_Float16*>(c + 64) = h1;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < a.M) yb[(long long)m * p.y_ld + n] = vv[r];
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int TERMS, int NST, bool UNI>
__global__ __launch_bounds__(256) void conv_split_kernel(const KernArgs a) { conv_split_body<WM, WN, TM, TN, TERMS, NST, UNI, 2>(a); }
// One LDS buffer per operand (a second barrier per chunk) and registers capped for OCC waves per SIMD: the 128 x 64
// tile then needs 24 KB instead of 48 and four blocks share a CU (the lever of conv_patch.hip).
template <int WM, int WN, int TM, int TN, int TERMS, bool UNI, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_split_kernel_occ(const KernArgs a) {
    conv_split_body<WM, WN, TM, TN, TERMS, 1, UNI, 1>(a);
}

template <int WM, int WN, int TM, int TN, int TERMS, int NST, bool UNI>
int launch_u(const KernArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t lds = 2 * (BM + BN) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<WM, WN, TM, TN, TERMS, NST, UNI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    KernArgs k = a;
    k.m_tiles = (a.M + BM - 1) / BM;
    k.n_tiles = (a.p.Cout + BN - 1) / BN;
    dim3 grid(k.m_tiles * k.n_tiles, a.p.groups);
    static const int occ = ff::tune_env("FF_SPLIT_OCC") ? atoi(ff::tune_env("FF_SPLIT_OCC")) : 1;
    if constexpr (NST == 1 && WM == 2 && TM == 2 && TN == 1) {
        if (occ) {
            conv_split_kernel_occ<WM, WN, TM, TN, TERMS, UNI, 4><<<grid, 256, lds / 2, s>>>(k);
            return ff::check_launch("ff_conv2d_fwd(split)");
        }
    }
    if constexpr (NST == 1 && WM == 4 && TN == 3) {      // 128 x 96 (Cout = 96 layers): 56 KB -> 28 KB, 2 -> 3 blocks
        if (occ) {
            conv_split_kernel_occ<WM, WN, TM, TN, TERMS, UNI, 3><<<grid, 256, lds / 2, s>>>(k);
            return ff::check_launch("ff_conv2d_fwd(split)");
        }
    }
    conv_split_kernel<WM, WN, TM, TN, TERMS, NST, UNI><<<grid, 256, lds, s>>>(k);
    return ff::check_launch("ff_conv2d_fwd(split)");
}

template <int WM, int WN, int TM, int TN, int TERMS, int NST>
int launch_n(const KernArgs& a, hipStream_t s) {
    const FFConvParams& p = a.p;
    static const bool allow = !ff::tune_env("FF_SPLIT_NO_UNI");
    bool uni = allow && a.Cin % 32 == 0 && p.KH * p.KW <= 64;
    long long max_bytes = (long long)p.Cout * a.w_row_bytes;
    for (int i = 0; i < FF_MAX_SEG; ++i) {
        if (p.x_c[i] % 32) uni = false;
        if (p.x_c[i]) max_bytes = std::max(max_bytes, (long long)p.B * p.H * p.W * p.x_ld[i] * 4);
    }
    if (max_bytes >= (1ll << 31)) uni = false;      // buffer resources address < 2 GiB each
    return uni ? launch_u<WM, WN, TM, TN, TERMS, NST, true>(a, s) : launch_u<WM, WN, TM, TN, TERMS, NST, false>(a, s);
}

template <int WM, int WN, int TM, int TN, int TERMS>
int launch(const KernArgs& a, hipStream_t s) {
    // Ring depth 1 = plain double buffering.  Depth 3 was measured SLOWER end to end for the 128-row
    // tiles (200+ registers -> 2 blocks/CU instead of 3) and +3% for the 64x64 tile; FF_SPLIT_NST
    // overrides for tuning runs.
    static const int nst = ff::tune_env("FF_SPLIT_NST") ? atoi(ff::tune_env("FF_SPLIT_NST")) : (TM * TN == 1 ? 3 : 1);
    if (nst >= 3) return launch_n<WM, WN, TM, TN, TERMS, 3>(a, s);
    if (nst == 2) return launch_n<WM, WN, TM, TN, TERMS, 2>(a, s);
    return launch_n<WM, WN, TM, TN, TERMS, 1>(a, s);
}

template <int TERMS>
int dispatch(const KernArgs& a, hipStream_t s) {
    const FFConvParams& p = a.p;
    const long long M = a.M, g = p.groups;
    static const int force = ff::tune_env("FF_SPLIT_TILE") ? atoi(ff::tune_env("FF_SPLIT_TILE")) : -1;   // tuning only
    if (force == 0) return launch<2, 2, 1, 1, TERMS>(a, s);
    if (force == 1) return launch<2, 2, 2, 1, TERMS>(a, s);
    if (force == 2) return launch<2, 2, 2, 2, TERMS>(a, s);
    if (force == 3) return p.Cout > 64 && p.Cout <= 96 ? launch<4, 1, 1, 3, TERMS>(a, s) : launch<2, 2, 2, 1, TERMS>(a, s);
    // With f16-rate MFMAs a chunk of compute no longer hides a global-load round trip, so residency
    // (blocks per CU) beats tile size: the 128x128 three-term tile needs 270 registers = 1 block/CU and
    // measured 1.8x slower end to end than 128x64 (3 blocks/CU).  Keep 128x128 for the 1-term mode only.
    auto blocks = [&](int bm, int bn) { return g * ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    static const bool big1 = ff::tune_env("FF_SPLIT_F16_128") && atoi(ff::tune_env("FF_SPLIT_F16_128")) == 1;     // one-term mode: the 128 x 128 tile lost to 128 x 64 at 4 blocks per CU (391 vs 224 us on the 1x1 fusion convs)
    if (big1 && TERMS == 1 && p.Cout > 96 && (p.Cout % 128 == 0 || p.Cout > 192) && blocks(128, 128) >= 512) return launch<2, 2, 2, 2, TERMS>(a, s);
    if (p.Cout > 64 && p.Cout <= 96 && blocks(128, 96) >= 512) return launch<4, 1, 1, 3, TERMS>(a, s);
    if (blocks(128, 64) >= 512) return launch<2, 2, 2, 1, TERMS>(a, s);
    return launch<2, 2, 1, 1, TERMS>(a, s);
}

// fp32 rows [rows][K] -> split rows [rows][ceil(K/32)][x0: 32 fp16 | x1: 32 fp16], zero padded
// One thread = 8 consecutive k of one row chunk: two 16-byte loads (when the row allows), two 16-byte stores (x0 part
// and x1 part of the 128-byte chunk row).  Same arithmetic as one element at a time: h0 = f16(s v), h1 = f16(s v - h0).
__global__ void pack_split_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long rows, int K,
                                  int nchunks, int vec) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const long long total = rows * nchunks * 4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int oct = (int)(i & 3);
        const long long rc = i >> 2;
        const int c = (int)(rc % nchunks);
        const long long r = rc / nchunks;
        const int k0 = c * 32 + oct * 8;
        float v[8];
        if (vec && k0 + 8 <= K) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + r * K + k0), b = *reinterpret_cast<const f32x4*>(src + r * K + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = a[e];
                v[4 + e] = b[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = k0 + e < K ? src[r * K + k0 + e] : 0.f;
        }
        h8 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sv = v[e] * ff::WSPLIT;
            h0[e] = (_Float16)sv;
            h1[e] = (_Float16)(sv - (float)h0[e]);
        }
        *reinterpret_cast<h8*>(dst + rc * 64 + oct * 8) = h0;
        *reinterpret_cast<h8*>(dst + rc * 64 + 32 + oct * 8) = h1;
    }
}

}  // namespace

namespace ff {
// called from ff_conv2d_fwd after argument validation
int conv2d_fwd_split(const FFConvParams& p, int M, int cin, hipStream_t s) {
    int rc = conv2d_fwd_dma(p, cin, s);               // split-pair inputs: patch by LDS-DMA, weights straight into registers (conv_dma.hip)
    if (rc != 1) return rc;
    const bool split_out = p.y_fmt != FF_FMT_F32 || p.y2;      // fp32 in, split-pair out: the patch kernel's or the im2col kernel's epilogue writes it
    rc = split_out ? 1 : conv2d_fwd_stem(p, cin, s);  // the encoders' 7x7 stride-2 stems over NHWC4 (conv_stem.hip)
    if (rc != 1) return rc;
    rc = conv2d_fwd_patch(p, cin, s);                 // stride-1 "same" convolutions: patch-stationary kernel (declines split outputs it cannot write)
    if (rc != 1) return rc;
    if (p.in_scale) return fail(FF_EINVAL, "ff_conv2d_fwd: in_scale/in_shift: the patch kernel declined this shape");
    if (p.ep_mode) return fail(FF_EINVAL, "ff_conv2d_fwd: ep_mode: the patch kernel declined this shape");
    if (p.stats_part) return fail(FF_EINVAL, "ff_conv2d_fwd: stats_part: the patch kernel declined this shape (ff_conv2d_stats_parts says which convolutions qualify)");
    KernArgs a;
    a.p = p;
    a.M = M;
    a.Cin = cin;
    a.K = p.KH * p.KW * cin;
    a.m_tiles = a.n_tiles = 0;
    a.w_row_bytes = (long long)((a.K + BK - 1) / BK) * ROWB;
    return p.w_format == FF_W_F16 ? dispatch<1>(a, s) : dispatch<3>(a, s);
}
}  // namespace ff

extern "C" int ff_pack_split_f16(const float* src, void* dst, long long rows, int K, void* stream) {
    FF_REQUIRE(src && dst && rows > 0 && K > 0, "ff_pack_split_f16: bad argument");
    FF_REQUIRE(ff::aligned16(dst), "ff_pack_split_f16: dst not 16-byte aligned");
    const int nchunks = (K + 31) / 32;
    const long long total = rows * nchunks * 4;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    const int vec = (K & 3) == 0 && ff::aligned16(src);
    pack_split_kernel<<<(unsigned)g, 256, 0, static_cast<hipStream_t>(stream)>>>(src, static_cast<_Float16*>(dst), rows, K, nchunks, vec);
    return ff::check_launch("ff_pack_split_f16");
}
