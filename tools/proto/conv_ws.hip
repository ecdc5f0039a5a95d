// PROTOTYPE (round 1-3 experiment, not built into libfocusflow_hip.so since round 4): measured on par with conv_patch.hip,
// 3 % slower end to end.  Kept for the record of what it measured.
// Wave-specialised, persistent, patch-stationary fp16x3 convolution (stride-1 "same" convs, Cin % 32 == 0).
//
// Findings that shaped it ("conv kernels"): in conv_patch.hip the memory phase and the MFMA phase of
// a block add up, and what bounds the memory phase is not bandwidth but the serial instruction stream of the
// loading waves (address arithmetic, segment selection, fp16 splitting): ~170 instructions per step per wave.
// CDNA4 runs a VALU/memory wave and an MFMA wave of the same SIMD concurrently, so the roles are split and the
// loader's per-step stream is cut to a few dozen instructions:
//
//   waves 0-3  "math"   : per step 12 ds_read_b128 + 12 MFMAs on the stationary patch; no global loads.
//   waves 4-7  "loader" : per step   store the weights of step g+1 (loaded D steps earlier) and reload the slot;
//                                    store slice t of the NEXT chunk's input patch (loaded one chunk earlier,
//                                    fp32 -> (x0, x1) fp16 split on the way) and reload the slot for the chunk
//                                    after that.  Tile geometry lives in registers (recomputed per tile, not
//                                    per step), the segment descriptor is rebuilt per chunk, the taps are
//                                    unrolled so every register-ring index is static, and every path issues the
//                                    same loads so the compiler's vmcnt bookkeeping stays exact: loads stay in
//                                    flight across the per-step barrier, which is a raw s_barrier behind
//                                    s_waitcnt lgkmcnt(0) only (__syncthreads() would drain vmcnt).
//
// A step = one kernel tap of one 32-channel chunk.  Blocks are persistent: each walks a list of output tiles
// (8x16 pixels x 64 output channels), so the loaders fetch the next tile's first patch while the math waves
// finish the current one and the per-block prologue is paid once per CU slot, not once per tile.
// LDS: patch double-buffered (2 x (8+KH-1)(16+KW-1) rows), weights double-buffered (2 x 64 rows); rows are
// 128 B of (x0 | x1) fp16 data at a 144-byte pitch.
#include <algorithm>
#include <cstdlib>
#include "ff_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16, BN = 64, ROWB = 128;
constexpr int ROWP = 144;   // LDS row pitch: 128 B of data + 16 B pad. 16 lanes reading 16 B at this pitch hit 64 distinct
                            // banks, and with no XOR swizzle every tap is a compile-time offset from one base register.
constexpr int OOB = 0x7fffffff;

struct WArgs {
    FFConvParams p;
    int Cin, nci;
    int tiles_x, tiles_y, n_tiles, total_tiles;
    long long w_row_bytes;
    int abl;                  // diagnostics (FF_WS_ABLATE) bits: 1 math waves idle, 2 loaders issue no memory traffic,
                              // 4 loaders skip their LDS stores, 8 math waves skip their LDS reads (results are garbage)
};

struct Tile { int y0, x0, bimg, n0, valid; };

__device__ __forceinline__ void split4(const f32x4 v, f16x4& h0, f16x4& h1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 a = (_Float16)v[j];
        h0[j] = a;
        h1[j] = (_Float16)(v[j] - (float)a);
    }
}

__device__ __forceinline__ void step_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS traffic is complete
    __builtin_amdgcn_s_barrier();                         // raw: global loads stay in flight across it
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ int sreg(int v) { return __builtin_amdgcn_readfirstlane(v); }
// mask arithmetic instead of ?: chains: LLVM turns a select chain on one index into a scratch lookup table
__device__ __forceinline__ int sel3(int seg, int v0, int v1, int v2) {
    return (v0 & -(int)(seg == 0)) | (v1 & -(int)(seg == 1)) | (v2 & -(int)(seg == 2));
}

template <int KH, int KW, int TERMS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_ws_kernel(const WArgs a) {
    constexpr int NT = KH * KW;                       // steps per chunk
    constexpr int D = (NT % 3 == 0) ? 3 : NT;         // weight ring depth = prefetch distance in steps (divides NT)
    constexpr int PW = TW + KW - 1, PH = TH + KH - 1, NPIX = PH * PW;
    // The patch is stored in NSL <= NT slices, one per step, each NS items (pixel x 16-byte group) per loader
    // thread with all 256 loader threads busy: fewer, fuller slices = fewer registers in the patch ring.
    constexpr int NS = (NPIX + 32 * NT - 1) / (32 * NT);
    constexpr int SPX = 32 * NS;                      // patch pixels per slice
    constexpr int NSL = (NPIX + SPX - 1) / SPX;       // slices actually used (steps NSL..NT-1 move weights only)
    constexpr int PATCH_BYTES = (NPIX * ROWP + 255) & ~255;
    constexpr int NPRO = (NPIX * 8 + 511) / 512;      // prologue items per thread
    static_assert(NT % D == 0 && D + 1 <= 2 * NT && NT % 2 == 1, "ring depth must divide the taps; odd tap count");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FFConvParams& p = a.p;
    constexpr int OFF_W = 2 * PATCH_BYTES;            // smem: [2][PATCH_BYTES] patch, then [2][BN][ROWP] weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sreg(tid >> 6);                  // scalar: the role branch below is wave-uniform
    const int H = p.H, W = p.W, nci = a.nci;

    // ---- this block's tile list: each XCD (blockIdx % 8) owns a contiguous range of tiles (halo rows and the
    // weights stay in that XCD's L2); within the XCD its blocks take tiles round-robin.
    const int nblk = gridDim.x, per = nblk >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int t8 = (a.total_tiles + 7) >> 3;
    const int rbeg = xcd * t8, rend = min(rbeg + t8, a.total_tiles);
    if (rbeg + slot >= rend) return;
    const int n_mine = (rend - rbeg - slot + per - 1) / per;        // tiles rbeg + slot + k * per
    const int Q = n_mine * nci;                       // chunks this block walks
    auto decode = [&](int k) {
        Tile t;
        t.valid = k < n_mine;
        int id = rbeg + slot + (t.valid ? k : 0) * per;
        const int nt = id % a.n_tiles; id /= a.n_tiles;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        t.y0 = (id % a.tiles_y) * TH; t.x0 = tx * TW; t.bimg = id / a.tiles_y; t.n0 = nt * BN;
        return t;
    };

    // Segment descriptors as explicit scalars (readfirstlane keeps the compiler from turning the per-chunk
    // segment choice into a dynamically indexed copy of the kernel arguments in scratch).
    const long long pix_total = (long long)p.B * H * W;
    const int c0 = sreg(p.x_c[0]), c01 = sreg(p.x_c[0] + p.x_c[1]);
    const int ldb0 = sreg(p.x_ld[0] * 4), ldb1 = sreg(p.x_ld[1] * 4), ldb2 = sreg(p.x_ld[2] * 4);
    const unsigned long long a0 = (unsigned long long)p.x[0], a1 = p.x[1] ? (unsigned long long)p.x[1] : a0,
                             a2 = p.x[2] ? (unsigned long long)p.x[2] : a0;
    const int b0l = sreg((int)a0), b0h = sreg((int)(a0 >> 32)), b1l = sreg((int)a1), b1h = sreg((int)(a1 >> 32)),
              b2l = sreg((int)a2), b2h = sreg((int)(a2 >> 32));
    const int nb0 = sreg((int)(pix_total * p.x_ld[0] * 4)), nb1 = sreg(p.x[1] ? (int)(pix_total * p.x_ld[1] * 4) : 0),
              nb2 = sreg(p.x[2] ? (int)(pix_total * p.x_ld[2] * 4) : 0);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long long)p.Cout * a.w_row_bytes), 0x00020000);
    struct Seg { __amdgpu_buffer_rsrc_t rs; int ldb, cbyte; };
    auto segment = [&](int c) {                       // 32-channel chunk c of the (virtually concatenated) input
        const int ci = c * 32;
        const int seg = (int)(ci >= c0) + (int)(ci >= c01);
        Seg s;
        const unsigned lo = (unsigned)sel3(seg, b0l, b1l, b2l), hi = (unsigned)sel3(seg, b0h, b1h, b2h);
        s.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 sel3(seg, nb0, nb1, nb2), 0x00020000);
        s.ldb = sel3(seg, ldb0, ldb1, ldb2);
        s.cbyte = (ci - sel3(seg, 0, c0, c01)) * 4;
        return s;
    };
    // image pixel index of patch pixel px of tile t, or -1 (outside the image / patch / tile list)
    auto patch_pixel = [&](const Tile& t, int px, bool ok) {
        const int py = px / PW, pxx = px - py * PW;   // PW is a compile-time constant
        const int yy = t.y0 - KH / 2 + py, xx = t.x0 - KW / 2 + pxx;
        return (ok && t.valid && px < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (t.bimg * H + yy) * W + xx : -1;
    };
    auto load_px = [&](const Seg& s, int pix, int kq16) {
        const int off = pix >= 0 ? (int)__umul24((unsigned)pix, (unsigned)s.ldb) + kq16 : OOB;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rs, off, s.cbyte, 0));
    };
    auto store_px = [&](int off, const f32x4 v) {     // off = LDS byte offset of the x0 half-group (x1: piece + 4)
        f16x4 h0, h1;
        split4(v * ff::XSPLIT, h0, h1);
        *reinterpret_cast<f16x4*>(smem + off) = h0;
        if (TERMS == 3) *reinterpret_cast<f16x4*>(smem + off + 64) = h1;
    };
    auto lds_px = [&](int px, int kq) { return px * ROWP + kq * 8; };

    // ---- prologue (all 512 threads): patch of chunk 0 of the first tile -> patch buffer 0, weights of step 0
    {
        const Tile t0 = decode(0);
        const Seg s0 = segment(0);
        f32x4 v[NPRO];
#pragma unroll
        for (int i = 0; i < NPRO; ++i) {
            const int item = tid + 512 * i;
            v[i] = load_px(s0, patch_pixel(t0, item >> 3, true), (item & 7) * 16);
        }
        const int row = tid >> 3, kq = tid & 7, n = t0.n0 + row;      // 64 rows x 8 pieces = 512 threads
        const f32x4 wv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsw, n < p.Cout ? (int)(n * a.w_row_bytes) + kq * 16 : OOB, 0, 0));
#pragma unroll
        for (int i = 0; i < NPRO; ++i) {
            const int item = tid + 512 * i;
            if (item < NPIX * 8) store_px(lds_px(item >> 3, item & 7), v[i]);
        }
        *reinterpret_cast<f32x4*>(smem + OFF_W + row * ROWP + kq * 16) = wv;
    }

    if (wave >= 4) {
        // ================= loader waves =================
        const int lt = tid - 256, kq = lt & 7, pl = lt >> 3;          // pl = pixel within a slice (item 0)
        const bool mem = !(a.abl & 2), lst = !(a.abl & 4);
        int wvA[2], wvB[2], wdst[2];                   // weight-row offsets of the current / next tile; LDS offsets
        int pix[NSL][NS];
        const int pdst = lds_px(pl, kq);               // slice t, item i: + (t * SPX + 32 * i) * ROWP (immediate)
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int row = pl + 32 * i; wdst[i] = OFF_W + row * ROWP + kq * 16; }
        auto set_tile = [&](const Tile& t) {           // geometry of the patch loads
#pragma unroll
            for (int tt = 0; tt < NSL; ++tt)
#pragma unroll
                for (int i = 0; i < NS; ++i) pix[tt][i] = patch_pixel(t, tt * SPX + pl + 32 * i, mem);
        };
        auto wrows = [&](int n0, int (&wv)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + pl + 32 * i;
                wv[i] = (mem && n < p.Cout) ? (int)(n * a.w_row_bytes) + kq * 16 : OOB;
            }
        };
        Tile cur = decode(0), nxt = decode(1);
        wrows(cur.n0, wvA);
        wrows(nxt.n0, wvB);
        f32x4 rw[D][2], rp[NSL][NS];
        // weight job j = the weights of step j+1: stored during step j, loaded D steps earlier (ring slot j % D).
        // The pipeline is primed by the load half of a virtual chunk -1 of the first tile, issued in exactly the
        // loop's order (jobs that do not exist load out of range), so that the pending-load counts the compiler
        // derives at the loop head are the steady-state ones and the prefetch distance is not cut short.
        {
            const Seg s1 = segment(1);                 // patch of chunk 1 (stored during chunk 0): same tile
            set_tile(cur);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = t - (NT - D);            // steps 1..D lie in chunks 0/1 of the first tile (nci >= 2)
                const int st = j + 1, cc = st / NT, tp = st % NT;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    rw[t % D][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsw, j >= 0 ? wvA[i] : OOB, j >= 0 ? (tp * nci + cc) * ROWB : 0, 0));
                if (t < NSL) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) rp[t][i] = load_px(s1, pix[t][i], kq * 16);
                }
            }
        }
        step_barrier();                                // prologue stores visible

        int c = 0;                                     // chunk q = k * nci + c of tile `cur`
        for (int q = 0; q < Q; ++q) {
            // patch loads of this iteration fetch chunk q+2: chunk c+2 of this tile or chunk c+2-nci of the next
            const bool pwrap = c + 2 >= nci;
            const int c2 = pwrap ? c + 2 - nci : c + 2;
            if (c2 == 0) set_tile(nxt);                // the loads enter the next tile (invalid past the list)
            const Seg s2 = segment(c2);
            // weight loads fetch the steps (q, t) + D + 1: chunk c, c+1 or c+2, possibly in the next tile
            const bool w1 = c + 1 >= nci, w2 = c + 2 >= nci;
            const int cw1 = w1 ? c + 1 - nci : c + 1, cw2 = w2 ? c + 2 - nci : c + 2;
            const int pbuf = ((q + 1) & 1) * PATCH_BYTES;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int wbuf = ((q * NT + t + 1) & 1) * (BN * ROWP);   // buffer of the step whose weights are stored now
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (lst || rw[t % D][i][0] == 12345.f) *reinterpret_cast<f32x4*>(smem + wbuf + wdst[i]) = rw[t % D][i];
                const int ts = t + D + 1, inc = ts / NT, tp = ts % NT;   // compile-time after unrolling
                const int cw = inc == 0 ? c : (inc == 1 ? cw1 : cw2);
                const bool ww = inc == 0 ? false : (inc == 1 ? w1 : w2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    rw[t % D][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsw, ww ? wvB[i] : wvA[i], (tp * nci + cw) * ROWB, 0));
                if (t < NSL) {                         // compile-time
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        if (t * SPX + pl + 32 * i < NPIX && (lst || rp[t][i][0] == 12345.f)) store_px(pbuf + pdst + (t * SPX + 32 * i) * ROWP, rp[t][i]);
                        rp[t][i] = load_px(s2, pix[t][i], kq * 16);
                    }
                }
                step_barrier();
            }
            if (++c == nci) {                          // next tile
                c = 0;
                cur = nxt;
                nxt = decode(q / nci + 2);
#pragma unroll
                for (int i = 0; i < 2; ++i) wvA[i] = wvB[i];
                wrows(nxt.n0, wvB);
            }
        }
        return;
    }

    // ================= math waves =================
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int lrow = li >> 4, lcol = li & 15;
    const int brow = wn * 32 + li;
    const bool idle = a.abl & 1, nolds = a.abl & 8;
    step_barrier();                                    // prologue stores visible
    int q = 0;
    for (int k = 0; k < n_mine; ++k) {
        const Tile tl = decode(k);
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int c = 0; c < nci; ++c, ++q) {
            // per-chunk base pointers; every tap / k-slice / term below is a compile-time offset from them
            // (folds into the ds_read offset field: no per-step address arithmetic, nothing to hoist and spill)
            const char* pa[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                pa[tt] = smem + (q & 1) * PATCH_BYTES + (((wm * 2 + tt) * 2 + lrow) * PW + lcol) * ROWP + lh * 16;
            const char* pw[2];                         // weight buffer of even / odd taps (NT is odd)
            pw[0] = smem + OFF_W + (q & 1) * (BN * ROWP) + brow * ROWP + lh * 16;
            pw[1] = smem + OFF_W + ((q & 1) ^ 1) * (BN * ROWP) + brow * ROWP + lh * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int dy = t / KW, dx = t % KW;
                if (!idle) {
                    // per 16-channel k-slice: its 6 fragment reads (distinct registers), then its 6 MFMAs behind
                    // counted waits.  Left to itself the scheduler funnels every fragment through one register
                    // and exposes each LDS latency; holding both k-slices (48 registers) spills.
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f16x8 x0[2], x1[2], w0, w1;
                        if (nolds) {                   // diagnostics: MFMAs on whatever the registers hold
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) { asm volatile("" : "=v"(x0[tt])); asm volatile("" : "=v"(x1[tt])); }
                            asm volatile("" : "=v"(w0)); asm volatile("" : "=v"(w1));
                        } else {
                        w0 = *reinterpret_cast<const f16x8*>(pw[t & 1] + 2 * s * 16);
                        if (TERMS == 3) w1 = *reinterpret_cast<const f16x8*>(pw[t & 1] + (4 + 2 * s) * 16);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            x0[tt] = *reinterpret_cast<const f16x8*>(pa[tt] + (dy * PW + dx) * ROWP + 2 * s * 16);
                            if (TERMS == 3) x1[tt] = *reinterpret_cast<const f16x8*>(pa[tt] + (dy * PW + dx) * ROWP + (4 + 2 * s) * 16);
                        }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[tt], w0, acc[tt], 0, 0, 0);
                        if (TERMS == 3) {
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[tt], w1, acc[tt], 0, 0, 0);
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[tt], w0, acc[tt], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                step_barrier();
            }
        }
        // epilogue of this tile (no LDS): the loaders are already streaming the next tile
        const int n = tl.n0 + wn * 32 + li;
        if (n < p.Cout) {
            const float bias = p.bias ? p.bias[n] : 0.f;
            const float cs = p.ch_scale ? p.ch_scale[n] : 1.f;
            const float ct = p.ch_scale ? p.ch_shift[n] : 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int y = tl.y0 + (wm * 2 + tt) * 2 + (pi >> 4), x = tl.x0 + (pi & 15);
                    if (y >= H || x >= W) continue;
                    const long long m = ((long long)tl.bimg * H + y) * W + x;
                    float v = acc[tt][r] * ff::SPLIT_INV + bias;
                    v *= p.out_scale;
                    if (p.ch_scale) v = v * cs + ct;
                    v = ff::apply_act(v, p.act);
                    if (p.res) v = ff::apply_act(v + p.res[m * p.res_ld + n], p.act_res);
                    p.y[m * p.y_ld + n] = v;
                }
            }
        }
    }
}

template <int KH, int KW, int TERMS>
int launch(const WArgs& a, hipStream_t s) {
    constexpr int NPIX = (TH + KH - 1) * (TW + KW - 1);
    constexpr size_t lds = 2 * (size_t)((NPIX * ROWP + 255) & ~255) + 2 * BN * ROWP;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<KH, KW, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        once = true;
    }
    // persistent grid: two 8-wave blocks per CU, a multiple of 8 so every XCD gets the same number of blocks
    static const int per_cu = getenv("FF_WS_BLOCKS_PER_CU") ? atoi(getenv("FF_WS_BLOCKS_PER_CU")) : 2;
    int blocks = std::min(256 * per_cu, (a.total_tiles + 7) / 8 * 8);
    blocks = std::max(8, blocks / 8 * 8);
    conv_ws_kernel<KH, KW, TERMS><<<(unsigned)blocks, 512, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_fwd(ws)");
}

}  // namespace

namespace ff {
// FF_OK if launched, 1 if not eligible
int conv2d_fwd_ws(const FFConvParams& p, int cin, hipStream_t s) {
    static const bool enabled = getenv("FF_WS_CONV") != nullptr ? atoi(getenv("FF_WS_CONV")) != 0 : false;   // opt-in: on par with conv_patch.hip, see DESIGN.md
    if (!enabled) return 1;
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1 || p.x_amax || p.in_scale) return 1;
    if (p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2) return 1;
    const int shape = p.KH * 16 + p.KW;
    if (shape != 0x33 && shape != 0x15 && shape != 0x51) return 1;
    if (cin % 32 || cin < 64) return 1;                       // at least two chunks (the loader's pipeline depth)
    const long long pix_total = (long long)p.B * p.H * p.W;
    if (pix_total >= (1ll << 24)) return 1;                  // 24-bit pixel index arithmetic
    long long max_bytes = 0;
    for (int i = 0; i < FF_MAX_SEG; ++i) {
        if (p.x_c[i] % 32) return 1;
        if (p.x_c[i]) {
            if ((long long)p.x_ld[i] * 4 >= (1 << 24)) return 1;
            max_bytes = std::max(max_bytes, pix_total * p.x_ld[i] * 4);
        }
    }
    WArgs a;
    a.p = p;
    a.Cin = cin;
    a.nci = cin / 32;
    a.tiles_x = (p.W + TW - 1) / TW;
    a.tiles_y = (p.H + TH - 1) / TH;
    a.n_tiles = (p.Cout + BN - 1) / BN;
    a.total_tiles = p.B * a.tiles_y * a.tiles_x * a.n_tiles;
    a.w_row_bytes = (long long)((p.KH * p.KW * cin + 31) / 32) * ROWB;
    max_bytes = std::max(max_bytes, (long long)p.Cout * a.w_row_bytes);
    if (max_bytes >= (1ll << 31)) return 1;
    static const int abl = getenv("FF_WS_ABLATE") ? atoi(getenv("FF_WS_ABLATE")) : 0;
    a.abl = abl;
    const bool t3 = p.w_format == FF_W_F16X3;
    if (shape == 0x33) return t3 ? launch<3, 3, 3>(a, s) : launch<3, 3, 1>(a, s);
    if (shape == 0x15) return t3 ? launch<1, 5, 3>(a, s) : launch<1, 5, 1>(a, s);
    return t3 ? launch<5, 1, 3>(a, s) : launch<5, 1, 1>(a, s);
}
}  // namespace ff
