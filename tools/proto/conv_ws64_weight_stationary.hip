// PROTOTYPE (round 5, not built): weight-stationary persistent kernel for the encoders' 64 -> 64 3x3 layers - the follow-up of
// conv_dma_f32_persistent.hip: a wave's weights (144 registers) loaded once per block, every other global read of the tile loop
// an LDS-DMA the kernel counts itself, unconditional buffer stores that are never waited for.  Built, parity-green on the shapes
// tried (3.6e-6 against fp64, statistics 4e-5), and measured against conv_patch.hip at 16 x 192 x 256 (tools/bench_ws64.py, same
// box, event-bracketed): plain 260 vs 251 us, normalise-on-load + statistics 363 vs 269 us, residual + BatchNorm fold 316 vs 282 us.
// Why it loses: 144 + 32 accumulator registers leave two waves per SIMD and the allocator still spills 4-31 registers (their
// reloads are vmcnt-counted loads behind the stores: the drain is back); with two waves per SIMD the fragment reads' latency is
// exposed unless the tap loop is software-pipelined by hand (it is not here).  Paste in front of conv_dma.hip's fp32-input kernels
// to build; tools/bench_ws64.py measures it.
// =====================================================================================================================
// WEIGHT-STATIONARY persistent kernel for the encoders' 64 -> 64 3x3 layers (round 5): extractor.py:48-56 at 1/2 resolution,
// the largest block of the step (sixteen launches per forward at 192 x 256, as many input gradients per training step).
//
// In-kernel stamps of the one-tile-per-block forms on this layer (tools/dma_f32_stamps.py, 16 x 192 x 256): prologue 1.9 us +
// main loop 11.3 us + epilogue 6.6 us per block - with TWO 32-channel chunks per tile there is no steady state to hide the
// first patch's round trip and the wait for the block's own stores in.  A persistent block that walks a list of tiles can
// overlap them - but `s_waitcnt vmcnt` retires in issue order, so ANY load that is issued behind the epilogue's stores and
// waited for soon after drains them (tools/proto/conv_dma_f32_persistent.hip: the per-tap weight loads did).  Hence:
//   * a wave's weights - 16 channels x 576 k x two halves = 144 registers - are loaded ONCE per block and stay (conv_stem.hip's
//     plan); at two waves per SIMD they fit beside the 32 accumulator registers;
//   * every other global read of the loop is an LDS-DMA that the kernel counts itself: the raw fp32 patch (converted to split
//     pairs in place, as conv_dma_body's XMODE 1 / 2) and - normalise-on-load - the image's (scale, shift) tables of the chunk;
//   * the epilogue's stores are unconditional buffer stores (a lane without a pixel points out of range), so their count per
//     tile is a constant NST, and the only wait between them and the next tile's second chunk is `vmcnt(NST)`: they have a
//     whole chunk (~5 us) to retire.
// Arithmetic and summation order are conv_dma_body's (chunk -> tap -> w0 x0, w1 x0, w0 x1 into one accumulator).
template <int XMODE, bool STATS>
__device__ __forceinline__ void conv_ws64_body(const DArgs& a) {
    constexpr bool INORM = XMODE == 2;
    constexpr int KH = 3, KW = 3, TH = 8, NW = 4, NCI = 2;
    constexpr int PW = 18, PH = 10, NPIX = PH * PW, NPIECE = (NPIX + 7) / 8, NPP = (NPIECE + NW - 1) / NW;
    constexpr int PBYTES = (NPIECE + 1) * 1024, COEF = NPIECE * 1024;         // one more piece per buffer: the chunk's (scale, shift) tables
    constexpr int NT = 9, NTHR = 64 * NW, NCV = (NPIX * 4 + NTHR - 1) / NTHR, UG = 1;
    constexpr int NST = TH + (STATS ? 4 : 0);           // store instructions of a wave per tile (all unconditional)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int n0 = wave * 16;
    const long long pix_total = (long long)p.B * H * W;
    const int total = a.n_tiles, G = gridDim.x;
    const bool xcd_runs = (G & 7) == 0;
    auto decode = [&](int v, int& bimg, int& y0, int& x0) {
        int pt = v;
        if (xcd_runs) {
            const int q8 = total >> 3, r8 = total & 7, x = v & 7;
            pt = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (v >> 3);
        }
        const int tx = pt % a.tiles_x; pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        bimg = pt / a.tiles_y;
        y0 = ty * TH; x0 = tx * 16;
    };
    const int ldb = p.x_ld[0] * 4;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[0]), 0, (int)(pix_total * ldb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INORM ? p.in_scale : p.x[0]), 0, INORM ? p.B * 64 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INORM ? p.in_shift : p.x[0]), 0, INORM ? p.B * 64 * 4 : 0, 0x00020000);
    auto issue_patch = [&](int bimg, int y0, int x0, int c, int buf) {
        const unsigned soff = (unsigned)(c * 128);
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            if ((wave + NW * j) < NPIECE) {       // wave-uniform
                const int r = (wave + NW * j) * 8 + (lane >> 3);
                const int py = r / PW, px = r - py * PW;
                const int yy = y0 - 1 + py, xx = x0 - 1 + px;
                const bool in = r < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const unsigned voff = in ? __umul24((unsigned)((bimg * H + yy) * W + xx), (unsigned)ldb) + (unsigned)((lane & 7) * 16) : OOB;
                dma_piece(voff, rsx, lds0 + buf * PBYTES + (wave + NW * j) * 1024, soff);
            }
        }
        if constexpr (INORM) {      // (every wave issues it - the same 256 bytes - so that every wave's count of pieces is the same)
            const unsigned cv = lane < 8 ? (unsigned)((bimg * 64 + c * 32) * 4 + lane * 16) : OOB;
            const unsigned ch = (lane >= 8 && lane < 16) ? (unsigned)((bimg * 64 + c * 32) * 4 + (lane - 8) * 16) : OOB;
            if (wave == 0) {
                dma_piece(cv, rsc, lds0 + buf * PBYTES + COEF, 0);
                dma_piece(ch, rsh, lds0 + buf * PBYTES + COEF, 0);
            }
        }
    };
    float xs = 1.f, xinv_in = 1.f;
    ff::input_scale(p.x_amax, xs, xinv_in);
    const int s4 = tid & 3;
    auto convert = [&](int y0, int x0, int buf) {
        // one item (= one patch row's k-group) at a time: the four lanes of a row read before they write, rows do not interact -
        // and the wave's 144 weight registers leave no room for three items' values at once
        const unsigned base = lds0 + buf * PBYTES + (unsigned)(tid >> 2) * 128;
        f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = m0, a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if constexpr (INORM) {      // the chunk's tables arrived with the patch: scale in bytes 0..127 of the coefficient piece, shift in 128..255
            const unsigned cb = lds0 + buf * PBYTES + COEF + s4 * 32;
            m0 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(cb);
            m1 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(cb + 16);
            a0 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(cb + 128);
            a1 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(cb + 144);
        }
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int r = (tid >> 2) + (NTHR / 4) * i;
            if (r < NPIX) {
                const int py = r / PW, px = r - py * PW;
                f32x4 v0 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + i * (NTHR * 32));
                f32x4 v1 = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + 16 + i * (NTHR * 32));
                if constexpr (INORM) {
                    v0 = __builtin_elementwise_fma(v0, m0, a0);
                    v1 = __builtin_elementwise_fma(v1, m1, a1);
                    if (p.in_act == FF_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] = v0[j] < 0.f ? 0.f : v0[j]; v1[j] = v1[j] < 0.f ? 0.f : v1[j]; }
                    }
                    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
                    if (!((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)) v0 = v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                ff::ff_f16x4 h00, h01, h10, h11;
                ff::split_pair4(v0 * xs, h00, h10);
                ff::split_pair4(v1 * xs, h01, h11);
                f16x8 x0v, x1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { x0v[j] = h00[j]; x0v[4 + j] = h01[j]; x1v[j] = h10[j]; x1v[4 + j] = h11[j]; }
                const unsigned key = (unsigned)((px >> 1) & 7);
                const unsigned row = base + i * (NTHR * 32);
                asm volatile("" ::: "memory");       // (the reads of this row before its writes, also after the optimiser)
                *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)s4 ^ key) << 4)) = x0v;
                *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)(4 + s4) ^ key) << 4)) = x1v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- the wave's weights: kc = t * NCI + c, two halves, loaded once
    const int i16 = lane & 15, g16 = lane >> 4;
    const bool frag = p.w_frag != nullptr;
    const int ntile16 = (p.Cout + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rsw = frag ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_frag), 0, ntile16 * a.nkc * 2048, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long long)p.Cout * a.w_row_bytes), 0x00020000);
    const int kc_stride = frag ? 2048 : 128, term_off = frag ? 1024 : 64;
    int woff;
    {
        const int n = n0 + i16;
        if (frag) woff = (n >> 4) < ntile16 ? ((n >> 4) * a.nkc * 2048 + lane * 16) : (int)(OOB - 2048);
        else woff = n < p.Cout ? (int)(n * a.w_row_bytes) + g16 * 16 : (int)(OOB - 2048);
    }
    f16x8 wk[NCI][NT][2];
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wk[c][t][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff, (t * NCI + c) * kc_stride, 0));
            wk[c][t][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + term_off, (t * NCI + c) * kc_stride, 0));
        }
    // (opaque to the optimiser from here on: left alone it RE-LOADS the loop-invariant weights inside the tile loop to save
    // registers - 36 loads per tile behind the epilogue's stores, exactly what this kernel exists to avoid)
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("" : "+v"(wk[c][t][0]));
            asm volatile("" : "+v"(wk[c][t][1]));
        }
    // ---- fragment addresses: one pair per dx (the swizzle key depends on the pixel column), everything else immediates
    const int pcol = PI16(i16);
    unsigned xa0[3];          // (the x1 half of a pixel's chunk sits four slots on: address ^ 64 - smem is 128-byte aligned)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int px = pcol + dx;
        const unsigned sw = (unsigned)((px >> 1) & 7);
        xa0[dx] = lds0 + (unsigned)(px * 128) + ((g16 ^ sw) << 4);
    }
    // ---- epilogue constants
    const int n4 = n0 + g16 * 4;
    const bool has_n = n4 < p.Cout;
    const int nb4 = min(n4, p.Cout - 4);         // (Cout % 4 == 0: a lane's four channels exist or none does)
    const float xinv = ff::SPLIT_INV * xinv_in;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(pix_total * p.y_ld * 4), 0x00020000);
    const int nparts = a.tiles_y * a.tiles_x;
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(STATS ? p.stats_part : p.y, 0, STATS ? (int)((long long)p.B * nparts * p.Cout * 16) : 0, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    int v = blockIdx.x;
    int cb, cy0, cx0, nb = 0, ny0 = 0, nx0 = 0;
    decode(v, cb, cy0, cx0);
    bool has_next = v + G < total;
    if (has_next) decode(v + G, nb, ny0, nx0);
    bool first = true;
    issue_patch(cb, cy0, cx0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // weights in registers (and the first patch) before the loop's counted waits start
    for (;;) {
        f32x4 acc[TH];
#pragma unroll
        for (int u = 0; u < TH; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCI; ++c) {
            // pieces of (tile, c) have landed once everything older than the wave's youngest ops has: at a tile's top those are
            // the NST stores of the previous tile's epilogue, at the second chunk nothing (and the stores are a chunk old)
            if (c == 0 && !first) wait_vm<NST>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            convert(cy0, cx0, c);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int dy = t / 3, dx = t - dy * 3;
                const unsigned ra = xa0[dx] + (unsigned)(c * PBYTES + dy * PW * 128), rb = (xa0[dx] ^ 64u) + (unsigned)(c * PBYTES + dy * PW * 128);
                auto rows = [&](int lo, int hi) {
#pragma unroll
                    for (int u = lo; u < hi; ++u) {
                        const f16x8 xa = lds_ld16(ra + u * PW * 128);
                        const f16x8 xb = lds_ld16(rb + u * PW * 128);
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[c][t][0], xa, acc[u], 0, 0, 0);
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[c][t][1], xa, acc[u], 0, 0, 0);
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[c][t][0], xb, acc[u], 0, 0, 0);
                    }
                };
                if (t == 0) {
                    rows(0, 2);
                    if (c + 1 < NCI) issue_patch(cb, cy0, cx0, c + 1, c + 1);
                    else if (has_next) issue_patch(nb, ny0, nx0, 0, 0);
                    rows(2, TH);
                } else {
                    rows(0, 4);
                    __builtin_amdgcn_sched_barrier(0);
                    rows(4, TH);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: acc[u][r] = channel n4 + r of pixel (cy0 + u, cx0 + pcol); stores unconditional, never waited for
        {
            const int x = cx0 + pcol;
            // (per tile, not per block: the wave's 144 weight registers leave no room to keep them across the main loop; these
            // loads are older than the tile's stores, so the counted wait at the next tile's top does not see them)
            f32x4 bias = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f}, ct = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + nb4);
            if (p.ch_scale) { cs = *reinterpret_cast<const f32x4*>(p.ch_scale + nb4); ct = *reinterpret_cast<const f32x4*>(p.ch_shift + nb4); }
            f32x4 st_p = {0.f, 0.f, 0.f, 0.f}, st_s1 = st_p, st_s2 = st_p;
            float st_n = 0.f;
#pragma unroll
            for (int ug = 0; ug < TH; ug += UG) {
                f32x4 rr[UG];
                int po[UG];
#pragma unroll
                for (int k = 0; k < UG; ++k) {
                    const int y = cy0 + ug + k;
                    po[k] = (has_n && y < H && x < W) ? (cb * H + y) * W + x : -1;
                    rr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (p.res && po[k] >= 0) rr[k] = *reinterpret_cast<const f32x4*>(p.res + (long long)po[k] * p.res_ld + n4);
                }
                f32x4 vv[UG];
#pragma unroll
                for (int k = 0; k < UG; ++k) {
                    f32x4 t = acc[ug + k] * xinv + bias;
                    t *= p.out_scale;
                    if (p.ch_scale) t = t * cs + ct;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r], p.act);
                    if (p.res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[r] = ff::apply_act(t[r] + rr[k][r], p.act_res);
                    }
                    vv[k] = t;
                }
                if constexpr (STATS) {
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
#pragma unroll
                for (int k = 0; k < UG; ++k) {
                    const unsigned off = po[k] >= 0 ? (unsigned)(po[k] * p.y_ld + n4) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv[k]), rsy, (int)off, 0, 0);
                }
            }
            if constexpr (STATS) {
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
                const int part = (cy0 / TH) * a.tiles_x + (cx0 >> 4);
                const unsigned e = (i16 == 0 && has_n) ? (unsigned)(((cb * nparts + part) * p.Cout + n4) * 16) : OOB;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){st_p[r], st_s1[r], st_s2[r], st_n}), rst, (int)(e == OOB ? OOB : e + 16u * r), 0, 0);
            }
        }
        first = false;
        if (!has_next) break;
        v += G;
        cb = nb; cy0 = ny0; cx0 = nx0;
        has_next = v + G < total;
        if (has_next) decode(v + G, nb, ny0, nx0);
    }
}

template <int XMODE, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_ws64_kernel(const DArgs a) {
    conv_ws64_body<XMODE, STATS>(a);
}

// eligibility of the weight-stationary kernel: 3x3 stride 1, 64 input channels in ONE segment, <= 64 output channels in whole groups of
// four, fp32 output, 16-byte aligned output / residual, and enough pixel tiles for the walk to have a steady state
bool ws64_ok(const FFConvParams& p, int cin) {
    static const bool enabled = !(getenv("FF_CONV_WS64") && atoi(getenv("FF_CONV_WS64")) == 0);      // A/B switch: 0 = conv_patch.hip
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    const long long pix = (long long)p.B * p.H * p.W;
    if (!enabled || p.KH != 3 || p.KW != 3 || p.stride != 1 || dlh != 1 || dlw != 1 || p.groups != 1 || p.pad_h != 1 || p.pad_w != 1) return false;
    if (p.w_format != FF_W_F16X3 || cin != 64 || p.x_c[0] != 64 || p.x_fmt[0] != FF_FMT_F32 || p.Cout > 64 || p.Cout % 4 || p.res2 || p.splitk > 1 || p.ep_mode || p.y_fmt || p.y2) return false;
    if (p.y_ld % 4 || !ff::aligned16(p.y) || pix * p.y_ld * 4 >= (1ll << 31) || pix * p.x_ld[0] * 4 >= (1ll << 31) || pix >= (1ll << 24)) return false;
    if (p.res && (p.res_ld % 4 || !ff::aligned16(p.res))) return false;
    if ((p.bias && !ff::aligned16(p.bias)) || (p.ch_scale && (!ff::aligned16(p.ch_scale) || !ff::aligned16(p.ch_shift)))) return false;
    if (p.in_scale && (p.x_amax || !ff::aligned16(p.in_scale) || !ff::aligned16(p.in_shift) || (p.in_act != FF_ACT_NONE && p.in_act != FF_ACT_RELU))) return false;
    if (p.stats_part && (p.x_amax || !ff::aligned16(p.stats_part))) return false;
    const long long tiles = (long long)p.B * ((p.H + 7) / 8) * ((p.W + 15) / 16);
    if (p.stats_part && tiles / p.B * p.B * p.Cout * 16 >= (1ll << 31)) return false;
    return tiles >= 1024;
}

int launch_ws64(DArgs& a, hipStream_t s) {
    constexpr int NPIECE = (10 * 18 + 7) / 8;
    constexpr size_t lds = 2 * (NPIECE + 1) * 1024;
    a.tiles_y = (a.p.H + 7) / 8;
    const long long total = (long long)a.p.B * a.tiles_y * a.tiles_x;
    a.n_tiles = (int)total;
    const unsigned gx = (unsigned)std::min<long long>(total, 2 * 256);        // two blocks per CU (two waves per SIMD hold the weights)
    const bool inorm = a.p.in_scale != nullptr, stats = a.p.stats_part != nullptr;
    if (inorm && stats) conv_ws64_kernel<2, true><<<gx, 256, lds, s>>>(a);
    else if (inorm) conv_ws64_kernel<2, false><<<gx, 256, lds, s>>>(a);
    else if (stats) conv_ws64_kernel<1, true><<<gx, 256, lds, s>>>(a);
    else conv_ws64_kernel<1, false><<<gx, 256, lds, s>>>(a);
    return ff::check_launch("ff_conv2d_fwd(weight-stationary 64-channel 3x3)");
}

