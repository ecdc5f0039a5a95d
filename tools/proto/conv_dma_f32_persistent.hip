// PROTOTYPE (round 5, not built): a persistent form of conv_dma.hip's fp32-input route - a block walks a list of pixel tiles, the
// next tile's first patch is DMA'd under the current tile's last chunk and the epilogue's (unconditional buffer) stores are
// meant to stay in flight.  Compiled clean (152-167 registers, no spills) and then read in the ISA: `s_waitcnt vmcnt(N)`
// retires in issue order, and every tap needs a weight set that was loaded one step earlier - i.e. AFTER the stores - so
// the first weight wait of the next tile (vmcnt(2), one step = 0.6 us behind the epilogue) drains the stores whatever the
// explicit counted wait at the tile's top allows.  Keeping the stores in flight for their 5-6 us needs a tile's weights in
// registers before them (144 registers for a 64-channel 3x3 layer: the weight-stationary form of conv_stem.hip), which is a
// different kernel.  In-kernel stamps of the one-tile-per-block form this was meant to beat (tools/dma_f32_stamps.py,
// 64 -> 64 at 16 x 192 x 256): prologue 1.9 us, main loop 11.3 us, epilogue 6.6 us per block; six-row tiles at four blocks
// per CU: 2.0 / 13.2 / 4.9 us and the same launch time - the loop phases share a saturated resource, occupancy is not the lever.
// (Paste into conv_dma.hip in front of the fp32-input kernels to build it: it uses that file's helpers.)
// =====================================================================================================================
// PERSISTENT fp32-input kernel (round 5).  In-kernel stamps of the one-tile-per-block form above on the encoders' layers
// (tools/dma_f32_stamps.py: 64 -> 64 at 16 x 192 x 256): prologue 1.9 us + main loop 11.3 us + epilogue 6.6 us per block -
// two fifths of a block's life are the first patch's round trip and the wait for its own stores, and with two 32-channel
// chunks per tile there is no steady state to hide them in.  Here a block walks a LIST of pixel tiles (grid.x ~ the number
// of resident blocks, grid.y = channel tiles): the next tile's first patch is DMA'd under the current tile's last chunk,
// the next tile's first weights are in registers when the epilogue starts, and the epilogue's stores are never waited for:
// `s_waitcnt vmcnt(N)` retires in issue order, so the wait at the next tile's top leaves exactly this tile's NST store
// instructions in flight - which is why every store of the epilogue is an UNCONDITIONAL buffer store (a lane with nothing
// to write points out of range and the hardware drops it): the count must not depend on the data.
template <int KH, int KW, int TH, int NW, int XMODE, bool STATS>
__device__ __forceinline__ void conv_f32p_body(const DArgs& a) {
    constexpr bool INORM = XMODE == 2;
    constexpr int PW = 16 + KW - 1, PH = TH + KH - 1, NPIX = PH * PW, NPIECE = (NPIX + 7) / 8, NPP = (NPIECE + NW - 1) / NW;
    constexpr int PBYTES = NPIECE * 1024, NT = KH * KW, NWL = 2, NTHR = 64 * NW, NCV = (NPIX * 4 + NTHR - 1) / NTHR;
    constexpr int UG = TH % 4 == 0 ? 4 : (TH % 3 == 0 ? 3 : TH);
    constexpr int NST = TH + (STATS ? 4 : 0);           // store instructions of a wave per tile (all unconditional)
    static_assert(PW % 2 == 0 && NT >= 3, "see conv_dma_body");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem;
    const FFConvParams& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int n0 = blockIdx.y * (16 * NW) + wave * 16;          // this WAVE's first output channel (fixed for the block's life)
    const long long pix_total = (long long)p.B * H * W;
    const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];
    const int nci = a.nci;
    const int total = a.n_tiles, G = gridDim.x;                    // pixel tiles in all, blocks walking them
    const bool xcd_runs = (G & 7) == 0;
    // tile v of this block's list -> (image, y0, x0); with G % 8 == 0 the tiles of one XCD (v % 8 = blockIdx.x % 8) are a contiguous run
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
    const int ld0 = p.x_ld[0] * 4, ld1 = p.x_ld[1] * 4, ld2 = p.x_ld[2] * 4;
    const unsigned long long xp0 = (unsigned long long)p.x[0], xp1 = (unsigned long long)p.x[1], xp2 = (unsigned long long)p.x[2];
    auto issue_patch = [&](int bimg, int y0, int x0, int c, int buf) {          // chunk c of tile (bimg, y0, x0) -> patch buffer buf, raw fp32
        const int ci = c * 32;
        const int in0 = -(int)(ci < c0), in1 = -(int)(ci >= c0 && ci < c01), in2 = -(int)(ci >= c01);
        const int ldb = (ld0 & in0) | (ld1 & in1) | (ld2 & in2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((ci - (c0 & (in1 | in2)) - ((c01 - c0) & in2)) * 4);
        const unsigned long long xp = (xp0 & (unsigned long long)(long long)in0) | (xp1 & (unsigned long long)(long long)in1) | (xp2 & (unsigned long long)(long long)in2);
        const unsigned long long xpu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)xp);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(xpu), 0, __builtin_amdgcn_readfirstlane((int)(pix_total * ldb)), 0x00020000);
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            if ((wave + NW * j) < NPIECE) {       // wave-uniform
                const int r = (wave + NW * j) * 8 + (lane >> 3);
                const int py = r / PW, px = r - py * PW;
                const int yy = y0 - p.pad_h + py, xx = x0 - p.pad_w + px;
                const bool in = r < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const unsigned voff = in ? __umul24((unsigned)((bimg * H + yy) * W + xx), (unsigned)ldb) + (unsigned)((lane & 7) * 16) : OOB;
                dma_piece(voff, rs, lds0 + buf * PBYTES + (wave + NW * j) * 1024, soff);
            }
        }
    };
    float xs = 1.f, xinv_in = 1.f;
    ff::input_scale(p.x_amax, xs, xinv_in);
    const int s4 = tid & 3;
    auto convert = [&](int bimg, int y0, int x0, int c, int buf) {       // in place: fp32 rows -> split pairs (see conv_dma_body)
        const unsigned base = lds0 + buf * PBYTES + (unsigned)(tid >> 2) * 128;
        f32x4 va[NCV], vb[NCV];
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            va[i] = vb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((tid >> 2) + (NTHR / 4) * i < NPIX) {
                va[i] = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + i * (NTHR * 32));
                vb[i] = *(__attribute__((address_space(3))) const f32x4*)(unsigned long)(base + s4 * 32 + 16 + i * (NTHR * 32));
            }
        }
        f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = m0, a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if constexpr (INORM) {
            const long long t = (long long)bimg * a.Cin + c * 32 + s4 * 8;
            m0 = *reinterpret_cast<const f32x4*>(p.in_scale + t);
            m1 = *reinterpret_cast<const f32x4*>(p.in_scale + t + 4);
            a0 = *reinterpret_cast<const f32x4*>(p.in_shift + t);
            a1 = *reinterpret_cast<const f32x4*>(p.in_shift + t + 4);
        }
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int r = (tid >> 2) + (NTHR / 4) * i;
            if (r >= NPIX) continue;
            const int py = r / PW, px = r - py * PW;
            f32x4 v0 = va[i], v1 = vb[i];
            if constexpr (INORM) {
                v0 = __builtin_elementwise_fma(v0, m0, a0);
                v1 = __builtin_elementwise_fma(v1, m1, a1);
                if (p.in_act == FF_ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v0[j] = v0[j] < 0.f ? 0.f : v0[j]; v1[j] = v1[j] < 0.f ? 0.f : v1[j]; }
                }
                const int yy = y0 - p.pad_h + py, xx = x0 - p.pad_w + px;
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
            *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)s4 ^ key) << 4)) = x0v;
            *(__attribute__((address_space(3))) f16x8*)(unsigned long)(row + (((unsigned)(4 + s4) ^ key) << 4)) = x1v;
        }
    };
    // ---- weights (fragment order or packed rows), as conv_dma_body
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
    f32x4 wr[2][2];     // [register set][term]
    auto issue_w = [&](auto set_tag, int kc) {
        constexpr int SET = decltype(set_tag)::value;
        const int soff = kc * kc_stride;
        wr[SET][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff, soff, 0));
        wr[SET][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + term_off, soff, 0));
    };
    const int pcol = PI16(i16);
    f32x4 acc[TH];
#pragma unroll
    for (int u = 0; u < TH; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- epilogue operands that do not change over the tiles
    const int n4 = n0 + g16 * 4;
    const bool has_n = n4 < p.Cout;            // (Cout % 4 == 0 on this route: a channel group is whole or absent)
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f}, ct = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = min(n4 + r, p.Cout - 1);
        if (p.bias) bias[r] = p.bias[n];
        if (p.ch_scale) { cs[r] = p.ch_scale[n]; ct[r] = p.ch_shift[n]; }
    }
    const float xinv = ff::SPLIT_INV * xinv_in;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(pix_total * p.y_ld * 4), 0x00020000);
    const int nparts = a.tiles_y * a.tiles_x;
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(p.stats_part, 0, STATS ? (int)((long long)p.B * nparts * p.Cout * 16) : 0, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto epilogue = [&](int bimg, int y0, int x0) {
        const int x = x0 + pcol;
        f32x4 st_p = {0.f, 0.f, 0.f, 0.f}, st_s1 = st_p, st_s2 = st_p;
        float st_n = 0.f;
#pragma unroll
        for (int ug = 0; ug < TH; ug += UG) {
            f32x4 rr[UG];
            int po[UG];
#pragma unroll
            for (int k = 0; k < UG; ++k) {
                const int y = y0 + ug + k;
                po[k] = (has_n && y < H && x < W) ? (bimg * H + y) * W + x : -1;
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
            for (int k = 0; k < UG; ++k) {       // unconditional: a lane without a pixel stores out of range (dropped)
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
            const int part = (y0 / TH) * a.tiles_x + (x0 >> 4);
            const unsigned e = (i16 == 0 && has_n) ? (unsigned)(((bimg * nparts + part) * p.Cout + n4) * 16) : OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){st_p[r], st_s1[r], st_s2[r], st_n}), rst, (int)(e == OOB ? OOB : e + 16u * r), 0, 0);
        }
    };

    // ---- the walk: one step = one (tile, chunk c, tap t); two step bodies so that the weight register sets alternate without copies
    int v = blockIdx.x;
    int cb, cy0, cx0, nb = 0, ny0 = 0, nx0 = 0;
    decode(v, cb, cy0, cx0);
    bool has_next = v + G < total;
    if (has_next) decode(v + G, nb, ny0, nx0);
    int c = 0, t = 0, gc = 0;
    bool first = true, alive = true;
    issue_patch(cb, cy0, cx0, 0, 0);
    issue_w(std::integral_constant<int, 0>{}, 0);
    auto step = [&](auto set_tag) {
        constexpr int CUR = decltype(set_tag)::value, NXT = CUR ^ 1;
        const int dy = t / KW, dx = t - dy * KW;
        if (t == 0) {
            // the pieces of this chunk have landed once everything older than the wave's youngest ops has: those are the NWL
            // weight loads of this step - or, at a tile's top, the NST stores of the previous tile's epilogue
            if (c == 0 && !first) wait_vm<NST>(); else wait_vm<NWL>();
            __builtin_amdgcn_s_barrier();
            convert(cb, cy0, cx0, c, gc & 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads(): its fence would drain the stores)
        }
        {   // weights of the step after this one: the next tap, the next chunk's first, or - past the tile - the next tile's first
            int tn = t + 1, cn = c;
            if (tn == NT) { tn = 0; cn = c + 1 == nci ? 0 : c + 1; }
            issue_w(std::integral_constant<int, NXT>{}, tn * nci + cn);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int px = pcol + dx;
        const unsigned sw = (unsigned)((px >> 1) & 7);
        const unsigned rowb = lds0 + (unsigned)(gc & 1) * PBYTES + (unsigned)((dy * PW + px) * 128);
        const unsigned xa0 = rowb + ((g16 ^ sw) << 4), xa1 = rowb + (((4 + g16) ^ sw) << 4);
        auto rows = [&](auto lo_tag, auto hi_tag) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
            for (int u = LO; u < HI; ++u) {
                const f16x8 xa = lds_ld16(xa0 + u * PW * 128);
                const f16x8 xb = lds_ld16(xa1 + u * PW * 128);
                const f16x8 w0 = __builtin_bit_cast(f16x8, wr[CUR][0]);
                const f16x8 w1 = __builtin_bit_cast(f16x8, wr[CUR][1]);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[u], 0, 0, 0);
            }
            constexpr int N = HI - LO, LEAD = N < 2 ? N : 2;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * LEAD, 0);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (u + LEAD < N) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
        };
        constexpr int SPLIT = TH >= 4 ? 2 : TH;
        if (t == 0) {
            rows(std::integral_constant<int, 0>{}, std::integral_constant<int, SPLIT>{});
            if (c + 1 < nci) issue_patch(cb, cy0, cx0, c + 1, (gc + 1) & 1);
            else if (has_next) issue_patch(nb, ny0, nx0, 0, (gc + 1) & 1);
            rows(std::integral_constant<int, SPLIT>{}, std::integral_constant<int, TH>{});
        } else {
            rows(std::integral_constant<int, 0>{}, std::integral_constant<int, TH>{});
        }
        if (++t == NT) {
            t = 0;
            ++gc;
            if (++c == nci) {
                epilogue(cb, cy0, cx0);
                c = 0;
                first = false;
                if (!has_next) { alive = false; return; }
#pragma unroll
                for (int u = 0; u < TH; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                v += G;
                cb = nb; cy0 = ny0; cx0 = nx0;
                has_next = v + G < total;
                if (has_next) decode(v + G, nb, ny0, nx0);
            }
        }
    };
    while (true) {
        step(std::integral_constant<int, 0>{});
        if (!alive) break;
        step(std::integral_constant<int, 1>{});
        if (!alive) break;
    }
}

template <int KH, int KW, int TH, int OCC, int NW, int XMODE, bool STATS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_f32p_kernel(const DArgs a) {
    conv_f32p_body<KH, KW, TH, NW, XMODE, STATS>(a);
}

