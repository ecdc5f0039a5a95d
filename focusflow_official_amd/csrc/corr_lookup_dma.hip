// CorrBlock.__call__ (corr.py:29-50 + bilinear_sampler, utils.py:57-71) on the tiled pyramid of corr_layout.h, round-3
// kernel: the windows travel HBM -> LDS by LDS-DMA (buffer_load ... lds), nothing is staged through registers.
//
// What the round-2 kernel (corr_lookup_tiled.hip) spent its time on, by its counters: 245 vector + 106 scalar + 44 LDS
// instructions per query (two hardware divisions, eight exec-masked 16-byte LDS stores with their own branches, 64-bit
// plane addresses, shuffles through the LDS crossbar), the vector ALU 57 % and the LDS pipe 45 % busy for a kernel whose
// memory traffic is worth 2 us.  This one keeps the plan - one wave per query, taps -> window in LDS -> blend - and removes
// instructions and LDS bytes:
//   * ONE WAVE PER BLOCK, wave-private LDS, no barrier; every LDS address is a compile-time constant per lane;
//   * lanes are grouped 16 per level: the nine (level, offset) tap lanes of a level, its twelve window-row lanes and its five
//     window-chunk lanes are the same hardware row, so window geometry moves by DPP row broadcasts;
//   * both tap chains of a lane run as ONE chain of packed fp32 operations (v_pk_mul / v_pk_add / v_pk_fma round each half on
//     its own: the same numbers as two scalar chains); the division of the sampler's normalisation, x / (n - 1) with n - 1
//     a small integer, is five fused operations (reciprocal refinement with two exact remainders) instead of the eleven
//     of the hardware expansion - still the correctly rounded quotient (see taps_a), so the taps stay bit-identical
//     to the reference;
//   * the window of a level is COMPACT: exactly the 12 plane rows from the first tap row on, 16 (fp32) / 24 (fp16) columns
//     from the 16-byte chunk of the first tap column on, row-major at an 80-byte pitch - 960 bytes per level instead of
//     1792, so windows AND tap tables are double-buffered in 9.5 KB per wave: 16 waves per CU.  The four levels sit behind
//     ONE buffer resource (byte offsets < 4 GB from the lowest level), so the 240 16-byte pieces of a query are 4 DMA
//     instructions whose lane -> (level, window row, chunk) map is fixed (a table in the code object); per query a lane
//     adds the row's and the chunk's byte offsets (two small LDS tables written by the row / chunk lanes of the level, a
//     saturating add: either one out of range = out of range).  Rows / chunks outside the tile grid or beyond the last tap
//     get an out-of-range offset: the hardware returns zeros (grid_sample's zero padding) and fetches nothing;
//   * the DMA of query k + 1 is in flight during the whole of iteration k (counted s_waitcnt vmcnt, never 0 inside the
//     loop); the tap chains of query k + 2 run while the blend's LDS reads are in flight; coordinates arrive by scalar loads;
//   * outputs leave through a buffer resource over the output tensor: row base in a scalar register, no address
//     arithmetic on the vector ALU; the loop body is one basic block (no exec-masked branch).
// Measured variants kept behind switches (8 x 48 x 64 queries inside bench.py, same box): the window wait that also
// covered the previous query's three output stores (vmcnt NDMA instead of NDMA + 3) 19.9 -> 19.1 us; four query-waves
// per block (FF_LOOKUP_BLOCK_WAVES=4: a quarter of the workgroups to place) 19.9 vs 19.9 us; a third window / table
// buffer with two queries in flight at 11 waves per CU (FF_LOOKUP_DEPTH=2) 23.0 vs 19.4 us - the kernel is not short of
// bytes in flight, it needs its 16 waves.  A memory-only kernel of the same launch shape, byte count and segment size
// (probe.hip / tools/proto/hbm_gather.hip) takes 13.9-14.6 us.
#pragma clang fp contract(off)
#include <cstdlib>
#include "ff_common.h"
#include "corr_layout.h"

namespace {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// LDS accesses by absolute 32-bit address (one wave per block: the dynamic LDS segment starts at 0)
template <typename T>
__device__ __forceinline__ T lds_ld(unsigned addr) { return *(__attribute__((address_space(3))) const T*)(unsigned long)addr; }
template <typename T>
__device__ __forceinline__ void lds_st(unsigned addr, T v) { *(__attribute__((address_space(3))) T*)(unsigned long)addr = v; }

struct DArgs {
    const char* base;          // lowest of the four level pointers
    unsigned lvl_off[4];       // byte offset of level l from base
    unsigned plane_bytes[4];
    unsigned total_bytes;      // extent of the resource over all levels
    int ntx[4], nty[4];
    float nm1x[4], nm1y[4];    // (float)(w_l - 1), (float)(h_l - 1)
    float r2x[4], r2y[4];      // 2 * RN(1 / (w_l - 1)), 2 * RN(1 / (h_l - 1))
    unsigned qdiv, qrem;       // queries / grid, queries % grid
    const float* coords;
    float* out;
    int* taps;
    unsigned queries;
    unsigned out_bytes;
    int out_ld;
};

constexpr unsigned OOB = 0xffff0000u;      // + any in-range offset saturates / stays beyond every resource (total_bytes < OOB)

// ---- LDS map of a wave (bytes) ----
constexpr int ROWS = 12;                     // window rows per level: first tap row .. last tap row + 1 (<= 12 apart)
constexpr int CH = 5;                        // 16-byte chunks per window row (fp32: 4 data + 1 pad, fp16: 3 data + 2 pad)
constexpr int PITCH = CH * 16;               // 80 bytes: consecutive rows start 20 banks apart
constexpr int LVL_BYTES = ROWS * PITCH;      // 960
constexpr int NPOS = 4 * ROWS * CH;          // 240 pieces per query
constexpr int NDMA = 4;                      // 256 lanes: the last 16 write 256 bytes of spill behind the window
constexpr int WIN = 4096;                    // window stride: 3840 bytes + spill
constexpr int TAB = 2 * 36 * 8;              // x entries, then y entries (8 bytes each: LDS offset, weight of tap 1)
constexpr int OFF_ROW = 0;                   // row offsets  [4][12] + one word that is always out of range
constexpr int OFF_ROW_OOB = OFF_ROW + 48 * 4;
constexpr int OFF_COL = OFF_ROW + 52 * 4;    // chunk offsets [4][8]
constexpr int OFF_TAB = OFF_COL + 32 * 4;    // NBUF table buffers, then NBUF windows (NBUF - 1 = queries in flight ahead of the blend)
constexpr int off_win(int nbuf) { return OFF_TAB + nbuf * TAB; }
constexpr int wave_lds(int nbuf) { return off_win(nbuf) + nbuf * WIN; }
static_assert(wave_lds(2) == 9680 && wave_lds(2) * 16 <= 160 * 1024, "two buffers: 16 waves per CU");
static_assert(wave_lds(3) == 14352 && wave_lds(3) * 11 <= 160 * 1024, "three buffers: 11 waves per CU");
static_assert(off_win(2) % 16 == 0 && off_win(3) % 16 == 0 && OFF_TAB % 16 == 0, "DMA destinations are 16-byte aligned");


// ---- per-lane constants that do not depend on the launch: a table in the code object ----
// Output passes: lane L of pass j computes channel k = 2 (L + 64 (j >> 1)) + (j & 1) - a lane owns PAIRS of neighbouring
// channels, so a query leaves as three 8-byte stores per lane (three 512-byte instructions) instead of six 4-byte ones.
struct LaneTab {
    unsigned v[64][20];      // [0..3] row-offset word of DMA d, [4..7] chunk-offset word, [8..13] x entry of output pass j, [14..19] y entry
};
constexpr LaneTab make_lane_tab() {
    LaneTab t{};
    for (int lane = 0; lane < 64; ++lane) {
        for (int d = 0; d < NDMA; ++d) {
            const int pos = lane + 64 * d;
            if (pos < NPOS) {
                const int l = pos / (ROWS * CH), rem = pos % (ROWS * CH);
                t.v[lane][d] = OFF_ROW + (l * ROWS + rem / CH) * 4;
                t.v[lane][4 + d] = OFF_COL + (l * 8 + rem % CH) * 4;
            } else {
                t.v[lane][d] = OFF_ROW_OOB;
                t.v[lane][4 + d] = OFF_COL;
            }
        }
        for (int j = 0; j < 6; ++j) {
            const int k0 = 2 * (lane + 64 * (j >> 1)) + (j & 1);
            const int k = k0 < 323 ? k0 : 323;
            const int l = k / 81, rem = k % 81;
            t.v[lane][8 + j] = OFF_TAB + (l * 9 + rem / 9) * 8;
            t.v[lane][14 + j] = OFF_TAB + 288 + (l * 9 + rem % 9) * 8;
        }
    }
    return t;
}
__device__ const LaneTab g_lane_tab = make_lane_tab();

template <typename T>
__device__ __forceinline__ T sel4(int i, T a, T b, T c, T d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }

// (the empty statement keeps the broadcast a v_mov_b32_dpp of its own: folded into the consuming add / subtract by the
// compiler's DPP combiner - v_add_u32_dpp / v_subrev_u32_dpp with row_newbcast - the kernel returned wrong window rows)
__device__ __forceinline__ int row_bcast0(int v) {
    int r = __builtin_amdgcn_update_dpp(0, v, 0x150, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int row_bcast8(int v) {
    int r = __builtin_amdgcn_update_dpp(0, v, 0x158, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}

// ABL: timing-only ablation bits (FF_LOOKUP_ABLATE3, WRONG results): 1 no output stores, 4 no blend, 8 no tap chains inside
// the loop, 16 no DMA inside the loop.
// NW: query-waves per block (1 or 4).  Every wave keeps its private WAVE_LDS bytes and never meets the others (no barrier):
// four-wave blocks only quarter the number of workgroups the dispatcher has to place (4 096 -> 1 024 per launch).
// NBUF: window + table buffers of a wave; the DMA of query k + NBUF - 1 is issued before the blend of query k.
template <bool HALF, bool DBG, int ABL = 0, int NW = 1, int NBUF = 2>
__global__ __launch_bounds__(64 * NW) void lookup_dma_kernel(const DArgs a) {
    constexpr int WAVE_LDS = wave_lds(NBUF), OFF_WIN = off_win(NBUF), D = NBUF - 1;
    static_assert(ABL == 0 || NBUF == 2, "the ablations exist for the two-buffer kernel");
    constexpr int TSH = HALF ? 3 : 2, TH = 1 << TSH, ESZ = HALF ? 2 : 4;
    constexpr int CSH = HALF ? 3 : 2;            // log2(columns per 16-byte chunk)
    constexpr int DATA_CH = HALF ? 3 : 4;        // chunks of a window row that carry data
    constexpr int XMAX = DATA_CH * (1 << CSH) - 2;
    constexpr int ROWB = 8 * ESZ;                // bytes of one tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NW * WAVE_LDS bytes, addressed absolutely below (base 0)
    asm volatile("" ::"v"((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem));

    const int lane = NW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const unsigned wib = NW == 1 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wbase = wib * WAVE_LDS;                                 // this wave's LDS region
    const unsigned wave = blockIdx.x * NW + wib, nwaves = gridDim.x * NW;
    const unsigned count = a.qdiv + (wave < a.qrem ? 1u : 0u);             // the grid never exceeds the queries: count >= 1
    const unsigned qlast = wave + (count - 1) * nwaves;
    typedef const __attribute__((address_space(4))) f32x2* ccoords_t;     // scalar loads: off the vector memory counter
    const ccoords_t cptr = (ccoords_t)(a.coords);
    f32x2 cpro[NBUF];                                                     // the first NBUF queries of the wave
#pragma unroll
    for (int i = 0; i < NBUF; ++i) cpro[i] = cptr[min(wave + i * nwaves, qlast)];

    // ---- launch-independent lane roles (five 16-byte loads from the code object's table) ----
    const u32x4* tp = reinterpret_cast<const u32x4*>(&g_lane_tab.v[lane][0]);
    const u32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3], t4 = tp[4];
    const unsigned d_row[4] = {t0.x + wbase, t0.y + wbase, t0.z + wbase, t0.w + wbase};
    const unsigned d_col[4] = {t1.x + wbase, t1.y + wbase, t1.z + wbase, t1.w + wbase};
    const unsigned bx[6] = {t2.x + wbase, t2.y + wbase, t2.z + wbase, t2.w + wbase, t3.x + wbase, t3.y + wbase};
    const unsigned by[6] = {t3.z + wbase, t3.w + wbase, t4.x + wbase, t4.y + wbase, t4.z + wbase, t4.w + wbase};

    // ---- per-lane constants of the level row (lanes 16 l .. 16 l + 15 = level l): selected from the host's tables ----
    const int lv = lane >> 4, li = lane & 15;
    auto sel = [&](auto v0, auto v1, auto v2, auto v3) {
        auto r = v0;
        r = lv == 1 ? v1 : r;
        r = lv == 2 ? v2 : r;
        r = lv == 3 ? v3 : r;
        return r;
    };
    const float inv = __uint_as_float((unsigned)(127 - lv) << 23);          // 2^-l
    const float nm1x = sel(a.nm1x[0], a.nm1x[1], a.nm1x[2], a.nm1x[3]), nm1y = sel(a.nm1y[0], a.nm1y[1], a.nm1y[2], a.nm1y[3]);
    const float r2x = sel(a.r2x[0], a.r2x[1], a.r2x[2], a.r2x[3]), r2y = sel(a.r2y[0], a.r2y[1], a.r2y[2], a.r2y[3]);
    const float offf = (float)(min(li, 8) - 4);
    const f32x2 inv2 = {inv, inv}, off2 = {offf, offf}, r2v = {r2x, r2y}, dhv = {0.5f * nm1x, 0.5f * nm1y}, nm1v = {nm1x, nm1y};
    const int ntx = sel(a.ntx[0], a.ntx[1], a.ntx[2], a.ntx[3]);
    const int nty = sel(a.nty[0], a.nty[1], a.nty[2], a.nty[3]);
    const unsigned pb = sel(a.plane_bytes[0], a.plane_bytes[1], a.plane_bytes[2], a.plane_bytes[3]);
    const unsigned lo = sel(a.lvl_off[0], a.lvl_off[1], a.lvl_off[2], a.lvl_off[3]);
    const int is8 = li == 8 ? 1 : 0;
    const bool is_tap = li < 9;
    // row role (li < 12): window row li of the level; chunk role (li < DATA_CH): chunk li.  Lanes without the role compare
    // against an empty grid (always out of range) and write a word that is out of range anyway.
    const unsigned nty_r = li < ROWS ? (unsigned)nty : 0u, ntx_c = li < DATA_CH ? (unsigned)ntx : 0u;
    const unsigned rstride = (unsigned)ntx * 128u;
    const unsigned row_w = wbase + (li < ROWS ? OFF_ROW + (lv * ROWS + li) * 4 : OFF_ROW_OOB);
    const unsigned col_w = wbase + OFF_COL + (lv * 8 + min(li, 7)) * 4;           // li >= 8 rewrites chunk 7 (never read) with OOB
    // tap entries: lanes that are no tap lanes write theirs to a scratch place - the padding chunk of a window row of the
    // same buffer (idle while the taps run, never read) - so that the loop body has no exec-masked branch
    const unsigned tab_x = wbase + OFF_TAB + (lv * 9 + min(li, 8)) * 8;
    const unsigned scratch = wbase + OFF_WIN + (lv * 7 + max(li - 9, 0)) * PITCH + 64;
    unsigned tab_wx[NBUF], tab_wy[NBUF];
#pragma unroll
    for (int i = 0; i < NBUF; ++i) {
        tab_wx[i] = (is_tap ? tab_x : scratch) + i * (is_tap ? TAB : WIN);
        tab_wy[i] = (is_tap ? tab_x + 288 : scratch + 8) + i * (is_tap ? TAB : WIN);
    }
    const unsigned win_lv = wbase + OFF_WIN + lv * LVL_BYTES;
    // stores: pairs (2 L + 128 p, + 1) for p = 0, 1, 2; the third pair exists for lanes 0..33 only (channels 256..323)
    const unsigned st2 = lane < 34 ? (unsigned)(lane * 8 + 1024) : OOB;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.base), 0, (int)a.total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.base), 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);

    struct Taps { int x0, y0; float wx, wy; };
    // The two tap chains of query (cx, cy), packed.  g = 2 x / (n - 1) is computed as RN(t / d) for t = 2 x from x,
    // r2 = 2 RN(1 / d) and dh = d / 2 (exact scalings): q0 = RN(t r) is within 2 ulp of t / d; the remainder t - q0 d is
    // exactly representable, so fma gives it exactly (halved: exact too); q1 = RN(q0 + rem r) is a faithful quotient; one
    // more exact remainder and the same correction give the correctly rounded one (Markstein's theorem: r within half an
    // ulp of 1 / d, d's significand not all ones - d is an integer below 2^13).  The CPU tests carry the same five
    // operations next to a true division and compare them for every divisor up to 4096 (tests/test_*_golden.py).
    auto taps_a = [&](float cx, float cy) {
        const f32x2 c2 = {cx, cy};
        const f32x2 x = c2 * inv2 + off2;                        // corr.py:41-43 (contraction is off: two roundings)
        const f32x2 q0 = x * r2v;
        const f32x2 e0 = __builtin_elementwise_fma(-q0, dhv, x);
        const f32x2 q1 = __builtin_elementwise_fma(e0, r2v, q0);
        const f32x2 e1 = __builtin_elementwise_fma(-q1, dhv, x);
        const f32x2 g = __builtin_elementwise_fma(e1, r2v, q1) - 1.f;      // utils.py:61-62
        const f32x2 u = ((g + 1.f) * 0.5f) * nm1v;                // ATen's un-normalise (align_corners)
        const float fx = floorf(u.x), fy = floorf(u.y);
        Taps t;
        t.x0 = (int)fx;
        t.y0 = (int)fy;
        t.wx = __fsub_rn(u.x, fx);
        t.wy = __fsub_rn(u.y, fy);
        return t;
    };
    // window geometry, tap entries, row and chunk offsets of query q -> tab[buf], row / chunk tables
    auto taps_b = [&](unsigned q, const Taps& t, int buf) {
        // first tap (tap 0 of offset -4: lane 0 of the row) and last tap (tap 1 of offset +4: lane 8) of the level
        const int xlo = row_bcast0(t.x0), ylo = row_bcast0(t.y0);
        const int xhi = row_bcast8(t.x0 + is8), yhi = row_bcast8(t.y0 + is8);
        const int xa = xlo & ~((1 << CSH) - 1);                   // the window starts at the chunk of the first tap column
        // window coordinates of tap 0, clamped so that a wild coordinate (and its +1 neighbour) stays inside the window
        const int wxc = min(max(t.x0 - xa, 0), XMAX), wyc = min(max(t.y0 - ylo, 0), ROWS - 2);
        i32x2 ex, ey;
        ex.x = wxc * ESZ;
        ex.y = __float_as_int(t.wx);
        ey.x = __umul24(wyc, PITCH) + (win_lv + buf * WIN);
        ey.y = __float_as_int(t.wy);
        lds_st<i32x2>(tab_wx[buf], ex);
        lds_st<i32x2>(tab_wy[buf], ey);
        if (DBG) {
            if (is_tap && a.taps) {
                int* tp2 = a.taps + ((size_t)q * 4 + lv) * 18;
                tp2[li] = t.x0;
                tp2[9 + li] = t.y0;
            }
        }
        // window row li = plane row ylo + li: byte offset of its first element from the resource base, or out of range
        // (above / below the tile grid, or past the last tap row)
        const int y = ylo + li, ty = y >> TSH;
        const unsigned roff = __umul24(ty, rstride) + (unsigned)((y & (TH - 1)) * ROWB) + (__umul24(q, pb) + lo);
        lds_st<unsigned>(row_w, ((unsigned)ty < nty_r && y <= yhi) ? roff : OOB);
        // chunk li = plane columns xa + li * CW ...: byte offset inside a tile row, or out of range (left / right of the tile
        // grid, past the last tap column, or a padding chunk)
        const int xc = (xa >> CSH) + li;
        const int tx = HALF ? xc : xc >> 1;
        const unsigned coff = (unsigned)(tx << 7) + (HALF ? 0u : (unsigned)((xc & 1) << 4));
        lds_st<unsigned>(col_w, ((unsigned)tx < ntx_c && xc <= (xhi >> CSH)) ? coff : OOB);
    };
    // NDMA instructions in one statement: M0 (the LDS destination) is saved once, advanced by 1 KB per instruction
    auto dma = [&](int buf, __amdgpu_buffer_rsrc_t rs) {
        unsigned s[NDMA];
#pragma unroll
        for (int d = 0; d < NDMA; ++d) s[d] = __builtin_elementwise_add_sat(lds_ld<unsigned>(d_row[d]), lds_ld<unsigned>(d_col[d]));
        const unsigned dst = wbase + OFF_WIN + buf * WIN;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %5, 0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %5, 0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %5, 0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %5, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "s"(rs), "s"(dst)
                     : "memory", "scc");
    };
    // One iteration's LDS work in three phases - all table reads, all window reads, the arithmetic - so that a query costs
    // two LDS round trips instead of six; the tap chains of query k + 2 fill the first latency, its geometry / table
    // writes the second (LDS operations of a wave execute in order: the writes into tab[buf] follow the reads of it).
    auto blend_and_taps = [&](unsigned q, unsigned q2, float c2x, float c2y, int buf) {
        const unsigned so = q * (unsigned)(a.out_ld * 4);
        i32x2 ex[6], ey[6];
        if (!(ABL & 4)) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ex[j] = lds_ld<i32x2>(bx[j] + buf * TAB);
                ey[j] = lds_ld<i32x2>(by[j] + buf * TAB);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        Taps t;
        if (!(ABL & 8)) t = taps_a(c2x, c2y);
        __builtin_amdgcn_sched_barrier(0);
        float v00[6], v01[6], v10[6], v11[6];
        if (!(ABL & 4)) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const unsigned p = ey[j].x + ex[j].x;
                if (HALF) {
                    v00[j] = (float)lds_ld<_Float16>(p);
                    v01[j] = (float)lds_ld<_Float16>(p + 2);
                    v10[j] = (float)lds_ld<_Float16>(p + PITCH);
                    v11[j] = (float)lds_ld<_Float16>(p + PITCH + 2);
                } else {
                    v00[j] = lds_ld<float>(p);
                    v01[j] = lds_ld<float>(p + 4);
                    v10[j] = lds_ld<float>(p + PITCH);
                    v11[j] = lds_ld<float>(p + PITCH + 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) taps_b(q2, t, buf);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4)) {
            float o[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float wx = __int_as_float(ex[j].y), wy = __int_as_float(ey[j].y);
                // nw*s*e + ne*s*w + sw*n*e + se*n*w  (ATen's weight naming): s = 1 - wy, e = 1 - wx; the four weights and the
                // four products as packed multiplications (each half rounded on its own), the sum left to right
                const f32x2 ew = {__fsub_rn(1.f, wx), wx};
                const float s0 = __fsub_rn(1.f, wy);
                const f32x2 wa = ew * s0, wb = ew * wy;
                const f32x2 ta = f32x2{v00[j], v01[j]} * wa, tb = f32x2{v10[j], v11[j]} * wb;
                // (the empty statements keep the three additions scalar: the vectoriser paired them across two outputs and
                // paid six register moves per pair of outputs for three packed additions)
                float s1 = __fadd_rn(ta.x, ta.y);
                asm volatile("" : "+v"(s1));
                float s2 = __fadd_rn(s1, tb.x);
                asm volatile("" : "+v"(s2));
                o[j] = __fadd_rn(s2, tb.y);
            }
#pragma unroll
            for (int p2 = 0; p2 < 3; ++p2) {
                if (ABL & 1) asm volatile("" ::"v"(o[2 * p2]), "v"(o[2 * p2 + 1]));
                else {
                    u32x2 ov = {__float_as_uint(o[2 * p2]), __float_as_uint(o[2 * p2 + 1])};
                    __builtin_amdgcn_raw_buffer_store_b64(ov, rs_out, p2 < 2 ? (unsigned)(lane * 8 + p2 * 512) : st2, so, 0);
                }
            }
        }
    };

    // prologue: tables of the first NBUF queries, DMAs of the first NBUF - 1
#pragma unroll
    for (int i = 0; i < NBUF; ++i) {
        taps_b(min(wave + i * nwaves, qlast), taps_a(cpro[i].x, cpro[i].y), i);
        if (i < D) dma(i, (unsigned)i < count ? rs_in : rs_null);
    }
    // Iteration k: DMA of query k + D (its offsets were published by the previous iteration) -> wait for the window of
    // query k -> blend k, interleaved with the taps of query k + NBUF.  Past the end the same instructions run on the last
    // query with an empty resource (no traffic, no branches).
    // The wait: vector-memory operations complete in issue order (gfx9: one counter for loads and stores), and behind the
    // DMA of query k the wave has issued, per later iteration, three output stores and NDMA DMA instructions - none of
    // which it has to wait for: D * NDMA + min(k, D) * 3 operations may stay in flight.
    unsigned q = wave;
    for (unsigned k = 0; k < count; k += NBUF) {
#pragma unroll
        for (int P = 0; P < NBUF; ++P) {
            const unsigned kk = k + P;
            if (kk >= count) break;
            const unsigned qn = min(q + NBUF * nwaves, qlast);
            const f32x2 cn = cptr[qn];
            if (!(ABL & 16)) dma((P + D) % NBUF, kk + D < count ? rs_in : rs_null);
            if ((ABL & 1) || (P == 0 && k == 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NDMA) : "memory");
            else if (D == 2 && P == 1 && k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NDMA + 3) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NDMA + D * 3) : "memory");
            blend_and_taps(q, qn, cn.x, cn.y, P);
            q += nwaves;
        }
    }
}

}  // namespace

namespace ff {

// 0 = launched, 1 = not eligible (the caller falls back to the round-2 kernel), < 0 = error
int lookup_dma_fwd(const void* const* levels, int half, const float* coords, long long queries, int h0, int w0, float* out,
                   int out_ld, int* taps_dbg, hipStream_t s) {
    const CorrLayout Ly = corr_layout(h0, w0, half != 0);
    DArgs a;
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    for (int l = 0; l < 4; ++l) {
        const uintptr_t p = reinterpret_cast<uintptr_t>(levels[l]);
        const unsigned long long bytes = (unsigned long long)Ly.plane[l] * (half ? 2 : 4) * (unsigned long long)queries;
        lo = p < lo ? p : lo;
        hi = p + bytes > hi ? p + bytes : hi;
    }
    if (hi - lo >= 0xfff00000ull || queries >= (1ll << 24)) return 1;
    const unsigned long long out_bytes = (unsigned long long)queries * out_ld * 4;
    if (out_bytes >= 0xfff00000ull) return 1;
    a.base = reinterpret_cast<const char*>(lo);
    a.total_bytes = (unsigned)(hi - lo);
    for (int l = 0; l < 4; ++l) {
        a.lvl_off[l] = (unsigned)(reinterpret_cast<uintptr_t>(levels[l]) - lo);
        a.plane_bytes[l] = (unsigned)Ly.plane[l] * (half ? 2 : 4);
        if (a.plane_bytes[l] >= (1u << 24) || Ly.ntx[l] * 128 >= (1 << 24)) return 1;
        a.ntx[l] = Ly.ntx[l];
        a.nty[l] = Ly.nty[l];
    }
    for (int l = 0; l < 4; ++l) {
        a.nm1x[l] = (float)((w0 >> l) - 1);
        a.nm1y[l] = (float)((h0 >> l) - 1);
        const volatile float rx = 1.0f / a.nm1x[l], ry = 1.0f / a.nm1y[l];      // correctly rounded reciprocals (IEEE division)
        a.r2x[l] = rx + rx;
        a.r2y[l] = ry + ry;
    }
    a.coords = coords;
    a.out = out;
    a.taps = taps_dbg;
    a.queries = (unsigned)queries;
    a.out_bytes = (unsigned)out_bytes;
    a.out_ld = out_ld;
    // one wave per block; 16 blocks per CU with two buffers (9.5 KB of LDS each), 11 with three (14 KB)
    static const int env_wpc = ff::tune_env("FF_LOOKUP_WAVES_PER_CU") ? atoi(ff::tune_env("FF_LOOKUP_WAVES_PER_CU")) : 0;
    static const int env_depth = ff::tune_env("FF_LOOKUP_DEPTH") ? atoi(ff::tune_env("FF_LOOKUP_DEPTH")) : 1;      // A/B switch
    static const int env_bw = ff::tune_env("FF_LOOKUP_BLOCK_WAVES") ? atoi(ff::tune_env("FF_LOOKUP_BLOCK_WAVES")) : 1;
#ifdef FF_LAB      // timing-only ablations (WRONG results): lab build only (tools/build_lab.sh), not in libfocusflow_hip.so
    const char* abl_s = getenv("FF_LOOKUP_ABLATE3");
    const int abl = abl_s ? atoi(abl_s) : 0;
#else
    constexpr int abl = 0;
#endif
    const bool deep = env_depth == 2 && !taps_dbg && !abl;
    const int max_wpc = deep ? 11 : 16;
    const int wpc = env_wpc > 0 && env_wpc <= max_wpc ? env_wpc : max_wpc;
    long long waves = 256ll * wpc;
    if (waves > queries) waves = queries;
    // FF_LOOKUP_BLOCK_WAVES=4 (opt-in A/B switch): four query-waves per block.  Measured on 8 x 48 x 64 queries inside
    // bench.py: 19.9 us per launch either way - placing 4 096 one-wave workgroups is not what the ramp costs.
    const bool quad = env_bw == 4 && waves % 4 == 0 && waves >= 1024 && !taps_dbg && !deep;
    const unsigned blocks = (unsigned)(quad ? waves / 4 : waves);
    a.qdiv = (unsigned)(queries / waves);
    a.qrem = (unsigned)(queries % waves);
    hipEvent_t ev0, ev1;          // null unless ff_launch_timing_begin(FF_TIME_LOOKUP) is in effect
    launch_timing_events(FF_TIME_LOOKUP, &ev0, &ev1);
#define FF_LAUNCH3(H_, D_, A_) hipExtLaunchKernelGGL((lookup_dma_kernel<H_, D_, A_, 1, 2>), dim3(blocks), dim3(64), wave_lds(2), s, ev0, ev1, 0, a)
#define FF_LAUNCH4(H_) hipExtLaunchKernelGGL((lookup_dma_kernel<H_, false, 0, 4, 2>), dim3(blocks), dim3(256), 4 * wave_lds(2), s, ev0, ev1, 0, a)
#define FF_LAUNCH_DEEP(H_) hipExtLaunchKernelGGL((lookup_dma_kernel<H_, false, 0, 1, 3>), dim3(blocks), dim3(64), wave_lds(3), s, ev0, ev1, 0, a)
#define FF_LAUNCH(H_, D_) FF_LAUNCH3(H_, D_, 0)
#ifdef FF_LAB
#define FF_ABL(V_) if (abl == V_ && !half && !quad) { FF_LAUNCH3(false, false, V_); return check_launch("ff_corr_lookup_tiled_fwd (dma, ablated)"); }
    FF_ABL(1) FF_ABL(4) FF_ABL(8) FF_ABL(16) FF_ABL(20) FF_ABL(28) FF_ABL(12)
#undef FF_ABL
#endif
    if (deep) {
        if (half) FF_LAUNCH_DEEP(true); else FF_LAUNCH_DEEP(false);
    } else if (quad) {
        if (half) FF_LAUNCH4(true); else FF_LAUNCH4(false);
    } else if (half) {
        if (taps_dbg) FF_LAUNCH(true, true); else FF_LAUNCH(true, false);
    } else {
        if (taps_dbg) FF_LAUNCH(false, true); else FF_LAUNCH(false, false);
    }
#undef FF_LAUNCH
#undef FF_LAUNCH3
#undef FF_LAUNCH4
#undef FF_LAUNCH_DEEP
    return check_launch("ff_corr_lookup_tiled_fwd (dma)");
}

}  // namespace ff
