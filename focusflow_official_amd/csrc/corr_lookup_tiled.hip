// CorrBlock.__call__ (corr.py:29-50 + bilinear_sampler, utils.py:57-71) on the TILED pyramid of corr_layout.h: the entry point
// of the forward (the kernel is corr_lookup_dma.hip's; the round-2 tile-fetching kernel that lived here was its fallback for
// pyramids beyond one 4 GB buffer resource - CorrBlock now builds such batches in chunks, and it is gone), the BACKWARD
// (scatter into tiled fp32 gradient planes) and the pooling backward chain on tiled planes.  The coordinate arithmetic is
// the separately rounded fp32 replay of corr.hip (tap indices bit-identical to the reference).
#pragma clang fp contract(off)
#include <cstdlib>
#include "ff_common.h"
#include "corr_layout.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One separately-rounded replay of the sampler's coordinate chain (corr.py:41-43, utils.py:61-62, ATen's un-normalise).
__device__ __forceinline__ void tap_1d(float c, float inv_scale, int off, int n, int& i0, float& w1) {
    const float cl = __fmul_rn(c, inv_scale);
    const float x = __fadd_rn(cl, (float)off);
    const float nm1 = (float)(n - 1);
    const float g = __fsub_rn(__fdiv_rn(__fmul_rn(2.f, x), nm1), 1.f);
    const float u = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.f), 0.5f), nm1);
    const float f = floorf(u);
    i0 = (int)f;
    w1 = __fsub_rn(u, f);
}

// ---------------------------------------------------------------------------
// Lookup backward on tiled fp32 gradient planes: dlevel[l][q][y][x] += dout[q][k] * bilinear weight.
// One wave per query: the 324 output gradients are scattered into the LDS tile image of the four windows (ds_add_f32),
// then every touched tile is updated with whole-line read-modify-writes (8 lanes x 16 bytes per tile) - race free,
// because inside one launch every (query, plane element) belongs to exactly one lane; launches of successive iterations
// are ordered by the stream.  Tiles outside the plane are skipped, elements beyond the plane inside an edge tile are
// masked, so the pad of a gradient plane stays zero.
// ---------------------------------------------------------------------------
struct TLookupBwdArgs {
    float* dlvl[4];
    long long plane_elems[4];
    int h[4], w[4], ntx[4], nty[4];
    const float* coords;
    const float* dout;
    long long queries;
    int dout_ld;
};

constexpr int BTILES = 12;     // backward: 4 tile rows x 3 tile columns per level

__global__ __launch_bounds__(64) void lookup_tiled_bwd_kernel(const TLookupBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float win[4 * BTILES * 32];
    __shared__ int tab_o[4][2][2][9];
    __shared__ float tab_w[4][2][9];
    __shared__ int geo[4][4];
    const int lane = threadIdx.x;
    const int t_lv = min(lane / 9, 3), t_o = lane - (lane / 9) * 9;
    const float t_inv = 1.f / (float)(1 << t_lv);
    const int t_h = a.h[0] >> t_lv, t_w = a.w[0] >> t_lv;
    const int k8 = lane >> 3, piece = lane & 7;
    for (long long q = blockIdx.x; q < a.queries; q += gridDim.x) {
        int x0, y0;
        float wx, wy;
        tap_1d(a.coords[q * 2], t_inv, t_o - 4, t_w, x0, wx);
        tap_1d(a.coords[q * 2 + 1], t_inv, t_o - 4, t_h, y0, wy);
        const int xlo = __shfl(x0, t_lv * 9), ylo = __shfl(y0, t_lv * 9);
        const int xhi = __shfl(x0, t_lv * 9 + 8) + 1, yhi = __shfl(y0, t_lv * 9 + 8) + 1;
        const int tx0 = xlo >> 3, ty0 = ylo >> 2;
        if (lane < 36) {
            const int tc0 = min(max((x0 >> 3) - tx0, 0), 2), tc1 = min(max(((x0 + 1) >> 3) - tx0, 0), 2);
            const int tr0 = min(max((y0 >> 2) - ty0, 0), 3), tr1 = min(max(((y0 + 1) >> 2) - ty0, 0), 3);
            tab_o[t_lv][0][0][t_o] = tc0 * 32 + (x0 & 7);                 // float index in the level image
            tab_o[t_lv][0][1][t_o] = tc1 * 32 + ((x0 + 1) & 7);
            tab_o[t_lv][1][0][t_o] = tr0 * 96 + (y0 & 3) * 8;
            tab_o[t_lv][1][1][t_o] = tr1 * 96 + ((y0 + 1) & 3) * 8;
            tab_w[t_lv][0][t_o] = wx;
            tab_w[t_lv][1][t_o] = wy;
            if (t_o == 0) {
                geo[t_lv][0] = tx0;
                geo[t_lv][1] = ty0;
                geo[t_lv][2] = min(max((xhi >> 3) - tx0 + 1, 1), 3);
                geo[t_lv][3] = min(max((yhi >> 2) - ty0 + 1, 1), 4);
            }
        }
        for (int e = lane; e < 4 * BTILES * 32; e += 64) win[e] = 0.f;
        __syncthreads();
        const float* drow = a.dout + q * a.dout_ld;
        // the six gradient values of this lane in ONE round trip (a load under `if (k < 324)` is waited for before the next
        // one is issued: six dependent L2 / HBM round trips per query were most of this kernel's time)
        float gk[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) gk[j] = drow[min(lane + 64 * j, 323)];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = lane + 64 * j;
            if (k < 324) {
                const int lv = k / 81, rem = k - lv * 81;
                const int ia = rem / 9, ib = rem - ia * 9;
                const int xo0 = tab_o[lv][0][0][ia], xo1 = tab_o[lv][0][1][ia];
                const int yo0 = tab_o[lv][1][0][ib], yo1 = tab_o[lv][1][1][ib];
                const float fx = tab_w[lv][0][ia], fy = tab_w[lv][1][ib];
                const float g = gk[j];
                const float ex = 1.f - fx, sy = 1.f - fy;
                float* p = &win[lv * BTILES * 32];
                atomicAdd(p + yo0 + xo0, g * (sy * ex));
                atomicAdd(p + yo0 + xo1, g * (sy * fx));
                atomicAdd(p + yo1 + xo0, g * (fy * ex));
                atomicAdd(p + yo1 + xo1, g * (fy * fx));
            }
        }
        __syncthreads();
        // whole-tile read-modify-write, two levels at a time (registers: 3 -> 4-5 waves per SIMD; the kernel is a chain of
        // dependent memory round trips per query, what it needs is more queries in flight): the loads of both levels
        // first (lanes with nothing to add re-read a valid piece), then the stores; the addends come from the LDS image again
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 cur[2][2];
            int idx[2][2], wo[2][2], nvv[2][2];
#pragma unroll
            for (int l2 = 0; l2 < 2; ++l2) {
                const int lv = 2 * half + l2;
                const int gx0 = geo[lv][0], gy0 = geo[lv][1], ntc = geo[lv][2], ntr = geo[lv][3];
                const int ntx = a.ntx[lv], nty = a.nty[lv];
                const float* pl = a.dlvl[lv] + q * a.plane_elems[lv];
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int nt = ntc * ntr;
                    const int k = min(k8 + 8 * rd, nt - 1);
                    const int tr = (k * (ntc == 3 ? 43 : (ntc == 2 ? 64 : 128))) >> 7, tc = k - tr * ntc;   // k / ntc, k < 12
                    const int gtx = gx0 + tc, gty = gy0 + tr;
                    const bool in = (unsigned)gtx < (unsigned)ntx && (unsigned)gty < (unsigned)nty && k8 + 8 * rd < nt;
                    const int ctx = min(max(gtx, 0), ntx - 1), cty = min(max(gty, 0), nty - 1);
                    const int o = (cty * ntx + ctx) * 32 + piece * 4;
                    cur[l2][rd] = *reinterpret_cast<const f32x4*>(pl + o);
                    const int ey = gty * 4 + (piece >> 1), ex = gtx * 8 + (piece & 1) * 4;
                    nvv[l2][rd] = ((unsigned)ey < (unsigned)(a.h[0] >> lv)) ? min(max((a.w[0] >> lv) - ex, 0), 4) : 0;
                    wo[l2][rd] = lv * BTILES * 32 + (tr * 3 + tc) * 32 + piece * 4;
                    idx[l2][rd] = in ? o : -1;
                }
            }
#pragma unroll
            for (int l2 = 0; l2 < 2; ++l2) {
                float* pl = a.dlvl[2 * half + l2] + q * a.plane_elems[2 * half + l2];
#pragma unroll
                for (int rd = 0; rd < 2; ++rd)
                    if (idx[l2][rd] >= 0) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(&win[wo[l2][rd]]);
#pragma unroll
                        for (int d = 0; d < 4; ++d) v[d] = d < nvv[l2][rd] ? v[d] : 0.f;
                        *reinterpret_cast<f32x4*>(pl + idx[l2][rd]) = cur[l2][rd] + v;
                    }
            }
        }
        __syncthreads();
    }
}

// pooling backward chain on tiled fp32 planes: dl2 += up(dl3)/4 ; dl1 += up(dl2)/4 ; dl0 += up(dl1)/4  (in place).
// One block per plane; the two middle levels pass through LDS (row-major there).
__global__ __launch_bounds__(256) void pyramid_tiled_bwd_kernel(float* __restrict__ d0, float* __restrict__ d1,
                                                                float* __restrict__ d2, const float* __restrict__ d3,
                                                                const ff::CorrLayout L) {
    extern __shared__ float sm[];
    const int h0 = L.h[0], w0 = L.w[0], h1 = L.h[1], w1 = L.w[1], h2 = L.h[2], w2 = L.w[2], h3 = L.h[3], w3 = L.w[3];
    float* s2 = sm;               // h2*w2
    float* s1 = sm + h2 * w2;     // h1*w1
    const long long plane = blockIdx.x;
    float* p0 = d0 + plane * L.plane[0];
    float* p1 = d1 + plane * L.plane[1];
    float* p2 = d2 + plane * L.plane[2];
    const float* p3 = d3 + plane * L.plane[3];
    for (int i = threadIdx.x; i < h2 * w2; i += 256) {
        const int y = i / w2, x = i - y * w2;
        const int o = ff::tiled_offset<4>(y, x, L.ntx[2]);
        float v = p2[o];
        if ((y >> 1) < h3 && (x >> 1) < w3) v += 0.25f * p3[ff::tiled_offset<4>(y >> 1, x >> 1, L.ntx[3])];
        s2[i] = v;
        p2[o] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h1 * w1; i += 256) {
        const int y = i / w1, x = i - y * w1;
        const int o = ff::tiled_offset<4>(y, x, L.ntx[1]);
        float v = p1[o];
        if ((y >> 1) < h2 && (x >> 1) < w2) v += 0.25f * s2[(y >> 1) * w2 + (x >> 1)];
        s1[i] = v;
        p1[o] = v;
    }
    __syncthreads();
    // level 0 in tile order: a thread owns 4 consecutive x of one tile row (16 bytes)
    const int n4 = L.plane[0] >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int tile = i >> 3, pc = i & 7;
        const int y = (tile / L.ntx[0]) * 4 + (pc >> 1), x = (tile % L.ntx[0]) * 8 + (pc & 1) * 4;
        if (y >= h0 || x >= w0) continue;
        f32x4 v = *reinterpret_cast<f32x4*>(p0 + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x + e < w0 && (y >> 1) < h1 && ((x + e) >> 1) < w1) v[e] += 0.25f * s1[(y >> 1) * w1 + ((x + e) >> 1)];
        *reinterpret_cast<f32x4*>(p0 + i * 4) = v;
    }
}

// ---------------------------------------------------------------------------
// Lookup backward of ALL iterations at once + the pooling backward chain: d(volume) of a query from the (coords_t, dout_t)
// of every lookup that read it.  The gradient of the pyramid is a sum over the iterations and nothing needs it before the
// corr build's own backward, so instead of twelve launches that each read-modify-write ~48 scattered tiles per query in a
// 400 MB gradient pyramid (170 us each, 1.8 TB/s of scattered RMW traffic) plus a zero fill and a pooling pass, ONE block
// per query keeps the query's four gradient planes in LDS (15 KB at 46 x 62), scatters the 324 x T gradients into them
// (ds_add_f32), folds level 3 -> 2 -> 1 -> 0 (avg_pool2d backward: 0.25 to each of the four children) and writes level 0
// once, in tile order, pad elements zero: dout is read once, d(volume) written once.
// ---------------------------------------------------------------------------
constexpr int LBA_MAXT = 32;
struct LBAArgs {
    const float* coords[LBA_MAXT];
    const float* dout[LBA_MAXT];
    int T, dout_ld;
    float* d0;
    long long queries;
    ff::CorrLayout L;
};

// 384 threads; thread k < 324 owns output channel k (its gradient) of every iteration.  Per iteration:
//   A  72 threads replay the coordinate chain of the 36 (level, offset) taps per axis into an LDS table and check, with
//      a neighbour compare and a ballot, that the nine taps of a level are CONSECUTIVE integers (they are, unless the
//      coordinate is so large that fp32 rounding skips or repeats a tap); the gradients go to LDS too;
//   C  regular levels: the bilinear scatter is separable and every window element has exactly one writer -
//        tmp[b][x] = g[x][b] (1 - fx[x]) + g[x-1][b] fx[x-1]     (9 x 10 per level),
//      irregular levels: the plain scatter with LDS atomics;
//   D  regular levels: plane[ylo + y][xlo + x] += tmp[y][x] (1 - fy[y]) + tmp[y-1][x] fy[y-1]   (10 x 10, no atomics).
// The all-atomics form of this kernel took 1.28 ms per launch (355 M ds_add_f32 for 12 x 22 816 queries), whatever its
// vector-instruction count or its memory round trips were.
constexpr int LBA_THREADS = 384;
__global__ __launch_bounds__(LBA_THREADS) void lookup_bwd_all_kernel(const LBAArgs a) {
    extern __shared__ float sm[];
    __shared__ int tab_i[2][36];             // [axis][level * 9 + offset]: first tap
    __shared__ float tab_w[2][36];           //                           : weight of the second tap
    __shared__ float gsh[324];
    __shared__ float tmp[4][9][10];
    __shared__ int regx[4], regy[4];         // per level: the nine taps of the axis are consecutive integers
    __shared__ float cs[2 * LBA_MAXT];       // the query's coordinates of every iteration
    const ff::CorrLayout& L = a.L;
    const int h0 = L.h[0], w0 = L.w[0], h1 = L.h[1], w1 = L.w[1], h2 = L.h[2], w2 = L.w[2], h3 = L.h[3], w3 = L.w[3];
    const int n0 = h0 * w0, n1 = h1 * w1, n2 = h2 * w2, n3 = h3 * w3, ntot = n0 + n1 + n2 + n3;
    float* s0 = sm;
    float* s1 = s0 + n0;
    float* s2 = s1 + n1;
    float* s3 = s2 + n2;
    const int tid = threadIdx.x;
    // table role: x axis = lanes 0..35 of wave 0, y axis = lanes 0..35 of wave 1 (a wave-local neighbour compare then tells
    // whether the taps are consecutive); entry e = level * 9 + offset
    const int t_axis = tid >> 6, t_e = min(tid & 63, 35), t_lv = t_e / 9, t_o = t_e - t_lv * 9;
    const bool t_role = tid < 128 && (tid & 63) < 36;
    const float t_inv = 1.f / (float)(1 << t_lv);
    const int t_n = ((t_axis & 1) == 0 ? w0 : h0) >> t_lv;
    // output role (tid < 324): k = level * 81 + a * 9 + b, a = x offset, b = y offset
    const int k = min(tid, 323), lv = k / 81, rem = k - lv * 81, ia = rem / 9, ib = rem - ia * 9;
    const int hl = h0 >> lv, wl = w0 >> lv;
    float* pl = lv == 0 ? s0 : (lv == 1 ? s1 : (lv == 2 ? s2 : s3));
    // stage-1 role (tid < 360): level c_lv, y offset c_b, window column c_x ; stage-2 roles: items tid and tid + 384 of 400
    const int c_lv = min(tid / 90, 3), c_rem = tid - c_lv * 90, c_b = c_rem / 10, c_x = c_rem - c_b * 10;
    for (long long q = blockIdx.x; q < a.queries; q += gridDim.x) {
        // One round trip for all coordinates, one per FOUR iterations for the gradients (issued a group ahead).
        for (int i = tid; i < ntot; i += LBA_THREADS) sm[i] = 0.f;
        if (tid < 2 * a.T) cs[tid] = a.coords[tid >> 1][q * 2 + (tid & 1)];
        float gc[4], gn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gc[j] = a.dout[min(j, a.T - 1)][q * a.dout_ld + k];
        __syncthreads();
        for (int t0 = 0; t0 < a.T; t0 += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gn[j] = a.dout[min(t0 + 4 + j, a.T - 1)][q * a.dout_ld + k];      // (clamped: unconditional loads)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j;
                if (t >= a.T) break;
                // A (+ B: are the nine taps of a level consecutive?)
                if (tid < 128) {
                    int i0 = 0;
                    float w1v = 0.f;
                    if (t_role) {
                        tap_1d(cs[t * 2 + t_axis], t_inv, t_o - 4, t_n, i0, w1v);
                        tab_i[t_axis][t_e] = i0;
                        tab_w[t_axis][t_e] = w1v;
                    }
                    const int prev = __shfl_up(i0, 1);
                    const unsigned long long bad = __ballot(t_role && t_o > 0 && i0 != prev + 1);
                    if ((tid & 63) < 4) (t_axis == 0 ? regx : regy)[tid & 63] = ((bad >> ((tid & 63) * 9)) & 0x1ffull) == 0;
                }
                if (tid < 324) gsh[tid] = gc[j];
                __syncthreads();
                // C
                if (tid < 360 && regx[c_lv] && regy[c_lv]) {
                    float v = 0.f;
                    if (c_x < 9) v = gsh[c_lv * 81 + c_x * 9 + c_b] * (1.f - tab_w[0][c_lv * 9 + c_x]);
                    if (c_x > 0) v += gsh[c_lv * 81 + (c_x - 1) * 9 + c_b] * tab_w[0][c_lv * 9 + c_x - 1];
                    tmp[c_lv][c_b][c_x] = v;
                }
                if (tid < 324 && !(regx[lv] && regy[lv])) {
                    const float g = gc[j];
                    const int x0 = tab_i[0][lv * 9 + ia], y0 = tab_i[1][lv * 9 + ib];
                    const float fx = tab_w[0][lv * 9 + ia], fy = tab_w[1][lv * 9 + ib];
                    const float ex = 1.f - fx, sy = 1.f - fy;
                    const bool x0ok = (unsigned)x0 < (unsigned)wl, x1ok = (unsigned)(x0 + 1) < (unsigned)wl;
                    const bool y0ok = (unsigned)y0 < (unsigned)hl, y1ok = (unsigned)(y0 + 1) < (unsigned)hl;
                    if (y0ok && x0ok) atomicAdd(pl + y0 * wl + x0, g * (sy * ex));
                    if (y0ok && x1ok) atomicAdd(pl + y0 * wl + x0 + 1, g * (sy * fx));
                    if (y1ok && x0ok) atomicAdd(pl + (y0 + 1) * wl + x0, g * (fy * ex));
                    if (y1ok && x1ok) atomicAdd(pl + (y0 + 1) * wl + x0 + 1, g * (fy * fx));
                }
                __syncthreads();
                // D
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int it = tid + rep * LBA_THREADS;
                    if (it < 400) {
                        const int dl = it / 100, dr = it - dl * 100, dy = dr / 10, dx = dr - dy * 10;
                        if (regx[dl] && regy[dl]) {
                            float v = 0.f;
                            if (dy < 9) v = tmp[dl][dy][dx] * (1.f - tab_w[1][dl * 9 + dy]);
                            if (dy > 0) v += tmp[dl][dy - 1][dx] * tab_w[1][dl * 9 + dy - 1];
                            const int Y = tab_i[1][dl * 9] + dy, X = tab_i[0][dl * 9] + dx;
                            const int hh = h0 >> dl, ww = w0 >> dl;
                            float* pp = dl == 0 ? s0 : (dl == 1 ? s1 : (dl == 2 ? s2 : s3));
                            if ((unsigned)Y < (unsigned)hh && (unsigned)X < (unsigned)ww) pp[Y * ww + X] += v;
                        }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) gc[j] = gn[j];
        }
        for (int i = tid; i < n2; i += LBA_THREADS) {          // pyramid_tiled_bwd_kernel's chain, in LDS
            const int y = i / w2, x = i - y * w2;
            if ((y >> 1) < h3 && (x >> 1) < w3) s2[i] += 0.25f * s3[(y >> 1) * w3 + (x >> 1)];
        }
        __syncthreads();
        for (int i = tid; i < n1; i += LBA_THREADS) {
            const int y = i / w1, x = i - y * w1;
            if ((y >> 1) < h2 && (x >> 1) < w2) s1[i] += 0.25f * s2[(y >> 1) * w2 + (x >> 1)];
        }
        __syncthreads();
        float* p0 = a.d0 + q * L.plane[0];
        const int n4 = L.plane[0] >> 2;
        for (int i = tid; i < n4; i += LBA_THREADS) {          // level 0 in tile order: 4 consecutive x of one tile row per thread
            const int tile = i >> 3, pc = i & 7;
            const int y = (tile / L.ntx[0]) * 4 + (pc >> 1), x = (tile % L.ntx[0]) * 8 + (pc & 1) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < h0) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x + e < w0) {
                        float u = s0[y * w0 + x + e];
                        if ((y >> 1) < h1 && ((x + e) >> 1) < w1) u += 0.25f * s1[(y >> 1) * w1 + ((x + e) >> 1)];
                        v[e] = u;
                    }
            }
            *reinterpret_cast<f32x4*>(p0 + i * 4) = v;
        }
        __syncthreads();
    }
}

}  // namespace

static int fill_levels(const ff::CorrLayout& L, int* h, int* w, int* ntx, int* nty) {
    for (int l = 0; l < 4; ++l) {
        h[l] = L.h[l];
        w[l] = L.w[l];
        ntx[l] = L.ntx[l];
        nty[l] = L.nty[l];
    }
    return 0;
}

extern "C" int ff_corr_lookup_tiled_fwd(const void* const* levels, int half, const float* coords, long long queries, int h0,
                                        int w0, float* out, int out_ld, int* taps_dbg, void* stream) {
    FF_REQUIRE(levels && coords && out, "ff_corr_lookup_tiled_fwd: null pointer");
    FF_REQUIRE(queries > 0 && out_ld >= 324, "ff_corr_lookup_tiled_fwd: out_ld %d < 324", out_ld);
    FF_REQUIRE((h0 >> 3) >= 2 && (w0 >> 3) >= 2, "ff_corr_lookup_tiled_fwd: level 3 is %dx%d; the sampler divides by (n-1)", h0 >> 3, w0 >> 3);
    for (int l = 0; l < 4; ++l)
        FF_REQUIRE(levels[l] != nullptr && ff::aligned16(levels[l]), "ff_corr_lookup_tiled_fwd: level %d null or misaligned", l);
    const int r = ff::lookup_dma_fwd(levels, half, coords, queries, h0, w0, out, out_ld, taps_dbg, static_cast<hipStream_t>(stream));
    FF_REQUIRE(r != 1, "ff_corr_lookup_tiled_fwd: the four levels must lie within 4 GB of each other (one buffer resource; allocate them "
                       "as ops.TiledPyramid does) with fewer than 2^24 queries per call - build and look up larger batches in chunks (CorrBlock does)");
    return r;
}

extern "C" int ff_corr_lookup_tiled_bwd(float* const* dlevels, const float* coords, const float* dout, int dout_ld,
                                        long long queries, int h0, int w0, void* stream) {
    FF_REQUIRE(dlevels && coords && dout && queries > 0 && dout_ld >= 324, "ff_corr_lookup_tiled_bwd: bad argument");
    FF_REQUIRE((h0 >> 3) >= 2 && (w0 >> 3) >= 2, "ff_corr_lookup_tiled_bwd: level 3 must be at least 2x2");
    const ff::CorrLayout L = ff::corr_layout(h0, w0, false);
    TLookupBwdArgs a;
    fill_levels(L, a.h, a.w, a.ntx, a.nty);
    for (int l = 0; l < 4; ++l) {
        FF_REQUIRE(dlevels[l] != nullptr && ff::aligned16(dlevels[l]), "ff_corr_lookup_tiled_bwd: level %d null or misaligned", l);
        a.dlvl[l] = dlevels[l];
        a.plane_elems[l] = L.plane[l];
    }
    a.coords = coords;
    a.dout = dout;
    a.dout_ld = dout_ld;
    a.queries = queries;
    static const int bwd_wpc = ff::tune_env("FF_LOOKUP_BWD_WAVES_PER_CU") ? atoi(ff::tune_env("FF_LOOKUP_BWD_WAVES_PER_CU")) : 20;
    const long long blocks = queries < 256ll * bwd_wpc ? queries : 256ll * bwd_wpc;
    lookup_tiled_bwd_kernel<<<(unsigned)blocks, 64, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_corr_lookup_tiled_bwd");
}

extern "C" int ff_corr_lookup_tiled_bwd_all(float* d0, const float* const* coords_list, const float* const* dout_list, int T, int dout_ld,
                                            long long queries, int h0, int w0, void* stream) {
    FF_REQUIRE(d0 && coords_list && dout_list && T > 0 && queries > 0 && dout_ld >= 324 && ff::aligned16(d0), "ff_corr_lookup_tiled_bwd_all: bad argument");
    FF_REQUIRE((h0 >> 3) >= 2 && (w0 >> 3) >= 2, "ff_corr_lookup_tiled_bwd_all: level 3 must be at least 2x2");
    if (T > LBA_MAXT) return 1;                       // more lookups than the argument block holds: the caller goes launch by launch
    LBAArgs a;
    a.L = ff::corr_layout(h0, w0, false);
    const size_t lds = (size_t)(a.L.h[0] * a.L.w[0] + a.L.h[1] * a.L.w[1] + a.L.h[2] * a.L.w[2] + a.L.h[3] * a.L.w[3]) * sizeof(float);
    if (lds > 64 * 1024) return 1;                    // the four planes of a query do not fit: launch by launch
    for (int t = 0; t < T; ++t) {
        FF_REQUIRE(coords_list[t] && dout_list[t], "ff_corr_lookup_tiled_bwd_all: null entry %d", t);
        a.coords[t] = coords_list[t];
        a.dout[t] = dout_list[t];
    }
    a.T = T;
    a.dout_ld = dout_ld;
    a.d0 = d0;
    a.queries = queries;
    FF_ALLOW_DYNAMIC_LDS((&lookup_bwd_all_kernel), 64 * 1024);
    const long long blocks = queries < 256ll * 64 ? queries : 256ll * 64;
    lookup_bwd_all_kernel<<<(unsigned)blocks, LBA_THREADS, lds, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_corr_lookup_tiled_bwd_all");
}

extern "C" int ff_corr_pyramid_tiled_bwd(float* d0, float* d1, float* d2, const float* d3, long long planes, int h0, int w0,
                                         void* stream) {
    FF_REQUIRE(d0 && d1 && d2 && d3 && planes > 0 && planes < (1ll << 31) && h0 >= 8 && w0 >= 8, "ff_corr_pyramid_tiled_bwd: bad argument");
    const ff::CorrLayout L = ff::corr_layout(h0, w0, false);
    const size_t lds = (size_t)(L.h[1] * L.w[1] + L.h[2] * L.w[2]) * sizeof(float);
    FF_REQUIRE(lds <= 64 * 1024, "ff_corr_pyramid_tiled_bwd: plane too large for LDS staging");
    pyramid_tiled_bwd_kernel<<<(unsigned)planes, 256, lds, static_cast<hipStream_t>(stream)>>>(d0, d1, d2, d3, L);
    return ff::check_launch("ff_corr_pyramid_tiled_bwd");
}
