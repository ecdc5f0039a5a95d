// CorrBlock.__call__ (corr.py:29-50 + bilinear_sampler, utils.py:57-71) on the TILED pyramid of corr_layout.h,
// fp32 or fp16 storage, 4 levels, radius 4; and its backward (scatter into tiled fp32 gradient planes) plus the pooling
// backward chain on tiled planes.
//
// Forward: ONE WAVE PER QUERY, software-pipelined as the row-major kernel was (taps -> window loads in flight during the
// previous query's blend -> LDS -> blend), but a window is fetched as whole 128-byte tiles: the 11 x 11 values a level
// needs lie in 2-3 x 3-4 tiles (fp32, 8 x 4) or 2-3 x 2-3 tiles (fp16, 8 x 8) instead of 11-15 row pieces of 128-byte
// lines.  8 lanes fetch one tile (16 bytes each); the tile range is computed from the taps actually needed (offset -4 ..
// offset +4, +1), tiles outside the plane are zeros (grid_sample's zero padding), elements of an edge tile beyond the
// plane are masked by coordinates (the pad of a tiled plane is not defined).  The coordinate arithmetic is the
// separately rounded fp32 replay of corr.hip (tap indices bit-identical to the reference).
#pragma clang fp contract(off)
#include <cstdlib>
#include "ff_common.h"
#include "corr_layout.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TLookupArgs {
    const char* lvl[4];
    long long plane_bytes[4];
    int h[4], w[4], ntx[4], nty[4];
    const float* coords;
    float* out;
    int* taps;
    long long queries;
    int out_ld;
};

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

constexpr int LTILES = 12;                 // LDS image of one level: 4 tile rows x 3 tile columns of 128 B
constexpr int LVL_BYTES = LTILES * 128;

// The window of one level as tile coordinates: first tile (tx0, ty0), ntc x ntr tiles.
struct WinGeom {
    int tx0, ty0, ntc, ntr;
};

template <bool HALF>
__global__ __launch_bounds__(64) void lookup_tiled_kernel(const TLookupArgs a) {
    constexpr int TSH = HALF ? 3 : 2;      // log2(tile height)
    constexpr int ESZ = HALF ? 2 : 4;
    constexpr int MAXR = HALF ? 3 : 4;     // tile rows a window can span
    __shared__ __attribute__((aligned(16))) char win[4 * LVL_BYTES];
    __shared__ int tab_o[2][4][2][2][9];   // [buf][level][axis][tap 0/1][offset]: byte offset of the tap in the level image
    __shared__ float tab_w[2][4][2][9];
    __shared__ int geo[2][4][4];           // [buf][level]{tx0, ty0, ntc, ntr}
    const int lane = threadIdx.x;
    const int t_lv = min(lane / 9, 3), t_o = lane - (lane / 9) * 9;   // tap role (lanes < 36)
    const float t_inv = 1.f / (float)(1 << t_lv);
    const int t_h = a.h[0] >> t_lv, t_w = a.w[0] >> t_lv;
    const int k8 = lane >> 3, piece = lane & 7;                       // staging role: tile k8 (+8), 16-byte piece

    auto publish_taps = [&](long long q, float cx, float cy, int buf) {
        int x0, y0;
        float wx, wy;
        tap_1d(cx, t_inv, t_o - 4, t_w, x0, wx);
        tap_1d(cy, t_inv, t_o - 4, t_h, y0, wy);
        const int xlo = __shfl(x0, t_lv * 9), ylo = __shfl(y0, t_lv * 9);            // taps of offset -4
        const int xhi = __shfl(x0, t_lv * 9 + 8) + 1, yhi = __shfl(y0, t_lv * 9 + 8) + 1;   // last tap read: offset +4, +1
        const int tx0 = xlo >> 3, ty0 = ylo >> TSH;
        const int ntc = min(max((xhi >> 3) - tx0 + 1, 1), 3), ntr = min(max((yhi >> TSH) - ty0 + 1, 1), MAXR);
        if (lane < 36) {
            // byte offset in the level image [tile row][3 tile columns][128 B]; clamped so a wild coordinate stays inside
            const int tc0 = min(max((x0 >> 3) - tx0, 0), 2), tc1 = min(max(((x0 + 1) >> 3) - tx0, 0), 2);
            const int tr0 = min(max((y0 >> TSH) - ty0, 0), MAXR - 1), tr1 = min(max(((y0 + 1) >> TSH) - ty0, 0), MAXR - 1);
            tab_o[buf][t_lv][0][0][t_o] = tc0 * 128 + (x0 & 7) * ESZ;
            tab_o[buf][t_lv][0][1][t_o] = tc1 * 128 + ((x0 + 1) & 7) * ESZ;
            tab_o[buf][t_lv][1][0][t_o] = tr0 * 384 + (y0 & ((1 << TSH) - 1)) * 8 * ESZ;
            tab_o[buf][t_lv][1][1][t_o] = tr1 * 384 + ((y0 + 1) & ((1 << TSH) - 1)) * 8 * ESZ;
            tab_w[buf][t_lv][0][t_o] = wx;
            tab_w[buf][t_lv][1][t_o] = wy;
            if (t_o == 0) {
                geo[buf][t_lv][0] = tx0;
                geo[buf][t_lv][1] = ty0;
                geo[buf][t_lv][2] = ntc;
                geo[buf][t_lv][3] = ntr;
            }
            if (a.taps) {
                int* t = a.taps + (q * 4 + t_lv) * 18;
                t[t_o] = x0;
                t[9 + t_o] = y0;
            }
        }
    };

    u32x4 rv[4][2];      // per level: two rounds of 8 tiles
    unsigned okm = 0;    // bit (2 lv + round): this lane's tile is inside the plane's tile grid (else: zeros = padding)
    unsigned wrm = 0;    // bit (2 lv + round): this lane's tile index lies inside the window (else: nothing to stage)
    int slot_lds[4][2];  // byte offset of this lane's piece in the level image
    unsigned emask[4][2];// per 16-byte piece: bit e set = element e (fp32: 4, fp16: 8) lies inside the plane
    auto issue_loads = [&](long long q, int buf) {
        okm = 0;
        wrm = 0;
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
            const int tx0 = geo[buf][lv][0], ty0 = geo[buf][lv][1], ntc = geo[buf][lv][2], ntr = geo[buf][lv][3];
            const int ntx = a.ntx[lv], nty = a.nty[lv];
            const char* pl = a.lvl[lv] + q * a.plane_bytes[lv];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int nt = ntc * ntr;
                const int k = min(k8 + 8 * rd, nt - 1);                   // lanes beyond the window re-read its last tile
                const int tr = (k * (ntc == 3 ? 43 : (ntc == 2 ? 64 : 128))) >> 7, tc = k - tr * ntc;   // k / ntc, k < 12
                const int gtx = tx0 + tc, gty = ty0 + tr;
                const bool wr = k8 + 8 * rd < nt;
                const bool in = (unsigned)gtx < (unsigned)ntx && (unsigned)gty < (unsigned)nty && wr;
                const int ctx = min(max(gtx, 0), ntx - 1), cty = min(max(gty, 0), nty - 1);
                rv[lv][rd] = *reinterpret_cast<const u32x4*>(pl + (size_t)((cty * ntx + ctx) * 128 + piece * 16));
                okm |= in ? 1u << (2 * lv + rd) : 0u;
                wrm |= wr ? 1u << (2 * lv + rd) : 0u;
                slot_lds[lv][rd] = (tr * 3 + tc) * 128 + piece * 16;
                // elements of the piece inside the plane: fp32 piece = row (piece >> 1), x = (piece & 1) * 4 .. +3 ;
                // fp16 piece = row piece, x = 0 .. 7
                const int ey = (gty << TSH) + (HALF ? piece : piece >> 1);
                const int ex = gtx * 8 + (HALF ? 0 : (piece & 1) * 4);
                const int nv = min(max((a.w[0] >> lv) - ex, 0), HALF ? 8 : 4);
                emask[lv][rd] = ((unsigned)ey < (unsigned)(a.h[0] >> lv)) ? (1u << nv) - 1u : 0u;
            }
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int lv = 0; lv < 4; ++lv)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const unsigned m = (okm >> (2 * lv + rd) & 1u) ? emask[lv][rd] : 0u;
                u32x4 v = rv[lv][rd];
                if (HALF) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        v[d] &= ((m >> (2 * d) & 1u) ? 0x0000ffffu : 0u) | ((m >> (2 * d + 1) & 1u) ? 0xffff0000u : 0u);
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) v[d] = (m >> d & 1u) ? v[d] : 0u;
                }
                // lanes whose tile index is past the window stage nothing (their slot aliases the window's last tile)
                if (wrm >> (2 * lv + rd) & 1u) *reinterpret_cast<u32x4*>(&win[lv * LVL_BYTES + slot_lds[lv][rd]]) = v;
            }
    };

    long long q = blockIdx.x;
    if (q >= a.queries) return;
    int cur = 0;
    publish_taps(q, a.coords[q * 2], a.coords[q * 2 + 1], 0);
    __syncthreads();
    issue_loads(q, 0);
    for (;;) {
        const long long qn = q + gridDim.x;
        const bool has_next = qn < a.queries;
        const long long qs = has_next ? qn : q;          // the last round re-stages its own query: nothing under a branch
        const float cxn = a.coords[qs * 2], cyn = a.coords[qs * 2 + 1];
        store_window();                                  // waits for this query's window loads
        publish_taps(qs, cxn, cyn, cur ^ 1);
        __syncthreads();                                 // win + both table sets visible
        issue_loads(qs, cur ^ 1);                        // in flight during the blend below
        float* orow = a.out + q * a.out_ld;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = lane + 64 * j;
            if (k < 324) {
                const int lv = k / 81, rem = k - lv * 81;
                const int ia = rem / 9, ib = rem - ia * 9;
                const int xo0 = tab_o[cur][lv][0][0][ia], xo1 = tab_o[cur][lv][0][1][ia];
                const int yo0 = tab_o[cur][lv][1][0][ib], yo1 = tab_o[cur][lv][1][1][ib];
                const float fx = tab_w[cur][lv][0][ia], fy = tab_w[cur][lv][1][ib];
                const char* p = &win[lv * LVL_BYTES];
                float v00, v01, v10, v11;
                if (HALF) {
                    v00 = (float)*reinterpret_cast<const _Float16*>(p + yo0 + xo0);
                    v01 = (float)*reinterpret_cast<const _Float16*>(p + yo0 + xo1);
                    v10 = (float)*reinterpret_cast<const _Float16*>(p + yo1 + xo0);
                    v11 = (float)*reinterpret_cast<const _Float16*>(p + yo1 + xo1);
                } else {
                    v00 = *reinterpret_cast<const float*>(p + yo0 + xo0);
                    v01 = *reinterpret_cast<const float*>(p + yo0 + xo1);
                    v10 = *reinterpret_cast<const float*>(p + yo1 + xo0);
                    v11 = *reinterpret_cast<const float*>(p + yo1 + xo1);
                }
                const float ex = __fsub_rn(1.f, fx), sy = __fsub_rn(1.f, fy);
                // nw*s*e + ne*s*w + sw*n*e + se*n*w  (ATen's weight naming)
                float o = __fmul_rn(v00, __fmul_rn(sy, ex));
                o = __fadd_rn(o, __fmul_rn(v01, __fmul_rn(sy, fx)));
                o = __fadd_rn(o, __fmul_rn(v10, __fmul_rn(fy, ex)));
                o = __fadd_rn(o, __fmul_rn(v11, __fmul_rn(fy, fx)));
                orow[k] = o;
            }
        }
        if (!has_next) break;
        __syncthreads();                                 // everyone done reading win before it is overwritten
        q = qn;
        cur ^= 1;
    }
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

__global__ __launch_bounds__(64) void lookup_tiled_bwd_kernel(const TLookupBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float win[4 * LTILES * 32];
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
        for (int e = lane; e < 4 * LTILES * 32; e += 64) win[e] = 0.f;
        __syncthreads();
        const float* drow = a.dout + q * a.dout_ld;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = lane + 64 * j;
            if (k < 324) {
                const int lv = k / 81, rem = k - lv * 81;
                const int ia = rem / 9, ib = rem - ia * 9;
                const int xo0 = tab_o[lv][0][0][ia], xo1 = tab_o[lv][0][1][ia];
                const int yo0 = tab_o[lv][1][0][ib], yo1 = tab_o[lv][1][1][ib];
                const float fx = tab_w[lv][0][ia], fy = tab_w[lv][1][ib];
                const float g = drow[k];
                const float ex = 1.f - fx, sy = 1.f - fy;
                float* p = &win[lv * LTILES * 32];
                atomicAdd(p + yo0 + xo0, g * (sy * ex));
                atomicAdd(p + yo0 + xo1, g * (sy * fx));
                atomicAdd(p + yo1 + xo0, g * (fy * ex));
                atomicAdd(p + yo1 + xo1, g * (fy * fx));
            }
        }
        __syncthreads();
        // whole-tile read-modify-write: all loads first (lanes with nothing to add re-read a valid piece), then the stores
        f32x4 cur[4][2], add[4][2];
        long long idx[4][2];
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
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
                const long long o = (long long)(cty * ntx + ctx) * 32 + piece * 4;
                cur[lv][rd] = *reinterpret_cast<const f32x4*>(pl + o);
                f32x4 v = *reinterpret_cast<const f32x4*>(&win[lv * LTILES * 32 + (tr * 3 + tc) * 32 + piece * 4]);
                const int ey = gty * 4 + (piece >> 1), ex = gtx * 8 + (piece & 1) * 4;
                const int nv = ((unsigned)ey < (unsigned)(a.h[0] >> lv)) ? min(max((a.w[0] >> lv) - ex, 0), 4) : 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) v[d] = d < nv ? v[d] : 0.f;
                add[lv][rd] = v;
                idx[lv][rd] = in ? o : -1;
            }
        }
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
            float* pl = a.dlvl[lv] + q * a.plane_elems[lv];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd)
                if (idx[lv][rd] >= 0) *reinterpret_cast<f32x4*>(pl + idx[lv][rd]) = cur[lv][rd] + add[lv][rd];
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
    const ff::CorrLayout L = ff::corr_layout(h0, w0, half != 0);
    TLookupArgs a;
    fill_levels(L, a.h, a.w, a.ntx, a.nty);
    for (int l = 0; l < 4; ++l) {
        FF_REQUIRE(levels[l] != nullptr && ff::aligned16(levels[l]), "ff_corr_lookup_tiled_fwd: level %d null or misaligned", l);
        a.lvl[l] = static_cast<const char*>(levels[l]);
        a.plane_bytes[l] = (long long)L.plane[l] * (half ? 2 : 4);
    }
    a.coords = coords;
    a.out = out;
    a.taps = taps_dbg;
    a.queries = queries;
    a.out_ld = out_ld;
    // one wave per block, 7.7 KB of LDS each: 20 blocks fit a CU; a grid of 256 x 16 runs as one resident round
    static const int wpc = getenv("FF_LOOKUP_WAVES_PER_CU") ? atoi(getenv("FF_LOOKUP_WAVES_PER_CU")) : 16;
    const long long blocks = queries < 256ll * wpc ? queries : 256ll * wpc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (half) lookup_tiled_kernel<true><<<(unsigned)blocks, 64, 0, s>>>(a);
    else lookup_tiled_kernel<false><<<(unsigned)blocks, 64, 0, s>>>(a);
    return ff::check_launch("ff_corr_lookup_tiled_fwd");
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
    const long long blocks = queries < 256ll * 16 ? queries : 256ll * 16;
    lookup_tiled_bwd_kernel<<<(unsigned)blocks, 64, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_corr_lookup_tiled_bwd");
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
