// Tiled layout of the correlation pyramid (CorrBlock, corr.py:12-27), shared by the build, lookup and backward kernels.
//
// Level l of query i is a plane of h_l x w_l values (h_l = h0 >> l: avg_pool2d floors).  A plane is stored as 128-byte
// 2-D tiles so that the 11 x 11 window a lookup reads touches few HBM lines:
//     fp32 storage: tile = 8 wide x 4 high floats        fp16 storage: tile = 8 wide x 8 high halfs
//     element (y, x)  ->  ((y / TH) * ntx + (x / 8)) * (8 * TH) + (y % TH) * 8 + (x % 8)
// The tile grid covers the plane padded to the build kernel's 16 x 8 patch of level-0 positions (wp0 = ceil16(w0),
// hp0 = ceil8(h0), level l: wp0 >> l by hp0 >> l), rounded up to whole tiles.  Pad elements are ZERO by contract
// (ff_corr_build clears what its patches do not cover; ff_corr_retile fills zero-initialised storage; the backward
// kernels mask their updates), so the lookup needs no per-element masks.
#pragma once

namespace ff {

struct CorrLayout {
    int h[4], w[4];        // plane sizes
    int ntx[4], nty[4];    // tile grid
    int plane[4];          // elements per plane (ntx * nty * 8 * th)
    int th;                // tile height: 4 (fp32) or 8 (fp16)
    int npx, npy;          // 16 x 8 patches of level 0
};

__host__ __device__ inline CorrLayout corr_layout(int h0, int w0, bool half) {
    CorrLayout L;
    L.th = half ? 8 : 4;
    L.npx = (w0 + 15) / 16;
    L.npy = (h0 + 7) / 8;
    for (int l = 0; l < 4; ++l) {
        L.h[l] = h0 >> l;
        L.w[l] = w0 >> l;
        const int wp = (L.npx * 16) >> l, hp = (L.npy * 8) >> l;
        L.ntx[l] = (wp + 7) / 8;
        L.nty[l] = (hp + L.th - 1) / L.th;
        L.plane[l] = L.ntx[l] * L.nty[l] * 8 * L.th;
    }
    return L;
}

// element offset inside a plane; TH = 4 or 8
template <int TH>
__host__ __device__ inline int tiled_offset(int y, int x, int ntx) {
    return ((y / TH) * ntx + (x >> 3)) * (8 * TH) + (y % TH) * 8 + (x & 7);
}

}  // namespace ff
