/* TEST INFRASTRUCTURE — plain-C restatement of the reference's CorrBlock
 * (core/models/ff-raft/FF_RAFT_Core/corr.py:12-60 + utils/utils.py:57-71).
 * Parity status: pinned — tests/test_oracle_golden.py checks it against
 * vectors produced by the reference (tests/golden/*.npz).
 * Build: gcc -O2 -ffp-contract=off (every fp32 op separately rounded).
 * Layouts follow the reference: fmaps NCHW, planes [B*Q][h][w], output (B,K,H,W). */
#include <math.h>
#include <stddef.h>

/* corr.py:52-60: vol[b][i][j] = sum_c f1[b][c][i]*f2[b][c][j] / sqrt(C) */
void orc_corr_volume(const float* f1, const float* f2, float* vol, int B, int C, int Q) {
    const float s = sqrtf((float)C);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < Q; ++i)
            for (int j = 0; j < Q; ++j) {
                float acc = 0.f;
                for (int c = 0; c < C; ++c) acc += f1[((size_t)b * C + c) * Q + i] * f2[((size_t)b * C + c) * Q + j];
                vol[((size_t)b * Q + i) * Q + j] = acc / s;
            }
}

/* corr.py:24-27: avg_pool2d(2, stride 2), floor on odd sizes */
void orc_avg_pool2(const float* src, float* dst, long planes, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    for (long p = 0; p < planes; ++p)
        for (int y = 0; y < ho; ++y)
            for (int x = 0; x < wo; ++x) {
                const float* s = src + (size_t)p * h * w + (size_t)(2 * y) * w + 2 * x;
                dst[(size_t)p * ho * wo + (size_t)y * wo + x] = (((s[0] + s[1]) + s[w]) + s[w + 1]) / 4.0f;
            }
}

/* utils.py:61-62 then ATen's align_corners un-normalise; returns floor index, *w1 = frac */
static int tap(float c, int level, int off, int n, float* w1) {
    const float cl = c / (float)(1 << level);          /* corr.py:41 */
    const float x = cl + (float)off;                   /* corr.py:43 */
    const float g = 2.f * x / (float)(n - 1) - 1.f;    /* utils.py:61 */
    const float u = ((g + 1.f) / 2.f) * (float)(n - 1);
    const float f = floorf(u);
    *w1 = u - f;
    return (int)f;
}

/* corr.py:29-50.  coords (B,2,H,W) [x;y]; out (B, L*(2r+1)^2, H, W);
 * taps (nullable) int32 [B*Q][L][2][2r+1]. */
void orc_corr_lookup(const float* const* levels, int L, int r, const float* coords, int B, int H, int W,
                     float* out, int* taps) {
    const int win = 2 * r + 1, Q = H * W, K = L * win * win;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Q; ++q) {
            const float cx = coords[((size_t)b * 2) * Q + q], cy = coords[((size_t)b * 2 + 1) * Q + q];
            int hl = H, wl = W;
            for (int l = 0; l < L; ++l) {
                const float* pl = levels[l] + ((size_t)b * Q + q) * hl * wl;
                for (int ia = 0; ia < win; ++ia)
                    for (int ib = 0; ib < win; ++ib) {
                        float wx, wy;
                        const int x0 = tap(cx, l, ia - r, wl, &wx), y0 = tap(cy, l, ib - r, hl, &wy);
                        if (taps) {
                            int* t = taps + (((size_t)b * Q + q) * L + l) * 2 * win;
                            t[ia] = x0;
                            t[win + ib] = y0;
                        }
                        const float ex = 1.f - wx, sy = 1.f - wy;
                        float acc = 0.f;
                        if (x0 >= 0 && x0 < wl && y0 >= 0 && y0 < hl) acc += pl[y0 * wl + x0] * (sy * ex);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 >= 0 && y0 < hl) acc += pl[y0 * wl + x0 + 1] * (sy * wx);
                        if (x0 >= 0 && x0 < wl && y0 + 1 >= 0 && y0 + 1 < hl) acc += pl[(y0 + 1) * wl + x0] * (wy * ex);
                        if (x0 + 1 >= 0 && x0 + 1 < wl && y0 + 1 >= 0 && y0 + 1 < hl) acc += pl[(y0 + 1) * wl + x0 + 1] * (wy * wx);
                        out[(((size_t)b * K) + (size_t)l * win * win + ia * win + ib) * Q + q] = acc;
                    }
                hl /= 2;
                wl /= 2;
            }
        }
}

/* The five fused operations csrc/corr_lookup_dma.hip uses for g = 2 x / (n - 1) (utils.py:61) next to the true
 * division, so that the GPU kernel's tap arithmetic is pinned on the CPU: r2 = 2 RN(1 / d), dh = d / 2. */
static float div5(float x, float r2, float dh) {
    const float q0 = x * r2;
    const float e0 = fmaf(-q0, dh, x);
    const float q1 = fmaf(e0, r2, q0);
    const float e1 = fmaf(-q1, dh, x);
    return fmaf(e1, r2, q1);
}

/* number of x in xs[0..n) for which div5 differs (bit pattern) from (2 x) / d; *first = index of the first one */
long orc_div5_mismatches(const float* xs, long n, int d, long* first) {
    const float df = (float)d, r = 1.0f / df, r2 = r + r, dh = 0.5f * df;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const float a = div5(xs[i], r2, dh), b = (2.f * xs[i]) / df;
        if (a != b && !(a != a && b != b)) {
            if (!bad && first) *first = i;
            ++bad;
        }
    }
    return bad;
}
