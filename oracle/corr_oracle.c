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

/* FF-PWC cost volume, SECOND restatement: a loop-level transcription of the reference's CUDA strings
 * (core/models/ff-pwcnet/PWCNet_Core/correlation.py) - index arithmetic, padding and summation order as written there,
 * one "thread" at a time.  oracle/pwc_ref.py states the same function tensor-wise (unfold style); the two share no code,
 * so a misreading of the kernel would have to be made twice in different forms to go unnoticed
 * (tests/test_oracle_golden.py compares them).
 *   kernel_Correlation_rearrange     :7-32   NCHW -> zero-padded NHWC rbot[B][H + 8][W + 8][C], pad 4
 *   kernel_Correlation_updateOutput  :34-102 one block per output pixel, 32 threads striding the channels,
 *                                            top[b][ch][y][x] = (sum over c) rbot0[b][y+4][x+4][c] * rbot1[b][y+4+s2p][x+4+s2o][c] / C
 *                                            with s2o = ch % 9 - 4, s2p = ch / 9 - 4 (:71-72); thread 0 adds the 32 partial sums (:90-98) */
#include <stdlib.h>
static void pwc_rearrange(const float* input, float* output, int B, int C, int H, int W) {   /* :7-32, grid (ceil(HW/16), C, B) */
    const int n = H * W;
    for (int intSample = 0; intSample < B; ++intSample)
        for (int intChannel = 0; intChannel < C; ++intChannel)
            for (int intIndex = 0; intIndex < n; ++intIndex) {
                const float fltValue = input[(((size_t)intSample * C + intChannel) * H * W) + intIndex];
                const int intPaddedY = (intIndex / W) + 4;
                const int intPaddedX = (intIndex % W) + 4;
                const int intRearrange = ((W + 8) * intPaddedY) + intPaddedX;
                /* SIZE_1(output) = H + 8, SIZE_2(output) = W + 8, SIZE_1(input) = C */
                output[(((size_t)intSample * (H + 8) * (W + 8)) + intRearrange) * C + intChannel] = fltValue;
            }
}

int orc_pwc_costvolume_kernel(const float* one, const float* two, float* top, int B, int C, int H, int W) {
    const size_t padded = (size_t)B * (H + 8) * (W + 8) * C;
    float* rbot0 = (float*)calloc(padded, sizeof(float));      /* torch.zeros, correlation.py:285-286 */
    float* rbot1 = (float*)calloc(padded, sizeof(float));
    float* patch_data = (float*)malloc((size_t)C * sizeof(float));
    if (!rbot0 || !rbot1 || !patch_data) return -1;
    pwc_rearrange(one, rbot0, B, C, H, W);
    pwc_rearrange(two, rbot1, B, C, H, W);
    const int S1 = H + 8, S2 = W + 8, S3 = C;                   /* SIZE_1..3(rbot0) */
    for (int item = 0; item < B; ++item)                         /* blockIdx.z */
        for (int by = 0; by < H; ++by)                           /* blockIdx.y */
            for (int bx = 0; bx < W; ++bx) {                     /* blockIdx.x */
                const int x1 = bx + 4, y1 = by + 4;
                for (int ch_off = 0; ch_off < 32; ++ch_off)      /* threadIdx.x: load the 1 x 1 x C patch (:52-61) */
                    for (int ch = ch_off; ch < S3; ch += 32) {
                        const size_t idx1 = (((size_t)item * S1 + y1) * S2 + x1) * S3 + ch;
                        patch_data[ch] = rbot0[idx1];
                    }
                for (int top_channel = 0; top_channel < 81; ++top_channel) {
                    float sum[32];
                    const int s2o = top_channel % 9 - 4;
                    const int s2p = top_channel / 9 - 4;
                    for (int ch_off = 0; ch_off < 32; ++ch_off) {
                        sum[ch_off] = 0;
                        for (int ch = ch_off; ch < S3; ch += 32) {
                            const int x2 = x1 + s2o;
                            const int y2 = y1 + s2p;
                            const size_t idx2 = (((size_t)item * S1 + y2) * S2 + x2) * S3 + ch;
                            sum[ch_off] += patch_data[ch] * rbot1[idx2];
                        }
                    }
                    float total_sum = 0;
                    for (int idx = 0; idx < 32; ++idx) total_sum += sum[idx];
                    const int sumelems = S3;
                    const size_t index = (((size_t)top_channel * H + by) * W) + bx;      /* SIZE_2(top) = H, SIZE_3(top) = W */
                    top[index + (size_t)item * 81 * H * W] = total_sum / (float)sumelems;
                }
            }
    free(rbot0);
    free(rbot1);
    free(patch_data);
    return 0;
}
