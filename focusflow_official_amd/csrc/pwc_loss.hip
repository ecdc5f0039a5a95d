// FF-PWC multi-scale losses (core/models/ff-pwcnet/losses/losses.py:19-261: EPELoss, CPCL, MixLoss; dense ground truth).
// Per pyramid level the reference area-interpolates the target to the level's size, bilinearly resizes the key-point
// mask, thresholds it (> 0) and convolves it with a Gaussian, forms an end-point-error map and reduces it.  Here one
// kernel builds the level's mask weight map (+ its sum) and one kernel does the rest, gradient included:
//
//   ff_pwc_loss_mask   gmask[b,y,x] = sum_k G[k] * [bilinear(mask)(y+ky-p, x+kx-p) > 0]   (zero padded); *msum += sum gmask
//   ff_pwc_loss_scale  E = |t - o|_2  (pretrain)  or  (|t - o|_1 + eps)^q ;  pixel weight = w_plain + w_mask * gmask,
//                      w_mask = w_mask_num / *msum (0 when *msum == 0 and zero_if_empty); mask_over_batch: gmask summed over
//                      the batch (the broadcasting of CPCL's (B,h,w) x (B,1,h,w) product, losses.py:114);
//                      *loss += sum weight * E ;  grad = d(that)/d(o)
//   ff_pwc_epe_mean    *out2 += { sum of E over all pixels, pixel count }   (the reported 'epe' on the resized output)
#include "ff_common.h"

namespace {

__device__ __forceinline__ float bilinear_src(int dst, float scale) {
    const float s = ((float)dst + 0.5f) * scale - 0.5f;
    return s < 0.f ? 0.f : s;
}

__global__ void pwc_loss_mask_kernel(const float* __restrict__ mask, const float* __restrict__ gauss, int ks, float* __restrict__ gmask,
                                     double* __restrict__ msum, int B, int H, int W, int h, int w) {
    const long long total = (long long)B * h * w;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const int pad = ks / 2;
    double local = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long b = i / ((long long)w * h);
        const float* m = mask + b * H * W;
        float acc = 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int yy = y + ky - pad;
            if ((unsigned)yy >= (unsigned)h) continue;
            const float fy = bilinear_src(yy, sy);
            const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = fy - (float)y0;
            for (int kx = 0; kx < ks; ++kx) {
                const int xx = x + kx - pad;
                if ((unsigned)xx >= (unsigned)w) continue;
                const float fx = bilinear_src(xx, sx);
                const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = fx - (float)x0;
                const float v = (1.f - ly) * ((1.f - lx) * m[y0 * W + x0] + lx * m[y0 * W + x1]) +
                                ly * ((1.f - lx) * m[y1 * W + x0] + lx * m[y1 * W + x1]);
                if (v > 0.f) acc += gauss[ky * ks + kx];
            }
        }
        gmask[i] = acc;
        local += acc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local != 0.0) atomicAdd(msum, local);
}

__device__ __forceinline__ float epe_value(float dx, float dy, int l1q, float eps, float q, float& gx, float& gy) {
    if (!l1q) {   // |d|_2 ; d/d(d) = d / |d|  (ATen's norm backward gives 0 at d = 0)
        const float n = sqrtf(dx * dx + dy * dy);
        gx = n > 0.f ? dx / n : 0.f;
        gy = n > 0.f ? dy / n : 0.f;
        return n;
    }
    const float s = fabsf(dx) + fabsf(dy) + eps;
    const float e = powf(s, q);
    const float de = q * powf(s, q - 1.f);
    gx = de * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f));
    gy = de * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f));
    return e;
}

// SPARSE (KITTI stage, losses.py:28-41, :58-67, :186-214): the target is down-sampled by sparse_max_pool (max of the
// positive values minus max of the negated negative ones over the adaptive window), pixels whose pooled target is
// exactly (0, 0) are invalid; `mask_over_batch` then means "the plain term counts valid pixels only" (EPELoss; MixLoss
// sums the plain term over every pixel and only the key-point term over valid ones).
template <bool SPARSE>
__global__ void pwc_loss_scale_kernel(const float* __restrict__ out, const float* __restrict__ target, const float* __restrict__ gmask,
                                      const double* __restrict__ msum, float w_plain, float w_mask_num, int zero_if_empty,
                                      int mask_over_batch, int l1q, float eps, float q, float* __restrict__ grad, double* __restrict__ loss, int B, int H, int W, int h,
                                      int w) {
    const long long total = (long long)B * h * w;
    float w_mask = 0.f;
    if (gmask) {
        const double ms = *msum;
        w_mask = (ms == 0.0 && zero_if_empty) ? 0.f : (float)((double)w_mask_num / ms);
    }
    double local = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long b = i / ((long long)w * h);
        // F.interpolate(mode='area') = adaptive average pooling: window [floor(i*H/h), ceil((i+1)*H/h))
        const int ys = (int)(((long long)y * H) / h), ye = (int)((((long long)y + 1) * H + h - 1) / h);
        const int xs = (int)(((long long)x * W) / w), xe = (int)((((long long)x + 1) * W + w - 1) / w);
        float tx = 0.f, ty = 0.f;
        const float* t0 = target + b * 2 * H * W;
        if (SPARSE) {
            float px = 0.f, nx = 0.f, py = 0.f, ny = 0.f;      // max(t * (t > 0)), max(-t * (t < 0)): both >= 0
            for (int yy = ys; yy < ye; ++yy)
                for (int xx = xs; xx < xe; ++xx) {
                    const float a = t0[(long long)yy * W + xx], c = t0[(long long)H * W + (long long)yy * W + xx];
                    px = fmaxf(px, a);
                    nx = fmaxf(nx, -a);
                    py = fmaxf(py, c);
                    ny = fmaxf(ny, -c);
                }
            tx = px - nx;
            ty = py - ny;
        } else {
            for (int yy = ys; yy < ye; ++yy)
                for (int xx = xs; xx < xe; ++xx) {
                    tx += t0[(long long)yy * W + xx];
                    ty += t0[(long long)H * W + (long long)yy * W + xx];
                }
            const float inv = 1.f / (float)((ye - ys) * (xe - xs));
            tx *= inv;
            ty *= inv;
        }
        const bool valid = !SPARSE || !(tx == 0.f && ty == 0.f);
        const long long o0 = (b * 2 * h + y) * w + x, o1 = o0 + (long long)h * w;
        float gx, gy;
        const float e = epe_value(tx - out[o0], ty - out[o1], l1q, eps, q, gx, gy);
        float gm = 0.f;
        if (gmask && SPARSE) {
            gm = valid ? gmask[i] : 0.f;
        } else if (gmask) {
            if (mask_over_batch) {          // CPCL multiplies a (B,h,w) error map by a (B,1,h,w) mask (:114): broadcasting
                const long long pix = i - b * (long long)h * w;            // pairs every sample with every sample's mask
                for (int bb = 0; bb < B; ++bb) gm += gmask[(long long)bb * h * w + pix];
            } else {
                gm = gmask[i];
            }
        }
        const float wp = ((SPARSE && mask_over_batch && !valid) ? 0.f : w_plain) + w_mask * gm;
        local += (double)(wp * e);
        if (grad) {           // d/d(out) = -d/d(d)
            grad[o0] = -wp * gx;
            grad[o1] = -wp * gy;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, local);
}

// sparse != 0: only pixels whose target is not exactly (0, 0) count (realEPE with sparse=True, losses.py:33-37)
__global__ void pwc_epe_mean_kernel(const float* __restrict__ pred, const float* __restrict__ target, int l1q, float eps, float q,
                                    double* __restrict__ out2, int B, int H, int W, int sparse) {
    const long long hw = (long long)H * W, total = B * hw;
    double local = 0.0, cnt = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / hw, p = i - b * hw;
        const float tx = target[b * 2 * hw + p], ty = target[b * 2 * hw + hw + p];
        if (sparse && tx == 0.f && ty == 0.f) continue;
        float gx, gy;
        local += epe_value(tx - pred[b * 2 * hw + p], ty - pred[b * 2 * hw + hw + p], l1q, eps, q, gx, gy);
        cnt += 1.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        local += __shfl_xor(local, o);
        cnt += __shfl_xor(cnt, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out2, local);
        atomicAdd(out2 + 1, cnt);
    }
}

inline unsigned grid_for(long long total) {
    const long long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int ff_pwc_loss_mask(const float* mask, const float* gauss, int ks, float* gmask, double* msum, int B, int H, int W, int h,
                                int w, void* stream) {
    FF_REQUIRE(mask && gauss && gmask && msum && ks >= 1 && (ks & 1) && B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "ff_pwc_loss_mask: bad argument");
    pwc_loss_mask_kernel<<<grid_for((long long)B * h * w), 256, 0, static_cast<hipStream_t>(stream)>>>(mask, gauss, ks, gmask, msum, B, H, W, h, w);
    return ff::check_launch("ff_pwc_loss_mask");
}

extern "C" int ff_pwc_loss_scale(const float* out, const float* target, const float* gmask, const double* msum, float w_plain,
                                 float w_mask_num, int zero_if_empty, int mask_over_batch, int l1q, float eps, float q, float* grad,
                                 double* loss, int B, int H, int W, int h, int w, void* stream) {
    FF_REQUIRE(out && target && loss && (!gmask || msum) && B > 0 && H >= h && W >= w && h > 0 && w > 0, "ff_pwc_loss_scale: bad argument");
    pwc_loss_scale_kernel<false><<<grid_for((long long)B * h * w), 256, 0, static_cast<hipStream_t>(stream)>>>(out, target, gmask, msum, w_plain, w_mask_num,
                                                                                                           zero_if_empty, mask_over_batch, l1q, eps, q, grad, loss, B, H, W, h, w);
    return ff::check_launch("ff_pwc_loss_scale");
}

extern "C" int ff_pwc_loss_scale_sparse(const float* out, const float* target, const float* gmask, const double* msum, float w_plain,
                                        float w_mask_num, int zero_if_empty, int plain_valid_only, int l1q, float eps, float q, float* grad,
                                        double* loss, int B, int H, int W, int h, int w, void* stream) {
    FF_REQUIRE(out && target && loss && (!gmask || msum) && B > 0 && H >= h && W >= w && h > 0 && w > 0, "ff_pwc_loss_scale_sparse: bad argument");
    pwc_loss_scale_kernel<true><<<grid_for((long long)B * h * w), 256, 0, static_cast<hipStream_t>(stream)>>>(out, target, gmask, msum, w_plain, w_mask_num,
                                                                                                          zero_if_empty, plain_valid_only, l1q, eps, q, grad, loss, B, H, W, h, w);
    return ff::check_launch("ff_pwc_loss_scale_sparse");
}

extern "C" int ff_pwc_epe_mean(const float* pred, const float* target, int l1q, float eps, float q, double* out2, int B, int H, int W,
                               void* stream) {
    FF_REQUIRE(pred && target && out2 && B > 0 && H > 0 && W > 0, "ff_pwc_epe_mean: bad argument");
    pwc_epe_mean_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(pred, target, l1q, eps, q, out2, B, H, W, 0);
    return ff::check_launch("ff_pwc_epe_mean");
}

extern "C" int ff_pwc_epe_mean_sparse(const float* pred, const float* target, int l1q, float eps, float q, double* out2, int B, int H, int W,
                                      void* stream) {
    FF_REQUIRE(pred && target && out2 && B > 0 && H > 0 && W > 0, "ff_pwc_epe_mean_sparse: bad argument");
    pwc_epe_mean_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(pred, target, l1q, eps, q, out2, B, H, W, 1);
    return ff::check_launch("ff_pwc_epe_mean_sparse");
}
