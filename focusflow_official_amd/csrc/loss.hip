// Fused sequence loss of FF-RAFT (core/models/ff-raft/losses/losses.py:18-130).
//
// EPELoss, CPCL and MixLoss are one weighted sequence-L1:
//     loss = sum_i gamma^(n-1-i) * sum_{b,c,y,x} w[b,y,x] * |pred_i - gt|
//     w    = valid * (a + lam * G*mask / sum(G*mask)),  a = 1/(B*2*H*W) (EPE term) or 0
// with valid = (valid>=0.5) & (|gt| < max_flow) and G a Gaussian box of the key-point mask.
// The reference runs ~8 elementwise/reduction kernels per prediction (12-32 full-resolution
// predictions); here one pass per prediction reads pred and gt once, accumulates the loss in
// fp64 and writes d(loss)/d(pred) at the same time (the loss is the terminal node of the tape).
#include "ff_common.h"

namespace {

// wmap[b,y,x] = valid ? 1 : 0 ; gconv[b,y,x] = sum_k G[k] * (mask>0)[y+dy,x+dx]; also sum(gconv) and
// per-call sum of squared error pieces are handled elsewhere
__global__ void loss_prepare_kernel(const float* __restrict__ gt, const float* __restrict__ valid,
                                    const float* __restrict__ mask, const float* __restrict__ gk, int ks,
                                    float max_flow, float* __restrict__ vmap, float* __restrict__ gconv,
                                    double* __restrict__ gsum, int B, int H, int W) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    double local = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW;
        const int p = (int)(i - b * HW), y = p / W, x = p - y * W;
        const float fx = gt[(b * 2) * HW + p], fy = gt[(b * 2 + 1) * HW + p];
        const float mag = sqrtf(fx * fx + fy * fy);
        vmap[i] = (valid[i] >= 0.5f && mag < max_flow) ? 1.f : 0.f;
        if (mask) {
            const int r = ks / 2;
            float acc = 0.f;
            for (int dy = 0; dy < ks; ++dy) {
                const int yy = y + dy - r;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = 0; dx < ks; ++dx) {
                    const int xx = x + dx - r;
                    if ((unsigned)xx < (unsigned)W && mask[b * HW + yy * W + xx] > 0.f) acc += gk[dy * ks + dx];
                }
            }
            gconv[i] = acc;
            local += (double)acc;
        }
    }
    if (mask) {
        __shared__ double red[256];
        red[threadIdx.x] = local;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(gsum, red[0]);
    }
}

// one prediction: loss += wgt_i * sum w*|pred-gt| ; grad = wgt_i * w * sign(pred-gt)
__global__ void loss_accum_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                  const float* __restrict__ vmap, const float* __restrict__ gconv,
                                  const double* __restrict__ gsum, float a_mean, float lam, float wgt,
                                  float* __restrict__ grad, double* __restrict__ loss, int B, int HW) {
    const long long total = (long long)B * 2 * HW;
    const float inv_g = gconv ? (float)(1.0 / *gsum) : 0.f;
    double local = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long bc = i / HW;
        const long long pix = (bc >> 1) * HW + (i - bc * HW);
        float w = vmap[pix] * a_mean;
        if (gconv) w += vmap[pix] * gconv[pix] * inv_g * lam;
        const float d = pred[i] - gt[i];
        local += (double)(w * fabsf(d));
        if (grad) grad[i] = wgt * w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    __shared__ double red[256];
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, red[0] * (double)wgt);
}

// end-point error of the last prediction over valid pixels: {sum epe, count}
__global__ void epe_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ vmap,
                           double* __restrict__ out, int B, int HW) {
    const long long total = (long long)B * HW;
    double s = 0.0, n = 0.0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (vmap[i] > 0.f) {
            const long long b = i / HW;
            const long long p = i - b * HW;
            const float dx = pred[(b * 2) * HW + p] - gt[(b * 2) * HW + p];
            const float dy = pred[(b * 2 + 1) * HW + p] - gt[(b * 2 + 1) * HW + p];
            s += (double)sqrtf(dx * dx + dy * dy);
            n += 1.0;
        }
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s;
    r2[threadIdx.x] = n;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(out, r1[0]); atomicAdd(out + 1, r2[0]); }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ff_loss_prepare(const float* flow_gt, const float* valid, const float* mask, const float* gauss, int ks,
                               float max_flow, float* vmap, float* gconv, double* gsum, int B, int H, int W,
                               void* stream) {
    FF_REQUIRE(flow_gt && valid && vmap && B > 0 && H > 0 && W > 0, "ff_loss_prepare: bad argument");
    FF_REQUIRE(!mask || (gauss && gconv && gsum && ks >= 1 && ks % 2 == 1), "ff_loss_prepare: mask needs an odd Gaussian kernel, gconv and gsum");
    loss_prepare_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(
        flow_gt, valid, mask, gauss, ks, max_flow, vmap, gconv, gsum, B, H, W);
    return ff::check_launch("ff_loss_prepare");
}

extern "C" int ff_loss_accumulate(const float* pred, const float* flow_gt, const float* vmap, const float* gconv,
                                  const double* gsum, float a_mean, float lam, float weight, float* grad, double* loss,
                                  int B, int H, int W, void* stream) {
    FF_REQUIRE(pred && flow_gt && vmap && loss && B > 0 && H > 0 && W > 0, "ff_loss_accumulate: bad argument");
    FF_REQUIRE(!gconv || gsum, "ff_loss_accumulate: gconv needs gsum");
    loss_accum_kernel<<<grid_for((long long)B * 2 * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(
        pred, flow_gt, vmap, gconv, gsum, a_mean, lam, weight, grad, loss, B, H * W);
    return ff::check_launch("ff_loss_accumulate");
}

extern "C" int ff_epe_metric(const float* pred, const float* flow_gt, const float* vmap, double* out2, int B, int H, int W,
                             void* stream) {
    FF_REQUIRE(pred && flow_gt && vmap && out2 && B > 0 && H > 0 && W > 0, "ff_epe_metric: bad argument");
    epe_kernel<<<grid_for((long long)B * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(pred, flow_gt, vmap, out2, B, H * W);
    return ff::check_launch("ff_epe_metric");
}
