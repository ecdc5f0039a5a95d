// init_mask modes other than 'point' (core/models/ff-raft/FF_RAFT_Core/ff_raft.py:23-72) fused with the
// input scaling (:142-145): one pass from the (B,1,H,W) key-point mask (+ image) to the NHWC4 tensor the
// condition branch consumes.
//   neighborG : m = conv2d(mask, G, pad k/2) ; m = m*255/max(m) ; 3 channels            (:57-66)
//   neighborE : m = (conv2d(mask/255, E, pad k/2) > 0) * 255 ; 3 channels               (:40-55)
//   context   : m = (conv2d(mask/255, E, pad k/2) > 0) * image                          (:24-30)
// G = get_kernel(KERNEL_SIZE, KERNEL_SIGMA) (:13-21), E = cv.getStructuringElement(MORPH_ELLIPSE, (k,k)) —
// both tables are built on the host and passed in.
// FF-PWC's init_mask (core/models/ff-pwcnet/PWCNet_Core/ff_pwcnet.py:61-110) is the same arithmetic on inputs that stay
// in [0,255]: mode bit 2 (FF_MASK_RAW) leaves the [0,255] -> [-1,1] scaling out, mode bit 3 (FF_MASK_IMAGE_NHWC4) reads
// the context image from an NHWC4 tensor (FF_PWCNET has already resized / repacked it).
#pragma clang fp contract(off)
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float scale255(float v) { return __fsub_rn(__fmul_rn(2.f, __fdiv_rn(v, 255.0f)), 1.0f); }

// phase 1: tmp[b,y,x] = sum_k table[k] * src(mask)[y+dy, x+dx]  (src = mask or mask/255), global max
__global__ void mask_conv_kernel(const float* __restrict__ mask, const float* __restrict__ table, int ks, float pre_div,
                                 float* __restrict__ tmp, unsigned int* __restrict__ gmax, int B, int H, int W) {
    const int HW = H * W, r = ks / 2;
    const long long total = (long long)B * HW;
    float lmax = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW;
        const int p = (int)(i - b * HW), y = p / W, x = p - y * W;
        float acc = 0.f;
        for (int dy = 0; dy < ks; ++dy) {
            const int yy = y + dy - r;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = 0; dx < ks; ++dx) {
                const int xx = x + dx - r;
                if ((unsigned)xx >= (unsigned)W) continue;
                const float m = mask[b * HW + yy * W + xx];
                if (m != 0.f) acc += (m / pre_div) * table[dy * ks + dx];
            }
        }
        tmp[i] = acc;
        lmax = fmaxf(lmax, acc);
    }
    // values are >= 0: float order == unsigned bit order
    atomicMax(gmax, __float_as_uint(lmax));
}

// phase 2: write the NHWC4 scaled tensor
__global__ void mask_finish_kernel(const float* __restrict__ tmp, const unsigned int* __restrict__ gmax,
                                   const float* __restrict__ image, int mode, int raw, int image_nhwc4, float* __restrict__ dst,
                                   int B, int HW) {
    const long long total = (long long)B * HW;
    const float mx = __uint_as_float(*gmax);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW, p = i - b * HW;
        f32x4 o;
        if (mode == 0) {                       // neighborG: m*255/max
            const float m = __fdiv_rn(__fmul_rn(tmp[i], 255.f), mx);
            const float v = raw ? m : scale255(m);
            o = (f32x4){v, v, v, 0.f};
        } else if (mode == 1) {                // neighborE: (dil > 0) * 255
            const float m = tmp[i] > 0.f ? 255.f : 0.f;
            const float v = raw ? m : scale255(m);
            o = (f32x4){v, v, v, 0.f};
        } else {                               // context: (dil > 0) * image
            const float k = tmp[i] > 0.f ? 1.f : 0.f;
            float s0, s1, s2;
            if (image_nhwc4) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(image + i * 4);
                s0 = s[0], s1 = s[1], s2 = s[2];
            } else {
                const float* s = image + b * 3 * HW + p;
                s0 = s[0], s1 = s[HW], s2 = s[2ll * HW];
            }
            o = raw ? (f32x4){k * s0, k * s1, k * s2, 0.f} : (f32x4){scale255(k * s0), scale255(k * s1), scale255(k * s2), 0.f};
        }
        *reinterpret_cast<f32x4*>(dst + i * 4) = o;
    }
}

}  // namespace

extern "C" int ff_mask_prepare(int mode, const float* mask, const float* image, const float* table, int ks, float* tmp,
                               unsigned int* gmax, float* dst_nhwc4, int B, int H, int W, void* stream) {
    const int raw = (mode >> 2) & 1, image_nhwc4 = (mode >> 3) & 1;
    mode &= 3;
    FF_REQUIRE(mode >= 0 && mode <= 2, "ff_mask_prepare: mode must be 0 (neighborG), 1 (neighborE) or 2 (context) [+ 4 raw, + 8 NHWC4 image]");
    FF_REQUIRE(mask && table && tmp && gmax && dst_nhwc4 && B > 0 && H > 0 && W > 0 && ks >= 1 && ks % 2 == 1,
               "ff_mask_prepare: bad argument (odd kernel size required)");
    FF_REQUIRE(mode != 2 || image, "ff_mask_prepare: context mode needs the image");
    FF_REQUIRE(ff::aligned16(dst_nhwc4), "ff_mask_prepare: dst alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long total = (long long)B * H * W;
    int g = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    (void)hipMemsetAsync(gmax, 0, sizeof(unsigned int), s);
    mask_conv_kernel<<<g, 256, 0, s>>>(mask, table, ks, mode == 0 ? 1.f : 255.f, tmp, gmax, B, H, W);
    mask_finish_kernel<<<g, 256, 0, s>>>(tmp, gmax, image, mode, raw, image_nhwc4, dst_nhwc4, B, H * W);
    return ff::check_launch("ff_mask_prepare");
}
