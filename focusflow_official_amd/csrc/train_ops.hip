// Element-wise steps of the update block's BACKWARD as the recorded update loop runs it (focusflow_official_amd/train_loop.py):
// one autograd node for all iterations of raft.py:218-231, its backward a fixed sequence of launches.  The three kernels below
// are what lies between the input-gradient convolutions of a SepConvGRU pass (update.py:45-60): each reads every gradient once,
// applies the gate derivatives, ADDS the contributions that autograd used to sum with separate launches (the state h has three
// consumers per pass, the motion features four per iteration) and leaves max|g| of what the next convolution gradient reads
// (FFConvParams.x_amax).  All HBM-bound, 16-byte accesses.
#include "ff_common.h"
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float amax4(float mx, const f32x4 v) {
    return fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// one skip-if-not-larger atomicMax per block and word (same-address atomics serialise at the memory side)
__device__ __forceinline__ void block_amax(float mx, unsigned int* word, float* wmax /* [4] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (mx > 0.f && mx < INFINITY && __float_as_uint(mx) > *reinterpret_cast<volatile unsigned int*>(word))
            atomicMax(word, __float_as_uint(mx));
    }
    __syncthreads();
}

struct GruBwdArgs {
    // blend step                                     rh step                              out step
    const float* dh_in; int dh_in_ld;        //  d h' so far (next iteration / heads)       d h so far                  d h so far
    const float* dzc; int dzc_ld;            //  [d h | d motion] from the LATER pass's z|r dgrad (nullable)   [d rh | d motion] of the q dgrad     [d h | d motion] of pass 1's z|r dgrad
    float* dm; int dm_ld;                    //  motion gradient accumulator (+= dzc's second half)            (= or +=) second half                 read
    const float* a; int a_ld;                //  z                                         r                           motion (forward output, relu'd in channels < Cm)
    const float* b; int b_ld;                //  q                                         h                           -
    const float* c; int c_ld;                //  h                                         -                           -
    float* g0; int g0_ld;                    //  g_z -> G_zr[:, :C]                        g_r -> G_zr[:, C:]          g_motion (C channels, zeros from Cm on)
    float* g1; int g1_ld;                    //  g_q                                       -                           -
    float* dh_out; int dh_out_ld;            //  d h' (1 - z)                              d h + d rh * r              d h + dzc's first half
    unsigned int* amax0; unsigned int* amax1;
    unsigned npix; int C, Cm, dm_init;
};

// h' = (1 - z) h + z q ,  q = tanh(.), z = sigmoid(.):   g_z = d h' (q - h) z (1 - z) ;  g_q = d h' z (1 - q^2) ;  d h = d h' (1 - z)
__global__ __launch_bounds__(256) void gru_bwd_blend_kernel(const GruBwdArgs a) {
    __shared__ float wmax[4];
    const unsigned cg = (unsigned)a.C >> 2, total = a.npix * cg;
    float mz = 0.f, mq = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / cg, c4 = (i - p * cg) * 4;
        f32x4 d = *reinterpret_cast<const f32x4*>(a.dh_in + (size_t)p * a.dh_in_ld + c4);
        const f32x4 z = *reinterpret_cast<const f32x4*>(a.a + (size_t)p * a.a_ld + c4);
        const f32x4 q = *reinterpret_cast<const f32x4*>(a.b + (size_t)p * a.b_ld + c4);
        const f32x4 h = *reinterpret_cast<const f32x4*>(a.c + (size_t)p * a.c_ld + c4);
        if (a.dzc) {
            d += *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + c4);
            f32x4* dm = reinterpret_cast<f32x4*>(a.dm + (size_t)p * a.dm_ld + c4);
            *dm = *dm + *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + a.C + c4);
        }
        const f32x4 gz = d * (q - h) * (z * (1.f - z));
        const f32x4 gq = d * z * (1.f - q * q);
        *reinterpret_cast<f32x4*>(a.g0 + (size_t)p * a.g0_ld + c4) = gz;
        *reinterpret_cast<f32x4*>(a.g1 + (size_t)p * a.g1_ld + c4) = gq;
        *reinterpret_cast<f32x4*>(a.dh_out + (size_t)p * a.dh_out_ld + c4) = d * (1.f - z);
        mz = amax4(mz, gz);
        mq = amax4(mq, gq);
    }
    block_amax(mz, a.amax0, wmax);
    block_amax(mq, a.amax1, wmax);
}

// rh = r h, r = sigmoid(.):   g_r = d rh h r (1 - r) ;  d h += d rh r ;  d motion (+)= the q dgrad's motion half
__global__ __launch_bounds__(256) void gru_bwd_rh_kernel(const GruBwdArgs a) {
    __shared__ float wmax[4];
    const unsigned cg = (unsigned)a.C >> 2, total = a.npix * cg;
    float mr = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / cg, c4 = (i - p * cg) * 4;
        const f32x4 drh = *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + c4);
        const f32x4 dmo = *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + a.C + c4);
        const f32x4 r = *reinterpret_cast<const f32x4*>(a.a + (size_t)p * a.a_ld + c4);
        const f32x4 h = *reinterpret_cast<const f32x4*>(a.b + (size_t)p * a.b_ld + c4);
        const f32x4 dh = *reinterpret_cast<const f32x4*>(a.dh_in + (size_t)p * a.dh_in_ld + c4);
        f32x4* dm = reinterpret_cast<f32x4*>(a.dm + (size_t)p * a.dm_ld + c4);
        f32x4 m = dmo;
        if (!a.dm_init) m += *dm;
        const f32x4 gr = drh * h * (r * (1.f - r));
        *reinterpret_cast<f32x4*>(a.g0 + (size_t)p * a.g0_ld + c4) = gr;
        *reinterpret_cast<f32x4*>(a.dh_out + (size_t)p * a.dh_out_ld + c4) = dh + drh * r;
        *dm = m;
        mr = amax4(mr, gr);
    }
    block_amax(mr, a.amax0, wmax);
}

// behind pass 1's z|r dgrad: d h (this iteration's input state) complete; the motion features' gradient complete and through
// the relu of the motion encoder's last convolution (channels >= Cm are the flow, which carries no gradient: raft.py:220)
__global__ __launch_bounds__(256) void gru_bwd_out_kernel(const GruBwdArgs a) {
    __shared__ float wmax[4];
    const unsigned cg = (unsigned)a.C >> 2, total = a.npix * cg;
    float mm = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / cg, c4 = (i - p * cg) * 4;
        const f32x4 dh = *reinterpret_cast<const f32x4*>(a.dh_in + (size_t)p * a.dh_in_ld + c4);
        const f32x4 dzh = *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + c4);
        const f32x4 dzm = *reinterpret_cast<const f32x4*>(a.dzc + (size_t)p * a.dzc_ld + a.C + c4);
        const f32x4 dm = *reinterpret_cast<const f32x4*>(a.dm + (size_t)p * a.dm_ld + c4);
        const f32x4 mo = *reinterpret_cast<const f32x4*>(a.a + (size_t)p * a.a_ld + c4);
        f32x4 g = dm + dzm;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((int)c4 + j >= a.Cm || !(mo[j] > 0.f)) g[j] = 0.f;
        *reinterpret_cast<f32x4*>(a.dh_out + (size_t)p * a.dh_out_ld + c4) = dh + dzh;
        *reinterpret_cast<f32x4*>(a.g0 + (size_t)p * a.g0_ld + c4) = g;
        mm = amax4(mm, g);
    }
    block_amax(mm, a.amax0, wmax);
}

// dst[i] = sum_t src[t][i]: the gradient of a tensor every iteration of the loop reads (the context features' share of the gates)
__global__ __launch_bounds__(256) void sum_stack_kernel(const float* __restrict__ src, int T, long long n4, float* __restrict__ dst) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 s = reinterpret_cast<const f32x4*>(src)[i];
        for (int t = 1; t < T; ++t) s += reinterpret_cast<const f32x4*>(src)[(long long)t * n4 + i];
        reinterpret_cast<f32x4*>(dst)[i] = s;
    }
}

inline int blocks_for(long long items, int cap) { return (int)std::min<long long>(std::max<long long>((items + 255) / 256, 1), cap); }

inline bool ok16(const void* p, int ld) { return p && ff::aligned16(p) && ld % 4 == 0; }

}  // namespace

extern "C" int ff_gru_bwd_blend(const float* dh_in, int dh_in_ld, const float* dzc, int dzc_ld, float* dm, int dm_ld, const float* z, int z_ld,
                                const float* q, int q_ld, const float* h, int h_ld, float* gz, int gz_ld, float* gq, int gq_ld, float* dh_out,
                                int dh_out_ld, unsigned int* amax_zr, unsigned int* amax_q, long long npix, int C, void* stream) {
    FF_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && npix * (C / 4) < (1ll << 31) && amax_zr && amax_q, "ff_gru_bwd_blend: bad argument");
    FF_REQUIRE(ok16(dh_in, dh_in_ld) && ok16(z, z_ld) && ok16(q, q_ld) && ok16(h, h_ld) && ok16(gz, gz_ld) && ok16(gq, gq_ld) && ok16(dh_out, dh_out_ld) &&
               (!dzc || (ok16(dzc, dzc_ld) && dzc_ld >= 2 * C && ok16(dm, dm_ld))), "ff_gru_bwd_blend: 16-byte aligned NHWC tensors, lds % 4 == 0");
    GruBwdArgs a{};
    a.dh_in = dh_in; a.dh_in_ld = dh_in_ld; a.dzc = dzc; a.dzc_ld = dzc_ld; a.dm = dm; a.dm_ld = dm_ld;
    a.a = z; a.a_ld = z_ld; a.b = q; a.b_ld = q_ld; a.c = h; a.c_ld = h_ld;
    a.g0 = gz; a.g0_ld = gz_ld; a.g1 = gq; a.g1_ld = gq_ld; a.dh_out = dh_out; a.dh_out_ld = dh_out_ld;
    a.amax0 = amax_zr; a.amax1 = amax_q; a.npix = (unsigned)npix; a.C = C;
    gru_bwd_blend_kernel<<<blocks_for(npix * (C / 4), 1024), 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_gru_bwd_blend");
}

extern "C" int ff_gru_bwd_rh(const float* dqc, int dqc_ld, const float* r, int r_ld, const float* h, int h_ld, float* gr, int gr_ld,
                             float* dh, int dh_ld, float* dm, int dm_ld, int dm_init, unsigned int* amax_zr, long long npix, int C, void* stream) {
    FF_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && npix * (C / 4) < (1ll << 31) && amax_zr && dqc_ld >= 2 * C, "ff_gru_bwd_rh: bad argument");
    FF_REQUIRE(ok16(dqc, dqc_ld) && ok16(r, r_ld) && ok16(h, h_ld) && ok16(gr, gr_ld) && ok16(dh, dh_ld) && ok16(dm, dm_ld), "ff_gru_bwd_rh: 16-byte aligned NHWC tensors, lds % 4 == 0");
    GruBwdArgs a{};
    a.dzc = dqc; a.dzc_ld = dqc_ld; a.a = r; a.a_ld = r_ld; a.b = h; a.b_ld = h_ld; a.g0 = gr; a.g0_ld = gr_ld;
    a.dh_in = dh; a.dh_in_ld = dh_ld; a.dh_out = dh; a.dh_out_ld = dh_ld; a.dm = dm; a.dm_ld = dm_ld; a.dm_init = dm_init;
    a.amax0 = amax_zr; a.npix = (unsigned)npix; a.C = C;
    gru_bwd_rh_kernel<<<blocks_for(npix * (C / 4), 1024), 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_gru_bwd_rh");
}

extern "C" int ff_gru_bwd_out(const float* dh, int dh_ld, const float* dzc, int dzc_ld, const float* dm, int dm_ld, const float* motion, int mo_ld,
                              float* dh_out, int dh_out_ld, float* gm, int gm_ld, int Cm, unsigned int* amax_m, long long npix, int C, void* stream) {
    FF_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && Cm <= C && npix * (C / 4) < (1ll << 31) && amax_m && dzc_ld >= 2 * C, "ff_gru_bwd_out: bad argument");
    FF_REQUIRE(ok16(dh, dh_ld) && ok16(dzc, dzc_ld) && ok16(dm, dm_ld) && ok16(motion, mo_ld) && ok16(dh_out, dh_out_ld) && ok16(gm, gm_ld), "ff_gru_bwd_out: 16-byte aligned NHWC tensors, lds % 4 == 0");
    GruBwdArgs a{};
    a.dh_in = dh; a.dh_in_ld = dh_ld; a.dzc = dzc; a.dzc_ld = dzc_ld; a.dm = const_cast<float*>(dm); a.dm_ld = dm_ld; a.a = motion; a.a_ld = mo_ld;
    a.dh_out = dh_out; a.dh_out_ld = dh_out_ld; a.g0 = gm; a.g0_ld = gm_ld; a.amax0 = amax_m; a.npix = (unsigned)npix; a.C = C; a.Cm = Cm;
    gru_bwd_out_kernel<<<blocks_for(npix * (C / 4), 1024), 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_gru_bwd_out");
}

extern "C" int ff_sum_stack(const float* src, int T, long long n, float* dst, void* stream) {
    FF_REQUIRE(src && dst && T > 0 && n > 0 && n % 4 == 0 && ff::aligned16(src) && ff::aligned16(dst), "ff_sum_stack: bad argument");
    sum_stack_kernel<<<blocks_for(n / 4, 2048), 256, 0, static_cast<hipStream_t>(stream)>>>(src, T, n / 4, dst);
    return ff::check_launch("ff_sum_stack");
}
