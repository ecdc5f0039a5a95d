// Direct 3x3 convolution for 1 or 2 output channels (flow head conv2 256 -> 2, update.py:13-14; SA's 2 -> 1
// spatial map; FF-PWC's flow estimators): on a 64-wide MFMA tile such a conv wastes 97 % of the matrix work
// (41 us for 0.23 GFLOP at 48x64x8).  Here it is a dot product on the vector ALU, exact fp32:
//
//   one wave = a run of RUN output pixels of one image row; lane = one group of 4 input channels (Cin <= 256 per
//   pass, more passes for wider inputs); the 3x3 weights of the lane's channels stay in registers for the run;
//   the input window slides along x, so each new pixel costs 3 coalesced float4 loads (1 KB per wave) instead
//   of 9; per pixel and output channel the 64 partial sums are folded with xor-shuffles.
#include "ff_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RUN = 8;

struct SArgs {
    FFConvParams p;
    int Cin, runs_x;
};

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); return fmaf(a[3], b[3], acc);
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_small_kernel(const SArgs a) {
    const FFConvParams& p = a.p;
    const int lane = threadIdx.x & 63;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b, y, run)
    const long long ntask = (long long)p.B * p.H * a.runs_x;
    if (task >= ntask) return;
    const int run = (int)(task % a.runs_x);
    const int y = (int)((task / a.runs_x) % p.H);
    const int b = (int)(task / ((long long)a.runs_x * p.H));
    const int x0 = run * RUN;
    const int H = p.H, W = p.W, K = 9 * a.Cin;

    float mine[COUT];                        // lane i (< RUN) collects pixel x0 + i
#pragma unroll
    for (int c = 0; c < COUT; ++c) mine[c] = 0.f;

    // channel passes: lane's 4 channels at cpos = pass*256 + lane*4 of the concatenated input
    for (int cbase = 0; cbase < a.Cin; cbase += 256) {
        const int cpos = cbase + lane * 4;
        const bool cok = cpos < a.Cin;
        // which segment holds cpos (segments are multiples of 4 channels, so a float4 never straddles two)
        const float* xp = p.x[0];            // lanes past the last channel load a valid dummy (segment 0, channel 0)
        int ld = p.x_ld[0], cs = 0;
        if (cok) {
            const int c0 = p.x_c[0], c01 = p.x_c[0] + p.x_c[1];
            cs = cpos;
            if (cs < c0) { xp = p.x[0]; ld = p.x_ld[0]; }
            else if (cs < c01) { xp = p.x[1]; ld = p.x_ld[1]; cs -= c0; }
            else { xp = p.x[2]; ld = p.x_ld[2]; cs -= c01; }
        }
        f32x4 w[COUT][9];
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                w[c][t] = cok ? *reinterpret_cast<const f32x4*>(p.w + (long long)c * K + t * a.Cin + cpos) : (f32x4){0.f, 0.f, 0.f, 0.f};
        // no load under a branch (the compiler would wait for each before issuing the next): out-of-image taps read a
        // clamped pixel and are zeroed by a select
        auto load = [&](int yy, int xx) {
            const bool ok = cok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + ((long long)(b * H + yc) * W + xc) * ld + cs);
            return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        f32x4 col[3][3];                     // col[slot][dy]: window columns x-1, x, x+1 rotate through 3 slots
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) { col[0][dy] = load(y - 1 + dy, x0 - 1); col[1][dy] = load(y - 1 + dy, x0); }
#pragma unroll
        for (int i = 0; i < RUN; ++i) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) col[(i + 2) % 3][dy] = load(y - 1 + dy, x0 + i + 1);
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                float s = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) s = dot4(col[(i + dx) % 3][dy], w[c][dy * 3 + dx], s);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);      // fold the 64 lanes right away
                if (lane == i) mine[c] += s;
            }
        }
    }
    const int x = x0 + lane;
    if (lane < RUN && x < W) {
        const long long m = ((long long)b * H + y) * W + x;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            float v = mine[c] + (p.bias ? p.bias[c] : 0.f);
            v *= p.out_scale;
            if (p.ch_scale) v = v * p.ch_scale[c] + p.ch_shift[c];
            v = ff::apply_act(v, p.act);
            if (p.res) v = ff::apply_act(v + p.res[m * p.res_ld + c], p.act_res);
            p.y[m * p.y_ld + c] = v;
            mine[c] = v;
        }
        if constexpr (COUT == 2) {
            if (p.ep_mode == FF_EP_COORDS) {      // raft.py:223 coords1 = coords1 + delta_flow, :219 flow = coords1 - coords0 (ff_coords_step's roundings)
                float* c1 = const_cast<float*>(p.ep_a) + m * 2;
                const float cx = __fadd_rn(c1[0], mine[0]), cy = __fadd_rn(c1[1], mine[1]);
                c1[0] = cx;
                c1[1] = cy;
                *reinterpret_cast<f32x4*>(const_cast<float*>(p.ep_b) + m * 4) = (f32x4){__fsub_rn(cx, (float)x), __fsub_rn(cy, (float)y), 0.f, 0.f};
            }
        }
    }
}

}  // namespace

namespace ff {
// FF_OK if launched, 1 if the shape is not this kernel's (fp32 rows, 3x3, stride 1, pad 1, Cout <= 2, groups 1)
int conv2d_fwd_small(const FFConvParams& p, int cin, hipStream_t s) {
    const int dlh = p.dil_h ? p.dil_h : 1, dlw = p.dil_w ? p.dil_w : 1;
    if (p.w_format != FF_W_F32 || p.Cout > 2 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1 ||
        dlh != 1 || dlw != 1 || p.groups != 1)
        return 1;
    SArgs a;
    a.p = p;
    a.Cin = cin;
    a.runs_x = (p.W + RUN - 1) / RUN;
    const long long tasks = (long long)p.B * p.H * a.runs_x;
    const unsigned blocks = (unsigned)((tasks + 3) / 4);
    if (p.Cout == 1) conv_small_kernel<1><<<blocks, 256, 0, s>>>(a);
    else conv_small_kernel<2><<<blocks, 256, 0, s>>>(a);
    return check_launch("ff_conv2d_fwd(small)");
}
}  // namespace ff

// ---------------------------------------------------------------------------------------------------------------------
// nn.ConvTranspose2d(Cin, Cout <= 2, kernel 4, stride 2, padding 1) - FF-PWC's netUpflow / netUpfeat
// (ff_pwcnet.py:243-244) - without the zero-dilated copy of the input and without a matrix tile: on the MFMA path the
// 640 -> 2 layer at 56 x 128 took 257 us (a 4x larger, three-quarters-zero input and 62 of 64 tile columns empty).
// Written as the equivalent forward convolution over the dilated input D (D[2i][2j] = x[i][j], 4 x 4 kernel Wf = the
// flipped, transposed parameter, pad 2): out[oy][ox] = sum_{a,b} D[oy-2+a][ox-2+b] Wf[a][b]; D is non-zero only at even
// coordinates, so an output pixel sees 2 x 2 taps: a = oy mod 2 (+2), input row (oy - 2 + a) / 2, likewise in x.
// One wave = a run of 8 output pixels of one output row (6 input columns x 2 input rows); lane = 4 input channels
// (256 channels per pass); the 2 x 4 taps of the row parity stay in registers; exact fp32.
namespace {

struct DArgs {
    const float* x; int x_ld;
    const float* w;          // [Cout][16 * Cin] fp32 rows, k = (a * 4 + b) * Cin + ci  (ff_pack_conv_weight of Wf)
    const float* bias;
    float* y; int y_ld;
    int B, H, W, Cin, runs_x;
};

template <int COUT>
__global__ __launch_bounds__(256) void deconv4x4s2_small_kernel(const DArgs a) {
    const int lane = threadIdx.x & 63;
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b, oy, run)
    if (task >= (long long)a.B * Ho * a.runs_x) return;
    const int run = (int)(task % a.runs_x);
    const int oy = (int)((task / a.runs_x) % Ho);
    const int b = (int)(task / ((long long)a.runs_x * Ho));
    const int ox0 = run * RUN, n0 = ox0 >> 1;               // RUN = 8 outputs = input columns n0 - 1 .. n0 + 4
    const int a0 = oy & 1;                                   // kernel rows a0, a0 + 2 -> input rows iy0, iy0 + 1
    const int iy0 = (oy - 2 + a0) >> 1;                      // arithmetic shift: -1 for oy = 0
    const int K = 16 * a.Cin;

    float mine[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) mine[c] = 0.f;
    for (int cbase = 0; cbase < a.Cin; cbase += 256) {
        const int cpos = cbase + lane * 4;
        const bool cok = cpos < a.Cin;
        f32x4 w[COUT][2][4];                                  // [co][row tap a0 + 2 r][b]
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
                    w[c][r][bb] = cok ? *reinterpret_cast<const f32x4*>(a.w + (long long)c * K + ((a0 + 2 * r) * 4 + bb) * a.Cin + cpos)
                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load = [&](int iy, int ix) {                     // no load under a branch: clamp, then select
            const bool ok = cok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int yc = min(max(iy, 0), a.H - 1), xc = min(max(ix, 0), a.W - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((long long)(b * a.H + yc) * a.W + xc) * a.x_ld + (cok ? cpos : 0));
            return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        f32x4 col[6][2];                                       // input columns n0 - 1 + j, rows iy0 + r
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 2; ++r) col[j][r] = load(iy0 + r, n0 - 1 + j);
#pragma unroll
        for (int i = 0; i < RUN; ++i) {                       // ox = ox0 + i: b0 = i & 1, input columns (ox - 2 + b0) / 2 (+1)
            const int b0 = i & 1, j0 = ((i - 2 + b0) >> 1) + 1;   // column slot of the first tap: (ox - 2 + b0) / 2 - (n0 - 1)
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) s = dot4(col[j0 + t][r], w[c][r][b0 + 2 * t], s);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if (lane == i) mine[c] += s;
            }
        }
    }
    const int ox = ox0 + lane;
    if (lane < RUN && ox < Wo) {
        const long long m = ((long long)b * Ho + oy) * Wo + ox;
#pragma unroll
        for (int c = 0; c < COUT; ++c) a.y[m * a.y_ld + c] = mine[c] + (a.bias ? a.bias[c] : 0.f);
    }
}

}  // namespace

extern "C" int ff_deconv4x4s2_small(const float* x, int x_ld, int B, int H, int W, int Cin, const float* w, const float* bias,
                                    int Cout, float* y, int y_ld, void* stream) {
    FF_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && (Cout == 1 || Cout == 2),
               "ff_deconv4x4s2_small: Cin must be a multiple of 4, Cout 1 or 2");
    FF_REQUIRE(x_ld >= Cin && x_ld % 4 == 0 && y_ld >= Cout && ff::aligned16(x) && ff::aligned16(w), "ff_deconv4x4s2_small: ld/alignment");
    DArgs a;
    a.x = x; a.x_ld = x_ld; a.w = w; a.bias = bias; a.y = y; a.y_ld = y_ld;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin;
    a.runs_x = (2 * W + RUN - 1) / RUN;
    const long long tasks = (long long)B * 2 * H * a.runs_x;
    const unsigned blocks = (unsigned)((tasks + 3) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cout == 1) deconv4x4s2_small_kernel<1><<<blocks, 256, 0, s>>>(a);
    else deconv4x4s2_small_kernel<2><<<blocks, 256, 0, s>>>(a);
    return ff::check_launch("ff_deconv4x4s2_small");
}
