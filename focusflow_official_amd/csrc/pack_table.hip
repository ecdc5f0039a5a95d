// Weight layouts of MANY convolutions in one launch (training: every parameter changes every step, so every PackedConv of
// the model re-packs its forward rows, its bias vector and its input-gradient rows once per step - about 680 launches of
// ff_pack_conv_weight / ff_pack_split_f16 / ff_pack_conv_weight_dgrad / copies that took 3-5 us each on an otherwise idle
// chip), and the way back for the gradients of one PackedConv (packed dW rows + db -> OIHW gradients and bias gradients
// of its member convolutions, one launch instead of one per member and one per bias).
//
// Same numbers as the per-convolution entry points: forward rows [Cout][KH][KW][cin_pad], input-gradient rows
// [cin_pad][KH-1-kh][KW-1-kw][cout_pad], fp32 or split (h0 = f16(s v), h1 = f16(s v - h0): ff_pack_split_f16's arithmetic).
#include "ff_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// source channel of packed input channel ci (cin_slices of the host: the packed channels are the concatenation of the
// slices [lo, hi) of the parameter's input channels)
__device__ __forceinline__ int src_channel(const FFPackJob& J, int ci) {
    if (J.nslice == 0) return ci;
    int o = 0;
    for (int s = 0; s < J.nslice; ++s) {
        const int n = J.slice_hi[s] - J.slice_lo[s];
        if (ci < o + n) return J.slice_lo[s] + ci - o;
        o += n;
    }
    return -1;
}

__device__ __forceinline__ int member_of(const FFPackJob& J, int co) {
    int j = 0;
    for (int m = 1; m < J.nmem; ++m)
        if (co >= J.off[m]) j = m;
    return j;
}

__device__ __forceinline__ void store8(void* dst, int format, long long row_chunk, int oct, long long row, int K, int k0, const float (&v)[8]) {
    if (format == FF_W_F32) {
        float* d = static_cast<float*>(dst) + row * K + k0;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < K) d[e] = v[e];
        return;
    }
    h8 h0, h1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float sv = v[e] * ff::WSPLIT;
        h0[e] = (_Float16)sv;
        h1[e] = (_Float16)(sv - (float)h0[e]);
    }
    _Float16* d = static_cast<_Float16*>(dst) + row_chunk * 64 + oct * 8;
    *reinterpret_cast<h8*>(d) = h0;
    *reinterpret_cast<h8*>(d + 32) = h1;
}

__global__ __launch_bounds__(256) void pack_table_kernel(const FFPackJob* __restrict__ jobs, int njobs) {
    // the job of this block: the last one whose first block is <= blockIdx.x
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FFPackJob& J = jobs[lo];
    long long i = ((long long)blockIdx.x - J.block0) * 256 + threadIdx.x;
    const int khw = J.KH * J.KW;
    if (i < J.items_fwd) {                              // forward rows: 8 consecutive k of one row
        const int K = khw * J.cin_pad, nchunks = (K + 31) / 32;
        const int oct = (int)(i & 3);
        const long long rc = i >> 2;
        const int c = (int)(rc % nchunks);
        const int r = (int)(rc / nchunks);
        const int j = member_of(J, r);
        const float* w = J.w[j] + (long long)(r - J.off[j]) * J.cin_src * khw;
        const int k0 = c * 32 + oct * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            const int ci = k % J.cin_pad, t = k / J.cin_pad;         // t = kh * KW + kw
            const int cs = (k < K && ci < J.cin) ? src_channel(J, ci) : -1;
            v[e] = cs >= 0 ? w[(long long)cs * khw + t] : 0.f;
        }
        store8(J.fwd, J.fwd_format, rc, oct, r, K, k0, v);
        if (c == 0 && oct == 0 && J.bias_dst) J.bias_dst[r] = J.bias[j] ? J.bias[j][r - J.off[j]] : 0.f;
        return;
    }
    i -= J.items_fwd;
    if (i < J.items_dgrad) {                            // input-gradient rows: 8 consecutive (tap, output channel) of one input channel
        const int K = khw * J.cout_pad, nchunks = (K + 31) / 32;
        const int oct = (int)(i & 3);
        const long long rc = i >> 2;
        const int c = (int)(rc % nchunks);
        const int ci = (int)(rc / nchunks);
        const int cs = ci < J.cin ? src_channel(J, ci) : -1;
        const int k0 = c * 32 + oct * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            const int co = k % J.cout_pad, tf = k / J.cout_pad;      // tf = flipped tap
            float x = 0.f;
            if (k < K && cs >= 0 && co < J.cout) {
                const int j = member_of(J, co);
                x = J.w[j][((long long)(co - J.off[j]) * J.cin_src + cs) * khw + (khw - 1 - tf)];
            }
            v[e] = x;
        }
        store8(J.dgrad, J.dgrad_format, rc, oct, ci, K, k0, v);
    }
}

struct UnpackArgs {
    const float* dw;
    const float* db;
    float* dst;
    int nmem, cout[FF_PACK_MAX_MEMBERS], off[FF_PACK_MAX_MEMBERS], has_bias[FF_PACK_MAX_MEMBERS];
    long long dst0[FF_PACK_MAX_MEMBERS + 1];     // first element of member j in dst (weights, then its bias)
    int cin_src, nslice, slice_lo[FF_PACK_MAX_SLICES], slice_hi[FF_PACK_MAX_SLICES], KH, KW, cin_pad;
};

__global__ __launch_bounds__(256) void unpack_group_kernel(const UnpackArgs a) {
    const long long total = a.dst0[a.nmem];
    const int khw = a.KH * a.KW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int j = 0;
        for (int m = 1; m < a.nmem; ++m)
            if (i >= a.dst0[m]) j = m;
        const long long li = i - a.dst0[j], nw = (long long)a.cout[j] * a.cin_src * khw;
        if (li >= nw) {                                 // bias gradient
            a.dst[i] = a.db[a.off[j] + (int)(li - nw)];
            continue;
        }
        const int t = (int)(li % khw);
        const long long q = li / khw;
        const int cs = (int)(q % a.cin_src), co = (int)(q / a.cin_src);
        int ci = cs;                                    // packed channel of source channel cs (or none: gradient zero)
        if (a.nslice) {
            ci = -1;
            int o = 0;
            for (int s = 0; s < a.nslice; ++s) {
                if (cs >= a.slice_lo[s] && cs < a.slice_hi[s]) ci = o + cs - a.slice_lo[s];
                o += a.slice_hi[s] - a.slice_lo[s];
            }
        }
        a.dst[i] = ci >= 0 ? a.dw[((long long)(a.off[j] + co) * khw + t) * a.cin_pad + ci] : 0.f;
    }
}

}  // namespace

extern "C" int ff_pack_weights_table(const FFPackJob* jobs_dev, int njobs, long long total_blocks, void* stream) {
    FF_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "ff_pack_weights_table: bad argument");
    pack_table_kernel<<<(unsigned)total_blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(jobs_dev, njobs);
    return ff::check_launch("ff_pack_weights_table");
}

extern "C" int ff_pack_job_check(const FFPackJob* j) {
    FF_REQUIRE(j, "ff_pack_job_check: null job");
    FF_REQUIRE(j->nmem >= 1 && j->nmem <= FF_PACK_MAX_MEMBERS && j->nslice >= 0 && j->nslice <= FF_PACK_MAX_SLICES, "ff_pack_job_check: members / slices");
    FF_REQUIRE(j->KH > 0 && j->KW > 0 && j->cin > 0 && j->cin_pad >= j->cin && j->cin_pad % 4 == 0 && j->cin_src > 0 && j->cout > 0, "ff_pack_job_check: shape");
    int off = 0, sl = 0;
    for (int m = 0; m < j->nmem; ++m) {
        FF_REQUIRE(j->w[m] && j->off[m] == off && j->cout_m[m] > 0, "ff_pack_job_check: members must tile the output channels in order");
        off += j->cout_m[m];
    }
    FF_REQUIRE(off == j->cout, "ff_pack_job_check: member channels do not add up");
    for (int s = 0; s < j->nslice; ++s) {
        FF_REQUIRE(j->slice_lo[s] >= 0 && j->slice_hi[s] > j->slice_lo[s] && j->slice_hi[s] <= j->cin_src, "ff_pack_job_check: slice");
        sl += j->slice_hi[s] - j->slice_lo[s];
    }
    FF_REQUIRE(j->nslice ? sl == j->cin : j->cin == j->cin_src, "ff_pack_job_check: cin does not match the slices");
    const long long kf = (long long)j->KH * j->KW * j->cin_pad;
    FF_REQUIRE(j->items_fwd == 0 || (j->fwd && j->items_fwd == (long long)j->cout * ((kf + 31) / 32) * 4), "ff_pack_job_check: items_fwd");
    FF_REQUIRE(j->items_fwd == 0 || j->fwd_format == FF_W_F32 || ff::aligned16(j->fwd), "ff_pack_job_check: fwd rows not 16-byte aligned");
    if (j->items_dgrad) {
        FF_REQUIRE(j->dgrad && j->cout_pad >= j->cout && j->cout_pad % 4 == 0, "ff_pack_job_check: dgrad rows");
        const long long kd = (long long)j->KH * j->KW * j->cout_pad;
        FF_REQUIRE(j->items_dgrad == (long long)j->cin_pad * ((kd + 31) / 32) * 4, "ff_pack_job_check: items_dgrad");
        FF_REQUIRE(j->dgrad_format == FF_W_F32 || ff::aligned16(j->dgrad), "ff_pack_job_check: dgrad rows not 16-byte aligned");
    }
    FF_REQUIRE(j->block0 >= 0, "ff_pack_job_check: block0");
    return FF_OK;
}

extern "C" int ff_unpack_wgrad_group(const float* dw_packed, const float* db_packed, int nmem, const int* cout, const int* off,
                                     const int* has_bias, int cin_src, int nslice, const int* slice_lo, const int* slice_hi,
                                     int KH, int KW, int cin_pad, float* dst, void* stream) {
    FF_REQUIRE(dw_packed && dst && cout && off && has_bias, "ff_unpack_wgrad_group: null pointer");
    FF_REQUIRE(nmem >= 1 && nmem <= FF_PACK_MAX_MEMBERS && nslice >= 0 && nslice <= FF_PACK_MAX_SLICES && (nslice == 0 || (slice_lo && slice_hi)),
               "ff_unpack_wgrad_group: members / slices");
    FF_REQUIRE(cin_src > 0 && KH > 0 && KW > 0 && cin_pad > 0, "ff_unpack_wgrad_group: bad shape");
    UnpackArgs a;
    a.dw = dw_packed;
    a.db = db_packed;
    a.dst = dst;
    a.nmem = nmem;
    a.cin_src = cin_src;
    a.nslice = nslice;
    a.KH = KH;
    a.KW = KW;
    a.cin_pad = cin_pad;
    int cin = nslice ? 0 : cin_src;
    for (int s = 0; s < nslice; ++s) {
        FF_REQUIRE(slice_lo[s] >= 0 && slice_hi[s] > slice_lo[s] && slice_hi[s] <= cin_src, "ff_unpack_wgrad_group: slice");
        a.slice_lo[s] = slice_lo[s];
        a.slice_hi[s] = slice_hi[s];
        cin += slice_hi[s] - slice_lo[s];
    }
    FF_REQUIRE(cin <= cin_pad, "ff_unpack_wgrad_group: cin_pad too small");
    long long at = 0;
    for (int m = 0; m < nmem; ++m) {
        FF_REQUIRE(cout[m] > 0 && off[m] >= 0 && (!has_bias[m] || db_packed), "ff_unpack_wgrad_group: member");
        a.cout[m] = cout[m];
        a.off[m] = off[m];
        a.has_bias[m] = has_bias[m];
        a.dst0[m] = at;
        at += (long long)cout[m] * cin_src * KH * KW + (has_bias[m] ? cout[m] : 0);
    }
    a.dst0[nmem] = at;
    for (int m = nmem + 1; m <= FF_PACK_MAX_MEMBERS; ++m) a.dst0[m] = at;
    const long long blocks = (at + 255) / 256;
    unpack_group_kernel<<<(unsigned)(blocks > 2048 ? 2048 : blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    return ff::check_launch("ff_unpack_wgrad_group");
}
