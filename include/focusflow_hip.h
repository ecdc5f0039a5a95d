/*
 * focusflow_hip.h — C ABI of libfocusflow_hip.so, the MI355X (gfx950) native
 * implementation of FocusFlow's FF-RAFT hot path.
 *
 * The reference has no FFI of its own: its boundary is the torch.nn.Module API
 * (SURVEY.md §8b).  These entry points are what a Python maintainer would bind
 * with ctypes from the reference's modules; each one
 * names the reference lines it replaces, relative to
 * core/models/ff-raft/FF_RAFT_Core/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless said otherwise; fp32.
 *   - activations are NHWC: element (b,y,x,c) of a tensor with `ld` floats per
 *     pixel lives at  ptr[((b*H + y)*W + x)*ld + c]   (ld >= C lets a tensor be
 *     a channel slice of a wider buffer).  Pointers and `ld`s that feed a
 *     convolution input must be multiples of 4 floats (16-byte vector loads).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Calls only enqueue work; they never synchronise or allocate.
 *   - return value: 0 on success, negative FF_E* on failure; ff_last_error()
 *     returns a thread-local message for the last failure.
 */
#ifndef FOCUSFLOW_HIP_H
#define FOCUSFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FF_OK 0
#define FF_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define FF_EHIP (-2)     /* HIP runtime reported an error */

#define FF_ACT_NONE 0
#define FF_ACT_RELU 1
#define FF_ACT_SIGMOID 2
#define FF_ACT_TANH 3
#define FF_ACT_LEAKY 4      /* LeakyReLU(0.1), FF-PWC (ff_pwcnet.py:129 ff.) */

#define FF_MAX_SEG 3

/* weight formats of FFConvParams.w */
#define FF_W_F32 0     /* fp32 rows [Cout][K]: exact fp32 MFMA (v_mfma_f32_32x32x2_f32)            */
#define FF_W_F16X3 1   /* rows from ff_pack_split_f16: 16 w = w0 + w1 in fp16, 3 f16 MFMAs per       */
                       /* product term set - fp32-level accuracy at 5.3x the matrix rate           */
#define FF_W_F16 2     /* same rows, only the x0*w0 term: plain fp16 operands (reduced precision)  */

const char* ff_last_error(void);
/* The version this header describes; ff_abi_version() returns the library's.  A caller must compare the two (a struct
 * that grew since the caller was built would be read past its end) and must zero-initialise every FFConvParams it
 * passes: new trailing fields mean "feature off" when zero.
 *   2 (round 2): FFConvParams + res2, res2_ld, res_split, splitk_ws, splitk; ff_norm_bwd + dx_amax; ff_corr_lookup_bwd and
 *                ff_corr_pyramid_bwd (row-major) removed
 *   3 (round 3): FFConvParams + ep_mode, ep_split, ep_a, ep_a_ld, ep_b, ep_b_ld, stats_part; + ff_conv2d_stats_parts,
 *                ff_norm_stats_finish, ff_launch_timing_begin / _end, ff_mask_upsample_pack / _fwd,
 *                ff_corr_lookup_tiled_bwd_all
 *   5 (round 4): FFConvParams + x_fmt[], y_fmt, y_fmt_from, y2, y2_ld (the split-pair activation format between the
 *                layers of the update block); + ff_split_copy; FF_EP_MOTION_TAIL
 *   6 (round 4): entry points only: ff_fusion_pair_fwd / ff_fusion_pair_tile (FFFusionPair)
 *   7 (round 5): entry points only: ff_gru_bwd_blend / _rh / _out, ff_sum_stack, ff_upsample_flow_bwd_ex (the recorded
 *                update loop's backward), ff_gru_pass_rec */
#define FF_ABI_VERSION 7
int ff_abi_version(void);

/* Kernel-timestamp timing of one class of the library's launches (measurement only; bench.py's roofline uses it).
 * Between _begin and _end every launch of the class is made through hipExtLaunchKernelGGL with a start/stop event pair
 * bound to the dispatch, so the distance is the kernel's own execution time - what rocprofv3 --kernel-trace reports -
 * on whatever stream the caller launches on.  _end waits for the launches, returns their count, summed / shortest /
 * longest duration in microseconds (min_us / max_us may be NULL) and switches the timing off again.  Not for use
 * while a stream is being captured into a hipGraph. */
#define FF_TIME_LOOKUP 1        /* ff_corr_lookup_tiled_fwd's kernel */
#define FF_TIME_CORR_BUILD 2    /* ff_corr_build's kernel            */
#define FF_TIME_PROBE 3         /* ff_probe_memory_kernel's kernel   */
#define FF_TIME_KINDS 3
int ff_launch_timing_begin(int which);
int ff_launch_timing_end(int which, long long* launches, double* total_us, double* min_us, double* max_us);
/* Measurement aid (probe.hip): a memory-only kernel with the lookup's launch shape (`blocks` one-wave blocks) and access
 * pattern - per trip and wave five 1 KB loads from random aligned seg_bytes-segments of src (salt picks the addresses)
 * and four 1 KB stores to the wave's own stream in dst (dst_bytes >= blocks * trips * 4096 + 4).  Reports the bytes it
 * moves; time it with FF_TIME_PROBE.  What it reaches is the part's ceiling for a launch of that size. */
int ff_probe_memory_kernel(const void* src, long long src_bytes, void* dst, long long dst_bytes, int seg_bytes, int blocks,
                           int trips, unsigned int salt, long long* bytes_read, long long* bytes_written, void* stream);

/* ------------------------------------------------------------------------
 * Convolution (implicit GEMM on fp32 MFMA), replaces every nn.Conv2d call of
 * the path: extractor.py:48-56, parallel_fusion.py:87-95/:211-247,
 * update.py:13-14/:45-60/:89-97/:121-135 — and, with per-sample "weights",
 * the all-pairs matmul of corr.py:52-60.
 *
 * Input = up to 3 channel segments read as if concatenated (torch.cat along C
 * in the reference: update.py:46,54,95,128).  Output element:
 *     v = sum_k x*w ; v += bias[c] ; v *= out_scale ;
 *     v = v*ch_scale[c] + ch_shift[c] ; v = act(v) ; v += res ; v = act_res(v)
 * (null pointers skip their step; act_res only applies when res is given).
 * ---------------------------------------------------------------------- */
typedef struct FFConvParams {
    const float* x[FF_MAX_SEG];        /* input segments (NHWC)                         */
    int x_ld[FF_MAX_SEG];              /* floats per pixel of each segment buffer       */
    int x_c[FF_MAX_SEG];               /* channels taken from each segment (mult. of 4) */
    long long x_gstride[FF_MAX_SEG];   /* floats between groups (0 unless groups > 1)   */
    int groups;                        /* outer batch with its own weights (corr: B)    */
    int B, H, W;                       /* per-group input: B images of H x W            */
    const float* w;                    /* packed [Cout][KH][KW][Cin], Cin = sum x_c     */
    long long w_gstride;               /* floats between groups' weights (0 = shared)   */
    const float* bias;                 /* [Cout] or NULL                                */
    const float* ch_scale;             /* [Cout] or NULL                                */
    const float* ch_shift;             /* [Cout] or NULL (required iff ch_scale)        */
    float out_scale;
    const float* res;                  /* residual, NHWC [B*Ho*Wo][res_ld], or NULL     */
    int res_ld;
    float* y;                          /* output NHWC [B*Ho*Wo][y_ld]                   */
    int y_ld;
    long long y_gstride;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w;
    int act;                           /* FF_ACT_*, before the residual add             */
    int act_res;                       /* FF_ACT_*, after the residual add              */
    int w_format;                      /* FF_W_*; for the split formats w_gstride counts 4-byte words */
    int dil_h, dil_w;                  /* dilation (0 = 1): FF-PWC refiner, ff_pwcnet.py:350-364          */
    const unsigned int* x_amax;        /* NULL, or device word = bits of max|x| over the input (ff_act_bwd): the split
                                        * formats then read x * 2^k (k puts the maximum at 2^10) and undo it in the
                                        * epilogue - gradients lie far below fp16's range (dgrad on the f16 pipe)   */
    const float* in_scale;             /* NULL, or per (image, input channel) tables [B][Cin] from ff_norm_coeffs: the   */
    const float* in_shift;             /* convolution reads in_act(x * in_scale + in_shift) - the normalisation (+ ReLU) */
    int in_act;                        /* of its producer applied while loading, zero padding AFTER it.  One segment,    */
                                       /* groups == 1, split formats, stride-1 3x3 with Cin % 32 == 0 (patch kernel);    */
                                       /* in_act: FF_ACT_NONE or FF_ACT_RELU                                             */
    const float* res2;                 /* NULL, or a second residual for output channels >= res_split (column n reads    */
    int res2_ld;                       /* res2[m * res2_ld + n - res_split]): both 1x1 convs of a FusionUnit              */
    int res_split;                     /* (parallel_fusion.py:142-150: img' = img + conv(mask), mask' = mask + conv(img)) */
                                       /* as ONE launch over the segments [img, mask] with an anti-diagonal weight.       */
                                       /* Split formats, 1x1 kernels (the im2col kernel's epilogue).                      */
    float* splitk_ws;                  /* NULL, or a workspace of splitk * B*Ho*Wo * Cout floats: the reduction over the  */
    int splitk;                        /* input channels is cut into `splitk` ranges computed by separate blocks, summed   */
                                       /* in a fixed order by a second launch (deterministic).  Use ff_conv2d_splitk_hint: */
                                       /* it pays for small planes with long reductions (FF-PWC decoders, 7x16..56x128).   */
    int ep_mode;                       /* FF_EP_*: an element-wise step of SepConvGRU (update.py:45-60) applied to the      */
    int ep_split;                      /* finished output value v = act_res(act(conv) + res) before it is stored:           */
    const float* ep_a;                 /*   FF_EP_GRU_RH    channels n >= ep_split: v * ep_a[m][n - ep_split]  (r * h: the  */
    int ep_a_ld;                       /*                   z|r convolution writes [z | r*h], update.py:47-48)              */
    const float* ep_b;                 /*   FF_EP_GRU_BLEND (1 - ep_a[m][n]) * ep_b[m][n] + ep_a[m][n] * v  (the new state  */
    int ep_b_ld;                       /*                   from the q convolution: a = z, b = h, update.py:49)             */
                                       /* Same roundings as ff_gru_rh / ff_gru_blend.  Split-format stride-1 convolutions   */
                                       /* with Cin % 32 == 0 and a 3x3 / 1x5 / 5x1 kernel (the patch kernel); ep_a / ep_b    */
                                       /* NHWC, 16-byte aligned, ld % 4 == 0.                                               */
                                       /*   FF_EP_COORDS    the flow head's 2-channel 3x3 convolution (fp32 rows, update.py:13-14)*/
                                       /*                   also takes RAFT's coordinate step (raft.py:219-223): ep_a = coords1 */
                                       /*                   [B][H][W][2], READ AND WRITTEN through the pointer (coords1 += v),  */
                                       /*                   ep_b = flow4 [B][H][W][4], WRITTEN: (coords1 - pixel grid, 0, 0) -  */
                                       /*                   the two launches of ff_coords_step that follow the flow head.       */
    float* stats_part;                 /* NULL, or [B][parts][Cout][4] floats that receive partial statistics of the OUTPUT    */
                                       /* ({pivot, sum(v - pivot), sum((v - pivot)^2), count} per entry; parts =                */
                                       /* ff_conv2d_stats_parts(p) > 0): the convolution's epilogue replaces the ff_norm_stats  */
                                       /* pass over its output (extractor.py:48-56: every conv of the encoder is followed by a  */
                                       /* norm); ff_norm_stats_finish turns the parts into the {sum, sum of squares} table.    */
    /* --- split-pair activations (round 4) ------------------------------------------------------------------------------
     * A tensor in FF_FMT_SPLIT has the same shape, `ld` and size as its fp32 form, but every 32-channel chunk of a pixel
     * (128 bytes, chunk k at byte (pixel * ld + 32 k) * 4) holds [x0: 32 fp16 | x1: 32 fp16] with x0 = fp16(4 v),
     * x1 = fp16(4 v - x0): exactly what the convolution loaders make of an fp32 value on their way to LDS, so a convolution
     * that reads a split tensor computes bit for bit what it computes from the fp32 tensor - but its input patch travels
     * L2 -> LDS by LDS-DMA (conv_dma.hip) instead of load -> convert -> split -> ds_write through registers.  Element-wise
     * consumers keep fp32 (the GRU state is written in both forms: y and y2).                                              */
    int x_fmt[FF_MAX_SEG];             /* FF_FMT_F32 / FF_FMT_SPLIT per input segment (x_c % 32 == 0 for split segments)     */
    int y_fmt;                         /* FF_FMT_SPLIT: output channels >= y_fmt_from are written in the split format         */
    int y_fmt_from;                    /* (multiple of 32; the z|r convolution keeps z in fp32 and writes r*h split)          */
    float* y2;                         /* NULL, or a second copy of the WHOLE output in FF_FMT_SPLIT (the GRU's new state:    */
    int y2_ld;                         /* fp32 for the next blend, split for the next convolutions)                           */
    const void* w_frag;                /* NULL, or the same split weights as `w` in MFMA-fragment order (ff_pack_frag16): the  */
                                       /* split-pair kernel then loads a wave's 16-channel x 32-k operand as ONE contiguous KB */
} FFConvParams;
#define FF_FMT_F32 0
#define FF_FMT_SPLIT 1
#define FF_EP_NONE 0
#define FF_EP_GRU_RH 1
#define FF_EP_GRU_BLEND 2
#define FF_EP_COORDS 3
#define FF_EP_MOTION_TAIL 4   /* the motion encoder's last convolution (update.py:95-97, Cout = 126 into a 128-channel buffer) also
                               * writes torch.cat([out, flow])'s two flow channels: ep_a = coords1 [B][H][W][2] (read),
                               * flow = coords1 - pixel grid (raft.py:219) into output channels Cout, Cout + 1 */

int ff_conv2d_fwd(const FFConvParams* p, void* stream);

/* nn.Conv2d weight (OIHW, as stored in the state_dict) -> packed
 * [Cout][KH][KW][cin_pad] rows, written at row `cout_offset` of a destination
 * holding `dst_rows` rows (lets convz|convr share one packed matrix). */
int ff_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW,
                        float* dst, int cin_pad, int cout_offset, void* stream);
/* fp32 rows [rows][K] -> split rows [rows][ceil(K/32)]{x0: 32 fp16, x1: 32 fp16} (128 B per
 * 32-k chunk, zero padded) for FF_W_F16X3 / FF_W_F16: x0 = fp16(16 v), x1 = fp16(16 v - x0), both halves on the
 * same scale (one accumulator serves all three product terms).  Values must satisfy |v| < 4094; the convolutions'
 * activations (split the same way at scale 4 inside the kernels) |x| < 16376. */
int ff_pack_split_f16(const float* src_rows, void* dst, long long rows, int K, void* stream);
/* Split rows (ff_pack_split_f16: [rows][nkc][x0: 32 fp16 | x1: 32 fp16]) -> MFMA-fragment order for FFConvParams.w_frag:
 * [ceil(rows / 16)][nkc][term][lane 0..63][16 bytes], lane = 16 g + i holding k-group g (8 halfs) of row 16 tile + i -
 * the A operand of v_mfma_f32_16x16x32_f16 as it sits in registers; rows past the end are zero.
 * dst: ceil(rows / 16) * nkc * 2048 bytes, 16-byte aligned. */
int ff_pack_frag16(const void* split_rows, void* dst, int rows, int nkc, void* stream);
/* Number of K splits ff_conv2d_fwd would put to use for this convolution (0: none - leave splitk_ws NULL).  A plain
 * return value, not a status.  Does not launch anything. */
int ff_conv2d_splitk_hint(const FFConvParams* p);
/* Entries per (image, channel) that ff_conv2d_fwd writes to FFConvParams.stats_part for this convolution; 0 = it cannot
 * (run ff_norm_stats over the output).  A plain return value, not a status. */
int ff_conv2d_stats_parts(const FFConvParams* p);

/* ------------------------------------------------------------------------
 * Normalisation: nn.InstanceNorm2d (extractor.py:28-32, per-sample statistics,
 * no affine) and nn.BatchNorm2d (extractor.py:22-26; train = batch statistics).
 *   ff_norm_stats : stats[s][c] = {sum, sum of squares} in fp64, s = sample
 *                   (per_sample=1) or 0 (batch statistics).  `stats` must be
 *                   zeroed by the caller (hipMemsetAsync) before the call.
 *   ff_norm_apply : y = act((x-mean)*rstd*gamma + beta) ; if res: y = relu(y+res)
 *   ff_norm_coeffs: the (scale, shift) ff_norm_apply would use, as fp32 tables [S][C] (S = B if per_sample else 1),
 *                   for a consumer convolution that normalises while loading (FFConvParams.in_scale / in_shift)
 *   ff_bn_fold    : eval-mode BatchNorm as a per-channel scale/shift for the
 *                   convolution epilogue.
 *   ff_bn_update_running : running-stat update of a train-mode BatchNorm.
 * ---------------------------------------------------------------------- */
int ff_norm_stats(const float* x, int ld, int B, int HW, int C, int per_sample,
                  double* stats, void* stream);
/* stats[b][c] += {sum, sum of squares} (fp64; zero the table first, as for ff_norm_stats) from the partial entries a
 * convolution left in FFConvParams.stats_part ([B][parts][C][4]); per-sample statistics (InstanceNorm). */
int ff_norm_stats_finish(const float* parts, int B, int nparts, int C, double* stats, void* stream);
int ff_norm_apply(const float* x, int ld, float* y, int y_ld, int B, int HW, int C,
                  const double* stats, int per_sample, float eps,
                  const float* gamma, const float* beta, int act,
                  const float* res, int res_ld, void* stream);
int ff_norm_coeffs(const double* stats, int S, int C, long long count, float eps, const float* gamma, const float* beta,
                   float* scale, float* shift, void* stream);
int ff_bn_fold(const float* running_mean, const float* running_var, const float* gamma,
               const float* beta, float eps, float* ch_scale, float* ch_shift, int C, void* stream);
int ff_bn_update_running(const double* stats, long long count, float momentum,
                         float* running_mean, float* running_var, int C, void* stream);

/* ------------------------------------------------------------------------
 * Input preparation: ff_raft.py:31-38 ('point' masks: 1 -> 3 channels, second
 * mask = 255) and :142-145 (x -> 2*(x/255) - 1), NCHW [0,255] -> NHWC, 4
 * channels per pixel (channel 3 = 0).  src == NULL fills with `fill_value`
 * before scaling (mask2).
 * ---------------------------------------------------------------------- */
int ff_prep_input(const float* src_nchw, int src_c, float fill_value, float* dst_nhwc4,
                  int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------
 * CorrBlock (corr.py:12-60), row-major planes (any radius / level count; the exact-fp32 and plain-f16 conv precisions
 * and the tests use it - the product path is the tiled set below).  The volume itself is ff_conv2d_fwd with
 * groups = B (fmap2 as per-sample 1x1 weights, out_scale = 1/sqrt(C)).
 *   ff_corr_pyramid    : levels 1..3 by 2x2 average pooling (corr.py:24-27),
 *                        planes are [B*Q][h_l][w_l] row-major, floor on odd sizes.
 *   ff_corr_lookup_fwd : radius-r bilinear window lookup (corr.py:29-50 +
 *                        utils.py:57-71), out NHWC [B*Q][out_ld], channel
 *                        k = level*(2r+1)^2 + a*(2r+1) + b with a = x-offset index.
 *                        taps_dbg (nullable): int32 [B*Q][levels][2][2r+1]
 *                        floor indices (x then y) — used by the bit-exactness tests.
 * ---------------------------------------------------------------------- */
int ff_corr_pyramid(const float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                    long long planes, int h0, int w0, void* stream);
int ff_corr_lookup_fwd(const float* const* levels /* HOST array of 4 device ptrs */,
                       int num_levels, int radius, const float* coords /* [B*Q][2] x,y */,
                       long long queries, int h0, int w0, float* out, int out_ld,
                       int* taps_dbg, void* stream);

/* ------------------------------------------------------------------------
 * CorrBlock on a TILED pyramid - the product path (corr.py:12-60, utils.py:57-71).
 * Level l of query i (h_l x w_l values, h_l = h0 >> l) is a grid of 128-byte 2-D tiles:
 *     fp32 storage: tile = 8 wide x 4 high floats ; fp16 storage: tile = 8 wide x 8 high halfs ;
 *     element (y, x) at ((y / TH) * ntx + x / 8) * (8 * TH) + (y % TH) * 8 + (x % 8),
 *     ntx = ceil((ceil16(w0) >> l) / 8), nty = ceil((ceil8(h0) >> l) / TH).  Pad elements (inside the tile grid, beyond
 *     the plane) MUST BE ZERO: ff_corr_build and ff_corr_retile (into zero-initialised storage) guarantee it, the lookup
 *     relies on it (grid_sample's zero padding), the backward kernels keep it.
 *   ff_corr_plane_elems       elements per plane of `level` (the only non-status return value besides ff_abi_version)
 *   ff_corr_build             CorrBlock.__init__ (corr.py:12-27) + CorrBlock.corr (:52-60) in ONE launch: volume
 *                             <fmap1[i], fmap2[j]> / sqrt(C) on the f16 matrix pipe from operands pre-split by
 *                             ff_pack_split_f16 ([B*Q][C] fp32 rows -> [B*Q][4 C bytes]), the three avg_pool2d(2,2) levels
 *                             (ATen's summation order) from the accumulators, all four levels stored once.  half != 0:
 *                             every level is rounded to fp16 (RNE) and the next one is pooled from the ROUNDED values
 *                             in fp32, as torch autocast does.  C must be 256; levels = HOST array of 4 device pointers.
 *   ff_corr_retile            layout conversion of one level: row-major fp32 planes <-> tiled (fp32 / fp16) planes
 *   ff_corr_tile_rows         feature rows [B][Q][C] <-> rows in the tile order of a level-0 fp32 plane [B][P][C]
 *   ff_corr_lookup_tiled_fwd  CorrBlock.__call__ (corr.py:29-50), 4 levels, radius 4: out NHWC [B*Q][out_ld], channel
 *                             k = level*81 + a*9 + b with a = x-offset index; taps_dbg as in ff_corr_lookup_fwd
 *   ff_corr_lookup_tiled_bwd  d(out) scattered into tiled fp32 gradient planes (accumulates; pads stay zero)
 *   ff_corr_pyramid_tiled_bwd pooling backward chain in place on tiled fp32 planes; afterwards d0 = d(volume)
 * ---------------------------------------------------------------------- */
int ff_corr_plane_elems(int h0, int w0, int level, int half);
int ff_corr_build(const void* fmap1_split, const void* fmap2_split, void* const* levels /* HOST array */,
                  int B, int h0, int w0, int C, int half, void* stream);
int ff_corr_retile(float* rowmajor, void* tiled, long long planes, int h0, int w0, int level, int half,
                   int to_tiled, void* stream);
int ff_corr_tile_rows(const float* src, float* dst, int B, int h0, int w0, int C, int to_tiled, void* stream);
int ff_corr_lookup_tiled_fwd(const void* const* levels /* HOST array of 4 device ptrs */, int half,
                             const float* coords /* [B*Q][2] x,y */, long long queries, int h0, int w0,
                             float* out, int out_ld, int* taps_dbg, void* stream);
int ff_corr_lookup_tiled_bwd(float* const* dlevels /* HOST array */, const float* coords, const float* dout,
                             int dout_ld, long long queries, int h0, int w0, void* stream);
/* The lookup backward of ALL T iterations of a CorrBlock at once + the avg_pool2d backward chain: d0 [queries][plane_0]
 * (tiled fp32, every element written - no zero fill needed) = d(volume) from coords_list[t] ([queries][2]) and
 * dout_list[t] ([queries][dout_ld >= 324]), t < T <= 32 (both HOST arrays of device pointers).  One block per query
 * keeps its four gradient planes in LDS: returns 1 (nothing launched) when they do not fit 64 KB or T > 32 - then call
 * ff_corr_lookup_tiled_bwd per iteration and ff_corr_pyramid_tiled_bwd.  Replaces T launches of the former + one of the
 * latter + the zero fill (corr.py:29-50 backward, :24-27 backward). */
int ff_corr_lookup_tiled_bwd_all(float* d0, const float* const* coords_list, const float* const* dout_list, int T, int dout_ld,
                                 long long queries, int h0, int w0, void* stream);
int ff_corr_pyramid_tiled_bwd(float* d0, float* d1, float* d2, const float* d3, long long planes, int h0, int w0,
                              void* stream);

/* ------------------------------------------------------------------------
 * Update-block glue (raft.py:205-231, update.py:45-60).
 * ---------------------------------------------------------------------- */
/* dst[pix][0..C) = act(src[pix][0..C))  with independent lds (torch.split + tanh/relu) */
int ff_act_copy(const float* src, int src_ld, float* dst, int dst_ld, long long npix, int C,
                int act, void* stream);
/* fp32 -> FF_FMT_SPLIT (FFConvParams: x0 = fp16(4 v), x1 = fp16(4 v - x0) per 32-channel chunk) with an activation on the
 * way (the hidden state's tanh, raft.py:208).  C % 32 == 0, 16-byte aligned pointers, lds % 4 == 0.  to_split = 0 converts
 * back, v = (x0 + x1) / 4 (22 significant bits; act must be FF_ACT_NONE) - tests and debugging. */
int ff_split_copy(const float* src, int src_ld, float* dst, int dst_ld, long long npix, int C, int act, int to_split, void* stream);
/* The always-on range guard of the drop-in module: max|x| of an NHWC tensor (C % 4 == 0) as float bits into *word by
 * atomicMax (the caller zeroes the word; several probes may share it); a NaN or an infinity leaves +inf.  The fp16-split
 * conv formats need |x| < 16376 at every convolution input - the module probes its two encoder outputs once per forward
 * and raises instead of returning inf / NaN flow. */
int ff_range_probe(const float* x, int ld, long long npix, int C, unsigned int* word, void* stream);
/* coords_grid (utils.py:74-77): coords[b][y][x] = (x, y) (+ flow_init NHWC2 if given) */
int ff_coords_init(float* coords, const float* flow_init_nchw, int B, int H, int W, void* stream);
/* coords1 += delta (if delta) ; flow = coords1 - coords0 written to
 * flow4 [npix][4] (zero padded, conv input) and to motion[...,126:128] style
 * slot `slot` ([npix][slot_ld], 2 floats) if non-null.   raft.py:219,223 */
int ff_coords_step(float* coords1, const float* delta, int delta_ld, float* flow4,
                   float* slot, int slot_ld, int B, int H, int W, void* stream);
/* One pass of SepConvGRU (update.py:45-60) as one launch, inference on split-pair activations (FF_FMT_SPLIT):
 *   z | r = sigmoid(conv_zr([h, motion]) + zr_pre) ; q = tanh(conv_q([r * h, motion]) + q_pre) ; h' = (1 - z) h + z q
 * dir 0 = the (1,5) convolutions of pass 1, dir 1 = the (5,1) convolutions of pass 2.  hs / motion: split-pair, 128 channels;
 * h: the same state in fp32; zr_pre [..][256] / q_pre [..][128]: the loop-invariant context share of the gates (fp32, no
 * bias); wzr_frag / wq_frag: the packed [Cout][5 taps][256] weights of the two convolutions over [h, motion] in fragment
 * order (ff_pack_frag16); bzr [256], bq [128]; w_format FF_W_F16X3 / FF_W_F16.  Writes the new state twice: y fp32, y2
 * split-pair.  r * h and z never leave the CU.  Bit-identical to the two ff_conv2d_fwd launches with FF_EP_GRU_RH /
 * FF_EP_GRU_BLEND that it replaces. */
int ff_gru_pass(int dir, const float* hs, int hs_ld, const float* motion, int mo_ld, const float* h, int h_ld,
                const float* zr_pre, int zr_pre_ld, const float* q_pre, int q_pre_ld, const void* wzr_frag, const void* wq_frag,
                const float* bzr, const float* bq, int w_format, float* y, int y_ld, float* y2, int y2_ld, int B, int H, int W,
                void* stream);
/* The same for RECORDED passes (round 5: the update loop as one autograd node, train_loop.py): the pass also leaves what its
 * backward differentiates through - z, r and q = tanh(.) on the tile's pixels, fp32 [..][gate_ld >= 128] each (all three or
 * none) - next to the new state; r * h is their product with the incoming state and is never stored. */
int ff_gru_pass_rec(int dir, const float* hs, int hs_ld, const float* motion, int mo_ld, const float* h, int h_ld,
                    const float* zr_pre, int zr_pre_ld, const float* q_pre, int q_pre_ld, const void* wzr_frag, const void* wq_frag,
                    const float* bzr, const float* bq, int w_format, float* y, int y_ld, float* y2, int y2_ld, float* z, float* r,
                    float* q, int gate_ld, int B, int H, int W, void* stream);
/* ------------------------------------------------------------------------
 * One fusion unit of the Condition Control Encoder, type '1x1conv', both directions (parallel_fusion.py:98-150):
 *     y[0] = v0 + conv1x1(v1; w_frag[0]) + bias[0]        (img'  = img  + mask2img(mask))
 *     y[1] = v1 + conv1x1(v0; w_frag[1]) + bias[1]        (mask' = mask + img2mask(img))
 * in one launch that reads every input once and keeps the residual in registers (csrc/fusion_pair.hip).  Branch i's
 * value v_i is x[i] itself, or - LAZY inputs, scale[i] != NULL - what the normalisation pass in front of the unit would
 * have written (extractor.py:44-56, parallel_fusion.py:211-217):
 *     t = in_act(fma(x[i], scale[i][b][c], shift[i][b][c]))  ;  v_i = xres[i] ? relu(xres[i] + t) : t
 * with the [B][C] coefficient tables of ff_norm_coeffs - the operations of ff_norm_apply, bit for bit.
 * C = 64 or 96 channels per branch (ff_fusion_pair_tile(C) = pixels per tile, 0 = no instance: use ff_conv2d_fwd);
 * HW = pixels per image, a multiple of the tile; w_frag[i]: the C x C weights as split rows (ff_pack_conv_weight +
 * ff_pack_split_f16) re-ordered by ff_pack_frag16; bias[i]: [C] or NULL; w_format FF_W_F16X3 / FF_W_F16; all tensors
 * NHWC fp32, 16-byte aligned, leading dimensions multiples of 4.  An output may alias its OWN branch's x / xres
 * (every pixel is read before it is written), never the other branch's.
 * ---------------------------------------------------------------------- */
typedef struct FFFusionPair {
    const float* x[2];     int x_ld[2];
    const float* xres[2];  int xres_ld[2];
    const float* scale[2]; const float* shift[2];
    int in_act;                      /* FF_ACT_* applied to fma(x, scale, shift) */
    const void* w_frag[2];
    const float* bias[2];
    int w_format;
    float* y[2];           int y_ld[2];
    int B, HW, C;
} FFFusionPair;
int ff_fusion_pair_tile(int C);
int ff_fusion_pair_fwd(const FFFusionPair* p, void* stream);
/* GRU gates (update.py:47-49): rh = r*h ; h' = (1-z)*h + z*q */
int ff_gru_rh(const float* r, int r_ld, const float* h, int h_ld, float* rh, int rh_ld,
              long long npix, int C, void* stream);
int ff_gru_blend(const float* z, int z_ld, const float* q, int q_ld, const float* h, int h_ld,
                 float* h_new, int hn_ld, long long npix, int C, void* stream);
/* update.py:121-124 (mask[2], 1x1 256 -> 576) + the ".25 *" of update.py:133 + raft.py:159-170 (soft-max over the nine
 * neighbours, convex combination of the 8x flow) in ONE launch: the (B, H, W, 576) mask is never written.  hid: the mask
 * head's hidden tensor (B, H, W, 256 channels, leading dimension hid_ld); w_stage: the 1x1 weights - split rows
 * (ff_pack_conv_weight + ff_pack_split_f16, w_format FF_W_F16X3 or FF_W_F16) rearranged once per weight version by
 * ff_mask_upsample_pack into the kernel's stage-major image (576 * 1024 bytes, like the split rows); bias [576] or NULL;
 * flow (B, H, W, >= 2); out (B, 2, 8H, 8W).  Same arithmetic as ff_conv2d_fwd + ff_upsample_flow up to the summation
 * order of the products and the soft-max's v_exp_f32 / single reciprocal. */
int ff_mask_upsample_pack(const void* w_split, void* w_stage, void* stream);
int ff_mask_upsample_fwd(const float* hid, int hid_ld, const void* w_stage, int w_format, const float* bias, float out_scale,
                         const float* flow, int flow_ld, float* out, int B, int H, int W, void* stream);
/* convex 8x upsampling (raft.py:159-170): flow NHWC [B*H*W][flow_ld] (2 ch),
 * mask NHWC [B*H*W][mask_ld] (576 ch) -> out NCHW (B,2,8H,8W) */
int ff_upsample_flow(const float* flow, int flow_ld, const float* mask, int mask_ld,
                     float* out_nchw, int B, int H, int W, void* stream);
/* NHWC [npix][ld] (C ch) -> NCHW (B,C,H,W) */
int ff_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int H, int W, int C, void* stream);

/* ========================================================================
 * Backward (autograd of raft.py:173-236).
 * ======================================================================== */
/* dW[co][kh][kw][ci] (packed layout, fp32 atomics, CALLER ZEROES dw) =
 *   out_scale * sum_pixels dY[pix][co] * x[pix @ (kh,kw)][ci].
 * `p` describes the forward conv (x segments, geometry); p->y / p->y_ld carry dY
 * (channels rounded up to a multiple of 4).  groups = B with a 1x1 kernel gives
 * d(corr volume)/d(fmap2) (BmmBackward of corr.py:58). */
/* p->w_format = FF_W_F16X3 / FF_W_F16 (groups == 1) runs it on the f16 matrix pipe with split operands; dY is
 * then scaled by the power of two derived from p->x_amax (= bits of max|dY|, from ff_act_bwd), and `db`
 * (nullable, CALLER ZEROES) receives the bias gradient sum_pixels dY[pix][co].  db must be NULL for FF_W_F32. */
int ff_conv2d_wgrad(const FFConvParams* p, float* dw, long long dw_gstride, float* db, void* stream);
/* packed dW rows [cout_offset, cout_offset+Cout) -> OIHW gradient of one nn.Conv2d */
int ff_unpack_conv_wgrad(const float* packed, int Cout, int Cin, int KH, int KW, int cin_pad,
                         int cout_offset, float* dw_oihw, void* stream);
/* input gradient = ff_conv2d_fwd over dY with these weights:
 * dst[ci][KH-1-kh][KW-1-kw][cout_offset+co] = w[co][ci][kh][kw]  (CALLER ZEROES dst) */
int ff_pack_conv_weight_dgrad(const float* w_oihw, int Cout, int Cin, int KH, int KW, float* dst,
                              int cout_pad, int cout_offset, void* stream);
/* ---- every weight layout of a training step in ONE launch (pack_table.hip) ----
 * A training step changes every parameter, so every convolution re-packs its forward rows (ff_pack_conv_weight +
 * ff_pack_split_f16), its bias vector and its input-gradient rows (ff_pack_conv_weight_dgrad + ff_pack_split_f16) once
 * per step: ~680 launches of 3-5 us each for FF-RAFT.  One FFPackJob describes all of that for one packed convolution
 * (up to FF_PACK_MAX_MEMBERS nn.Conv2d concatenated along Cout - the GRU's z and r gates, the flow and mask heads -
 * optionally over slices of their input channels); the table lives in DEVICE memory and is reused as long as no pointer
 * in it changes.  Results are bit-identical to the per-convolution entry points. */
#define FF_PACK_MAX_MEMBERS 4
#define FF_PACK_MAX_SLICES 4
typedef struct FFPackJob {
    const float* w[FF_PACK_MAX_MEMBERS];      /* OIHW parameters [cout_m][cin_src][KH][KW]                              */
    const float* bias[FF_PACK_MAX_MEMBERS];   /* nullable per member                                                   */
    int cout_m[FF_PACK_MAX_MEMBERS], off[FF_PACK_MAX_MEMBERS];   /* member channels and their first packed output row  */
    int nmem, cout;                           /* cout = sum of cout_m                                                  */
    int cin_src;                              /* input channels of the parameters                                      */
    int nslice, slice_lo[FF_PACK_MAX_SLICES], slice_hi[FF_PACK_MAX_SLICES];   /* packed channels = these ranges of the
                                                 source channels, concatenated (nslice 0: all of them)                  */
    int cin, cin_pad, KH, KW;                 /* cin = packed input channels (sum of the slices), padded to cin_pad      */
    void* fwd;                                /* forward rows [cout][KH*KW*cin_pad] in fwd_format (FF_W_*), or unused    */
    float* bias_dst;                          /* [cout] bias vector (0 where a member has none), nullable               */
    void* dgrad;                              /* input-gradient rows [cin_pad][KH*KW*cout_pad] in dgrad_format, or unused */
    int fwd_format, dgrad_format, cout_pad, reserved;
    long long items_fwd;                      /* cout * ceil(KH*KW*cin_pad / 32) * 4, or 0: no forward rows              */
    long long items_dgrad;                    /* cin_pad * ceil(KH*KW*cout_pad / 32) * 4, or 0: no input-gradient rows   */
    long long block0;                         /* first block of this job: sum over the earlier jobs of
                                                 ceil((items_fwd + items_dgrad) / 256)                                  */
} FFPackJob;
/* host-side validation of one job (the table itself is device memory: the launch cannot check it) */
int ff_pack_job_check(const FFPackJob* job);
/* jobs_dev: njobs jobs in device memory, ordered by block0; total_blocks = block0 + blocks of the last job */
int ff_pack_weights_table(const FFPackJob* jobs_dev, int njobs, long long total_blocks, void* stream);
/* The way back for one packed convolution: packed dW rows (+ db) -> dst = for every member, its OIHW gradient
 * [cout_m][cin_src][KH][KW] (zero outside the slices) followed, if has_bias[m], by its bias gradient [cout_m]. */
int ff_unpack_wgrad_group(const float* dw_packed, const float* db_packed, int nmem, const int* cout, const int* off,
                          const int* has_bias, int cin_src, int nslice, const int* slice_lo, const int* slice_hi,
                          int KH, int KW, int cin_pad, float* dst, void* stream);
/* g = dy * act'(y) * scale (activation derivative from the forward OUTPUT), zero-padded to Cpad; amax (nullable,
 * CALLER ZEROES): bits of max|g|.  g == dy is allowed (then nothing is stored: the call only measures max|dy|, for
 * activation-free convolutions whose gradient needs no copy) */
int ff_act_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float* g, int g_ld,
               long long npix, int C, int Cpad, int act, float scale, unsigned int* amax, void* stream);
/* zero-dilation by 2: dst[b][2y][2x][:] = src[b][y][x][:] (input gradient of stride-2 convs) */
int ff_dilate2(const float* src, int src_ld, float* dst, int B, int Ho, int Wo, int Hd, int Wd, int C,
               void* stream);
/* nn.ConvTranspose2d(Cin, Cout <= 2, kernel 4, stride 2, padding 1) as a direct fp32 kernel: FF-PWC's netUpflow /
 * netUpfeat (ff_pwcnet.py:243-244, called at :283-284).  x: NHWC [B][H][W][x_ld] (Cin % 4 == 0), y: [B][2H][2W][y_ld];
 * w: the fp32 rows ff_pack_conv_weight makes of the EQUIVALENT forward convolution's weight (the parameter transposed to
 * [Cout][Cin][4][4] and flipped in both kernel axes): [Cout][16 * Cin], k = (a * 4 + b) * Cin + ci.  Replaces
 * ff_dilate2 + ff_conv2d_fwd for these layers (same result to fp32 rounding). */
int ff_deconv4x4s2_small(const float* x, int x_ld, int B, int H, int W, int Cin, const float* w, const float* bias,
                         int Cout, float* y, int y_ld, void* stream);
/* Instance/BatchNorm backward for y = relu?(norm(x)) [; y = relu(y + res)].
 * bstats (fp64 [S][C][2], CALLER ZEROES) returns {sum g, sum g*xhat} = {dbeta, dgamma}.
 * dx_amax (nullable, a zeroed word): receives the bits of max|dx| - what ff_act_bwd would measure in a pass of its own
 * before the producing convolution's gradient kernels (FFConvParams.x_amax). */
int ff_norm_bwd(const float* x, int x_ld, const float* dy, int dy_ld, const float* y, int y_ld,
                const double* fstats, double* bstats, int per_sample, int fixed_stats, float eps,
                const float* gamma, const float* beta, int relu, float* dx, int dx_ld,
                float* dres, int dres_ld, int B, int HW, int C, unsigned int* dx_amax, void* stream);
/* GridSampler2DBackward (w.r.t. the pyramid only: coords are detached, raft.py:216) and the AvgPool2DBackward chain:
 * ff_corr_lookup_tiled_bwd / ff_corr_pyramid_tiled_bwd above (gradient planes share the tiled fp32 layout). */
int ff_gru_rh_bwd(const float* drh, int drh_ld, const float* r, int r_ld, const float* h, int h_ld,
                  float* dr, int dr_ld, float* dh, int dh_ld, long long npix, int C, void* stream);
int ff_gru_blend_bwd(const float* dhn, int dhn_ld, const float* z, int z_ld, const float* q, int q_ld,
                     const float* h, int h_ld, float* dz, int dz_ld, float* dq, int dq_ld,
                     float* dh, int dh_ld, long long npix, int C, void* stream);
/* convex upsampling backward: dflow NHWC [B*H*W][2] (CALLER ZEROES), dmask NHWC [B*H*W][576] */
int ff_upsample_flow_bwd(const float* dout_nchw, const float* flow, int flow_ld, const float* mask,
                         int mask_ld, float* dflow, float* dmask, int B, int H, int W, void* stream);
/* The same for the recorded update loop (below): dflow with its own leading dimension (the 4-channel gradient tensor the flow
 * head's convolution gradients read; CALLER ZEROES), dmask multiplied by mask_scale on the way out (the ".25 *" of
 * update.py:133) and max|dmask| left in *dmask_amax (nullable; a zeroed word - FFConvParams.x_amax of the mask head's
 * gradient convolutions). */
int ff_upsample_flow_bwd_ex(const float* dout_nchw, const float* flow, int flow_ld, const float* mask, int mask_ld,
                            float* dflow, int dflow_ld, float* dmask, float mask_scale, unsigned int* dmask_amax,
                            int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------
 * Backward of SepConvGRU (update.py:45-60) between its input-gradient convolutions, as the recorded update loop issues it
 * (one autograd node for all iterations of raft.py:218-231; focusflow_official_amd/train_loop.py).  Per pass, in backward order:
 *   ff_gru_bwd_blend   d h' (+ the h half of a [d h | d motion] tensor from the LATER pass's z|r input gradient, whose motion
 *                      half is added to the motion accumulator dm) -> g_z = d h' (q - h) z (1 - z), g_q = d h' z (1 - q^2),
 *                      dh_out = d h' (1 - z)         [h' = (1 - z) h + z q]
 *   (input gradient of the q convolution over [r h, motion]  ->  dqc = [d rh | d motion])
 *   ff_gru_bwd_rh      g_r = d rh h r (1 - r) ; dh += d rh r ; dm = (dm_init ? 0 : dm) + d motion
 *   (input gradient of the z|r convolution over [h, motion]  ->  dzc = [d h | d motion])
 * and once per iteration, behind pass 1:
 *   ff_gru_bwd_out     dh_out = dh + dzc[:, :C] ; g_motion = (dm + dzc[:, C:]) * [motion > 0] in channels < Cm, 0 from Cm on
 *                      (the motion encoder's last convolution has Cm = 126 outputs + relu; channels 126, 127 are the flow,
 *                      which carries no gradient: raft.py:220)
 * Every g leaves max|g| in a zeroed device word (atomicMax of the float bits): FFConvParams.x_amax of the gradient convolutions
 * that read it.  g_z and g_r are the two halves of ONE tensor (the z|r convolution's output gradient) and share amax_zr.
 * All tensors NHWC fp32, 16-byte aligned, leading dimensions multiples of 4; in-place where the names coincide.
 *   ff_sum_stack       dst[i] = sum_t src[t * n + i], t < T (n % 4 == 0): gradient of a tensor every iteration reads
 * ---------------------------------------------------------------------- */
int ff_gru_bwd_blend(const float* dh_in, int dh_in_ld, const float* dzc, int dzc_ld, float* dm, int dm_ld, const float* z, int z_ld,
                     const float* q, int q_ld, const float* h, int h_ld, float* gz, int gz_ld, float* gq, int gq_ld, float* dh_out,
                     int dh_out_ld, unsigned int* amax_zr, unsigned int* amax_q, long long npix, int C, void* stream);
int ff_gru_bwd_rh(const float* dqc, int dqc_ld, const float* r, int r_ld, const float* h, int h_ld, float* gr, int gr_ld,
                  float* dh, int dh_ld, float* dm, int dm_ld, int dm_init, unsigned int* amax_zr, long long npix, int C, void* stream);
int ff_gru_bwd_out(const float* dh, int dh_ld, const float* dzc, int dzc_ld, const float* dm, int dm_ld, const float* motion, int mo_ld,
                   float* dh_out, int dh_out_ld, float* gm, int gm_ld, int Cm, unsigned int* amax_m, long long npix, int C, void* stream);
int ff_sum_stack(const float* src, int T, long long n, float* dst, void* stream);

/* ========================================================================
 * FF-PWC native component (core/models/ff-pwcnet/PWCNet_Core/): the 81-channel cost volume
 * of correlation.py:7-232 (the reference's CuPy CUDA kernels) and backwarp, ff_pwcnet.py:27-47.
 * NHWC fp32, C a multiple of 4.
 *   fwd : out[b,y,x,(p+4)*9+(o+4)] = 1/C sum_c one[b,y,x,c] * two[b,y+p,x+o,c], p,o in [-4,4]
 *   bwd : grad[b,y,x,c] = 1/C sum_{p,o} g[b,y,x,(p,o)] * other[b,y+p,x+o,c]
 *         gradOne = bwd(gOut, two) ; gradTwo = bwd(ff_pwc_gout_transpose(gOut), one)
 * ======================================================================== */
int ff_pwc_costvolume_fwd(const float* one, int one_ld, const float* two, int two_ld, float* out,
                          int out_ld, int B, int H, int W, int C, void* stream);
/* The same with the activation that follows the volume everywhere (leaky_relu, ff_pwcnet.py:289) applied on the way out, and -
 * splits > 1, ws = splits * B*H*W * 81 floats - the channel reduction cut into ranges computed by separate blocks and
 * added in a fixed order by a second launch: the coarse levels are a handful of tiles with 96-196 channels. */
int ff_pwc_costvolume_fwd_ex(const float* one, int one_ld, const float* two, int two_ld, float* out, int out_ld,
                             int B, int H, int W, int C, int act, float* ws, int splits, void* stream);
int ff_pwc_costvolume_bwd(const float* g, int g_ld, const float* other, int other_ld, float* grad,
                          int grad_ld, int B, int H, int W, int C, void* stream);
int ff_pwc_gout_transpose(const float* g, int g_ld, float* gt, int gt_ld, int B, int H, int W, void* stream);
/* out = grid_sample(in, grid + flow, bilinear, zeros, align_corners=False) * (warped ones > 0.999) */
int ff_pwc_backwarp(const float* in, int in_ld, const float* flow, int flow_ld, float flow_scale,
                    float* out, int out_ld, int B, int H, int W, int C, void* stream);
/* its backward (GridSampler2DBackward times the validity mask, whose own gradient is zero: ff_pwcnet.py:45 overwrites
 * it with constants): din (nullable, CALLER ZEROES, fp32 atomics) and dflow (nullable, channels 0/1 written) */
int ff_pwc_backwarp_bwd(const float* in, int in_ld, const float* flow, int flow_ld, float flow_scale, const float* gout,
                        int gout_ld, float* din, int din_ld, float* dflow, int dflow_ld, int B, int H, int W, int C,
                        void* stream);

/* ========================================================================
 * Fused sequence loss (core/models/ff-raft/losses/losses.py:18-130: EPELoss, CPCL, MixLoss).
 * All tensors NCHW fp32 as the reference's loss receives them: preds/flow_gt (B,2,H,W),
 * valid (B,H,W), mask (B,1,H,W).
 *   ff_loss_prepare    vmap = (valid>=0.5)&(|gt|<max_flow); gconv = G*(mask>0) (zero padded),
 *                      *gsum += sum(gconv)  (fp64, CALLER ZEROES); mask == NULL skips the G part
 *   ff_loss_accumulate *loss += weight * sum w*|pred-gt|, w = vmap*(a_mean + lam*gconv/gsum);
 *                      grad (nullable) = weight * w * sign(pred-gt)     (*loss fp64, CALLER ZEROES)
 *   ff_epe_metric      out2 += {sum of end-point errors over vmap, count}
 * ======================================================================== */
int ff_loss_prepare(const float* flow_gt, const float* valid, const float* mask, const float* gauss, int ks,
                    float max_flow, float* vmap, float* gconv, double* gsum, int B, int H, int W, void* stream);
int ff_loss_accumulate(const float* pred, const float* flow_gt, const float* vmap, const float* gconv,
                       const double* gsum, float a_mean, float lam, float weight, float* grad, double* loss,
                       int B, int H, int W, void* stream);
int ff_epe_metric(const float* pred, const float* flow_gt, const float* vmap, double* out2, int B, int H, int W,
                  void* stream);

/* FF-PWC multi-scale losses (core/models/ff-pwcnet/losses/losses.py:19-261: EPELoss / CPCL / MixLoss, dense ground truth).
 * All tensors NCHW fp32: out (B,2,h,w) one pyramid level, target (B,2,H,W), mask (B,1,H,W), gmask (B,h,w).
 *   ff_pwc_loss_mask   gmask = G * (bilinear(mask -> h x w, align_corners=False) > 0), zero padded (:107-112, :187-191);
 *                      *msum (fp64, CALLER ZEROES) += sum gmask
 *   ff_pwc_loss_scale  target area-interpolated to h x w (:66, :149); E = |t-o|_2 (l1q = 0, 'pretrain') or (|t-o|_1 + eps)^q;
 *                      *loss (fp64, CALLER ZEROES) += sum (w_plain + w_mask * gmask) * E, w_mask = w_mask_num / *msum
 *                      (0 if *msum == 0 and zero_if_empty: MixLoss :182-183); mask_over_batch: gmask summed over the batch
 *                      for every sample (CPCL's (B,h,w) x (B,1,h,w) broadcast, :114); grad (nullable) = d(that)/d(out)
 *   ff_pwc_epe_mean    out2 (fp64, CALLER ZEROES) += { sum of E over all pixels of (B,2,H,W) pred vs target, pixel count } */
int ff_pwc_loss_mask(const float* mask, const float* gauss, int ks, float* gmask, double* msum, int B, int H, int W, int h,
                     int w, void* stream);
int ff_pwc_loss_scale(const float* out, const float* target, const float* gmask, const double* msum, float w_plain,
                      float w_mask_num, int zero_if_empty, int mask_over_batch, int l1q, float eps, float q, float* grad,
                      double* loss, int B, int H, int W, int h, int w, void* stream);
int ff_pwc_epe_mean(const float* pred, const float* target, int l1q, float eps, float q, double* out2, int B, int H, int W,
                    void* stream);
/* Sparse ground truth (KITTI stage, ff-pwcnet/train.py:287-312 passes sparse=True; losses.py:28-41, :58-67, :186-214):
 *   ff_pwc_loss_scale_sparse  as ff_pwc_loss_scale with the target down-sampled by sparse_max_pool (:44-57) and pixels
 *                             whose pooled target is exactly (0, 0) invalid: the key-point term counts valid pixels only;
 *                             plain_valid_only != 0: so does the plain term (EPELoss; MixLoss keeps it over every pixel)
 *   ff_pwc_epe_mean_sparse    out2 += { sum of E over pixels whose target is not exactly (0, 0), their count } */
int ff_pwc_loss_scale_sparse(const float* out, const float* target, const float* gmask, const double* msum, float w_plain,
                             float w_mask_num, int zero_if_empty, int plain_valid_only, int l1q, float eps, float q, float* grad,
                             double* loss, int B, int H, int W, int h, int w, void* stream);
int ff_pwc_epe_mean_sparse(const float* pred, const float* target, int l1q, float eps, float q, double* out2, int B, int H, int W,
                           void* stream);

/* init_mask modes neighborG (0) / neighborE (1) / context (2), ff_raft.py:23-72, fused with the
 * [0,255] -> [-1,1] scaling: mask (B,1,H,W) [+ image (B,3,H,W)] -> NHWC4.  `table` = host-built k x k
 * Gaussian (get_kernel, :13-21) or ellipse structuring element; tmp (B*H*W floats) and gmax (1 word)
 * are caller-provided scratch.  mode: 0 neighborG, 1 neighborE, 2 context; + 4: leave the values in [0,255] (FF-PWC's
 * init_mask, core/models/ff-pwcnet/PWCNet_Core/ff_pwcnet.py:61-110, does not normalise); + 8: `image` is NHWC4. */
int ff_mask_prepare(int mode, const float* mask, const float* image, const float* table, int ks, float* tmp,
                    unsigned int* gmax, float* dst_nhwc4, int B, int H, int W, void* stream);

/* FF-PWC plumbing: raw (unscaled) NCHW -> NHWC4 (ff_pwcnet.py:405-410 consumes [0,255]); and
 * F.interpolate(bilinear, align_corners=False) NHWC -> NCHW with channels 0/1 multiplied by mul0/mul1
 * (test_mode output resize + flow rescale, ff_pwcnet.py:427-431). */
int ff_nchw_to_nhwc4(const float* src_nchw, int src_c, float fill, float* dst_nhwc4, int B, int H, int W, void* stream);
int ff_resize_bilinear(const float* src_nhwc, int ld, int C, int Hi, int Wi, float* dst_nchw, int B, int Ho, int Wo,
                       float mul0, float mul1, void* stream);
/* FF_PWCNET.preprocess (ff_pwcnet.py:391-403): the same bilinear resize of a 1- or 3-channel NCHW input to
 * (Ho, Wo), written as NHWC4 (one channel is repeated to three; channel 3 = 0). */
int ff_resize_to_nhwc4(const float* src_nchw, int src_c, int Hi, int Wi, float* dst_nhwc4, int B, int Ho, int Wo,
                       void* stream);

/* ========================================================================
 * SA / CA fusion units (parallel_fusion.py:14-73) — the parts that are not convolutions.  NHWC fp32,
 * deterministic reductions.
 *   ff_chan_stats_fwd      st[p][0] = mean_c x[p][c], st[p][1] = max_c, st[p][2..st_ld) = 0 (the padded
 *                          input of SA.s_map's 2->1 conv, :66-69); argmax[p] = first maximal channel
 *   ff_chan_stats_bwd      gx[p][c] = g[p][0] / C + (c == argmax[p]) * g[p][1]
 *   ff_spatial_stats_fwd   avg[b][c], mx[b][c] = AdaptiveAvg/MaxPool2d(1) (CA, :42-43); argmax = pixel index;
 *                          scratch: FF_SPATIAL_SLABS * B * C * 3 floats
 *   ff_spatial_stats_bwd   gx[b][p][c] = gavg[b][c] / HW + (p == argmax[b][c]) * gmax[b][c]
 *   ff_scale_add_fwd       out = sv * v + q (q nullable); mode 0: sv = s[pixel * s_ld] (SA, :70-71);
 *                          mode 1: sv = s[b*C + c] + s2[b*C + c] (s2 nullable) (CA, :44-46)
 *   ff_scale_add_bwd       gv = sv * gout; gs = sum over channels (mode 0: one float per pixel) or over
 *                          pixels (mode 1: gs[b*C + c], duplicated at gs[B*C + ...] when s2 is given;
 *                          scratch as above) of v * gout
 * ======================================================================== */
#define FF_SPATIAL_SLABS 64
int ff_chan_stats_fwd(const float* x, int x_ld, int C, long long npix, float* st, int st_ld, int* argmax, void* stream);
int ff_chan_stats_bwd(const float* g, int g_ld, const int* argmax, int C, long long npix, float* gx, int gx_ld,
                      void* stream);
int ff_spatial_stats_fwd(const float* x, int x_ld, int C, int B, int HW, float* avg, float* mx, int* argmax,
                         float* scratch, void* stream);
int ff_spatial_stats_bwd(const float* gavg, const float* gmax, const int* argmax, int C, int B, int HW, float* gx,
                         int gx_ld, void* stream);
int ff_scale_add_fwd(const float* v, int v_ld, const float* s, int s_ld, const float* s2, const float* q, int q_ld,
                     float* out, int out_ld, int C, int B, int HW, int mode, void* stream);
int ff_scale_add_bwd(const float* gout, int g_ld, const float* v, int v_ld, const float* s, int s_ld, const float* s2,
                     float* gv, int gv_ld, float* gs, float* scratch, int C, int B, int HW, int mode, void* stream);

/* Host-side (CPU) helper of the data loaders: PNG scan-line reconstruction for the 16-bit KITTI flow maps
 * (core/utils/frame_utils.py:102-120 decodes them with cv2).  raw = inflated IDAT stream, h rows of 1 + stride bytes;
 * out = h x stride reconstructed bytes; bpp = bytes per complete pixel.  HOST pointers; no GPU involved. */
int ff_png_unfilter(const unsigned char* raw, long long raw_len, int h, int stride, int bpp, unsigned char* out);

#ifdef __cplusplus
}
#endif
#endif /* FOCUSFLOW_HIP_H */
