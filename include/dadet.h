/*
 * dadet.h — C-ABI of libdadet_hip.so, the MI355X (gfx950) native operator library behind the
 * domain-adaptive Faster R-CNN training hot path.
 *
 * This header is the drop-in boundary.  Every entry point replaces one function of the
 * reference's native extension `maskrcnn_benchmark._C` (reference: maskrcnn_benchmark/csrc/vision.cpp:7-15,
 * vendored tree tools/cityscapes/maskrcnn_benchmark/csrc/vision.cpp:17-24) or one ATen operator the
 * reference's Python hot path calls (conv / linear / pooling / SGD), as cited per function.
 *
 * Conventions
 *   - plain pointers + sizes, no framework types; all pointers are DEVICE pointers unless the
 *     argument name ends in `_host`;
 *   - the caller owns and allocates every buffer (outputs, scratch); scratch sizes come from the
 *     matching `*_workspace_bytes` query;
 *   - `stream` is a hipStream_t passed as void*; every kernel is launched on it, nothing
 *     synchronises the device (the reference's nms does a blocking D2H copy, csrc/cuda/nms.cu:99-103;
 *     this one does not);
 *   - activations are fp32 NHWC ("channels_last" physical order: [N][H][W][C]); convolution
 *     weights are fp32 [Cout][KH][KW][Cin] (K contiguous).  The Python shim presents them with the
 *     reference's logical NCHW / OIHW shapes via torch.channels_last strides;
 *   - every function returns 0 on success or a negative DADET_E* code; dadet_last_error() gives a
 *     thread-local message.  Empty inputs (n == 0 / R == 0) return 0 without launching
 *     (reference: ROIAlign_cuda.cu:278-281, nms.h:17-18).
 */
#ifndef DADET_H_
#define DADET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DADET_OK 0
#define DADET_EINVAL (-1)   /* bad argument (shape / alignment / null pointer) */
#define DADET_ELAUNCH (-2)  /* hip launch / runtime error */
#define DADET_EWORKSPACE (-3) /* workspace too small */
#define DADET_EUNSUPPORTED (-4)

const char* dadet_last_error(void);
/* library / device probe: returns 0 and fills the fields when a gfx950 device is present. */
int dadet_version(void);
int dadet_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------------
 * NMS — replaces `_C.nms(dets[N,4], scores[N], thr) -> int64[K]`
 *   reference: csrc/nms.h:10-28, csrc/cpu/nms_cpu.cpp:6-75, csrc/cuda/nms.cu:23-131.
 * Greedy NMS with the reference's "+1" pixel convention.  Boxes are ranked by (score desc, index asc)
 * on the device, the 64x64-tiled IoU bitmask is built for the upper triangle only, and the greedy
 * sweep runs ON THE DEVICE (no D2H of the mask).  Kept ORIGINAL indices are written ascending
 * (reference nms_cpu.cpp:64, nms.cu:127-130).
 *   tie_rule 0: suppress when IoU >= thr (reference CPU rule, nms_cpu.cpp:60)
 *   tie_rule 1: suppress when IoU >  thr (reference CUDA rule, nms.cu:60)
 *   max_keep  : <=0 = unlimited; >0 stops the sweep after that many boxes are kept (the first
 *               max_keep in score order; equals the reference's keep[:max_proposals],
 *               structures/boxlist_ops.py:30-31, when the input is already score-sorted as in the RPN).
 *   keep_out  : int64[n] device; num_keep_out: int32[1] device.
 * ----------------------------------------------------------------------------------------------*/
/* scores == NULL: the boxes are already ranked (best first) by the caller — e.g. the RPN's stable top-k sort,
 * rpn/inference.py:88-101 — and the internal ranking pass is skipped; kept indices are positions in that order. */
int dadet_nms_workspace_bytes(int n, size_t* bytes_out);
int dadet_nms(const float* boxes_xyxy, const float* scores, int n, float thresh, int tie_rule,
              int max_keep, void* workspace, size_t workspace_bytes, int64_t* keep_out,
              int* num_keep_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ROIAlign — replaces `_C.roi_align_forward` / `_C.roi_align_backward`
 *   reference: csrc/ROIAlign.h:11-45, csrc/cpu/ROIAlign_cpu.cpp:114-257, csrc/cuda/ROIAlign_cuda.cu:65-254.
 * input  [B][H][W][C] NHWC, rois [R][5] = (batch_idx, x1, y1, x2, y2), output [R][PH][PW][C] NHWC.
 * No coordinate rounding, ROI min size 1, adaptive grid ceil(roi/pooled) when sampling_ratio == 0,
 * sample skipped when y < -1 || y > H || x < -1 || x > W.  Forward is evaluated in the reference's
 * operation order with FMA contraction disabled, so it is bit-comparable with ROIAlign_cpu.cpp.
 * Backward is a deterministic GATHER: one workgroup per grad_input pixel sums, in a fixed order, the
 * contributions of every ROI sample whose bilinear footprint touches it (the weights are separable), and
 * overwrites grad_input completely — no atomics, no pre-zeroing.  The reference's scatter form
 * (ROIAlign_cuda.cu:178-254, atomicAdd into a zero-filled tensor at :316) is kept as
 * dadet_roi_align_backward_atomic (the caller zero-fills) for A/B measurements.
 * ----------------------------------------------------------------------------------------------*/
int dadet_roi_align_forward(const float* input, const float* rois, float* output, int B, int C, int H,
                            int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                            int sampling_ratio, void* stream);
int dadet_roi_align_backward(const float* grad_output, const float* rois, float* grad_input, int B,
                             int C, int H, int W, int R, int pooled_h, int pooled_w,
                             float spatial_scale, int sampling_ratio, void* stream);
/* Forward with a caller-allocated scratch buffer (dadet_roi_align_workspace_bytes), result identical bit for bit: the ROIs
 * are PROCESSED in Z-order of their centres (one extra tiny launch) so that the workgroups of an XCD share feature rows
 * in its L2; every ROI is still written to its own output row. */
int dadet_roi_align_workspace_bytes(int B, int H, int W, int R, size_t* bytes);
int dadet_roi_align_forward_ws(const float* input, const float* rois, float* output, int B, int C, int H, int W, int R,
                               int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, void* workspace,
                               size_t workspace_bytes, void* stream);
int dadet_roi_align_backward_atomic(const float* grad_output, const float* rois, float* grad_input, int B,
                                    int C, int H, int W, int R, int pooled_h, int pooled_w,
                                    float spatial_scale, int sampling_ratio, void* stream);
/* Every bin_stride-th bin of the pooled_h x pooled_w grid only: output / grad_output are the COMPACT
 * [R][ceil(pooled_h / s)][ceil(pooled_w / s)][C] tensors of the bins (ph, pw) with ph % s == 0 and pw % s == 0, each bin
 * evaluated exactly as in the full grid.  For the reference's res5 ROI head with STRIDE_IN_1X1 (modeling/backbone/
 * resnet.py:236-262: the first block's 1x1 conv and its shortcut carry stride 2), which reads one of four bins of
 * roi_align's 14 x 14 output and back-propagates zeros into the other three: the same 7 x 7 values, a quarter of the
 * work and of the pooled tensor.  workspace as in dadet_roi_align_forward_ws (may be NULL).  The backward needs C % 4 ==
 * 0, 16-byte aligned buffers and a pooled grid of at most 14 x 14 (DADET_EUNSUPPORTED otherwise). */
int dadet_roi_align_forward_sub(const float* input, const float* rois, float* output, int B, int C, int H, int W, int R,
                                int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int bin_stride,
                                void* workspace, size_t workspace_bytes, void* stream);
int dadet_roi_align_backward_sub(const float* grad_output, const float* rois, float* grad_input, int B, int C, int H,
                                 int W, int R, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                 int bin_stride, void* stream);
/* Feature pyramids — replaces the per-level split of `Pooler.forward` (reference: modeling/poolers.py:91-121: LevelMapper,
 * one `nonzero` + `_C.roi_align_forward` + index_put per level).  `levels` [R] int64 on the device: the level of every
 * ROI (poolers.py:11-42).  Call once per level with that level's map / scale and the SAME rois, levels and output
 * [R][ph][pw][C]: the launch pools the ROIs of `level` into their own rows and leaves the other rows alone.  Backward:
 * grad_input [B][H][W][C] of the level's map from the ROIs of that level (deterministic gather, C % 4 == 0). */
int dadet_roi_align_forward_level(const float* input, const float* rois, const int64_t* levels, int level, float* output,
                                  int B, int C, int H, int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                  int sampling_ratio, void* workspace, size_t workspace_bytes, void* stream);
int dadet_roi_align_backward_level(const float* grad_output, const float* rois, const int64_t* levels, int level,
                                   float* grad_input, int B, int C, int H, int W, int R, int pooled_h, int pooled_w,
                                   float spatial_scale, int sampling_ratio, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SigmoidFocalLoss — replaces `_C.sigmoid_focalloss_forward/backward`
 *   reference: csrc/SigmoidFocalLoss.h:10-40, csrc/cuda/SigmoidFocalLoss_cuda.cu:21-101.
 * logits [N][C] fp32, targets [N] int32 in [0..C] (t == d+1 positive, t >= 0 && t != d+1 negative).
 * ----------------------------------------------------------------------------------------------*/
int dadet_sigmoid_focal_loss_forward(const float* logits, const int32_t* targets, float* losses, int N,
                                     int C, float gamma, float alpha, void* stream);
int dadet_sigmoid_focal_loss_backward(const float* logits, const int32_t* targets,
                                      const float* d_losses, float* d_logits, int N, int C,
                                      float gamma, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32) with fused epilogue.
 * Replaces the ATen conv2d + FrozenBatchNorm2d + relu_ (+ residual add) chains of
 *   reference: modeling/backbone/resnet.py:294-336 (Bottleneck / BaseStem), layers/batch_norm.py:19-24,
 *   modeling/rpn/rpn.py:39-46 (RPNHead), modeling/da_heads/da_heads.py:32-37 (DAImgHead) and the
 *   nn.Linear layers of da_heads.py:61-68 / roi_box_predictors.py:28-33 (as 1x1 convs on [R,1,1,C]).
 *
 *   y[n, ho*os, wo*os, co] = act( (sum_{r,s,ci} x[n, ho*stride-pad+r, wo*stride-pad+s, ci] * w[co,r,s,ci])
 *                                  * scale[co] + bias[co] + addend[...] )
 *   act: relu_mode 0 none | 1 max(v,0) | 2 (mask_ref[...] > 0 ? v : 0)  (backward ReLU gating)
 * `os` (out_spatial_stride) > 1 scatters rows into a larger, caller-zeroed output [N][OutH][OutW][Cout]
 * (dgrad of a strided 1x1 conv).  scale / bias / addend / mask_ref may be NULL.  addend may alias y.
 * Requirements: Cin % 4 == 0, (KH*KW*Cin) % 4 == 0, 16-byte aligned x / w.
 * ----------------------------------------------------------------------------------------------*/
typedef struct dadet_conv_desc {
  int N, H, W, Cin;        /* input  [N][H][W][Cin] */
  int Cout, KH, KW;        /* weight [Cout][KH][KW][Cin] */
  int stride, pad;         /* same in both spatial dims; dilation 1 */
  int Ho, Wo;              /* GEMM rows: M = N*Ho*Wo */
  int OutH, OutW;          /* physical output spatial dims (== Ho,Wo unless out_spatial_stride > 1) */
  int out_spatial_stride;  /* 1, or s for scatter */
  int relu_mode;           /* 0 | 1 | 2 */
} dadet_conv_desc;

int dadet_conv_forward(const dadet_conv_desc* d, const float* x, const float* w, const float* scale,
                       const float* bias, const float* addend, const float* mask_ref, float* y,
                       void* stream);

/* Contraction mode of dadet_conv_forward / dadet_conv_wgrad (process-wide).  Inputs, outputs and accumulation are fp32 in
 * every mode; the modes differ in how the fp32 products are formed on the matrix pipe:
 *   4 (DEFAULT, also for a consumer that never calls this) = each operand, scaled by a per-tensor power of two, is split
 *     into two fp16 terms; three v_mfma_f32_32x32x16_f16 per K=16 (error ~2^-24 |ab|: fp32 class).  The scale of a tensor
 *     comes from its largest magnitude (below); through dadet_conv_forward / dadet_conv_wgrad the library measures the
 *     operands itself (two extra passes), through the *_scaled entry points the caller hands the maxima over;
 *   3 = three bf16 terms, six v_mfma_f32_32x32x16_bf16 per K=16 (error ~2^-24 |ab|, fp32 class, no scales needed);
 *   0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *   2 = two bf16 terms, three MFMAs per K=16 (error ~2^-16 |ab|, experiments only). */
int dadet_set_gemm_mode(int mode);
int dadet_get_gemm_mode(void);

/* Large-tile kernel of mode 4 (256 x 256 output tile, 8 waves, the two waves of a SIMD alternating between the matrix pipe
 * and the operand staging; csrc/conv_big.hip) for dadet_conv_forward[_scaled] (process-wide):
 *   1 (DEFAULT) = on the layers where it is expected to win (csrc/conv_big.hip: big_variant — K >= 512, M >= 4096,
 *       Cout >= 128, Cin % 32 == 0, unit output stride, at least 96 output tiles: the 256 x 128-tile variant for
 *       Cout <= 256 — except Cout in (128, 256] with at least 96 row tiles, which fill the chip on the 256 x 256 tile —
 *       the 256 x 256 tile above; everything else stays on the 128 x 128 / weight-stationary kernels);
 *   0 = never;  2 = wherever it is applicable (Cin % 32 == 0, Cout % 4 == 0, 16-byte aligned tensors below 2 GB) — tests.
 * Results differ from the 128 x 128 kernel's only by fp32 summation order. */
int dadet_set_big_gemm(int mode);
int dadet_get_big_gemm(void);

/* Non-finite guard of mode 4.  A per-tensor scale from a stale or too small maximum overflows fp16 inside the operand split
 * (inf, then NaN against the zeros of a ReLU'd operand).  Every GEMM checks its sums once behind the K loop and records
 * the first offending launch in two device words.  dadet_nonfinite_poll waits for EVERY stream of the device
 * (hipDeviceSynchronize: side streams are non-blocking, the null stream does not order against them), takes the two words
 * with one atomic exchange each (read and clear are one step: a record set meanwhile is reported by the next poll, never
 * lost) and returns the number of wavefronts that saw non-finite sums since the last poll (0 = clean, < 0 = error) — a
 * synchronising call, made at the logging period; `msg` (may be NULL) receives a sentence naming the first launch — entry point, launch
 * number, M / N / K — from a ring of the last 8192 launch records.  DADET_NONFINITE_GUARD=0 (environment) switches the
 * device-side check off. */
int dadet_nonfinite_poll(char* msg, int cap);

/* Largest magnitudes for mode 4.  A "slot" holds max|t| over a tensor t (an upper bound within a few binades serves as
 * well: it only has to keep t / slot inside fp16's range without wasting it).  It is addressed by one pointer p and
 * consists of EIGHT floats, p[0], p[S], ..., p[7 S] with S = DADET_AMAX_STRIDE: the value is the maximum of the eight
 * (writers merge into the shard of their XCD, so that the chip's workgroups do not queue on one word).  Allocate slots as
 * columns of a zero-filled [8][S] float array — column i is slot i, p = array + i.
 * dadet_amax merges max|x[0..n)| into a slot (atomic max: zero the slot first, or reuse one to cover several tensors).
 * dadet_amax_batch does the same for n tensors in one launch; items_dev is device-resident, item k owns the workgroups
 * [first_block, first_block + blocks) of a grid of total_blocks. */
#define DADET_AMAX_STRIDE 65536
typedef struct dadet_amax_item {
  const void* x;           /* float[n], 16-byte aligned */
  void* slot;              /* a slot (see above) */
  long long n;
  int first_block, blocks;
} dadet_amax_item;
int dadet_amax(const float* x, long long n, float* slot, void* stream);
int dadet_amax_batch(const dadet_amax_item* items_dev, int n, int total_blocks, void* stream);

/* dadet_conv_forward with the operands' maxima handed over (amax_x, amax_w: slots, NULL = measured here) and,
 * when amax_y is not NULL, max|y| of the values this call stores merged into *amax_y by the epilogue itself (zero it
 * first) — the next GEMM's amax_x.  Modes other than 4 ignore all three. */
int dadet_conv_forward_scaled(const dadet_conv_desc* d, const float* x, const float* w, const float* scale,
                              const float* bias, const float* addend, const float* mask_ref, float* y,
                              const float* amax_x, const float* amax_w, float* amax_y, void* stream);

/* which tile variant dadet_conv_forward launches for this shape: 0 = 128x128 (conv_fwd_kernel<2,2>),
 * 1 = 128x64 (<2,1>), 2 = 64x64 (<1,1>).  Used by bench.py to attribute per-launch timings. */
int dadet_conv_forward_variant(const dadet_conv_desc* d);

/* weight gradient: dw[co][r][s][ci] = out_scale[co] * sum_m gy[m][co] * x[gather(m, r, s)][ci]
 * (+ dw_prev when accumulate != 0).  Deterministic split-K over m through `workspace`
 * (query with dadet_conv_wgrad_workspace_bytes).  gy is [N][Ho][Wo][Cout] dense. */
int dadet_conv_wgrad_workspace_bytes(const dadet_conv_desc* d, size_t* bytes_out);
/* which kernel dadet_conv_wgrad* launches for this shape (dense gy rows): 0 = 128 x 128 tiles, 1 = 256 x 256 tiles
 * (conv_wgrad_big_kernel, mode 4; see dadet_set_big_gemm).  Profiling labels. */
int dadet_conv_wgrad_variant(const dadet_conv_desc* d);
int dadet_conv_wgrad(const dadet_conv_desc* d, const float* x, const float* gy, const float* out_scale,
                     float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the reduction over the split partial results left to the caller: when the plan splits the reduction,
 * *pending_out describes the missing pass (splits > 1) and `workspace` must stay untouched until
 * dadet_conv_wgrad_reduce_batch has run on the same stream; otherwise pending_out->splits == 0 and dw is complete.
 * dadet_conv_wgrad_reduce_batch performs up to 8 such passes per launch — one launch for the 3 - 4 weight gradients of a
 * residual block instead of one 10 - 40 us launch each — with dadet_conv_wgrad's arithmetic (identical bits). */
typedef struct dadet_wgrad_pending {
  const float* partials;   /* [splits][Cout][K] */
  const float* out_scale;  /* [Cout] or NULL */
  float* dw;               /* [Cout][K] */
  long long count;         /* Cout * K */
  int K, splits, accumulate;
} dadet_wgrad_pending;
int dadet_conv_wgrad_partials(const dadet_conv_desc* d, const float* x, const float* gy, const float* out_scale,
                              float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                              dadet_wgrad_pending* pending_out, void* stream);
int dadet_conv_wgrad_reduce_batch(const dadet_wgrad_pending* items, int n, void* stream);
/* gy with rows of gy_ld floats, gy_ld = Cout rounded up to a multiple of 4 (Cout itself need not be one): the weight
 * gradient of a convolution whose output rows the forward pads to 16 bytes — the offset / modulation branch of a
 * deformable block (18 | 27 channels in rows of 20 | 28; reference: layers/misc.py:139-151 `self.offset`).  dw keeps its
 * own [Cout][KH][KW][Cin] shape, so the gradient can be accumulated straight into the parameter's buffer. */
int dadet_conv_wgrad_partials_ld(const dadet_conv_desc* d, const float* x, const float* gy, int gy_ld,
                                 const float* out_scale, float* dw, int accumulate, void* workspace,
                                 size_t workspace_bytes, dadet_wgrad_pending* pending_out, void* stream);

/* dadet_conv_wgrad_partials_ld (pending_out may be NULL: then the reduction pass runs here, as in dadet_conv_wgrad) with
 * the operands' maxima handed over; see dadet_conv_forward_scaled */
int dadet_conv_wgrad_scaled(const dadet_conv_desc* d, const float* x, const float* gy, int gy_ld,
                            const float* out_scale, float* dw, int accumulate, void* workspace,
                            size_t workspace_bytes, dadet_wgrad_pending* pending_out, const float* amax_x,
                            const float* amax_gy, void* stream);

/* The 1 - 4 weight gradients of ONE backward node in ONE launch (contraction mode 4): replaces the per-layer ATen
 * backward-weight calls of a bottleneck block (reference: maskrcnn_benchmark/modeling/backbone/resnet.py:294-314
 * `Bottleneck.forward` under autograd — conv1, conv2, conv3 and the downsample branch).  The chip's workgroup slots are
 * shared among the problems (every part of every problem reduces the same number of rows), so a problem parks a
 * fraction of the partial sums its own launch would, its parts are several times as long, and layers too small for a
 * launch of their own ride along.
 * dadet_conv_wgrad_group_plan -> the tile edge of the kernel that serves the WHOLE group — 256 (every problem has Cout,
 * K >= 256 and >= 2048 rows: conv_wgrad_big_group_kernel) or 128 (the 128 x 128 kernel's grouped form; all problems' maps
 * at least, or all less than, 32 pixels wide) — and per problem the number of parts and the workspace bytes ([splits]
 * [Cout][K] floats; 0 for one part, which writes dw itself); 0 when there is no grouped launch for these problems (K or
 * Cout not a multiple of 4, fewer than 128 rows, another contraction mode, more than 4): call dadet_conv_wgrad_scaled
 * per layer.  dadet_conv_wgrad_group: arrays of n entries; gy rows are Cout floats; pending_out[i] describes problem i's
 * reduction pass for dadet_conv_wgrad_reduce_batch (splits == 0: nothing to reduce); the dw pointers must differ.
 * Results equal the per-layer calls' up to the order of the fp32 partial sums. */
int dadet_conv_wgrad_group_plan(const dadet_conv_desc* descs, int n, int* splits_out, size_t* workspace_bytes_out);
int dadet_conv_wgrad_group(const dadet_conv_desc* descs, int n, const float* const* x, const float* const* gy,
                           const float* const* out_scale, float* const* dw, const int* accumulate,
                           void* const* workspace, const size_t* workspace_bytes, dadet_wgrad_pending* pending_out,
                           const float* const* amax_x, const float* const* amax_gy, void* stream);

/* weight re-layout for the data gradient: wt[ci][KH-1-r][KW-1-s][co] = w[co][r][s][ci] * scale[co].
 * dgrad of a stride-1 conv is then dadet_conv_forward(gy, wt) with pad' = K-1-pad. */
int dadet_conv_weight_transpose(const float* w, const float* scale, float* wt, int Cout, int KH, int KW,
                                int Cin, void* stream);
/* the same into rows of cout_pad >= Cout columns, the columns co >= Cout zero: wt[ci][tap][cout_pad] — the data-gradient
 * weights for an output gradient whose rows are padded (see dadet_conv_wgrad_partials_ld) */
int dadet_conv_weight_transpose_padded(const float* w, const float* scale, float* wt, int Cout, int KH, int KW,
                                       int Cin, int cout_pad, void* stream);

/* The same for many weights in ONE launch.  items_dev: device-resident table; item k serves blocks [first_block,
 * first_block + blocks_ci * blocks_co * KH * KW) of the grid, blocks_ci = ceil(Cin / 32), blocks_co = ceil(Cout / 32);
 * total_blocks = the sum.  Used once per optimizer step for every weight whose data gradient the backward pass needs. */
typedef struct dadet_transpose_item {
  const float* w;
  const float* scale;
  float* wt;
  int Cout, KH, KW, Cin;
  int first_block, blocks_ci, blocks_co;
  int cout_pad;            /* width of the output rows, >= Cout (0: Cout); blocks_co = ceil(max(Cout, cout_pad) / 32) */
} dadet_transpose_item;
int dadet_conv_weight_transpose_batch(const dadet_transpose_item* items_dev, int n, int total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Deformable convolution v1 / v2 — replaces the vendored tree's `_C.deform_conv_forward / _backward_input /
 * _backward_parameters` and `_C.modulated_deform_conv_forward / _backward`
 *   reference: tools/cityscapes/maskrcnn_benchmark/csrc/vision.cpp:17-24, csrc/cuda/deform_conv_cuda.cu:158-690,
 *   csrc/cuda/deform_conv_kernel_cuda.cu:92-775 (not bound in the reference's main tree).
 * The deformable sampling is one kernel that writes the sampled operand as an NHWC column tensor
 *   cols[n][ho][wo][tap][c] = mask[n][ho][wo][g][tap] * bilinear(x[n], p0 + p_tap + offset[n][ho][wo][g][tap])
 * (zero outside (-1,H)x(-1,W)); the contraction with the [Cout][KH][KW][Cin] weights, its data gradient and its
 * weight gradient are dadet_conv_forward / dadet_conv_wgrad calls on `cols` viewed as a 1x1 convolution over
 * K = KH*KW*Cin.  offset: [N][Ho][Wo][dg*2*KH*KW] (channel = g*2*T + 2*tap + {0: dy, 1: dx}, the reference's order);
 * mask: [N][Ho][Wo][dg*KH*KW] or NULL (v1).  Backward: gx is accumulated with atomics (caller zero-fills, may be
 * NULL), goffset / gmask are accumulated with atomics (caller zero-fills).
 * ----------------------------------------------------------------------------------------------*/
int dadet_deform_sample_forward(const float* x, const float* offset, const float* mask, float* cols, int N,
                                int H, int W, int C, int KH, int KW, int stride, int pad, int dil,
                                int deformable_groups, int Ho, int Wo, void* stream);
int dadet_deform_sample_backward(const float* x, const float* offset, const float* mask, const float* gcols,
                                 float* gx, float* goffset, float* gmask, int N, int H, int W, int C, int KH,
                                 int KW, int stride, int pad, int dil, int deformable_groups, int Ho, int Wo,
                                 void* stream);
/* The same two operators reading offsets / modulation straight out of the offset-predicting convolution's output and
 * writing their gradients straight into that convolution's output gradient (DFConv2d: vendored layers/misc.py:114-203
 * slices `[:, :18]` / `[:, -9:]` out of one conv output and autograd concatenates the slice gradients back).  `*_ld` =
 * floats per output pixel of the tensor the pointer points into (>= dg*2*KH*KW resp. dg*KH*KW); mask_is_logit != 0: the
 * mask tensor holds the conv's raw output, modulation = sigmoid(logit) is applied here and gmask is the gradient w.r.t.
 * the logit (misc.py:188 `.sigmoid()` folded in). */
int dadet_deform_sample_forward_ld(const float* x, const float* offset, int offset_ld, const float* mask, int mask_ld,
                                   int mask_is_logit, float* cols, int N, int H, int W, int C, int KH, int KW, int stride,
                                   int pad, int dil, int deformable_groups, int Ho, int Wo, void* stream);
int dadet_deform_sample_backward_ld(const float* x, const float* offset, int offset_ld, const float* mask, int mask_ld,
                                    int mask_is_logit, const float* gcols, float* gx, float* goffset, int goffset_ld,
                                    float* gmask, int gmask_ld, int N, int H, int W, int C, int KH, int KW, int stride,
                                    int pad, int dil, int deformable_groups, int Ho, int Wo, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* ... and max|gradient w.r.t. the offset conv's output| left in `amax_gom` (a zero-initialised slot in dadet_amax's
 * layout; contraction mode 4: that gradient is an operand of the offset conv's weight-gradient GEMM).  goffset / gmask are
 * column ranges of ONE zero-filled [N*Ho*Wo][goffset_ld] tensor of gom_floats floats starting at goffset. */
int dadet_deform_sample_backward_ld_m(const float* x, const float* offset, int offset_ld, const float* mask, int mask_ld,
                                      int mask_is_logit, const float* gcols, float* gx, float* goffset, int goffset_ld,
                                      float* gmask, int gmask_ld, int N, int H, int W, int C, int KH, int KW, int stride,
                                      int pad, int dil, int deformable_groups, int Ho, int Wo, void* workspace,
                                      size_t workspace_bytes, float* amax_gom, long long gom_floats, void* stream);
/* scratch for the gather form of the backward (per-cell lists of the samples whose bilinear corners land on a cell:
 * csrc/deform.hip); without it (NULL / too small) the atomic forms run */
int dadet_deform_sample_backward_workspace_bytes(int N, int H, int W, int deformable_groups, size_t* bytes_out);

/* ROIPool — replaces `_C.roi_pool_forward / roi_pool_backward` (csrc/vision.cpp:11-12, ROIPool.h:11-46,
 * cuda/ROIPool_cuda.cu:16-108).  input [B][H][W][C] NHWC, rois [R][5] = (batch, x1, y1, x2, y2), output and
 * argmax [R][PH][PW][C]; argmax = h*W + w of the first maximum of the bin, -1 for an empty bin (output 0).
 * Backward zero-fills grad_input [B][H][W][C] and routes every output gradient to its argmax cell. */
int dadet_roi_pool_forward(const float* input, const float* rois, float* output, int* argmax, int B, int C, int H,
                           int W, int R, int pooled_h, int pooled_w, float spatial_scale, void* stream);
int dadet_roi_pool_backward(const float* grad_output, const int* argmax, const float* rois, float* grad_input, int B,
                            int C, int H, int W, int R, int pooled_h, int pooled_w, void* stream);

/* Device-side input pipeline (image.hip): Pillow-exact bilinear resize + flip + BGR-255 + normalisation, replacing
 * the host transforms of maskrcnn_benchmark/data/transforms/transforms.py:32-97 (Resize -> torchvision F.resize ->
 * PIL BILINEAR, RandomHorizontalFlip, ToTensor, Normalize(to_bgr255)).  Images are uint8 [H][W][3] (RGB, device
 * memory); bounds [out][2] = (first input index, tap count) and coeffs [out][ksize] (22-bit fixed point) are Pillow's
 * per-axis tables (computed on the host, da_detect_amd/data/device_prep.py).  Pass 1 resamples rows to out_w and
 * rounds to 8 bits; pass 2 resamples columns (bounds NULL: none), optionally mirrors, converts and writes fp32
 * [out_h][out_row_stride][3] — a slot of the zero-filled padded batch tensor.  mean3 / std3 are HOST pointers. */
int dadet_image_resample_h(const unsigned char* image_hwc, int H, int W, const int* bounds, const int* coeffs,
                           int ksize, int out_w, unsigned char* out_hwc, void* stream);
int dadet_image_resample_v_normalize(const unsigned char* image_hwc, int in_h, int w, const int* bounds,
                                     const int* coeffs, int ksize, int out_h, int flip, int to_bgr255,
                                     const float* mean3, const float* std3, float* out_hw3, int out_row_stride,
                                     void* stream);

/* Fused detection losses: value and gradient in one single-workgroup launch (losses.hip).
 * dadet_rpn_loss replaces RPNLossComputation.__call__'s loss part (modeling/rpn/loss.py:125-143): objectness /
 * box_regression are the flattened NHWC prediction maps ([N*H*W*A] and [N*H*W*A][4], the order of
 * concat_box_prediction_layers, rpn/utils.py:17-43, for one level); losses_out[0] = BCE-with-logits mean over the
 * sampled anchors, losses_out[1] = smooth-L1(beta) sum over positives / num_sampled.  The gradient maps must be
 * zero-filled by the caller; the kernel writes d(loss0 + loss1)/d(prediction) at the sampled positions.
 * dadet_fast_rcnn_loss replaces FastRCNNLossComputation.__call__ (modeling/roi_heads/box_head/loss.py:165-221):
 * losses_out[0] = cross-entropy mean over the source-domain rows, losses_out[1] = smooth-L1(beta 1) over the
 * positives' class columns / num_src; map_inds [num_pos][4] are the regression columns of each positive. */
int dadet_rpn_loss(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                   const float* labels_sampled, int num_sampled, const int64_t* pos_inds,
                   const float* regression_targets_pos, int num_pos, float beta, float* losses_out,
                   float* grad_objectness, float* grad_box_regression, void* stream);
/* The RPN losses in ROW form (losses.hip): the gradient of the head's maps is zero outside the num_sampled sampled anchors,
 * so it is returned as num_sampled rows — row r = anchor sampled_inds[r] = pixel * A + a, holding d loss / d (objectness
 * | box regression of that pixel) for anchor a in the columns [a] and [A + 4a .. A + 4a + 3] of a zero row of ldg >= 5A
 * floats — with the row's pixel index n*H*W + h*W + w in pixels_out.  The positives are the first num_pos sampled rows
 * (rpn/loss.py:116-118).  Same loss values as dadet_rpn_loss.  dadet_gather_pixel_taps builds the operand rows of a
 * stride-1 KH x KW convolution's backward at those pixels (out [rows][KH*KW][C], zero outside the image; KH = KW = 1: a
 * row gather); dadet_scatter_pixel_taps_add adds y [rows][KH*KW][C] to dx [N][H][W][C] at pixel + tap offset (the caller
 * zero-fills dx; fp32 atomics).  With these the RPN head's backward (modeling/rpn/rpn.py:39-46 under autograd) runs on
 * num_sampled rows instead of N*H*W. */
int dadet_rpn_loss_rows(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                        const float* labels_sampled, int num_sampled, int num_pos, const float* regression_targets_pos,
                        int anchors_per_location, float beta, float* losses_out, float* grad_rows, int ldg,
                        int* pixels_out, void* stream);
/* One level of a feature pyramid: sampled_inds index the concatenation over levels on which the reference defines the RPN
 * losses (concat_box_prediction_layers, modeling/rpn/utils.py:10-45: image-major, anchors_per_image per image, this level's
 * level_anchors = H*W*A entries at level_offset, ordered (h, w, a)); objectness / box_regression are THIS level's maps.
 * Every level divides by the same num_sampled, so the levels' losses add up to rpn/loss.py:125-143.
 * shared_rows = 0: grad_rows / pixels_out belong to this launch; an anchor of another level leaves its row zero and its
 *   pixel -1 (dadet_gather_pixel_taps then writes zeros, dadet_scatter_pixel_taps_add skips the row).
 * shared_rows = 1: grad_rows (zero-filled by the caller once), pixels_out and row_level_out are shared by the launches of all
 *   levels; this launch writes the rows of its own anchors only and tags them with level_id in row_level_out.  The head's
 *   backward then runs ONCE over the rows of all levels: dadet_gather_pixel_taps_level / dadet_scatter_pixel_taps_add_level
 *   touch only the rows tagged `level` (the operand rows of the other levels are written by their own calls). */
int dadet_rpn_loss_rows_level(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                              const float* labels_sampled, int num_sampled, int num_pos,
                              const float* regression_targets_pos, int anchors_per_location, float beta,
                              int64_t anchors_per_image, int64_t level_offset, int64_t level_anchors, int level_id,
                              int* row_level_out, int shared_rows, float* losses_out, float* grad_rows, int ldg,
                              int* pixels_out, void* stream);
int dadet_gather_pixel_taps_level(const float* x, const int* pixels, const int* row_level, int level, int num_rows, int N,
                                  int H, int W, int C, int KH, int KW, int pad, float* out, void* stream);
int dadet_scatter_pixel_taps_add_level(const float* y, const int* pixels, const int* row_level, int level, int num_rows,
                                       int N, int H, int W, int C, int KH, int KW, int pad, float* dx, void* stream);
int dadet_gather_pixel_taps(const float* x, const int* pixels, int num_rows, int N, int H, int W, int C, int KH, int KW,
                            int pad, float* out, void* stream);
int dadet_scatter_pixel_taps_add(const float* y, const int* pixels, int num_rows, int N, int H, int W, int C, int KH,
                                 int KW, int pad, float* dx, void* stream);
int dadet_fast_rcnn_loss(const float* class_logits, const float* box_regression, int num_classes, int reg_cols,
                         const int64_t* src_rows, const int64_t* labels_src, int num_src, const int64_t* rows_pos,
                         const int64_t* map_inds, const float* regression_targets_pos, int num_pos,
                         float* losses_out, float* grad_class_logits, float* grad_box_regression, void* stream);

/* The same Fast R-CNN losses from per-row targets (no index lists): loss_labels[r] < 0 keeps row r out of both losses
 * (rows of target-domain images, box_head/loss.py:193-198), otherwise it is the row's class; the regression term
 * covers the rows with loss_labels > 0, columns 4*label .. 4*label+3 (columns 4..7 when reg_cols == 8, the
 * class-agnostic layout, box_head/loss.py:205-209); both losses are divided by the number of rows with
 * loss_labels >= 0.  Gradient maps zero-filled by the caller. */
int dadet_fast_rcnn_loss_rows(const float* class_logits, const float* box_regression, int num_rows, int num_classes,
                              int reg_cols, const int64_t* loss_labels, const float* regression_targets,
                              float* losses_out, float* grad_class_logits, float* grad_box_regression, void* stream);

/* Box-head proposal sampling of ONE image in one launch: replaces BalancedPositiveNegativeSampler.__call__
 * (modeling/balanced_positive_negative_sampler.py:25-68) + the union-mask nonzero and the per-field BoxList gather of
 * FastRCNNLossComputation.subsample (modeling/roi_heads/box_head/loss.py:95-130).  labels [n] (>= 1 positive,
 * 0 negative, < 0 ignored; NULL = all 0), regression_targets [n][4] (NULL = zeros), n <= 4096.  Takes
 * num_pos = min(#pos, max_pos) positives and num_neg = min(#neg, cap - num_pos) negatives, each a uniformly random
 * subset (smallest random keys of splitmix64(seed, index)), and writes them in ascending proposal order: idx_out,
 * boxes_out, labels_out, regression_targets_out, loss_labels_out (= label when is_source, else -1), domain_out
 * (= is_source), all [cap] rows; counts_out = {num_pos + num_neg, num_pos} (device memory). */
int dadet_sample_rois(const float* boxes, const int64_t* labels, const float* regression_targets, int n, int cap,
                      int max_pos, uint64_t seed, int is_source, int64_t* idx_out, float* boxes_out,
                      int64_t* labels_out, float* regression_targets_out, int64_t* loss_labels_out,
                      unsigned char* domain_out, int* counts_out, void* stream);

/* Multi-level proposal lists laid end to end (training-mode selection over a feature pyramid: rpn/inference.py:102-121 per
 * level, then the concatenation of :141-152).  Per (level, image) pair: boxes [n][4] / scores [n] in score order, the NMS
 * keep buffer and kept count (device memory, dadet_nms), and where the pair's cap = min(n, post_nms_top_n) slots start in the
 * image's output buffers: boxes_out[j] = boxes[keep[j]], scores_out[j] = scores[keep[j]] for j < count, score -1 behind.
 * At most 24 pairs per call; `entries` is a HOST array. */
typedef struct dadet_merge_entry {
  const float* boxes;
  const float* scores;
  const int64_t* keep;
  const int* count;
  float* boxes_out;
  float* scores_out;
  int n, cap;
} dadet_merge_entry;
int dadet_fpn_merge_levels(const dadet_merge_entry* entries, int n, void* stream);

/* NMS -> box-head sample of ONE image without leaving the device: proposal i = sorted_boxes[keep[i]] (objectness
 * sorted_scores[keep[i]]) for i < min(*count_dev, post_n), followed by `num_appended` appended boxes with objectness 1
 * (RPNPostProcessor.add_gt_proposals, rpn/inference.py:51-74); then, when is_source, IoU / Matcher / label rules /
 * BoxCoder.encode of every proposal against gt_boxes [G] exactly as dadet_box_match_encode, and the balanced random sample
 * exactly as dadet_sample_rois (same keys).  Replaces the kept-count read-back of rpn/inference.py:102 and the gathers /
 * concatenations behind it.  Besides dadet_sample_rois's outputs (+ objectness_out [cap]) it leaves the proposal list
 * itself in prop_boxes / prop_scores [post_n + num_appended] and its length in *n_props (device memory).
 * post_n + num_appended <= 4096, G <= 1024. */
int dadet_proposals_sample(const float* sorted_boxes, const float* sorted_scores, const int64_t* keep,
                           const int* count_dev, int post_n, const float* appended_boxes, int num_appended,
                           const float* gt_boxes, const int64_t* gt_labels, int G, float high_threshold,
                           float low_threshold, float wx, float wy, float ww, float wh, int cap, int max_pos, uint64_t seed,
                           int is_source, float* prop_boxes, float* prop_scores, int* n_props, int64_t* idx_out,
                           float* boxes_out, int64_t* labels_out, float* regression_targets_out, int64_t* loss_labels_out,
                           unsigned char* domain_out, float* objectness_out, int* counts_out, void* stream);

/* RPN anchor sampling of ONE image in one launch: replaces BalancedPositiveNegativeSampler.__call__ on the anchor
 * labels (modeling/balanced_positive_negative_sampler.py:25-68) + the nonzero / gathers of
 * RPNLossComputation.__call__ (modeling/rpn/loss.py:101-123).  labels [A] float (1 positive, 0 negative, -1 ignored),
 * regression_targets [A][4].  Takes num_pos = min(#pos, max_pos) positives and num_neg = min(#neg, cap - num_pos)
 * negatives, each a uniformly random subset (smallest splitmix64(seed, anchor) keys, radix select), cap <= 1024.
 * pos_inds_out / neg_inds_out [cap]: index_offset + anchor index, ascending (-1 past the count);
 * regression_targets_pos_out [cap][4]: the positives' targets; counts_out = {num_pos, num_neg} (device memory). */
int dadet_sample_anchors(const float* labels, const float* regression_targets, int A, int cap, int max_pos,
                         uint64_t seed, int64_t index_offset, int64_t* pos_inds_out, int64_t* neg_inds_out,
                         float* regression_targets_pos_out, int* counts_out, void* stream);

/* Sorted top-k of every row of scores[rows][row_stride] (first n entries of a row are valid), k <= min(n, 16384): the k
 * largest scores in descending order, EQUAL scores by ascending index (= torch.sort(descending=True, stable=True)[:k]).
 * Replaces objectness.topk(pre_nms_top_n, dim=1, sorted=True) of RPNPostProcessor.forward_for_single_feature_map
 * (modeling/rpn/inference.py:93-95).  out_scores [rows][k], out_idx [rows][k] int64.  One workgroup per row. */
int dadet_topk_sorted(const float* scores, int rows, int n, int64_t row_stride, int k, float* out_scores,
                      int64_t* out_idx, void* stream);

/* The same ranking for up to 16 rows of DIFFERENT lengths in one call — the per-level `objectness.topk(pre_nms_top_n)` of
 * the multi-level proposal selection (modeling/rpn/inference.py:124-152: one row per (pyramid level, image)).  Five
 * launches for all rows together, every radix pass spread over the chip (one workgroup per 8192 scores).  Row i: the
 * k_i <= min(n_i, 16384) largest of scores_i[0 .. n_i) in descending order, equal scores by ascending index, into
 * out_scores_i / out_idx_i [k_i].  workspace: dadet_topk_sorted_rows_workspace_bytes(rows, max k), 8-byte aligned. */
typedef struct dadet_topk_row {
  const float* scores;
  float* out_scores;
  int64_t* out_idx;
  int n, k;
} dadet_topk_row;
int dadet_topk_sorted_rows_workspace_bytes(int rows, int k_max, size_t* bytes_out);
int dadet_topk_sorted_rows(const dadet_topk_row* rows_in, int rows, void* workspace, size_t workspace_bytes, void* stream);

/* RPN anchor labelling in two launches: replaces boxlist_iou + Matcher(high, low, allow_low_quality_matches=True) +
 * the label rules of RPNLossComputation.prepare_targets (modeling/rpn/loss.py:57-98, modeling/matcher.py:42-112) +
 * BoxCoder((1,1,1,1)).encode.  visible[a] != 0: anchor inside the image.  labels: 1 matched, 0 below the low threshold,
 * -1 ignored (outside the image, or between the thresholds); anchors holding some ground-truth box's best IoU keep
 * their argmax match.  workspace_G: G uint32 scratch words. */
int dadet_rpn_anchor_targets(const float* anchors, const unsigned char* visible, int A, const float* gt_boxes, int G,
                             float high_threshold, float low_threshold, unsigned* workspace_G, float* labels,
                             float* regression_targets, void* stream);

/* Box-head target assignment in one launch: IoU of every proposal with the G ground-truth boxes, Matcher without
 * low-quality matches, label rules and regression targets — replaces the ATen chain boxlist_iou
 * (structures/boxlist_ops.py:56-91) -> Matcher.__call__ (modeling/matcher.py:42-92) -> prepare_targets label rules
 * (modeling/roi_heads/box_head/loss.py:69-93) -> BoxCoder.encode (modeling/box_coder.py:22-50).
 * matched_idxs[i] = argmax_g IoU (first maximum), or -1 (max < low) / -2 (low <= max < high);
 * labels[i] = 0 / -1 for those, else gt_labels[matched]; targets from gt[max(matched,0)] with weights (wx,wy,ww,wh). */
int dadet_box_match_encode(const float* proposals, int P, const float* gt_boxes, const int64_t* gt_labels, int G,
                           float high_threshold, float low_threshold, float wx, float wy, float ww, float wh,
                           int64_t* matched_idxs, int64_t* labels, float* regression_targets, void* stream);

/* Deformable position-sensitive ROI pooling — replaces the vendored tree's `_C.deform_psroi_pooling_forward /
 * _backward` (tools/cityscapes/maskrcnn_benchmark/csrc/vision.cpp:22-23, cuda/deform_pool_kernel_cuda.cu:30-264).
 * data [B][H][W][C] NHWC with C = output_dim*group_size^2; rois [R][5]; trans [R][num_classes*2][part][part]
 * (NULL with no_trans); out / top_count [R][P][P][output_dim] NHWC.  Backward accumulates grad_data and
 * grad_trans with atomics (caller zero-fills). */
int dadet_deform_psroi_pool_forward(const float* data, const float* rois, const float* trans, float* out,
                                    float* top_count, int B, int H, int W, int C, int R, int no_trans,
                                    float spatial_scale, int output_dim, int group_size, int pooled_size,
                                    int part_size, int sample_per_part, float trans_std, int num_classes,
                                    void* stream);
int dadet_deform_psroi_pool_backward(const float* grad_out, const float* top_count, const float* data,
                                     const float* rois, const float* trans, float* grad_data, float* grad_trans,
                                     int B, int H, int W, int C, int R, int no_trans, float spatial_scale,
                                     int output_dim, int group_size, int pooled_size, int part_size,
                                     int sample_per_part, float trans_std, int num_classes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / reduction helpers of the same path (all NHWC fp32).
 * ----------------------------------------------------------------------------------------------*/
/* g_out[m][c] = (y[m][c] > 0 ? g[m][c] : 0) ; g_scaled[m][c] = g_out * scale[c].  Either output may be
 * NULL; scale may be NULL (=1).  Outputs may alias g.   (ReLU + FrozenBN backward, batch_norm.py:19-24) */
int dadet_relu_bn_backward(const float* g, const float* y, const float* scale, float* g_out,
                           float* g_scaled, int64_t rows, int C, void* stream);
/* ... with max|g_out| / max|g_scaled| merged into the slots amax_out / amax_scaled (either may be NULL; zero them first):
 * both maps feed GEMMs (contraction mode 4, see dadet_conv_forward_scaled) */
int dadet_relu_bn_backward_m(const float* g, const float* y, const float* scale, float* g_out, float* g_scaled,
                             int64_t rows, int C, float* amax_out, float* amax_scaled, void* stream);
/* out[c] = sum_m g[m][c]  (bias gradient); workspace >= dadet_colsum_workspace_bytes */
int dadet_colsum_workspace_bytes(int64_t rows, int C, size_t* bytes_out);
int dadet_colsum_ld(const float* g, int ld, float* out, int64_t rows, int C, int accumulate, void* workspace,
                    size_t workspace_bytes, void* stream);   /* rows `ld` >= C floats apart; accumulate: out[c] += */
int dadet_colsum(const float* g, float* out, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                 void* stream);
/* y = x * scale[c] + bias[c]  (standalone FrozenBatchNorm2d, layers/batch_norm.py:19-24) */
int dadet_channel_affine(const float* x, const float* scale, const float* bias, float* y, int64_t rows,
                         int C, int relu, void* stream);
/* 3x3 stride-2 pad-1 max pool, NHWC (BaseStem, resnet.py:335); forward only (stem is frozen). */
int dadet_maxpool3x3s2_forward(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo,
                               void* stream);
/* global average pool over HW (nn.AvgPool2d(7) on 7x7 maps: roi_box_predictors.py:17,29; da_heads.py:89,403) */
int dadet_avgpool_forward(const float* x, float* y, int R, int HW, int C, void* stream);
int dadet_avgpool_backward(const float* gy, float* gx, int R, int HW, int C, void* stream);
/* NCHW 3-channel image -> NHWC4 zero-padded channel (stem input staging) */
int dadet_nchw3_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RPN proposal decode — replaces BoxCoder.decode + clip_to_image on the top-k anchors
 *   reference: modeling/box_coder.py:52-95, structures/bounding_box.py:214-224,
 *   modeling/rpn/inference.py:96-113.
 * For k in [0,K): a = topk_idx[k]; box = decode(deltas[a], anchors[a]) clipped to [0,im_w-1]x[0,im_h-1].
 * deltas is the RPN bbox_pred map in NHWC: [H][W][A*4] for one image => delta of anchor index
 * a = (h*W + w)*A + aa is at deltas[a*4 .. a*4+3] (same flattening as permute_and_flatten, rpn/utils.py:10-14).
 * ----------------------------------------------------------------------------------------------*/
int dadet_rpn_decode_clip(const float* deltas, const float* anchors, const int64_t* topk_idx, int K,
                          float wx, float wy, float ww, float wh, float xform_clip, float im_w,
                          float im_h, float* boxes_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused DA heads (reference: modeling/da_heads/da_heads.py:12-68,354-440, da_heads/loss.py:55-104,140-200,
 * layers/gradient_scalar_layer.py:4-13, layers/consistency_loss.py:3-27).
 *
 * Image-level domain classifier tail.  Given the hidden map t = relu(conv1_da(x)) [M][C1]
 * (M = num_images * rows_per_image, produced by dadet_conv_forward with its bias+ReLU epilogue):
 *   logit[m]          = sum_c t[m][c] * w2[c] + b2[0]                    (conv2_da, 1x1 -> 1 channel)
 *   sums[img][0]     += BCE-with-logits(logit[m], labels[img])           (da_heads/loss.py:95-97)
 *   sums[img][1]     += sigmoid(logit[m])                                (consistency_loss.py:12-14)
 * one wavefront per row, wave-reduced, one atomic per wavefront; the caller zeroes `sums` [num_images][2].
 * ----------------------------------------------------------------------------------------------*/
int dadet_da_img_head_loss_forward(const float* t, const float* w2, const float* b2, const float* labels,
                                   float* logits_out, float* sums_out, int num_images,
                                   int rows_per_image, int C1, void* stream);
/* backward.  coef [num_images][4] (device) = (a_bce_w, a_sig_w, a_bce_x, a_sig_x):
 *   g_logit_w = a_bce_w*(s-y) + a_sig_w*s*(1-s)  -> parameter gradients (g_w2, g_b2, and g_t_w for conv1_da's wgrad)
 *   g_logit_x = a_bce_x*(s-y) + a_sig_x*s*(1-s)  -> g_t_x for conv1_da's dgrad: the gradient-reversal
 *                                                 weights (GRL -w for the BCE path, +w for the consistency
 *                                                 path, da_heads.py:377-380) are folded into a_*_x.
 * g_w2 [C1] and g_b2 [1] are accumulated with atomics: the caller zeroes them.  g_t_x may be NULL. */
int dadet_da_img_head_loss_backward(const float* t, const float* w2, const float* logits,
                                    const float* labels, const float* coef, float* g_t_w, float* g_t_x,
                                    float* g_w2, float* g_b2, int num_images, int rows_per_image, int C1,
                                    void* stream);
/* the same with the four coefficients formed inside the kernel from the upstream gradients: g_bce [1] (gradient of the mean
 * BCE), g_mean_sig [num_images] or NULL (gradient of the per-image mean sigmoid), the adversarial reversal weight as a
 * device scalar (w_adv_dev, e.g. AdvGRL's adaptive weight) or, when NULL, the float w_adv; w_cst the consistency one */
int dadet_da_img_head_loss_backward_g(const float* t, const float* w2, const float* logits, const float* labels,
                                      const float* g_bce, const float* g_mean_sig, const float* w_adv_dev, float w_adv,
                                      float w_cst, float* g_t_w, float* g_t_x, float* g_w2, float* g_b2, int num_images,
                                      int rows_per_image, int C1, void* stream);
/* ... and with max|g_t_w| / max|g_t_x| merged into the slots amax_w / amax_x (each may be NULL; zero them first): the two
 * maps feed conv1_da's weight- and data-gradient GEMMs, whose mode-4 contraction wants their largest magnitudes (see
 * dadet_conv_forward_scaled) */
int dadet_da_img_head_loss_backward_gm(const float* t, const float* w2, const float* logits, const float* labels,
                                       const float* g_bce, const float* g_mean_sig, const float* w_adv_dev, float w_adv,
                                       float w_cst, float* g_t_w, float* g_t_x, float* g_w2, float* g_b2, int num_images,
                                       int rows_per_image, int C1, float* amax_w, float* amax_x, void* stream);
/* Domain-level triplet loss on NHWC maps [H][W][C] (one image each): L2 distance over the W axis with
 * eps, hinge with margin, loss_sum[0] += sum over (h,c) (the caller zeroes it and divides by H*C).
 * dist_out [H*C][2] keeps (d_ap, d_an) for the backward; g_scale[0] = upstream grad / (H*C).
 * reference: da_heads/loss.py:180-200 (nn.TripletMarginLoss(margin, p=2) on [1,C,H,W]). */
/* Instance-level domain classifier tail — replaces, per iteration, the last layer of `DAInsHead.forward`
 * (modeling/da_heads/da_heads.py:61-68), `F.binary_cross_entropy_with_logits` of the instance logits
 * (da_heads/loss.py:95-97) and `consistency_loss` (layers/consistency_loss.py:3-27) of the reference's TWO head passes
 * (behind GRL(-w) and GRL(+w), da_heads.py:421-424), forward and backward.  Rows of the adversarial pass ([R_bce]) and of
 * the consistency pass ([R_cst]) are stacked in h [R_bce + R_cst][C] (second hidden layer after dropout).
 *   forward : logits[r] = h[r] . w3 + b3;  sums[0] += sum_r<R_bce BCE(logits[r], labels[r]);
 *             sums[1] += sum over consistency rows j and levels l of |means[l][j < n_src ? 0 : 1] - sigmoid(logit)|
 *             (caller zero-fills sums; means [levels][2] = per-level mean sigmoid of the image head on image 0 / 1)
 *   backward: coef[0] / coef[1] (device) = d loss / d sums[0] / d sums[1];  g_z = gradient w.r.t. fc2's PRE-activation
 *             (dropout scale inv_keep and ReLU gate folded in); g_w3 / g_b3 / g_means are accumulated (caller zero-fills).
 * dadet_da_ins_dropout_rows: out[p][r] = h1[r] * masks[p][r] (the passes share the first layer up to its dropout mask).
 * dadet_da_ins_merge: backward through that shared layer — g_w = [h1 > 0] * sum_p masks[p] * g[p] (parameter gradients),
 *   g_x = [h1 > 0] * sum_p grl[p] * masks[p] * g[p] (towards the ROI features: the passes' gradient-reversal weights,
 *   layers/gradient_scalar_layer.py:4-13, device array grl[passes]). */
int dadet_da_ins_tail_forward(const float* h, const float* w3, const float* b3, const float* labels,
                              const float* means, float* logits, float* sums, int R_bce, int R_cst, int n_src,
                              int levels, int C, void* stream);
int dadet_da_ins_tail_backward(const float* h, const float* w3, const float* logits, const float* labels,
                               const float* means, const float* coef, float inv_keep, float* g_z, float* g_w3,
                               float* g_b3, float* g_means, int R_bce, int R_cst, int n_src, int levels, int C,
                               void* stream);
int dadet_da_ins_dropout_rows(const float* h1, const float* masks, float* out, int64_t numel_per_pass, int passes,
                              void* stream);
int dadet_da_ins_merge(const float* g, const float* masks, const float* h1, const float* grl, float* g_w, float* g_x,
                       int64_t numel_per_pass, int passes, void* stream);

int dadet_triplet_w_forward(const float* anchor, const float* positive, const float* negative, int H,
                            int W, int C, float margin, float eps, float* dist_out, float* loss_sum,
                            void* stream);
int dadet_triplet_w_backward(const float* anchor, const float* positive, const float* negative,
                             const float* dist, const float* g_scale, int H, int W, int C, float margin,
                             float eps, float* g_anchor, float* g_positive, float* g_negative,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor SGD with momentum — replaces torch.optim.SGD.step over one param group per tensor
 *   reference: solver/build.py:7-20, engine/trainer.py:237-239.
 *   d = g + wd*p ; buf = first ? d : mom*buf + d ; p -= lr*buf      (torch.optim.SGD, dampening 0)
 * Table entries are device pointers + per-tensor scalars, packed on the device by the caller.
 * ----------------------------------------------------------------------------------------------*/
typedef struct dadet_sgd_entry {
  float* p; const float* g; float* buf; int64_t numel; float lr; float weight_decay;
} dadet_sgd_entry;
int dadet_sgd_step(const dadet_sgd_entry* table_dev, int num_tensors, int64_t max_numel, float momentum,
                   int first_step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DADET_H_ */
