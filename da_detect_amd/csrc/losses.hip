// Fused detection losses: value AND gradient in one single-workgroup launch each (gfx950).
//
// The reference evaluates each of these as a chain of ATen calls (gather, log-softmax / BCE / smooth-L1, reductions,
// and the mirror chain in backward): ~80 launches of a few hundred elements for the RPN losses
// (modeling/rpn/loss.py:125-143) and ~60 for the Fast R-CNN losses (modeling/roi_heads/box_head/loss.py:165-221).
// They touch <= 512 sampled rows, so one 256-thread workgroup does the whole job: per-row terms, a deterministic
// block reduction (wave shuffle + LDS, fixed order), and the gradient rows scattered into zero-filled dense maps.
// Formulas: binary_cross_entropy_with_logits = max(x,0) - x*y + log1p(exp(-|x|)); smooth-L1 with beta
// (layers/smooth_l1_loss.py:6-16); cross_entropy = logsumexp(x) - x[label].
#include "common.h"

namespace dadet {

__device__ inline float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ inline float smooth_l1_term(float x, float t, float beta, float* grad) {
  const float d = x - t, n = fabsf(d);
  if (n < beta) {
    *grad = d / beta;
    return 0.5f * n * n / beta;
  }
  *grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return n - 0.5f * beta;
}

// objectness: flat logits (the NHWC map, index = ((n*H+h)*W+w)*A+a), box_regression: flat [.,4] (NHWC, 4A channels)
__global__ __launch_bounds__(256) void rpn_loss_kernel(const float* __restrict__ objectness,
                                                       const float* __restrict__ box_regression,
                                                       const int64_t* __restrict__ sampled_inds,
                                                       const float* __restrict__ labels_sampled, int S,
                                                       const int64_t* __restrict__ pos_inds,
                                                       const float* __restrict__ targets_pos, int Pn, float beta,
                                                       float* __restrict__ losses, float* __restrict__ g_obj,
                                                       float* __restrict__ g_reg) {
  __shared__ float red[4];
  const float inv = S > 0 ? 1.f / (float)S : 0.f;
  float bce = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) {
    const int64_t a = sampled_inds[i];
    const float x = objectness[a], y = labels_sampled[i];
    bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    g_obj[a] = (1.f / (1.f + expf(-x)) - y) * inv;
  }
  float box = 0.f;
  for (int i = threadIdx.x; i < Pn * 4; i += 256) {
    const int64_t a = pos_inds[i >> 2];
    float g;
    box += smooth_l1_term(box_regression[a * 4 + (i & 3)], targets_pos[i], beta, &g);
    g_reg[a * 4 + (i & 3)] = g * inv;
  }
  bce = block_sum_256(bce, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    losses[0] = bce * inv;   // mean over the sampled anchors
    losses[1] = box * inv;   // sum over positives / number of sampled anchors (rpn/loss.py:131-136)
  }
}

// class_logits [R][C], box_regression [R][4*Cb]; src rows (source-domain ROIs) with their labels; positives with their
// 4 regression columns (map_inds) and targets
__global__ __launch_bounds__(256) void frcnn_loss_kernel(const float* __restrict__ class_logits,
                                                         const float* __restrict__ box_regression, int C,
                                                         int reg_cols, const int64_t* __restrict__ src,
                                                         const int64_t* __restrict__ labels_src, int Ns,
                                                         const int64_t* __restrict__ rows_pos,
                                                         const int64_t* __restrict__ map_inds,
                                                         const float* __restrict__ targets_pos, int Pn,
                                                         float* __restrict__ losses, float* __restrict__ g_cls,
                                                         float* __restrict__ g_reg) {
  __shared__ float red[4];
  const float inv = Ns > 0 ? 1.f / (float)Ns : 0.f;
  float ce = 0.f;
  for (int i = threadIdx.x; i < Ns; i += 256) {
    const int64_t r = src[i];
    const float* x = class_logits + r * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(x[c] - m);
    const float lse = m + logf(z);
    const int lab = (int)labels_src[i];
    ce += lse - x[lab];
    for (int c = 0; c < C; ++c) g_cls[r * C + c] = (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * inv;
  }
  float box = 0.f;
  for (int i = threadIdx.x; i < Pn * 4; i += 256) {
    const int64_t r = rows_pos[i >> 2];
    const int64_t col = map_inds[i];
    float g;
    box += smooth_l1_term(box_regression[r * reg_cols + col], targets_pos[i], 1.f, &g);
    g_reg[r * reg_cols + col] = g * inv;
  }
  ce = block_sum_256(ce, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    losses[0] = ce * inv;    // mean over the source-domain rows
    losses[1] = box * inv;   // sum over positives / number of source-domain rows (box_head/loss.py:213-219)
  }
}

// the same losses from PER-ROW targets: loss_labels[r] < 0 = row outside the detection losses (target-domain image),
// else its class; reg_targets [R][4].  No index lists, so nothing has to be compacted on the way here.
__global__ __launch_bounds__(256) void frcnn_loss_rows_kernel(const float* __restrict__ class_logits,
                                                              const float* __restrict__ box_regression, int R, int C,
                                                              int reg_cols, const int64_t* __restrict__ loss_labels,
                                                              const float* __restrict__ reg_targets,
                                                              float* __restrict__ losses, float* __restrict__ g_cls,
                                                              float* __restrict__ g_reg) {
  __shared__ float red[4];
  __shared__ int s_rows;
  if (threadIdx.x == 0) s_rows = 0;
  __syncthreads();
  int mine = 0;
  for (int r = threadIdx.x; r < R; r += 256) mine += loss_labels[r] >= 0;
  if (mine) atomicAdd(&s_rows, mine);
  __syncthreads();
  const int Ns = s_rows;
  const float inv = Ns > 0 ? 1.f / (float)Ns : 0.f;
  float ce = 0.f, box = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const int lab = (int)loss_labels[r];
    if (lab < 0) continue;
    const float* x = class_logits + (size_t)r * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(x[c] - m);
    const float lse = m + logf(z);
    ce += lse - x[lab];
    for (int c = 0; c < C; ++c) g_cls[(size_t)r * C + c] = (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * inv;
    if (lab > 0) {
      const int col0 = reg_cols == 8 ? 4 : 4 * lab;   // class-agnostic regression keeps 2 x 4 columns
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g;
        box += smooth_l1_term(box_regression[(size_t)r * reg_cols + col0 + j], reg_targets[r * 4 + j], 1.f, &g);
        g_reg[(size_t)r * reg_cols + col0 + j] = g * inv;
      }
    }
  }
  ce = block_sum_256(ce, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    losses[0] = ce * inv;
    losses[1] = box * inv;
  }
}

// ---- RPN losses in ROW form -------------------------------------------------------------------------------------
// The RPN losses read the head's maps at the S sampled anchors only (<= 256 per labelled image), so the gradient of the
// maps is zero everywhere else: at most S of the N*H*W pixel rows of the head's backward GEMMs carry anything.  This
// kernel returns the gradient as S rows instead of two dense maps — row r belongs to anchor sampled_inds[r] = pixel * A + a
// and holds d loss / d (that pixel's A objectness + 4A regression outputs) restricted to anchor a — plus each row's pixel.
// Rows of the same pixel (two sampled anchors at one location) stay separate: everything downstream is linear in the
// rows.  The positives are the first Pn sampled rows (rpn/loss.py:116-118: cat([pos, neg])).  Loss values: the same
// arithmetic and reduction order as rpn_loss_kernel.
__global__ __launch_bounds__(256) void rpn_loss_rows_kernel(const float* __restrict__ objectness,
                                                            const float* __restrict__ box_regression,
                                                            const int64_t* __restrict__ sampled_inds,
                                                            const float* __restrict__ labels_sampled, int S, int Pn,
                                                            const float* __restrict__ targets_pos, int A, float beta,
                                                            float* __restrict__ losses, float* __restrict__ G, int ldg,
                                                            int* __restrict__ pixels, int64_t per_image, int64_t level_off,
                                                            int64_t level_cnt, int level_id, int* __restrict__ row_level,
                                                            int keep_rows) {
  // per_image > 0: ONE LEVEL of a feature pyramid.  sampled_inds index the concatenation over levels the losses are
  // defined on (concat_box_prediction_layers, rpn/utils.py:10-45: image-major, then level, then (h, w, a)); the maps
  // given here are this level's, an anchor of another level leaves its row zero and its pixel -1, and the sum of the
  // levels' launches is the loss (all of them divide by the same S)
  // keep_rows: G / pixels / row_level are SHARED by the launches of all levels (the caller zero-fills G once); this launch
  // writes the rows of its own anchors only and tags them with level_id — every sampled anchor lies on exactly one level
  __shared__ float red[4];
  const float inv = S > 0 ? 1.f / (float)S : 0.f;
  if (!keep_rows) {
    for (int i = threadIdx.x; i < S * ldg; i += 256) G[i] = 0.f;
    __syncthreads();
  }
  auto local = [&](int64_t idx) -> int64_t {
    if (per_image <= 0) return idx;
    const int64_t img = idx / per_image, rem = idx - img * per_image - level_off;
    return (rem >= 0 && rem < level_cnt) ? img * level_cnt + rem : -1;
  };
  float bce = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) {
    const int64_t idx = local(sampled_inds[i]);
    if (idx < 0) {
      if (!keep_rows) pixels[i] = -1;
      continue;
    }
    if (row_level) row_level[i] = level_id;
    const int a = (int)(idx % A);
    const float x = objectness[idx], y = labels_sampled[i];
    bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    G[(size_t)i * ldg + a] = (1.f / (1.f + expf(-x)) - y) * inv;
    pixels[i] = (int)(idx / A);
  }
  float box = 0.f;
  for (int i = threadIdx.x; i < Pn * 4; i += 256) {
    const int r = i >> 2, j = i & 3;
    const int64_t idx = local(sampled_inds[r]);
    if (idx < 0) continue;
    const int a = (int)(idx % A);
    float g;
    box += smooth_l1_term(box_regression[idx * 4 + j], targets_pos[i], beta, &g);
    G[(size_t)r * ldg + A + 4 * a + j] = g * inv;
  }
  bce = block_sum_256(bce, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    losses[0] = bce * inv;
    losses[1] = box * inv;
  }
}

// out[r][tap][c] = x[pixel_r + offset(tap)][c], 0 outside the image: the K-contiguous operand rows ("im2col" of S pixels)
// of a KH x KW stride-1 convolution's weight gradient; KH = KW = 1: plain row gather
__global__ __launch_bounds__(256) void gather_pixel_taps_kernel(const float4* __restrict__ x, const int* __restrict__ pixels,
                                                                int H, int W, int C4, int KH, int KW, int pad,
                                                                float4* __restrict__ out, const int* __restrict__ row_level,
                                                                int level) {
  const int r = blockIdx.x / (KH * KW), tap = blockIdx.x % (KH * KW);
  if (row_level && row_level[r] != level) return;      // a row of another pyramid level: written by that level's launch
  const int p = pixels[r];                  // -1: a row that belongs to another pyramid level (zeros)
  const int n = p / (H * W), rem = p - n * H * W;
  const int h = rem / W + tap / KW - pad, w = rem % W + tap % KW - pad;
  const bool inside = p >= 0 && h >= 0 && h < H && w >= 0 && w < W;
  const float4* src = x + ((size_t)(n * H + h) * W + w) * C4;
  float4* dst = out + (size_t)blockIdx.x * C4;
  for (int c = threadIdx.x; c < C4; c += 256) dst[c] = inside ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// dx[pixel_r + offset(tap)][c] += y[r][tap][c]: the data gradient of the same convolution from its S non-zero output rows
// (dx zero-filled by the caller; rows of neighbouring or equal pixels meet on the same dx row: hardware fp32 atomics)
__global__ __launch_bounds__(256) void scatter_pixel_taps_add_kernel(const float* __restrict__ y,
                                                                     const int* __restrict__ pixels, int H, int W, int C,
                                                                     int KH, int KW, int pad, float* __restrict__ dx,
                                                                     const int* __restrict__ row_level, int level) {
  const int r = blockIdx.x / (KH * KW), tap = blockIdx.x % (KH * KW);
  if (row_level && row_level[r] != level) return;
  const int p = pixels[r];
  if (p < 0) return;
  const int n = p / (H * W), rem = p - n * H * W;
  const int h = rem / W + tap / KW - pad, w = rem % W + tap % KW - pad;
  if (h < 0 || h >= H || w < 0 || w >= W) return;
  const float* src = y + (size_t)blockIdx.x * C;
  float* dst = dx + ((size_t)(n * H + h) * W + w) * C;
  for (int c = threadIdx.x; c < C; c += 256) atomicAdd(dst + c, src[c]);
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_rpn_loss(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                              const float* labels_sampled, int num_sampled, const int64_t* pos_inds,
                              const float* regression_targets_pos, int num_pos, float beta, float* losses_out,
                              float* grad_objectness, float* grad_box_regression, void* stream) {
  DADET_REQUIRE(num_sampled >= 0 && num_pos >= 0, "rpn_loss: negative count");
  DADET_REQUIRE(objectness && box_regression && losses_out && grad_objectness && grad_box_regression,
                "rpn_loss: null pointer");
  DADET_REQUIRE(num_sampled == 0 || (sampled_inds && labels_sampled), "rpn_loss: null sample arrays");
  DADET_REQUIRE(num_pos == 0 || (pos_inds && regression_targets_pos), "rpn_loss: null positive arrays");
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), objectness, box_regression,
                     sampled_inds, labels_sampled, num_sampled, pos_inds, regression_targets_pos, num_pos, beta,
                     losses_out, grad_objectness, grad_box_regression);
  return check_launch("rpn_loss");
}

extern "C" int dadet_fast_rcnn_loss(const float* class_logits, const float* box_regression, int num_classes,
                                    int reg_cols, const int64_t* src_rows, const int64_t* labels_src, int num_src,
                                    const int64_t* rows_pos, const int64_t* map_inds,
                                    const float* regression_targets_pos, int num_pos, float* losses_out,
                                    float* grad_class_logits, float* grad_box_regression, void* stream) {
  DADET_REQUIRE(num_src >= 0 && num_pos >= 0 && num_classes > 0 && reg_cols > 0, "fast_rcnn_loss: bad counts");
  DADET_REQUIRE(class_logits && box_regression && losses_out && grad_class_logits && grad_box_regression,
                "fast_rcnn_loss: null pointer");
  DADET_REQUIRE(num_src == 0 || (src_rows && labels_src), "fast_rcnn_loss: null source arrays");
  DADET_REQUIRE(num_pos == 0 || (rows_pos && map_inds && regression_targets_pos), "fast_rcnn_loss: null positives");
  hipLaunchKernelGGL(frcnn_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), class_logits, box_regression,
                     num_classes, reg_cols, src_rows, labels_src, num_src, rows_pos, map_inds,
                     regression_targets_pos, num_pos, losses_out, grad_class_logits, grad_box_regression);
  return check_launch("fast_rcnn_loss");
}

extern "C" int dadet_fast_rcnn_loss_rows(const float* class_logits, const float* box_regression, int num_rows,
                                         int num_classes, int reg_cols, const int64_t* loss_labels,
                                         const float* regression_targets, float* losses_out, float* grad_class_logits,
                                         float* grad_box_regression, void* stream) {
  DADET_REQUIRE(num_rows >= 0 && num_classes > 0 && reg_cols > 0, "fast_rcnn_loss_rows: bad counts");
  DADET_REQUIRE(reg_cols == 8 || reg_cols == 4 * num_classes,
                "fast_rcnn_loss_rows: %d regression columns for %d classes", reg_cols, num_classes);
  DADET_REQUIRE(losses_out && grad_class_logits && grad_box_regression, "fast_rcnn_loss_rows: null output");
  DADET_REQUIRE(num_rows == 0 || (class_logits && box_regression && loss_labels && regression_targets),
                "fast_rcnn_loss_rows: null input");
  hipLaunchKernelGGL(frcnn_loss_rows_kernel, dim3(1), dim3(256), 0, as_stream(stream), class_logits, box_regression,
                     num_rows, num_classes, reg_cols, loss_labels, regression_targets, losses_out, grad_class_logits,
                     grad_box_regression);
  return check_launch("fast_rcnn_loss_rows");
}

static int rpn_loss_rows_impl(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                              const float* labels_sampled, int num_sampled, int num_pos,
                              const float* regression_targets_pos, int anchors_per_location, float beta,
                              float* losses_out, float* grad_rows, int ldg, int* pixels_out, int64_t per_image,
                              int64_t level_off, int64_t level_cnt, int level_id, int* row_level_out, int keep_rows,
                              void* stream) {
  DADET_REQUIRE(num_sampled > 0 && num_pos >= 0 && num_pos <= num_sampled && anchors_per_location > 0,
                "rpn_loss_rows: bad counts (%d sampled, %d positive)", num_sampled, num_pos);
  DADET_REQUIRE(ldg >= 5 * anchors_per_location, "rpn_loss_rows: ldg=%d < 5 * %d", ldg, anchors_per_location);
  DADET_REQUIRE(objectness && box_regression && sampled_inds && labels_sampled && losses_out && grad_rows && pixels_out,
                "rpn_loss_rows: null pointer");
  DADET_REQUIRE(num_pos == 0 || regression_targets_pos, "rpn_loss_rows: null regression targets");
  DADET_REQUIRE(per_image == 0 || (per_image > 0 && level_off >= 0 && level_cnt > 0 && level_off + level_cnt <= per_image &&
                                   level_cnt % anchors_per_location == 0),
                "rpn_loss_rows: bad level window");
  hipLaunchKernelGGL(rpn_loss_rows_kernel, dim3(1), dim3(256), 0, as_stream(stream), objectness, box_regression,
                     sampled_inds, labels_sampled, num_sampled, num_pos, regression_targets_pos, anchors_per_location,
                     beta, losses_out, grad_rows, ldg, pixels_out, per_image, level_off, level_cnt, level_id, row_level_out,
                     keep_rows);
  return check_launch("rpn_loss_rows");
}

extern "C" int dadet_rpn_loss_rows(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                                   const float* labels_sampled, int num_sampled, int num_pos,
                                   const float* regression_targets_pos, int anchors_per_location, float beta,
                                   float* losses_out, float* grad_rows, int ldg, int* pixels_out, void* stream) {
  return rpn_loss_rows_impl(objectness, box_regression, sampled_inds, labels_sampled, num_sampled, num_pos,
                            regression_targets_pos, anchors_per_location, beta, losses_out, grad_rows, ldg, pixels_out, 0,
                            0, 0, 0, nullptr, 0, stream);
}

extern "C" int dadet_rpn_loss_rows_level(const float* objectness, const float* box_regression, const int64_t* sampled_inds,
                                         const float* labels_sampled, int num_sampled, int num_pos,
                                         const float* regression_targets_pos, int anchors_per_location, float beta,
                                         int64_t anchors_per_image, int64_t level_offset, int64_t level_anchors,
                                         int level_id, int* row_level_out, int shared_rows, float* losses_out,
                                         float* grad_rows, int ldg, int* pixels_out, void* stream) {
  DADET_REQUIRE(anchors_per_image > 0, "rpn_loss_rows_level: anchors_per_image must be positive");
  DADET_REQUIRE(!shared_rows || row_level_out, "rpn_loss_rows_level: shared rows need row_level_out");
  return rpn_loss_rows_impl(objectness, box_regression, sampled_inds, labels_sampled, num_sampled, num_pos,
                            regression_targets_pos, anchors_per_location, beta, losses_out, grad_rows, ldg, pixels_out,
                            anchors_per_image, level_offset, level_anchors, level_id, row_level_out, shared_rows, stream);
}

static int gather_pixel_taps_impl(const float* x, const int* pixels, int num_rows, int N, int H, int W, int C, int KH,
                                  int KW, int pad, float* out, const int* row_level, int level, void* stream) {
  DADET_REQUIRE(num_rows >= 0 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && KH > 0 && KW > 0 && pad >= 0,
                "gather_pixel_taps: bad dims (C=%d must be a multiple of 4)", C);
  if (num_rows == 0) return DADET_OK;
  DADET_REQUIRE(x && pixels && out && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "gather_pixel_taps: null or unaligned pointer");
  hipLaunchKernelGGL(gather_pixel_taps_kernel, dim3(num_rows * KH * KW), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), pixels, H, W, C / 4, KH, KW, pad, reinterpret_cast<float4*>(out),
                     row_level, level);
  return check_launch("gather_pixel_taps");
}

extern "C" int dadet_gather_pixel_taps(const float* x, const int* pixels, int num_rows, int N, int H, int W, int C, int KH,
                                       int KW, int pad, float* out, void* stream) {
  return gather_pixel_taps_impl(x, pixels, num_rows, N, H, W, C, KH, KW, pad, out, nullptr, 0, stream);
}

extern "C" int dadet_gather_pixel_taps_level(const float* x, const int* pixels, const int* row_level, int level, int num_rows,
                                             int N, int H, int W, int C, int KH, int KW, int pad, float* out, void* stream) {
  DADET_REQUIRE(row_level, "gather_pixel_taps_level: null row_level");
  return gather_pixel_taps_impl(x, pixels, num_rows, N, H, W, C, KH, KW, pad, out, row_level, level, stream);
}

static int scatter_pixel_taps_add_impl(const float* y, const int* pixels, int num_rows, int N, int H, int W, int C, int KH,
                                       int KW, int pad, float* dx, const int* row_level, int level, void* stream) {
  DADET_REQUIRE(num_rows >= 0 && N > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && pad >= 0,
                "scatter_pixel_taps_add: bad dims");
  if (num_rows == 0) return DADET_OK;
  DADET_REQUIRE(y && pixels && dx, "scatter_pixel_taps_add: null pointer");
  hipLaunchKernelGGL(scatter_pixel_taps_add_kernel, dim3(num_rows * KH * KW), dim3(256), 0, as_stream(stream), y, pixels,
                     H, W, C, KH, KW, pad, dx, row_level, level);
  return check_launch("scatter_pixel_taps_add");
}

extern "C" int dadet_scatter_pixel_taps_add(const float* y, const int* pixels, int num_rows, int N, int H, int W, int C,
                                            int KH, int KW, int pad, float* dx, void* stream) {
  return scatter_pixel_taps_add_impl(y, pixels, num_rows, N, H, W, C, KH, KW, pad, dx, nullptr, 0, stream);
}

extern "C" int dadet_scatter_pixel_taps_add_level(const float* y, const int* pixels, const int* row_level, int level,
                                                  int num_rows, int N, int H, int W, int C, int KH, int KW, int pad,
                                                  float* dx, void* stream) {
  DADET_REQUIRE(row_level, "scatter_pixel_taps_add_level: null row_level");
  return scatter_pixel_taps_add_impl(y, pixels, num_rows, N, H, W, C, KH, KW, pad, dx, row_level, level, stream);
}
