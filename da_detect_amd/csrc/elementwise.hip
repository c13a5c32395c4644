// HBM-bound helpers of the DA Faster R-CNN training path (NHWC fp32, gfx950).
// Every kernel moves 16 B per lane with lanes along the channel axis and a capped grid-stride launch.
// Each entry point cites the reference code it stands in for.
#include "box_match.h"
#include "conv_common.h"   // amax_publish (contraction mode 4)
#include <float.h>

namespace dadet {

static inline int stream_blocks(int64_t work_items, int threads) {
  int64_t b = ceil_div64(work_items, threads);
  if (b > kMaxStreamBlocks) b = kMaxStreamBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- ReLU (+ FrozenBN scale) backward -------------------------------------------------------
// reference: F.relu_ + FrozenBatchNorm2d.forward (layers/batch_norm.py:19-24) differentiated by autograd.
__global__ void relu_bn_backward_kernel(const float4* __restrict__ g, const float4* __restrict__ y,
                                        const float4* __restrict__ scale, float4* __restrict__ g_out,
                                        float4* __restrict__ g_scaled, int64_t total4, int C4,
                                        unsigned* __restrict__ amax_out, unsigned* __restrict__ amax_scaled) {
  float mo = 0.f, ms = 0.f;     // contraction mode 4: max|g_out|, max|g_scaled| (slots; both feed GEMMs)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 gv = g[i];
    float4 r = gv;
    if (y) {
      const float4 yv = y[i];
      r.x = yv.x > 0.f ? gv.x : 0.f;
      r.y = yv.y > 0.f ? gv.y : 0.f;
      r.z = yv.z > 0.f ? gv.z : 0.f;
      r.w = yv.w > 0.f ? gv.w : 0.f;
    }
    if (g_out) g_out[i] = r;
    mo = fmaxf(fmaxf(mo, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    if (g_scaled) {
      float4 s = r;
      if (scale) {
        const float4 sc = scale[i % C4];
        s.x *= sc.x; s.y *= sc.y; s.z *= sc.z; s.w *= sc.w;
      }
      g_scaled[i] = s;
      ms = fmaxf(fmaxf(ms, fmaxf(fabsf(s.x), fabsf(s.y))), fmaxf(fabsf(s.z), fabsf(s.w)));
    }
  }
  if (amax_out) amax_publish(amax_out, mo);
  if (amax_scaled) amax_publish(amax_scaled, ms);
}

// ---- column sum (bias gradient) -------------------------------------------------------------
// partial[s][c] = sum over the s-th row slab; deterministic two-pass.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g,
                                                             float* __restrict__ partial, int64_t rows,
                                                             int C, int64_t rows_per_split, int cb, int ld) {
  // block: cb columns x (256 / cb) row lanes, cb = 64, 32 or 16 — narrow matrices (the 20 / 28 offset channels of a
  // deformable conv, 8 .. 64 k rows) keep all 256 lanes busy instead of 20 of every 64
  __shared__ float red[256];
  const int lanes = 256 / cb;
  const int cl = threadIdx.x % cb, rl = threadIdx.x / cb;
  const int col = blockIdx.x * cb + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  int64_t r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (col < C)
    for (int64_t r = r0 + rl; r < r1; r += lanes) acc += g[r * ld + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int half = lanes / 2; half > 0; half >>= 1) {      // fixed-order tree over the row lanes
    if (rl < half) red[threadIdx.x] += red[threadIdx.x + half * cb];
    __syncthreads();
  }
  if (rl == 0 && col < C) partial[(size_t)blockIdx.y * C + col] = red[cl];
}
// one workgroup per column: the `splits` partial sums in a fixed-order tree (was: one thread per column walking up to 1024
// partials — 49 us for the 20-column case)
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C,
                                                          int splits, int accumulate) {
  const int c = blockIdx.x;
  float acc = 0.f;
  for (int s = threadIdx.x; s < splits; s += 64) acc += partial[(size_t)s * C + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + acc : acc;
}
static int colsum_cb(int C) { return C > 32 ? 64 : (C > 16 ? 32 : 16); }
static int colsum_splits(int64_t rows, int C) {
  const int col_blocks = ceil_div(C, colsum_cb(C));
  int64_t want = ceil_div64(kNumCU * 4, col_blocks);
  int64_t max_by_rows = ceil_div64(rows, 64);
  if (want > max_by_rows) want = max_by_rows;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

// ---- per-channel affine (standalone FrozenBatchNorm2d) ---------------------------------------
__global__ void channel_affine_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                      const float4* __restrict__ bias, float4* __restrict__ y,
                                      int64_t total4, int C4, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const float4 xv = x[i], s = scale[c], b = bias[c];
    float4 r;
    r.x = xv.x * s.x + b.x; r.y = xv.y * s.y + b.y; r.z = xv.z * s.z + b.z; r.w = xv.w * s.w + b.w;
    if (relu) {
      r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
    }
    y[i] = r;
  }
}

// ---- 3x3 / stride 2 / pad 1 max pool (BaseStem, resnet.py:335) -------------------------------
__global__ void maxpool3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N, int H,
                                    int W, int C4, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t p = i / C4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 - 1 + s;
        if (wi < 0 || wi >= W) continue;
        const float4 v = x[((int64_t)(n * H + hi) * W + wi) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    y[i] = m;
  }
}

// ---- global average pool over HW (nn.AvgPool2d(7) on 7x7 maps) --------------------------------
// sums in raster order like ATen's avg_pool2d CPU kernel (sum then divide by the window size).
__global__ void avgpool_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, int R, int HW,
                                   int C4) {
  const int64_t total = (int64_t)R * C4;
  const float inv = (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const int64_t r = i / C4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < HW; ++p) {
      const float4 v = x[(r * HW + p) * C4 + c];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    a.x /= inv; a.y /= inv; a.z /= inv; a.w /= inv;
    y[i] = a;
  }
}
__global__ void avgpool_bwd_kernel(const float4* __restrict__ gy, float4* __restrict__ gx, int R, int HW,
                                   int C4) {
  const int64_t total = (int64_t)R * HW * C4;
  const float inv = (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const int64_t r = i / ((int64_t)HW * C4);
    float4 g = gy[r * C4 + c];
    g.x /= inv; g.y /= inv; g.z /= inv; g.w /= inv;
    gx[i] = g;
  }
}

// ---- NCHW(3) -> NHWC(4) staging for the stem -------------------------------------------------
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ y, int N,
                                      int64_t HW) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / HW, p = i % HW;
    const float* b = x + n * 3 * HW + p;
    y[i] = make_float4(b[0], b[HW], b[2 * HW], 0.f);
  }
}

// ---- RPN decode + clip ----------------------------------------------------------------------
// reference: BoxCoder.decode (modeling/box_coder.py:52-95) then BoxList.clip_to_image
// (structures/bounding_box.py:214-224), in the reference's operation order.
__global__ void rpn_decode_clip_kernel(const float4* __restrict__ deltas, const float4* __restrict__ anchors,
                                       const int64_t* __restrict__ topk_idx, int K, float wx, float wy,
                                       float ww, float wh, float xform_clip, float im_w, float im_h,
                                       float4* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t a = topk_idx[k];
  const float4 b = anchors[a];
  const float4 d = deltas[a];
  const float width = b.z - b.x + 1.f;
  const float height = b.w - b.y + 1.f;
  const float ctr_x = b.x + 0.5f * width;
  const float ctr_y = b.y + 0.5f * height;
  const float dx = d.x / wx, dy = d.y / wy;
  float dw = d.z / ww, dh = d.w / wh;
  dw = fminf(dw, xform_clip);
  dh = fminf(dh, xform_clip);
  const float pcx = dx * width + ctr_x;
  const float pcy = dy * height + ctr_y;
  const float pw = expf(dw) * width;
  const float ph = expf(dh) * height;
  float x1 = pcx - 0.5f * pw;
  float y1 = pcy - 0.5f * ph;
  float x2 = pcx + 0.5f * pw - 1.f;
  float y2 = pcy + 0.5f * ph - 1.f;
  x1 = fminf(fmaxf(x1, 0.f), im_w - 1.f);
  y1 = fminf(fmaxf(y1, 0.f), im_h - 1.f);
  x2 = fminf(fmaxf(x2, 0.f), im_w - 1.f);
  y2 = fminf(fmaxf(y2, 0.f), im_h - 1.f);
  out[k] = make_float4(x1, y1, x2, y2);
}

// ---- RPN anchor labelling: IoU vs ground truth -> Matcher WITH low-quality matches -> labels -> encode ------------
// reference chain: boxlist_iou (structures/boxlist_ops.py:56-91), Matcher(0.7, 0.3, allow_low_quality_matches=True)
// (modeling/matcher.py:42-112), RPNLossComputation.prepare_targets label rules (modeling/rpn/loss.py:78-96: matched -> 1,
// below-low -> 0, outside the image -> -1, between thresholds -> -1), BoxCoder(1,1,1,1).encode.  Two passes over the
// anchors: (1) best IoU of every ground-truth box (atomicMax on the float's bit pattern, IoU >= 0), (2) the per-anchor
// decision, where an anchor whose IoU with some box EQUALS that box's best keeps its argmax ("low-quality match").
// Both passes evaluate the IoU with the same instruction sequence, so the equality test is exact.
__device__ inline float iou_ref_order(const float4 g, float garea, const float4 p, float parea) {
  const float w = fmaxf(fminf(g.z, p.z) - fmaxf(g.x, p.x) + 1.f, 0.f);
  const float h = fmaxf(fminf(g.w, p.w) - fmaxf(g.y, p.y) + 1.f, 0.f);
  const float inter = w * h;
  return inter / (garea + parea - inter);
}

__global__ void rpn_gt_best_kernel(const float4* __restrict__ anchors, int A, const float4* __restrict__ gts, int G,
                                   unsigned* __restrict__ best_bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* g_box = reinterpret_cast<float4*>(smem);
  float* g_area = reinterpret_cast<float*>(g_box + G);
  unsigned* g_best = reinterpret_cast<unsigned*>(g_area + G);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const float4 b = gts[g];
    g_box[g] = b;
    g_area[g] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
    g_best[g] = 0u;
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A; i += gridDim.x * blockDim.x) {
    const float4 p = anchors[i];
    const float area = (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
    for (int g = 0; g < G; ++g) {
      const float v = iou_ref_order(g_box[g], g_area[g], p, area);
      if (v > 0.f) atomicMax(&g_best[g], __float_as_uint(v));
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    if (g_best[g]) atomicMax(&best_bits[g], g_best[g]);
}

__global__ void rpn_anchor_targets_kernel(const float4* __restrict__ anchors, const unsigned char* __restrict__ visible,
                                          int A, const float4* __restrict__ gts, int G,
                                          const unsigned* __restrict__ best_bits, float high, float low,
                                          float* __restrict__ labels, float4* __restrict__ targets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* g_box = reinterpret_cast<float4*>(smem);
  float* g_area = reinterpret_cast<float*>(g_box + G);
  float* g_best = g_area + G;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const float4 b = gts[g];
    g_box[g] = b;
    g_area[g] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
    g_best[g] = __uint_as_float(best_bits[g]);
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A) return;
  const float4 p = anchors[i];
  const float area = (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
  float best = -1.f;
  int arg = 0;
  bool low_quality = false;
  for (int g = 0; g < G; ++g) {
    const float v = iou_ref_order(g_box[g], g_area[g], p, area);
    if (v > best) {
      best = v;
      arg = g;
    }
    low_quality = low_quality || (v == g_best[g]);
  }
  int m = arg;
  if (best < low) m = -1;
  else if (best < high) m = -2;
  if (low_quality) m = arg;
  float lab = m >= 0 ? 1.f : 0.f;
  if (m == -1) lab = 0.f;
  if (!visible[i]) lab = -1.f;
  if (m == -2) lab = -1.f;
  labels[i] = lab;
  const float4 r = g_box[m < 0 ? 0 : m];
  const float ew = p.z - p.x + 1.f, eh = p.w - p.y + 1.f;
  const float ecx = p.x + 0.5f * ew, ecy = p.y + 0.5f * eh;
  const float gw = r.z - r.x + 1.f, gh = r.w - r.y + 1.f;
  const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
  targets[i] = make_float4((gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh));
}

// ---- box-head target assignment: IoU vs ground truth -> Matcher -> labels -> BoxCoder.encode, one launch --------
// reference chain (each step a handful of ATen launches there): boxlist_iou (structures/boxlist_ops.py:56-91), Matcher
// without low-quality matches (modeling/matcher.py:42-92), label rules of FastRCNNLossComputation.prepare_targets
// (roi_heads/box_head/loss.py:69-93: below-low -> 0, between thresholds -> -1) and BoxCoder.encode
// (modeling/box_coder.py:22-50).  Operation order is the reference's (contraction is off for this file), first
// maximum wins on equal IoU.  One thread per proposal; the G ground-truth boxes sit in LDS.
__global__ void box_match_encode_kernel(const float4* __restrict__ props, int P, const float4* __restrict__ gts,
                                        const int64_t* __restrict__ gt_labels, int G, float high, float low,
                                        float wx, float wy, float ww, float wh, int64_t* __restrict__ matched,
                                        int64_t* __restrict__ labels, float4* __restrict__ targets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* g_box = reinterpret_cast<float4*>(smem);
  float* g_area = reinterpret_cast<float*>(g_box + G);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const float4 b = gts[g];
    g_box[g] = b;
    g_area[g] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float4 reg;
  int64_t m;
  labels[i] = match_encode_one(props[i], g_box, g_area, gt_labels, G, high, low, wx, wy, ww, wh, &reg, &m);
  matched[i] = m;
  targets[i] = reg;
}

// ---- sigmoid focal loss (reference: csrc/cuda/SigmoidFocalLoss_cuda.cu:21-101) ----------------
__global__ void focal_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                 float* __restrict__ losses, int64_t total, int C, float gamma,
                                 float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / C), d = (int)(i % C);
    const int t = targets[n];
    const float c1 = (t == (d + 1)) ? 1.f : 0.f;
    const float c2 = ((t >= 0) & (t != (d + 1))) ? 1.f : 0.f;
    const float zn = 1.f - alpha, zp = alpha;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
    const float xpos = (x >= 0.f) ? 1.f : 0.f;
    const float term2 = powf(p, gamma) * (-1.f * x * xpos - logf(1.f + expf(x - 2.f * x * xpos)));
    float l = 0.f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}
__global__ void focal_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                 const float* __restrict__ d_losses, float* __restrict__ d_logits,
                                 int64_t total, int C, float gamma, float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / C), d = (int)(i % C);
    const int t = targets[n];
    const float c1 = (t == (d + 1)) ? 1.f : 0.f;
    const float c2 = ((t >= 0) & (t != (d + 1))) ? 1.f : 0.f;
    const float zn = 1.f - alpha, zp = alpha;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, FLT_MIN))));
    const float xpos = (x >= 0.f) ? 1.f : 0.f;
    const float term2 =
        powf(p, gamma) * ((-1.f * x * xpos - logf(1.f + expf(x - 2.f * x * xpos))) * (1.f - p) * gamma - p);
    float g = 0.f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}

// ---- fused multi-tensor SGD -------------------------------------------------------------------
// reference: torch.optim.SGD.step over solver/build.py:7-20's one-group-per-tensor list.
// grid.y = tensor, grid.x strides over that tensor's elements in units of 1024 float4 per workgroup (4 per lane, their
// 12 loads issued before the first store).  grid.x is sized for the LARGEST tensor (up to 512): the workgroups a smaller
// tensor does not need leave at once.  (It was capped at 64 per tensor: the RPN conv's 9.4 M weights — 190 MB of the
// step's 730 — were then walked by 64 workgroups, an eighth of the chip's slots, one float4 per lane in flight.)
__global__ __launch_bounds__(256) void sgd_kernel(const dadet_sgd_entry* __restrict__ table, float momentum, int first_step,
                                                  float grad_scale) {
  const dadet_sgd_entry e = table[blockIdx.y];
  const int64_t n = e.numel;
  const bool vec = ((reinterpret_cast<uintptr_t>(e.p) | reinterpret_cast<uintptr_t>(e.g) |
                     reinterpret_cast<uintptr_t>(e.buf)) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  if ((int64_t)blockIdx.x * 1024 >= n4 && ((int64_t)blockIdx.x * 256 >= n - n4 * 4)) return;
  const float lr = e.lr, wd = e.weight_decay;
  float4* p4 = reinterpret_cast<float4*>(e.p);
  const float4* g4 = reinterpret_cast<const float4*>(e.g);
  float4* b4 = reinterpret_cast<float4*>(e.buf);
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += (int64_t)gridDim.x * 1024) {
    float4 p[4], g[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) {
        p[u] = p4[i];
        g[u] = g4[i];
        if (!first_step) b[u] = b4[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * 256 + threadIdx.x;
      if (i >= n4) continue;
      float4 d;
      d.x = g[u].x * grad_scale + wd * p[u].x; d.y = g[u].y * grad_scale + wd * p[u].y;
      d.z = g[u].z * grad_scale + wd * p[u].z; d.w = g[u].w * grad_scale + wd * p[u].w;
      float4 nb;
      if (first_step) {
        nb = d;
      } else {
        nb.x = momentum * b[u].x + d.x; nb.y = momentum * b[u].y + d.y;
        nb.z = momentum * b[u].z + d.z; nb.w = momentum * b[u].w + d.w;
      }
      b4[i] = nb;
      float4 q = p[u];
      q.x -= lr * nb.x; q.y -= lr * nb.y; q.z -= lr * nb.z; q.w -= lr * nb.w;
      p4[i] = q;
    }
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float p = e.p[i];
    const float d = e.g[i] * grad_scale + wd * p;
    const float b = first_step ? d : momentum * e.buf[i] + d;
    e.buf[i] = b;
    e.p[i] = p - lr * b;
  }
}

}  // namespace dadet

using namespace dadet;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int dadet_relu_bn_backward(const float* g, const float* y, const float* scale, float* g_out,
                                      float* g_scaled, int64_t rows, int C, void* stream) {
  return dadet_relu_bn_backward_m(g, y, scale, g_out, g_scaled, rows, C, nullptr, nullptr, stream);
}

extern "C" int dadet_relu_bn_backward_m(const float* g, const float* y, const float* scale, float* g_out,
                                        float* g_scaled, int64_t rows, int C, float* amax_out, float* amax_scaled,
                                        void* stream) {
  DADET_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0, "relu_bn_backward: C=%d must be a multiple of 4", C);
  if (rows == 0) return DADET_OK;
  DADET_REQUIRE(g && (g_out || g_scaled), "relu_bn_backward: null pointer");
  DADET_REQUIRE(aligned16(g) && aligned16(y) && aligned16(scale) && aligned16(g_out) && aligned16(g_scaled),
                "relu_bn_backward: pointers must be 16-byte aligned");
  const int64_t total4 = rows * (C / 4);
  hipLaunchKernelGGL(relu_bn_backward_kernel, dim3(stream_blocks(total4, 256)), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float4*>(g),
                     reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(scale),
                     reinterpret_cast<float4*>(g_out), reinterpret_cast<float4*>(g_scaled), total4, C / 4,
                     reinterpret_cast<unsigned*>(g_out ? amax_out : nullptr),
                     reinterpret_cast<unsigned*>(g_scaled ? amax_scaled : nullptr));
  return check_launch("relu_bn_backward");
}

extern "C" int dadet_colsum_workspace_bytes(int64_t rows, int C, size_t* bytes_out) {
  DADET_REQUIRE(rows >= 0 && C > 0 && bytes_out, "colsum_workspace_bytes: bad args");
  *bytes_out = rows == 0 ? 0 : sizeof(float) * (size_t)colsum_splits(rows, C) * C;
  return DADET_OK;
}

// rows of g may be longer than C (`ld` floats apart: the padded 20 / 28-float rows of a deformable block's offset
// gradient, of which the parameter owns 18 / 27); accumulate: out[c] += (the bias gradient lands in the parameter's slot
// of the flat gradient bucket, utils.streams.direct_bias_target — no autograd accumulation launch behind it)
extern "C" int dadet_colsum_ld(const float* g, int ld, float* out, int64_t rows, int C, int accumulate, void* workspace,
                               size_t workspace_bytes, void* stream) {
  DADET_REQUIRE(rows >= 0 && C > 0 && ld >= C && out, "colsum: bad args (rows=%lld C=%d ld=%d)", (long long)rows, C, ld);
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * C, st);
    return check_launch("colsum(empty)");
  }
  DADET_REQUIRE(g, "colsum: null input");
  const int splits = colsum_splits(rows, C);
  if (workspace_bytes < sizeof(float) * (size_t)splits * C || !workspace) {
    set_error("colsum: workspace too small");
    return DADET_EWORKSPACE;
  }
  const int64_t rps = ceil_div64(rows, splits);
  const int cb = colsum_cb(C);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(C, cb), splits), dim3(256), 0, st, g,
                     static_cast<float*>(workspace), rows, C, rps, cb, ld);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, st, static_cast<const float*>(workspace), out, C,
                     splits, accumulate);
  return check_launch("colsum");
}

extern "C" int dadet_colsum(const float* g, float* out, int64_t rows, int C, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return dadet_colsum_ld(g, C, out, rows, C, 0, workspace, workspace_bytes, stream);
}

extern "C" int dadet_channel_affine(const float* x, const float* scale, const float* bias, float* y,
                                    int64_t rows, int C, int relu, void* stream) {
  DADET_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0, "channel_affine: C=%d must be a multiple of 4", C);
  if (rows == 0) return DADET_OK;
  DADET_REQUIRE(x && scale && bias && y, "channel_affine: null pointer");
  DADET_REQUIRE(aligned16(x) && aligned16(scale) && aligned16(bias) && aligned16(y),
                "channel_affine: pointers must be 16-byte aligned");
  const int64_t total4 = rows * (C / 4);
  hipLaunchKernelGGL(channel_affine_kernel, dim3(stream_blocks(total4, 256)), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float4*>(x),
                     reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(bias),
                     reinterpret_cast<float4*>(y), total4, C / 4, relu);
  return check_launch("channel_affine");
}

extern "C" int dadet_maxpool3x3s2_forward(const float* x, float* y, int N, int H, int W, int C, int Ho,
                                          int Wo, void* stream) {
  DADET_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad dims");
  DADET_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool: Ho/Wo mismatch");
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(x && y && aligned16(x) && aligned16(y), "maxpool: bad pointers");
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(stream_blocks(total, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), N, H, W, C / 4, Ho,
                     Wo);
  return check_launch("maxpool3x3s2");
}

extern "C" int dadet_avgpool_forward(const float* x, float* y, int R, int HW, int C, void* stream) {
  DADET_REQUIRE(R >= 0 && HW > 0 && C > 0 && C % 4 == 0, "avgpool: bad dims");
  if (R == 0) return DADET_OK;
  DADET_REQUIRE(x && y && aligned16(x) && aligned16(y), "avgpool: bad pointers");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(stream_blocks((int64_t)R * (C / 4), 256)), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), R,
                     HW, C / 4);
  return check_launch("avgpool_forward");
}
extern "C" int dadet_avgpool_backward(const float* gy, float* gx, int R, int HW, int C, void* stream) {
  DADET_REQUIRE(R >= 0 && HW > 0 && C > 0 && C % 4 == 0, "avgpool_backward: bad dims");
  if (R == 0) return DADET_OK;
  DADET_REQUIRE(gy && gx && aligned16(gy) && aligned16(gx), "avgpool_backward: bad pointers");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(stream_blocks((int64_t)R * HW * (C / 4), 256)), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const float4*>(gy), reinterpret_cast<float4*>(gx),
                     R, HW, C / 4);
  return check_launch("avgpool_backward");
}

extern "C" int dadet_nchw3_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream) {
  DADET_REQUIRE(N >= 0 && H > 0 && W > 0, "nchw3_to_nhwc4: bad dims");
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(x && y && aligned16(y), "nchw3_to_nhwc4: bad pointers");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(stream_blocks((int64_t)N * HW, 256)), dim3(256), 0,
                     as_stream(stream), x, reinterpret_cast<float4*>(y), N, HW);
  return check_launch("nchw3_to_nhwc4");
}

extern "C" int dadet_rpn_decode_clip(const float* deltas, const float* anchors, const int64_t* topk_idx,
                                     int K, float wx, float wy, float ww, float wh, float xform_clip,
                                     float im_w, float im_h, float* boxes_out, void* stream) {
  DADET_REQUIRE(K >= 0, "rpn_decode_clip: K < 0");
  if (K == 0) return DADET_OK;
  DADET_REQUIRE(deltas && anchors && topk_idx && boxes_out, "rpn_decode_clip: null pointer");
  DADET_REQUIRE(aligned16(deltas) && aligned16(anchors) && aligned16(boxes_out),
                "rpn_decode_clip: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(rpn_decode_clip_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(deltas), reinterpret_cast<const float4*>(anchors),
                     topk_idx, K, wx, wy, ww, wh, xform_clip, im_w, im_h,
                     reinterpret_cast<float4*>(boxes_out));
  return check_launch("rpn_decode_clip");
}

extern "C" int dadet_rpn_anchor_targets(const float* anchors, const unsigned char* visible, int A,
                                        const float* gt_boxes, int G, float high_threshold, float low_threshold,
                                        unsigned* workspace_G, float* labels, float* regression_targets,
                                        void* stream) {
  DADET_REQUIRE(A >= 0 && G > 0, "rpn_anchor_targets: needs A >= 0 anchors and G > 0 ground-truth boxes");
  if (A == 0) return DADET_OK;
  DADET_REQUIRE(anchors && visible && gt_boxes && workspace_G && labels && regression_targets,
                "rpn_anchor_targets: null pointer");
  DADET_REQUIRE(aligned16(anchors) && aligned16(gt_boxes) && aligned16(regression_targets),
                "rpn_anchor_targets: box pointers must be 16-byte aligned");
  DADET_REQUIRE(G <= 3000, "rpn_anchor_targets: G=%d ground-truth boxes exceed the LDS table", G);
  hipStream_t st = as_stream(stream);
  (void)hipMemsetAsync(workspace_G, 0, sizeof(unsigned) * (size_t)G, st);
  int blocks = ceil_div(A, 256);
  if (blocks > kNumCU * 2) blocks = kNumCU * 2;
  hipLaunchKernelGGL(rpn_gt_best_kernel, dim3(blocks), dim3(256), (size_t)G * 24, st,
                     reinterpret_cast<const float4*>(anchors), A, reinterpret_cast<const float4*>(gt_boxes), G,
                     workspace_G);
  hipLaunchKernelGGL(rpn_anchor_targets_kernel, dim3(ceil_div(A, 256)), dim3(256), (size_t)G * 24, st,
                     reinterpret_cast<const float4*>(anchors), visible, A, reinterpret_cast<const float4*>(gt_boxes),
                     G, workspace_G, high_threshold, low_threshold, labels,
                     reinterpret_cast<float4*>(regression_targets));
  return check_launch("rpn_anchor_targets");
}

extern "C" int dadet_box_match_encode(const float* proposals, int P, const float* gt_boxes, const int64_t* gt_labels,
                                      int G, float high_threshold, float low_threshold, float wx, float wy, float ww,
                                      float wh, int64_t* matched_idxs, int64_t* labels, float* regression_targets,
                                      void* stream) {
  DADET_REQUIRE(P >= 0 && G > 0, "box_match_encode: needs P >= 0 proposals and G > 0 ground-truth boxes");
  if (P == 0) return DADET_OK;
  DADET_REQUIRE(proposals && gt_boxes && gt_labels && matched_idxs && labels && regression_targets,
                "box_match_encode: null pointer");
  DADET_REQUIRE(aligned16(proposals) && aligned16(gt_boxes) && aligned16(regression_targets),
                "box_match_encode: box pointers must be 16-byte aligned");
  DADET_REQUIRE(G <= 3000, "box_match_encode: G=%d ground-truth boxes exceed the LDS table", G);
  hipLaunchKernelGGL(box_match_encode_kernel, dim3(ceil_div(P, 256)), dim3(256), (size_t)G * 20, as_stream(stream),
                     reinterpret_cast<const float4*>(proposals), P, reinterpret_cast<const float4*>(gt_boxes),
                     gt_labels, G, high_threshold, low_threshold, wx, wy, ww, wh, matched_idxs, labels,
                     reinterpret_cast<float4*>(regression_targets));
  return check_launch("box_match_encode");
}

extern "C" int dadet_sigmoid_focal_loss_forward(const float* logits, const int32_t* targets, float* losses,
                                                int N, int C, float gamma, float alpha, void* stream) {
  DADET_REQUIRE(N >= 0 && C > 0, "focal_forward: bad dims");
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(logits && targets && losses, "focal_forward: null pointer");
  const int64_t total = (int64_t)N * C;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(stream_blocks(total, 256)), dim3(256), 0, as_stream(stream),
                     logits, targets, losses, total, C, gamma, alpha);
  return check_launch("sigmoid_focal_loss_forward");
}
extern "C" int dadet_sigmoid_focal_loss_backward(const float* logits, const int32_t* targets,
                                                 const float* d_losses, float* d_logits, int N, int C,
                                                 float gamma, float alpha, void* stream) {
  DADET_REQUIRE(N >= 0 && C > 0, "focal_backward: bad dims");
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(logits && targets && d_losses && d_logits, "focal_backward: null pointer");
  const int64_t total = (int64_t)N * C;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(stream_blocks(total, 256)), dim3(256), 0, as_stream(stream),
                     logits, targets, d_losses, d_logits, total, C, gamma, alpha);
  return check_launch("sigmoid_focal_loss_backward");
}

extern "C" int dadet_sgd_step(const dadet_sgd_entry* table_dev, int num_tensors, int64_t max_numel,
                              float momentum, int first_step, float grad_scale, void* stream) {
  DADET_REQUIRE(num_tensors >= 0 && max_numel >= 0, "sgd_step: bad args");
  if (num_tensors == 0 || max_numel == 0) return DADET_OK;
  DADET_REQUIRE(table_dev, "sgd_step: null table");
  DADET_REQUIRE(num_tensors <= 65535, "sgd_step: too many tensors for one launch");
  int bx = (int)ceil_div64(ceil_div64(max_numel, 4), 1024);
  if (bx > 512) bx = 512;  // sized for the largest tensor; smaller tensors' surplus workgroups exit at once
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(sgd_kernel, dim3(bx, num_tensors), dim3(256), 0, as_stream(stream), table_dev,
                     momentum, first_step, grad_scale);
  return check_launch("sgd_step");
}
