/*
 * dadet_oracle.c — CPU restatement (plain C, single thread) of the native operators on the DA Faster R-CNN
 * hot path.  TEST INFRASTRUCTURE ONLY: this file is the checker the HIP kernels are compared against
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under da_detect_amd/ may
 * import, link or execute it; the product path has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks these functions against (a) the reference's own
 * known-answer vectors (tests/test_nms.py:11-217 of the reference, re-typed as data in tests/golden/) and
 * (b) outputs of the reference's compiled CPU operators (oracle/_ref, built from
 * /root/reference/maskrcnn_benchmark/csrc by oracle/build_ref.py) and of the imported Python reference,
 * stored as fixtures under tests/golden/ by tests/golden/make_golden.py.
 *
 * Each function cites the reference lines it follows.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off;
 * contraction is off so the float operation order below is what the reference's x86-64 build executes).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * NMS — reference: maskrcnn_benchmark/csrc/cpu/nms_cpu.cpp:6-65 (rule 0, `ovr >= thr`, :60) and
 * maskrcnn_benchmark/csrc/cuda/nms.cu:13-21,60 (rule 1, `IoU > thr`).
 * order: indices sorted by score descending; ties broken by ascending index (what a stable sort gives;
 * the reference's ATen sort leaves tie order unspecified).  Returns the number kept; keep[] holds kept
 * ORIGINAL indices ascending (nms_cpu.cpp:64 `nonzero(suppressed == 0)`).
 * ----------------------------------------------------------------------------------------------*/
typedef struct { float score; int idx; } oracle_key;
static int key_cmp(const void* a, const void* b) {
  const oracle_key* x = (const oracle_key*)a;
  const oracle_key* y = (const oracle_key*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

int oracle_nms(const float* boxes, const float* scores, int n, float thresh, int tie_rule,
               int64_t* keep) {
  if (n <= 0) return 0;
  oracle_key* keys = (oracle_key*)malloc(sizeof(oracle_key) * (size_t)n);
  float* area = (float*)malloc(sizeof(float) * (size_t)n);
  unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
  for (int i = 0; i < n; ++i) {
    keys[i].score = scores[i];
    keys[i].idx = i;
    /* areas = (x2 - x1 + 1) * (y2 - y1 + 1)   nms_cpu.cpp:23 */
    area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0] + 1.f) * (boxes[4 * i + 3] - boxes[4 * i + 1] + 1.f);
  }
  qsort(keys, (size_t)n, sizeof(oracle_key), key_cmp);
  for (int a = 0; a < n; ++a) {
    const int i = keys[a].idx;
    if (dead[i]) continue;
    const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    for (int b = a + 1; b < n; ++b) {
      const int j = keys[b].idx;
      if (dead[j]) continue;
      const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
      const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
      const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
      const float inter = w * h;
      const float ovr = inter / (area[i] + area[j] - inter);
      if (tie_rule == 0 ? (ovr >= thresh) : (ovr > thresh)) dead[j] = 1;
    }
  }
  int k = 0;
  for (int i = 0; i < n; ++i)
    if (!dead[i]) keep[k++] = i;
  free(keys);
  free(area);
  free(dead);
  return k;
}

/* ------------------------------------------------------------------------------------------------
 * ROIAlign — reference: maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:18-219 (forward) and
 * maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:124-254 (backward; the reference has no CPU backward,
 * csrc/ROIAlign.h:44).  NCHW tensors, exactly like the reference.
 * ----------------------------------------------------------------------------------------------*/
typedef struct { int p1, p2, p3, p4; float w1, w2, w3, w4; } oracle_tap;

/* ROIAlign_cpu.cpp:46-104 / ROIAlign_cuda.cu:16-62 */
static oracle_tap tap_at(float y, float x, int H, int W) {
  oracle_tap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.p1 = t.p2 = t.p3 = t.p4 = -1;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - (float)yl, lx = x - (float)xl;
  const float hy = 1.f - ly, hx = 1.f - lx;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  t.p1 = yl * W + xl; t.p2 = yl * W + xh; t.p3 = yh * W + xl; t.p4 = yh * W + xh;
  return t;
}

typedef struct { int batch, gh, gw; float sw, sh, bw, bh, count; } oracle_roi;
/* ROIAlign_cpu.cpp:140-172 */
static oracle_roi roi_geom(const float* r, float scale, int ph, int pw, int sampling_ratio) {
  oracle_roi g;
  g.batch = (int)r[0];
  g.sw = r[1] * scale;
  g.sh = r[2] * scale;
  const float ew = r[3] * scale, eh = r[4] * scale;
  const float rw = fmaxf(ew - g.sw, 1.f), rh = fmaxf(eh - g.sh, 1.f);
  g.bh = rh / (float)ph;
  g.bw = rw / (float)pw;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
  g.count = (float)(g.gh * g.gw);
  return g;
}

void oracle_roi_align_forward(const float* input, const float* rois, float* output, int B, int C, int H,
                              int W, int R, int ph, int pw, float scale, int sampling_ratio) {
  (void)B;
  for (int n = 0; n < R; ++n) {
    const oracle_roi g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
    for (int c = 0; c < C; ++c) {
      const float* plane = input + ((size_t)g.batch * C + c) * H * W;
      for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
          float acc = 0.f;
          for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.sh + (float)i * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
              const float x = g.sw + (float)j * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
              const oracle_tap t = tap_at(y, x, H, W);
              if (t.p1 < 0) continue; /* reference adds 0*data[0] */
              /* ROIAlign_cpu.cpp:201-204 */
              acc += t.w1 * plane[t.p1] + t.w2 * plane[t.p2] + t.w3 * plane[t.p3] + t.w4 * plane[t.p4];
            }
          }
          output[(((size_t)n * C + c) * ph + i) * pw + j] = acc / g.count;
        }
    }
  }
}

/* ROIAlign_cuda.cu:178-254: g_k = top_diff * w_k / count scattered to the four neighbours. */
void oracle_roi_align_backward(const float* grad_out, const float* rois, float* grad_in, int B, int C,
                               int H, int W, int R, int ph, int pw, float scale, int sampling_ratio) {
  memset(grad_in, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < R; ++n) {
    const oracle_roi g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
    for (int c = 0; c < C; ++c) {
      float* plane = grad_in + ((size_t)g.batch * C + c) * H * W;
      for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
          const float go = grad_out[(((size_t)n * C + c) * ph + i) * pw + j];
          for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.sh + (float)i * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
              const float x = g.sw + (float)j * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
              const oracle_tap t = tap_at(y, x, H, W);
              if (t.p1 < 0) continue;
              plane[t.p1] += go * t.w1 / g.count;
              plane[t.p2] += go * t.w2 / g.count;
              plane[t.p3] += go * t.w3 / g.count;
              plane[t.p4] += go * t.w4 / g.count;
            }
          }
        }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Sigmoid focal loss — reference: maskrcnn_benchmark/csrc/cuda/SigmoidFocalLoss_cuda.cu:21-101
 * (the only native implementation; the Python CPU fallback layers/sigmoid_focal_loss.py:40-52 lacks the
 * FLT_MIN clamp and the stable log(1-p)).
 * ----------------------------------------------------------------------------------------------*/
void oracle_sigmoid_focal_loss_forward(const float* logits, const int32_t* targets, float* losses, int N,
                                       int C, float gamma, float alpha) {
  for (int n = 0; n < N; ++n)
    for (int d = 0; d < C; ++d) {
      const int t = targets[n];
      const float c1 = (t == d + 1) ? 1.f : 0.f;
      const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
      const float x = logits[(size_t)n * C + d];
      const float p = 1.f / (1.f + expf(-x));
      const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
      const float xp = x >= 0.f ? 1.f : 0.f;
      const float term2 = powf(p, gamma) * (-1.f * x * xp - logf(1.f + expf(x - 2.f * x * xp)));
      float l = 0.f;
      l += -c1 * term1 * alpha;
      l += -c2 * term2 * (1.f - alpha);
      losses[(size_t)n * C + d] = l;
    }
}

void oracle_sigmoid_focal_loss_backward(const float* logits, const int32_t* targets, const float* d_losses,
                                        float* d_logits, int N, int C, float gamma, float alpha) {
  for (int n = 0; n < N; ++n)
    for (int d = 0; d < C; ++d) {
      const int t = targets[n];
      const float c1 = (t == d + 1) ? 1.f : 0.f;
      const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
      const float x = logits[(size_t)n * C + d];
      const float p = 1.f / (1.f + expf(-x));
      const float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, FLT_MIN))));
      const float xp = x >= 0.f ? 1.f : 0.f;
      const float term2 =
          powf(p, gamma) * ((-1.f * x * xp - logf(1.f + expf(x - 2.f * x * xp))) * (1.f - p) * gamma - p);
      float g = 0.f;
      g += -c1 * term1 * alpha;
      g += -c2 * term2 * (1.f - alpha);
      d_logits[(size_t)n * C + d] = g * d_losses[(size_t)n * C + d];
    }
}

/* ------------------------------------------------------------------------------------------------
 * RPN box decode + clip — reference: maskrcnn_benchmark/modeling/box_coder.py:52-95 and
 * maskrcnn_benchmark/structures/bounding_box.py:214-224.
 * ----------------------------------------------------------------------------------------------*/
void oracle_decode_clip(const float* deltas, const float* anchors, int K, float wx, float wy, float ww,
                        float wh, float xform_clip, float im_w, float im_h, float* out) {
  for (int k = 0; k < K; ++k) {
    const float* b = anchors + 4 * k;
    const float* d = deltas + 4 * k;
    const float width = b[2] - b[0] + 1.f, height = b[3] - b[1] + 1.f;
    const float cx = b[0] + 0.5f * width, cy = b[1] + 0.5f * height;
    const float dx = d[0] / wx, dy = d[1] / wy;
    float dw = d[2] / ww, dh = d[3] / wh;
    if (dw > xform_clip) dw = xform_clip;
    if (dh > xform_clip) dh = xform_clip;
    const float pcx = dx * width + cx, pcy = dy * height + cy;
    const float pw = expf(dw) * width, ph = expf(dh) * height;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
    float x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
    x1 = fminf(fmaxf(x1, 0.f), im_w - 1.f);
    y1 = fminf(fmaxf(y1, 0.f), im_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), im_w - 1.f);
    y2 = fminf(fmaxf(y2, 0.f), im_h - 1.f);
    out[4 * k] = x1; out[4 * k + 1] = y1; out[4 * k + 2] = x2; out[4 * k + 3] = y2;
  }
}

/* ---------------------------------------------------------------------------------------------------------------
 * ROIPool forward / backward, NCHW.  The reference has no CPU ROIPool (csrc/ROIPool.h:20-22 "Not implemented on
 * the CPU"), so this restates its CUDA kernels (csrc/cuda/ROIPool_cuda.cu:16-75 forward, :77-108 backward) and is
 * NOT pinned against reference outputs ("parity unpinned"); tests pin it by properties (whole-map ROI == global
 * max, 1x1-bin ROI == the pixel, adaptive_max_pool2d on an exactly divisible ROI). */
void oracle_roi_pool_forward(const float* input, const float* rois, float* output, int* argmax, int B, int C, int H,
                             int W, int R, int PH, int PW, float scale) {
  (void)B;
  for (int n = 0; n < R; ++n) {
    const float* r = rois + (size_t)n * 5;
    const int b = (int)r[0];
    const int sw = (int)roundf(r[1] * scale), sh = (int)roundf(r[2] * scale);
    const int ew = (int)roundf(r[3] * scale), eh = (int)roundf(r[4] * scale);
    const int rw = (ew - sw + 1) > 1 ? (ew - sw + 1) : 1, rh = (eh - sh + 1) > 1 ? (eh - sh + 1) : 1;
    const float bin_h = (float)rh / (float)PH, bin_w = (float)rw / (float)PW;
    for (int c = 0; c < C; ++c)
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          int hs = (int)floorf((float)ph * bin_h) + sh, he = (int)ceilf((float)(ph + 1) * bin_h) + sh;
          int ws = (int)floorf((float)pw * bin_w) + sw, we = (int)ceilf((float)(pw + 1) * bin_w) + sw;
          hs = hs < 0 ? 0 : (hs > H ? H : hs);
          he = he < 0 ? 0 : (he > H ? H : he);
          ws = ws < 0 ? 0 : (ws > W ? W : ws);
          we = we < 0 ? 0 : (we > W ? W : we);
          const int empty = (he <= hs) || (we <= ws);
          float best = empty ? 0.f : -FLT_MAX;
          int idx = -1;
          const float* plane = input + ((size_t)b * C + c) * H * W;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w)
              if (plane[h * W + w] > best) {
                best = plane[h * W + w];
                idx = h * W + w;
              }
          const size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
          output[o] = best;
          argmax[o] = idx;
        }
  }
}

void oracle_roi_pool_backward(const float* grad_out, const int* argmax, const float* rois, float* grad_in, int B,
                              int C, int H, int W, int R, int PH, int PW) {
  memset(grad_in, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < R; ++n) {
    const int b = (int)rois[(size_t)n * 5];
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < PH * PW; ++k) {
        const size_t o = ((size_t)n * C + c) * PH * PW + k;
        if (argmax[o] >= 0) grad_in[((size_t)b * C + c) * H * W + argmax[o]] += grad_out[o];
      }
  }
}
