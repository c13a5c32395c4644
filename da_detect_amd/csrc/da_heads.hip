// Fused domain-adaptation head kernels (gfx950): image-level domain classifier tail + BCE +
// consistency statistics with the gradient-reversal weights folded into the backward, and the
// domain-level triplet (metric regularisation) loss on NHWC feature maps.
//
// Reference being restated:
//   DAImgHead.forward            modeling/da_heads/da_heads.py:32-37   (conv1_da -> relu -> conv2_da)
//   GradientScalarLayer          layers/gradient_scalar_layer.py:4-13  (identity fwd, weight*grad bwd)
//   image BCE loss               modeling/da_heads/loss.py:80-98 / :140-167 (BCE-with-logits, label = is_source,
//                                mean over B*H*W)
//   consistency statistic        layers/consistency_loss.py:12-14      (mean over H*W of sigmoid(logit) per image)
//   triplet image loss           modeling/da_heads/loss.py:180-200     (nn.TripletMarginLoss(p=2) on
//                                [1,C,H,W] maps: L2 distance over the LAST axis W, eps 1e-6, hinge, mean over C*H)
//
// The reference runs the image head twice per iteration on identical values (once behind GRL(-w) for the
// adversarial loss, once behind GRL(+w) for the consistency loss; da_heads.py:409-419).  Forward values of
// the two passes are identical, so the head is evaluated ONCE here and the two reversal weights are applied
// in the backward: weight gradients use g_adv + g_cst, the feature gradient uses w_adv*g_adv + w_cst*g_cst.
//
// The 1024->512 1x1 conv (the FLOPs) runs on the MFMA implicit-GEMM kernel with its bias+ReLU epilogue; the
// kernels here consume its output t[M][C1] one wavefront per row: 64 lanes x float4 along the channel
// axis, DPP/shuffle wave reduction, one atomic per wavefront for the loss sums.
#include "conv_common.h"   // amax_publish (contraction mode 4)

namespace dadet {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// numerically-stable BCE-with-logits as ATen computes it (Loss.cpp binary_cross_entropy_with_logits):
//   loss = (1 - y) * x + max(-x, 0) + log(exp(-max(-x,0)) + exp(-x - max(-x,0)))
__device__ inline float bce_with_logits(float x, float y) {
  const float mx = fmaxf(-x, 0.f);
  return (1.f - y) * x + mx + logf(expf(-mx) + expf(-x - mx));
}
__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// grid-stride over rows, one wavefront per row.  sums_out[img][0] += bce, sums_out[img][1] += sigmoid.
__global__ __launch_bounds__(256) void da_img_fwd_kernel(const float* __restrict__ t,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ b2,
                                                         const float* __restrict__ labels,
                                                         float* __restrict__ logits_out,
                                                         float* __restrict__ sums_out, int num_images,
                                                         int rows_per_image, int C1) {
  // per-image sums are collected in LDS first: one global atomic per workgroup, image and quantity (4096 same-address
  // global atomics — one pair per wavefront — serialised in L2 and made this 33 MB kernel take 112 us)
  constexpr int kMaxImg = 8;
  __shared__ float s_acc[kMaxImg * 2];
  const bool use_lds = num_images <= kMaxImg;
  if (threadIdx.x < kMaxImg * 2) s_acc[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int64_t M = (int64_t)num_images * rows_per_image;
  const float bias = b2[0];
  auto flush = [&](int img, float bce, float sig) {
    if (use_lds) {
      atomicAdd(&s_acc[img * 2 + 0], bce);
      atomicAdd(&s_acc[img * 2 + 1], sig);
    } else {
      atomicAdd(&sums_out[img * 2 + 0], bce);
      atomicAdd(&sums_out[img * 2 + 1], sig);
    }
  };
  // a wavefront walks rows m = wave_global, wave_global + nwaves, ...; partial sums are flushed whenever the
  // image index changes (rows of one image are contiguous)
  float bce_acc = 0.f, sig_acc = 0.f;
  int cur_img = -1;
  for (int64_t m = wave_global; m < M; m += nwaves) {
    const float* row = t + m * C1;
    float dot = 0.f;
    for (int c = lane * 4; c < C1; c += 256) {
      const float4 tv = *reinterpret_cast<const float4*>(row + c);
      const float4 wv = *reinterpret_cast<const float4*>(w2 + c);
      dot += tv.x * wv.x + tv.y * wv.y + tv.z * wv.z + tv.w * wv.w;
    }
    dot = wave_sum(dot);
    const float logit = dot + bias;
    const int img = (int)(m / rows_per_image);
    if (img != cur_img) {
      if (cur_img >= 0 && lane == 0) flush(cur_img, bce_acc, sig_acc);
      bce_acc = 0.f;
      sig_acc = 0.f;
      cur_img = img;
    }
    if (lane == 0) {
      logits_out[m] = logit;
      bce_acc += bce_with_logits(logit, labels[img]);
      sig_acc += sigmoidf(logit);
    }
  }
  if (cur_img >= 0 && lane == 0) flush(cur_img, bce_acc, sig_acc);
  __syncthreads();
  if (use_lds && threadIdx.x < num_images * 2 && s_acc[threadIdx.x] != 0.f)
    atomicAdd(&sums_out[threadIdx.x], s_acc[threadIdx.x]);
}

// backward.  coef[img] = (a_bce_w, a_sig_w, a_bce_x, a_sig_x):
//   g_logit_w = a_bce_w*(s - y) + a_sig_w*s*(1-s)     -> head parameter gradients
//   g_logit_x = a_bce_x*(s - y) + a_sig_x*s*(1-s)     -> feature gradient (reversal weights folded in)
//   g_t_w[m][c] = g_logit_w * w2[c] * (t>0),  g_t_x[m][c] = g_logit_x * w2[c] * (t>0)
//   g_w2[c] += sum_m g_logit_w * t[m][c],     g_b2 += sum_m g_logit_w
__global__ __launch_bounds__(256) void da_img_bwd_kernel(
    const float* __restrict__ t, const float* __restrict__ w2, const float* __restrict__ logits,
    const float* __restrict__ labels, const float* __restrict__ coef, float* __restrict__ g_t_w,
    float* __restrict__ g_t_x, float* __restrict__ g_w2, float* __restrict__ g_b2, int num_images,
    int rows_per_image, int C1, const float* __restrict__ g_bce, const float* __restrict__ g_sig,
    const float* __restrict__ w_adv_dev, float w_adv, float w_cst, unsigned* __restrict__ amax_w,
    unsigned* __restrict__ amax_x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_w2g = reinterpret_cast<float*>(smem);  // [C1] workgroup partial of g_w2
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw_block = blockDim.x >> 6;
  const int64_t M = (int64_t)num_images * rows_per_image;
  for (int c = threadIdx.x; c < C1; c += blockDim.x) s_w2g[c] = 0.f;
  __syncthreads();
  // requires C1 <= 1024: each lane owns float4 slots c = lane*4 + 256*k, k < 4
  float4 wacc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wacc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bacc = 0.f;
  float mw = 0.f, mxx = 0.f;    // contraction mode 4: max|g_t_w|, max|g_t_x| of what this lane stores (slots amax_w / amax_x)
  const int wave_global = blockIdx.x * nw_block + wave;
  const int nwaves = gridDim.x * nw_block;
  for (int64_t m = wave_global; m < M; m += nwaves) {
    const int img = (int)(m / rows_per_image);
    const float y = labels[img];
    const float s = sigmoidf(logits[m]);
    // coef == nullptr: the coefficients are formed here from the upstream gradients (what the host chain
    // `g / (N H W)`, expand, stack, multiply by the reversal weights took six launches for)
    float4 cf;
    if (coef) {
      cf = *reinterpret_cast<const float4*>(coef + img * 4);
    } else {
      cf.x = g_bce[0] / (float)((int64_t)num_images * rows_per_image);
      cf.y = g_sig ? g_sig[img] / (float)rows_per_image : 0.f;
      cf.z = cf.x * (w_adv_dev ? w_adv_dev[0] : w_adv);
      cf.w = cf.y * w_cst;
    }
    const float d1 = s - y, d2 = s * (1.f - s);
    const float gw = cf.x * d1 + cf.y * d2;
    const float gx = cf.z * d1 + cf.w * d2;
    bacc += gw;
    const float* row = t + m * C1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < C1) {
        const float4 tv = *reinterpret_cast<const float4*>(row + c);
        const float4 wv = *reinterpret_cast<const float4*>(w2 + c);
        float4 ow, ox;
        ow.x = tv.x > 0.f ? gw * wv.x : 0.f; ow.y = tv.y > 0.f ? gw * wv.y : 0.f;
        ow.z = tv.z > 0.f ? gw * wv.z : 0.f; ow.w = tv.w > 0.f ? gw * wv.w : 0.f;
        ox.x = tv.x > 0.f ? gx * wv.x : 0.f; ox.y = tv.y > 0.f ? gx * wv.y : 0.f;
        ox.z = tv.z > 0.f ? gx * wv.z : 0.f; ox.w = tv.w > 0.f ? gx * wv.w : 0.f;
        *reinterpret_cast<float4*>(g_t_w + m * C1 + c) = ow;
        if (g_t_x) *reinterpret_cast<float4*>(g_t_x + m * C1 + c) = ox;
        mw = fmaxf(fmaxf(mw, fmaxf(fabsf(ow.x), fabsf(ow.y))), fmaxf(fabsf(ow.z), fabsf(ow.w)));
        mxx = fmaxf(fmaxf(mxx, fmaxf(fabsf(ox.x), fabsf(ox.y))), fmaxf(fabsf(ox.z), fabsf(ox.w)));
        wacc[k].x += gw * tv.x; wacc[k].y += gw * tv.y; wacc[k].z += gw * tv.z; wacc[k].w += gw * tv.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane * 4 + 256 * k;
    if (c < C1) {
      atomicAdd(&s_w2g[c + 0], wacc[k].x);
      atomicAdd(&s_w2g[c + 1], wacc[k].y);
      atomicAdd(&s_w2g[c + 2], wacc[k].z);
      atomicAdd(&s_w2g[c + 3], wacc[k].w);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C1; c += blockDim.x) atomicAdd(&g_w2[c], s_w2g[c]);
  if (lane == 0) atomicAdd(g_b2, bacc);  // bacc is wavefront-uniform
  if (amax_w) amax_publish(amax_w, mw);
  if (amax_x && g_t_x) amax_publish(amax_x, mxx);
}

// ---- instance-level domain classifier tail -------------------------------------------------------
// Reference: DAInsHead.forward's last layer (modeling/da_heads/da_heads.py:61-68: fc3_da 1024 -> 1), the instance BCE
// (da_heads/loss.py:95-97: BCE-with-logits against the ROI's domain label, mean over the ROIs) and the consistency
// regulariser (layers/consistency_loss.py:3-27: |mean_hw sigmoid(image logits of the ROI's image) - sigmoid(instance
// logit)|, one column per feature level, mean over ROIs x levels; ROIs of the source image come first, batch of 2).
// The reference evaluates the head twice per iteration — behind GRL(-w) for the BCE, behind GRL(+w) for the consistency
// term, each pass with its own dropout masks (da_heads.py:421-424) — and runs fc3, squeeze, BCE, sigmoid, repeat / cat /
// abs / mean as ~25 ATen launches forward and backward.  Here the rows of both passes are stacked ([R_bce] adversarial
// rows, then [R_cst] consistency rows) and ONE launch evaluates fc3 and both loss sums, one wavefront per row (64 lanes x
// float4 along the 1024 hidden units, wave reduction), and ONE launch does the whole backward of the tail: the gradient
// w.r.t. fc2's pre-activation (dropout scale and ReLU gate folded in: h = relu(z) * mask with mask in {0, 1 / keep}, so
// h != 0 <=> gate open and mask = 1 / keep), fc3's weight / bias gradients and the gradient of the per-image means.
// The two reversal weights act where the passes merge again, in da_ins_merge_kernel below.
//   h       [R_bce + R_cst][C]   second hidden layer AFTER dropout
//   means   [L][2]               per level: mean sigmoid of image 0 (source) and image 1 (target); null when R_cst == 0
//   sums    [2]                  += sum of BCE terms, += sum over consistency rows and levels of |mean - sigmoid|
__global__ __launch_bounds__(256) void da_ins_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w3,
                                                         const float* __restrict__ b3,
                                                         const float* __restrict__ labels,
                                                         const float* __restrict__ means, float* __restrict__ logits,
                                                         float* __restrict__ sums, int R_bce, int R_cst, int n_src,
                                                         int L, int C) {
  __shared__ float s_acc[2];
  if (threadIdx.x < 2) s_acc[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int R = R_bce + R_cst;
  const float bias = b3[0];
  float bce = 0.f, cst = 0.f;
  for (int r = wave_global; r < R; r += nwaves) {
    const float* row = h + (size_t)r * C;
    float dot = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 hv = *reinterpret_cast<const float4*>(row + c);
      const float4 wv = *reinterpret_cast<const float4*>(w3 + c);
      dot += hv.x * wv.x + hv.y * wv.y + hv.z * wv.z + hv.w * wv.w;
    }
    const float logit = wave_sum(dot) + bias;
    if (lane == 0) {
      logits[r] = logit;
      if (r < R_bce) {
        bce += bce_with_logits(logit, labels[r]);
      } else {
        const int img = (r - R_bce) < n_src ? 0 : 1;
        const float sg = sigmoidf(logit);
        for (int l = 0; l < L; ++l) cst += fabsf(means[l * 2 + img] - sg);
      }
    }
  }
  if (lane == 0) {
    if (bce != 0.f) atomicAdd(&s_acc[0], bce);
    if (cst != 0.f) atomicAdd(&s_acc[1], cst);
  }
  __syncthreads();
  if (threadIdx.x < 2 && s_acc[threadIdx.x] != 0.f) atomicAdd(&sums[threadIdx.x], s_acc[threadIdx.x]);
}

// coef[0] = d loss / d (BCE sum), coef[1] = d loss / d (consistency sum) (device scalars: upstream gradients x the
// mean factors).  g_z[r][c] = dlogit_r * w3[c] * inv_keep * [h[r][c] != 0];  g_w3[c] += sum_r dlogit_r * h[r][c];
// g_b3 += sum_r dlogit_r;  g_means[l][img] += coef[1] * sign(mean - sigmoid).
__global__ __launch_bounds__(256) void da_ins_bwd_kernel(const float* __restrict__ h, const float* __restrict__ w3,
                                                         const float* __restrict__ logits,
                                                         const float* __restrict__ labels,
                                                         const float* __restrict__ means,
                                                         const float* __restrict__ coef, float inv_keep,
                                                         float* __restrict__ g_z, float* __restrict__ g_w3,
                                                         float* __restrict__ g_b3, float* __restrict__ g_means,
                                                         int R_bce, int R_cst, int n_src, int L, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_w = reinterpret_cast<float*>(smem);   // [C] workgroup partial of g_w3, then [L * 2] of g_means, [1] g_b3
  float* s_m = s_w + C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw_block = blockDim.x >> 6;
  for (int c = threadIdx.x; c < C + 2 * L + 1; c += blockDim.x) s_w[c] = 0.f;
  __syncthreads();
  const int R = R_bce + R_cst;
  const float a_bce = coef[0], a_cst = coef[1];
  float4 wacc[4];     // C <= 1024: lane owns float4 slots c = lane*4 + 256*k
#pragma unroll
  for (int k = 0; k < 4; ++k) wacc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bacc = 0.f;
  for (int r = blockIdx.x * nw_block + wave; r < R; r += gridDim.x * nw_block) {
    const float sg = sigmoidf(logits[r]);
    float dl;
    if (r < R_bce) {
      dl = a_bce * (sg - labels[r]);
    } else {
      const int img = (r - R_bce) < n_src ? 0 : 1;
      float sgn = 0.f;    // d |m - s| / d s summed over levels
      for (int l = 0; l < L; ++l) {
        const float d = means[l * 2 + img] - sg;
        const float sd = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        sgn -= sd;
        if (lane == 0 && sd != 0.f) atomicAdd(&s_m[l * 2 + img], a_cst * sd);
      }
      dl = a_cst * sgn * sg * (1.f - sg);
    }
    bacc += dl;
    const float* row = h + (size_t)r * C;
    const float gk = dl * inv_keep;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < C) {
        const float4 hv = *reinterpret_cast<const float4*>(row + c);
        const float4 wv = *reinterpret_cast<const float4*>(w3 + c);
        float4 o;
        o.x = hv.x != 0.f ? gk * wv.x : 0.f; o.y = hv.y != 0.f ? gk * wv.y : 0.f;
        o.z = hv.z != 0.f ? gk * wv.z : 0.f; o.w = hv.w != 0.f ? gk * wv.w : 0.f;
        *reinterpret_cast<float4*>(g_z + (size_t)r * C + c) = o;
        wacc[k].x += dl * hv.x; wacc[k].y += dl * hv.y; wacc[k].z += dl * hv.z; wacc[k].w += dl * hv.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane * 4 + 256 * k;
    if (c < C) {
      atomicAdd(&s_w[c + 0], wacc[k].x);
      atomicAdd(&s_w[c + 1], wacc[k].y);
      atomicAdd(&s_w[c + 2], wacc[k].z);
      atomicAdd(&s_w[c + 3], wacc[k].w);
    }
  }
  if (lane == 0 && bacc != 0.f) atomicAdd(&s_m[2 * L], bacc);   // bacc is wavefront-uniform
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    if (s_w[c] != 0.f) atomicAdd(&g_w3[c], s_w[c]);
  if (g_means && threadIdx.x < 2 * L && s_m[threadIdx.x] != 0.f) atomicAdd(&g_means[threadIdx.x], s_m[threadIdx.x]);
  if (threadIdx.x == 0 && s_m[2 * L] != 0.f) atomicAdd(g_b3, s_m[2 * L]);
}

// the two passes share the head's first layer up to its dropout mask: h1s = [h1 * mask_a; h1 * mask_b]
__global__ __launch_bounds__(256) void da_ins_dropout_rows_kernel(const float4* __restrict__ h1,
                                                                  const float4* __restrict__ masks,
                                                                  float4* __restrict__ out, int64_t n4_per_pass,
                                                                  int passes) {
  const int64_t total = n4_per_pass * passes;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = h1[i % n4_per_pass], m = masks[i];
    out[i] = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
  }
}

// where the passes merge again (backward through the shared first layer).  g [P][R][C] = gradient w.r.t. the dropped
// hidden layer of each pass; the gradient w.r.t. fc1's pre-activation is summed over the passes ONCE unweighted (the
// head's own parameter gradients) and ONCE with each pass's gradient-reversal weight (what flows on into the ROI
// features: layers/gradient_scalar_layer.py:4-13 sits in front of fc1, and everything in between is linear in the
// gradient), both gated by h1's ReLU:   g_w = [h1 > 0] * sum_p mask_p * g_p,   g_x = [h1 > 0] * sum_p grl[p] * mask_p * g_p
__global__ __launch_bounds__(256) void da_ins_merge_kernel(const float4* __restrict__ g, const float4* __restrict__ masks,
                                                           const float4* __restrict__ h1, const float* __restrict__ grl,
                                                           float4* __restrict__ g_w, float4* __restrict__ g_x,
                                                           int64_t n4_per_pass, int passes) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_pass;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 hv = h1[i];
    float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ax = aw;
    for (int p = 0; p < passes; ++p) {
      const float4 gv = g[p * n4_per_pass + i], mv = masks[p * n4_per_pass + i];
      const float w = grl[p];
      const float4 t = make_float4(gv.x * mv.x, gv.y * mv.y, gv.z * mv.z, gv.w * mv.w);
      aw.x += t.x; aw.y += t.y; aw.z += t.z; aw.w += t.w;
      ax.x += w * t.x; ax.y += w * t.y; ax.z += w * t.z; ax.w += w * t.w;
    }
    g_w[i] = make_float4(hv.x > 0.f ? aw.x : 0.f, hv.y > 0.f ? aw.y : 0.f, hv.z > 0.f ? aw.z : 0.f, hv.w > 0.f ? aw.w : 0.f);
    if (g_x) g_x[i] = make_float4(hv.x > 0.f ? ax.x : 0.f, hv.y > 0.f ? ax.y : 0.f, hv.z > 0.f ? ax.z : 0.f,
                                  hv.w > 0.f ? ax.w : 0.f);
  }
}

// ---- triplet loss over the W axis of NHWC maps ------------------------------------------------
// a, p, n: [H][W][C] (one image each).  For every (h, c): d_ap = ||a - p + eps||_2 over w, d_an likewise;
// loss_sum += max(d_ap - d_an + margin, 0).  A lane owns one (h, c) pair — lanes run along c, so every
// w step is a coalesced row read and no cross-lane reduction is needed until the final sum.
// dist_out[(h*C + c)*2 + {0,1}] keeps (d_ap, d_an) for the backward.
__global__ __launch_bounds__(256) void triplet_w_fwd_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ p,
                                                            const float* __restrict__ n, int H, int W, int C,
                                                            float margin, float eps,
                                                            float* __restrict__ dist_out,
                                                            float* __restrict__ loss_sum) {
  const int64_t total = (int64_t)H * C;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(i / C), c = (int)(i % C);
    const int64_t base = (int64_t)h * W * C + c;
    float sap = 0.f, san = 0.f;
    for (int w = 0; w < W; ++w) {
      const float av = a[base + (int64_t)w * C];
      const float dp = av - p[base + (int64_t)w * C] + eps;
      const float dn = av - n[base + (int64_t)w * C] + eps;
      sap += dp * dp;
      san += dn * dn;
    }
    const float dap = sqrtf(sap), dan = sqrtf(san);
    dist_out[i * 2 + 0] = dap;
    dist_out[i * 2 + 1] = dan;
    acc += fmaxf(dap - dan + margin, 0.f);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, acc);
}

// g_scale[0] = upstream gradient / (H*C)  (mean reduction).  d/da = g*( (a-p+eps)/d_ap - (a-n+eps)/d_an ),
// d/dp = -g*(a-p+eps)/d_ap, d/dn = +g*(a-n+eps)/d_an, all gated by the hinge.
__global__ __launch_bounds__(256) void triplet_w_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n,
    const float* __restrict__ dist, const float* __restrict__ g_scale, int H, int W, int C, float margin,
    float eps, float* __restrict__ ga, float* __restrict__ gp, float* __restrict__ gn) {
  const int64_t total = (int64_t)H * C;
  const float g = g_scale[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(i / C), c = (int)(i % C);
    const int64_t base = (int64_t)h * W * C + c;
    const float dap = dist[i * 2 + 0], dan = dist[i * 2 + 1];
    const bool active = (dap - dan + margin) > 0.f;
    const float ip = (active && dap > 0.f) ? g / dap : 0.f;
    const float in_ = (active && dan > 0.f) ? g / dan : 0.f;
    for (int w = 0; w < W; ++w) {
      const int64_t o = base + (int64_t)w * C;
      const float av = a[o];
      const float dp = (av - p[o] + eps) * ip;
      const float dn = (av - n[o] + eps) * in_;
      if (ga) ga[o] = dp - dn;
      if (gp) gp[o] = -dp;
      if (gn) gn[o] = dn;
    }
  }
}

}  // namespace dadet

using namespace dadet;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int dadet_da_img_head_loss_forward(const float* t, const float* w2, const float* b2,
                                              const float* labels, float* logits_out, float* sums_out,
                                              int num_images, int rows_per_image, int C1, void* stream) {
  DADET_REQUIRE(num_images >= 0 && rows_per_image > 0 && C1 > 0 && C1 % 4 == 0,
                "da_img_head_loss_forward: bad dims");
  if (num_images == 0) return DADET_OK;
  DADET_REQUIRE(t && w2 && b2 && labels && logits_out && sums_out && a16(t) && a16(w2),
                "da_img_head_loss_forward: bad pointers");
  const int64_t M = (int64_t)num_images * rows_per_image;
  int64_t blocks = ceil_div64(M, 4 * 8);  // 4 waves per block, ~8 rows per wave
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(da_img_fwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), t, w2, b2,
                     labels, logits_out, sums_out, num_images, rows_per_image, C1);
  return check_launch("da_img_head_loss_forward");
}

extern "C" int dadet_da_img_head_loss_backward(const float* t, const float* w2, const float* logits,
                                               const float* labels, const float* coef, float* g_t_w,
                                               float* g_t_x, float* g_w2, float* g_b2, int num_images,
                                               int rows_per_image, int C1, void* stream) {
  DADET_REQUIRE(num_images >= 0 && rows_per_image > 0 && C1 > 0 && C1 % 4 == 0 && C1 <= 1024,
                "da_img_head_loss_backward: C1 must be a multiple of 4 and <= 1024");
  if (num_images == 0) return DADET_OK;
  DADET_REQUIRE(t && w2 && logits && labels && coef && g_t_w && g_w2 && g_b2 && a16(t) && a16(w2) &&
                    a16(g_t_w) && a16(g_t_x) && a16(coef),
                "da_img_head_loss_backward: bad pointers");
  const int64_t M = (int64_t)num_images * rows_per_image;
  int64_t blocks = ceil_div64(M, 4 * 16);
  if (blocks > kNumCU * 4) blocks = kNumCU * 4;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(da_img_bwd_kernel, dim3((int)blocks), dim3(256), sizeof(float) * C1,
                     as_stream(stream), t, w2, logits, labels, coef, g_t_w, g_t_x, g_w2, g_b2, num_images,
                     rows_per_image, C1, nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr);
  return check_launch("da_img_head_loss_backward");
}

extern "C" int dadet_da_img_head_loss_backward_g(const float* t, const float* w2, const float* logits,
                                                 const float* labels, const float* g_bce, const float* g_mean_sig,
                                                 const float* w_adv_dev, float w_adv, float w_cst, float* g_t_w,
                                                 float* g_t_x, float* g_w2, float* g_b2, int num_images,
                                                 int rows_per_image, int C1, void* stream) {
  return dadet_da_img_head_loss_backward_gm(t, w2, logits, labels, g_bce, g_mean_sig, w_adv_dev, w_adv, w_cst, g_t_w, g_t_x,
                                            g_w2, g_b2, num_images, rows_per_image, C1, nullptr, nullptr, stream);
}

extern "C" int dadet_da_img_head_loss_backward_gm(const float* t, const float* w2, const float* logits,
                                                  const float* labels, const float* g_bce, const float* g_mean_sig,
                                                  const float* w_adv_dev, float w_adv, float w_cst, float* g_t_w,
                                                  float* g_t_x, float* g_w2, float* g_b2, int num_images,
                                                  int rows_per_image, int C1, float* amax_w, float* amax_x,
                                                  void* stream) {
  DADET_REQUIRE(num_images >= 0 && rows_per_image > 0 && C1 > 0 && C1 % 4 == 0 && C1 <= 1024,
                "da_img_head_loss_backward_g: C1 must be a multiple of 4 and <= 1024");
  if (num_images == 0) return DADET_OK;
  DADET_REQUIRE(t && w2 && logits && labels && g_bce && g_t_w && g_w2 && g_b2 && a16(t) && a16(w2) && a16(g_t_w) &&
                    a16(g_t_x),
                "da_img_head_loss_backward_g: bad pointers");
  const int64_t M = (int64_t)num_images * rows_per_image;
  int64_t blocks = ceil_div64(M, 4 * 16);
  if (blocks > kNumCU * 4) blocks = kNumCU * 4;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(da_img_bwd_kernel, dim3((int)blocks), dim3(256), sizeof(float) * C1, as_stream(stream), t, w2,
                     logits, labels, nullptr, g_t_w, g_t_x, g_w2, g_b2, num_images, rows_per_image, C1, g_bce,
                     g_mean_sig, w_adv_dev, w_adv, w_cst, reinterpret_cast<unsigned*>(amax_w),
                     reinterpret_cast<unsigned*>(amax_x));
  return check_launch("da_img_head_loss_backward_g");
}

extern "C" int dadet_triplet_w_forward(const float* anchor, const float* positive, const float* negative,
                                       int H, int W, int C, float margin, float eps, float* dist_out,
                                       float* loss_sum, void* stream) {
  DADET_REQUIRE(H > 0 && W > 0 && C > 0, "triplet_w_forward: bad dims");
  DADET_REQUIRE(anchor && positive && negative && dist_out && loss_sum, "triplet_w_forward: null pointer");
  int64_t blocks = ceil_div64((int64_t)H * C, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(triplet_w_fwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), anchor,
                     positive, negative, H, W, C, margin, eps, dist_out, loss_sum);
  return check_launch("triplet_w_forward");
}

extern "C" int dadet_triplet_w_backward(const float* anchor, const float* positive, const float* negative,
                                        const float* dist, const float* g_scale, int H, int W, int C,
                                        float margin, float eps, float* g_anchor, float* g_positive,
                                        float* g_negative, void* stream) {
  DADET_REQUIRE(H > 0 && W > 0 && C > 0, "triplet_w_backward: bad dims");
  DADET_REQUIRE(anchor && positive && negative && dist && g_scale, "triplet_w_backward: null pointer");
  int64_t blocks = ceil_div64((int64_t)H * C, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(triplet_w_bwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), anchor,
                     positive, negative, dist, g_scale, H, W, C, margin, eps, g_anchor, g_positive,
                     g_negative);
  return check_launch("triplet_w_backward");
}

extern "C" int dadet_da_ins_tail_forward(const float* h, const float* w3, const float* b3, const float* labels,
                                         const float* means, float* logits, float* sums, int R_bce, int R_cst,
                                         int n_src, int levels, int C, void* stream) {
  DADET_REQUIRE(R_bce >= 0 && R_cst >= 0 && C > 0 && C % 4 == 0 && levels >= 0, "da_ins_tail_forward: bad dims");
  if (R_bce + R_cst == 0) return DADET_OK;
  DADET_REQUIRE(h && w3 && b3 && logits && sums && a16(h) && a16(w3) && (R_bce == 0 || labels) &&
                    (R_cst == 0 || (means && levels > 0)),
                "da_ins_tail_forward: bad pointers");
  int blocks = ceil_div(R_bce + R_cst, 4);
  if (blocks > kNumCU * 2) blocks = kNumCU * 2;
  hipLaunchKernelGGL(da_ins_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), h, w3, b3, labels, means,
                     logits, sums, R_bce, R_cst, n_src, levels, C);
  return check_launch("da_ins_tail_forward");
}

extern "C" int dadet_da_ins_tail_backward(const float* h, const float* w3, const float* logits, const float* labels,
                                          const float* means, const float* coef, float inv_keep, float* g_z,
                                          float* g_w3, float* g_b3, float* g_means, int R_bce, int R_cst, int n_src,
                                          int levels, int C, void* stream) {
  DADET_REQUIRE(R_bce >= 0 && R_cst >= 0 && C > 0 && C % 4 == 0 && C <= 1024 && levels >= 0 && levels <= 16,
                "da_ins_tail_backward: C must be a multiple of 4 and <= 1024");
  if (R_bce + R_cst == 0) return DADET_OK;
  DADET_REQUIRE(h && w3 && logits && coef && g_z && g_w3 && g_b3 && a16(h) && a16(w3) && a16(g_z) &&
                    (R_bce == 0 || labels) && (R_cst == 0 || (means && levels > 0)),
                "da_ins_tail_backward: bad pointers");
  int blocks = ceil_div(R_bce + R_cst, 4 * 4);
  if (blocks > kNumCU) blocks = kNumCU;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(da_ins_bwd_kernel, dim3(blocks), dim3(256), sizeof(float) * (C + 2 * levels + 1), as_stream(stream),
                     h, w3, logits, labels, means, coef, inv_keep, g_z, g_w3, g_b3, g_means, R_bce, R_cst, n_src, levels,
                     C);
  return check_launch("da_ins_tail_backward");
}

extern "C" int dadet_da_ins_dropout_rows(const float* h1, const float* masks, float* out, int64_t numel_per_pass,
                                         int passes, void* stream) {
  DADET_REQUIRE(numel_per_pass >= 0 && numel_per_pass % 4 == 0 && passes > 0, "da_ins_dropout_rows: bad dims");
  if (numel_per_pass == 0) return DADET_OK;
  DADET_REQUIRE(h1 && masks && out && a16(h1) && a16(masks) && a16(out), "da_ins_dropout_rows: bad pointers");
  int64_t blocks = ceil_div64(numel_per_pass / 4 * passes, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(da_ins_dropout_rows_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(h1), reinterpret_cast<const float4*>(masks),
                     reinterpret_cast<float4*>(out), numel_per_pass / 4, passes);
  return check_launch("da_ins_dropout_rows");
}

extern "C" int dadet_da_ins_merge(const float* g, const float* masks, const float* h1, const float* grl, float* g_w,
                                  float* g_x, int64_t numel_per_pass, int passes, void* stream) {
  DADET_REQUIRE(numel_per_pass >= 0 && numel_per_pass % 4 == 0 && passes > 0, "da_ins_merge: bad dims");
  if (numel_per_pass == 0) return DADET_OK;
  DADET_REQUIRE(g && masks && h1 && grl && g_w && a16(g) && a16(masks) && a16(h1) && a16(g_w) && a16(g_x),
                "da_ins_merge: bad pointers");
  int64_t blocks = ceil_div64(numel_per_pass / 4, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(da_ins_merge_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(masks),
                     reinterpret_cast<const float4*>(h1), grl, reinterpret_cast<float4*>(g_w),
                     reinterpret_cast<float4*>(g_x), numel_per_pass / 4, passes);
  return check_launch("da_ins_merge");
}
