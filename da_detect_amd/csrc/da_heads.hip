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
#include "common.h"

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
    int rows_per_image, int C1) {
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
  const int wave_global = blockIdx.x * nw_block + wave;
  const int nwaves = gridDim.x * nw_block;
  for (int64_t m = wave_global; m < M; m += nwaves) {
    const int img = (int)(m / rows_per_image);
    const float y = labels[img];
    const float s = sigmoidf(logits[m]);
    const float4 cf = *reinterpret_cast<const float4*>(coef + img * 4);
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
                     rows_per_image, C1);
  return check_launch("da_img_head_loss_backward");
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
