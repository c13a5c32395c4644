// Deformable convolution v1 / v2 (modulated) sampling kernels for NHWC fp32 tensors on gfx950.
//
// Reference being restated (vendored tree of the reference; the main tree never binds these):
//   tools/cityscapes/maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu
//     deformable_im2col_gpu_kernel              :198-250   (sample = bilinear(x, p + p_k + dp_k), zero outside (-1,H)x(-1,W))
//     modulated_deformable_im2col_gpu_kernel    :578-640   (sample * mask)
//     deformable_col2im_gpu_kernel              :287-342   (input gradient, atomicAdd scatter)
//     deformable_col2im_coord_gpu_kernel        :381-443   (offset gradient through get_coordinate_weight :153-195)
//     modulated_deformable_col2im_coord_gpu_kernel :703-775 (offset + mask gradients)
//   and the host glue tools/cityscapes/maskrcnn_benchmark/csrc/cuda/deform_conv_cuda.cu (im2col + addmm_ per group).
//
// Decomposition here: the reference materialises a [C*kh*kw, N*Ho*Wo] column buffer per im2col_step chunk and
// calls a library GEMM.  Here the sampled operand is written ONCE as an NHWC "column" tensor
// cols[n][ho][wo][tap][c] whose flattened last two axes are exactly the K axis of the implicit-GEMM convolution
// kernel ([Cout][KH][KW][Cin] weights), so the contraction — forward, data gradient and weight gradient — runs on
// the same MFMA kernels as every other convolution (as a 1x1 conv over K = kh*kw*Cin).  Lanes run along the
// channel axis (16 B each): every bilinear corner is one coalesced row read, offsets / masks are read once per
// (pixel, tap) instead of once per (pixel, tap, channel).
#include "common.h"

namespace dadet {

struct DeformGeom {
  int N, H, W, C, KH, KW, stride, pad, dil, dg, Ho, Wo;
};

struct Corner {
  bool valid;           // sample inside (-1,H) x (-1,W)
  int hl, wl;           // floor coordinates
  float lh, lw;         // fractional parts
  bool in1, in2, in3, in4;  // (hl,wl) (hl,wh) (hh,wl) (hh,wh) inside the map
};

// deform_conv_kernel_cuda.cu:92-123 (bilinear) and :236 (validity)
__device__ inline Corner corner_of(float h, float w, int H, int W) {
  Corner c;
  c.valid = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
  const float fh = floorf(h), fw = floorf(w);
  c.hl = (int)fh;
  c.wl = (int)fw;
  c.lh = h - fh;
  c.lw = w - fw;
  const int hh = c.hl + 1, wh = c.wl + 1;
  c.in1 = c.valid && c.hl >= 0 && c.wl >= 0;
  c.in2 = c.valid && c.hl >= 0 && wh <= W - 1;
  c.in3 = c.valid && hh <= H - 1 && c.wl >= 0;
  c.in4 = c.valid && hh <= H - 1 && wh <= W - 1;
  return c;
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one workgroup per output pixel m = (n, ho, wo); loop over taps; lanes over channels
__global__ __launch_bounds__(256) void deform_sample_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ mask,
                                                                float* __restrict__ cols, DeformGeom g) {
  const int m = blockIdx.x;
  const int wo = m % g.Wo;
  const int ho = (m / g.Wo) % g.Ho;
  const int n = m / (g.Wo * g.Ho);
  const int T = g.KH * g.KW;
  const int cpg = g.C / g.dg;
  const float* __restrict__ img = x + (size_t)n * g.H * g.W * g.C;
  const float* __restrict__ off_m = offset + (size_t)m * g.dg * 2 * T;
  const float* __restrict__ msk_m = mask ? mask + (size_t)m * g.dg * T : nullptr;
  float* __restrict__ col_m = cols + (size_t)m * T * g.C;
  for (int tap = 0; tap < T; ++tap) {
    const int i = tap / g.KW, j = tap - i * g.KW;
    for (int c = threadIdx.x * 4; c < g.C; c += blockDim.x * 4) {
      const int grp = c / cpg;
      const float oh = off_m[grp * 2 * T + 2 * tap], ow = off_m[grp * 2 * T + 2 * tap + 1];
      const float mk = msk_m ? msk_m[grp * T + tap] : 1.f;
      const float h_im = (float)(ho * g.stride - g.pad + i * g.dil) + oh;
      const float w_im = (float)(wo * g.stride - g.pad + j * g.dil) + ow;
      const Corner k = corner_of(h_im, w_im, g.H, g.W);
      const float4 v1 = k.in1 ? ld4(img + ((size_t)k.hl * g.W + k.wl) * g.C + c) : zero4();
      const float4 v2 = k.in2 ? ld4(img + ((size_t)k.hl * g.W + k.wl + 1) * g.C + c) : zero4();
      const float4 v3 = k.in3 ? ld4(img + ((size_t)(k.hl + 1) * g.W + k.wl) * g.C + c) : zero4();
      const float4 v4 = k.in4 ? ld4(img + ((size_t)(k.hl + 1) * g.W + k.wl + 1) * g.C + c) : zero4();
      const float hh = 1.f - k.lh, hw = 1.f - k.lw;
      const float w1 = hh * hw, w2 = hh * k.lw, w3 = k.lh * hw, w4 = k.lh * k.lw;
      float4 r;
      r.x = (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * mk;
      r.y = (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * mk;
      r.z = (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * mk;
      r.w = (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * mk;
      *reinterpret_cast<float4*>(col_m + (size_t)tap * g.C + c) = r;
    }
  }
}

// backward of the sampling: gx += w_k * mask * gcol (atomic scatter, as the reference), goffset / gmask by a
// reduction over the channels of the deformable group (wavefront reduce, one atomic per wavefront).
__global__ __launch_bounds__(256) void deform_sample_bwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ gcols,
                                                                float* __restrict__ gx,
                                                                float* __restrict__ goffset,
                                                                float* __restrict__ gmask, DeformGeom g) {
  const int m = blockIdx.x;
  const int wo = m % g.Wo;
  const int ho = (m / g.Wo) % g.Ho;
  const int n = m / (g.Wo * g.Ho);
  const int T = g.KH * g.KW;
  const int cpg = g.C / g.dg;
  const float* __restrict__ img = x + (size_t)n * g.H * g.W * g.C;
  float* __restrict__ gimg = gx ? gx + (size_t)n * g.H * g.W * g.C : nullptr;
  const float* __restrict__ off_m = offset + (size_t)m * g.dg * 2 * T;
  const float* __restrict__ msk_m = mask ? mask + (size_t)m * g.dg * T : nullptr;
  const float* __restrict__ gcol_m = gcols + (size_t)m * T * g.C;
  float* __restrict__ goff_m = goffset + (size_t)m * g.dg * 2 * T;
  float* __restrict__ gmsk_m = gmask ? gmask + (size_t)m * g.dg * T : nullptr;
  const bool wave_uniform_group = (cpg % 256) == 0 || g.dg == 1;  // all 64 lanes of a wave in one group
  for (int tap = 0; tap < T; ++tap) {
    const int i = tap / g.KW, j = tap - i * g.KW;
    for (int c0 = 0; c0 < g.C; c0 += blockDim.x * 4) {
      const int c = c0 + threadIdx.x * 4;
      const bool active = c < g.C;
      const int cc = active ? c : 0;
      const int grp = cc / cpg;
      const float oh = off_m[grp * 2 * T + 2 * tap], ow = off_m[grp * 2 * T + 2 * tap + 1];
      const float mk = msk_m ? msk_m[grp * T + tap] : 1.f;
      const float h_im = (float)(ho * g.stride - g.pad + i * g.dil) + oh;
      const float w_im = (float)(wo * g.stride - g.pad + j * g.dil) + ow;
      const Corner k = corner_of(h_im, w_im, g.H, g.W);
      float d_h = 0.f, d_w = 0.f, d_m = 0.f;
      if (active) {
        const float4 gc = ld4(gcol_m + (size_t)tap * g.C + c);
        const size_t p1 = ((size_t)k.hl * g.W + k.wl) * g.C + c, p2 = p1 + g.C;
        const size_t p3 = p1 + (size_t)g.W * g.C, p4 = p3 + g.C;
        const float4 v1 = k.in1 ? ld4(img + p1) : zero4();
        const float4 v2 = k.in2 ? ld4(img + p2) : zero4();
        const float4 v3 = k.in3 ? ld4(img + p3) : zero4();
        const float4 v4 = k.in4 ? ld4(img + p4) : zero4();
        const float hh = 1.f - k.lh, hw = 1.f - k.lw;
        const float w1 = hh * hw, w2 = hh * k.lw, w3 = k.lh * hw, w4 = k.lh * k.lw;
        const float gv[4] = {gc.x, gc.y, gc.z, gc.w};
        const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w};
        const float a3[4] = {v3.x, v3.y, v3.z, v3.w}, a4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ge = gv[e] * mk;  // gradient wrt the unmasked sample
          if (gimg) {
            if (k.in1) unsafeAtomicAdd(gimg + p1 + e, w1 * ge);
            if (k.in2) unsafeAtomicAdd(gimg + p2 + e, w2 * ge);
            if (k.in3) unsafeAtomicAdd(gimg + p3 + e, w3 * ge);
            if (k.in4) unsafeAtomicAdd(gimg + p4 + e, w4 * ge);
          }
          // get_coordinate_weight (deform_conv_kernel_cuda.cu:153-195): d sample / d h, d sample / d w
          d_h += ge * (hw * (a3[e] - a1[e]) + k.lw * (a4[e] - a2[e]));
          d_w += ge * (hh * (a2[e] - a1[e]) + k.lh * (a4[e] - a3[e]));
          d_m += gv[e] * (w1 * a1[e] + w2 * a2[e] + w3 * a3[e] + w4 * a4[e]);
        }
        if (!k.valid) d_h = d_w = 0.f;
      }
      if (wave_uniform_group) {
        d_h = wave_sum_f(d_h);
        d_w = wave_sum_f(d_w);
        d_m = wave_sum_f(d_m);
        if ((threadIdx.x & 63) == 0 && c0 + (threadIdx.x & ~63) * 4 < g.C) {
          atomicAdd(goff_m + grp * 2 * T + 2 * tap, d_h);
          atomicAdd(goff_m + grp * 2 * T + 2 * tap + 1, d_w);
          if (gmsk_m) atomicAdd(gmsk_m + grp * T + tap, d_m);
        }
      } else if (active) {
        atomicAdd(goff_m + grp * 2 * T + 2 * tap, d_h);
        atomicAdd(goff_m + grp * 2 * T + 2 * tap + 1, d_w);
        if (gmsk_m) atomicAdd(gmsk_m + grp * T + tap, d_m);
      }
    }
  }
}

static int deform_check(const char* who, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                        int dil, int dg, int Ho, int Wo) {
  DADET_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && dil > 0 && dg > 0,
                "%s: bad dims", who);
  DADET_REQUIRE(C % dg == 0 && (C / dg) % 4 == 0, "%s: channels per deformable group must be a multiple of 4", who);
  DADET_REQUIRE(Ho == (H + 2 * pad - (dil * (KH - 1) + 1)) / stride + 1 &&
                    Wo == (W + 2 * pad - (dil * (KW - 1) + 1)) / stride + 1,
                "%s: Ho/Wo do not match the convolution geometry", who);
  return DADET_OK;
}

}  // namespace dadet

using namespace dadet;

static bool al16d(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int dadet_deform_sample_forward(const float* x, const float* offset, const float* mask, float* cols,
                                           int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                                           int dil, int deformable_groups, int Ho, int Wo, void* stream) {
  int rc = deform_check("deform_sample_forward", N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo);
  if (rc) return rc;
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(x && offset && cols && al16d(x) && al16d(cols), "deform_sample_forward: bad pointers");
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo};
  const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(deform_sample_fwd_kernel, dim3((unsigned)(N * Ho * Wo)), dim3(threads), 0,
                     as_stream(stream), x, offset, mask, cols, g);
  return check_launch("deform_sample_forward");
}

extern "C" int dadet_deform_sample_backward(const float* x, const float* offset, const float* mask,
                                            const float* gcols, float* gx, float* goffset, float* gmask, int N,
                                            int H, int W, int C, int KH, int KW, int stride, int pad, int dil,
                                            int deformable_groups, int Ho, int Wo, void* stream) {
  int rc = deform_check("deform_sample_backward", N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo);
  if (rc) return rc;
  if (N == 0) return DADET_OK;
  DADET_REQUIRE(x && offset && gcols && goffset && al16d(x) && al16d(gcols), "deform_sample_backward: bad pointers");
  DADET_REQUIRE(!mask == !gmask || !gmask, "deform_sample_backward: gmask needs mask");
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo};
  const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(deform_sample_bwd_kernel, dim3((unsigned)(N * Ho * Wo)), dim3(threads), 0,
                     as_stream(stream), x, offset, mask, gcols, gx, goffset, gmask, g);
  return check_launch("deform_sample_backward");
}
