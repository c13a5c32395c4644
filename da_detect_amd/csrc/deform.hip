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
#include <cstdlib>

#include "conv_common.h"

namespace dadet {

struct DeformGeom {
  int N, H, W, C, KH, KW, stride, pad, dil, dg, Ho, Wo;
  // floats per output pixel of the offset / mask / offset-gradient / mask-gradient tensors (dense: dg*2*T and dg*T).
  // Larger strides let the kernels read offsets and modulation logits straight out of the offset-predicting conv's
  // (channel-padded) output and write their gradients straight into that conv's output-gradient tensor — no slices,
  // no concatenation (DFConv2d, layers/misc.py:187-188 of the vendored tree slices [:, :18] and [:, -9:])
  int off_ld, mask_ld, goff_ld, gmask_ld;
  int mask_sigmoid;   // the mask tensor holds LOGITS: modulation = sigmoid(logit), gmask is the gradient w.r.t. the logit
  int ablate;         // experiments only (DADET_DEFORM_ABLATE): 1 = pass A without the list appends, 2 = without the channel loop
};

__device__ inline float modulation(const float* __restrict__ msk_m, int idx, int sig) {
  const float v = msk_m[idx];
  return sig ? 1.f / (1.f + expf(-v)) : v;
}

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

// XCD-aware work order, MEASURED and left out (round 3, R-101-FPN-DCN shapes): workgroups are dealt to the 8 XCDs round-robin
// and each XCD has its own L2, so contiguous eighths of the pixels per XCD, and bands of 4 / 8 / 16 image rows dealt
// round-robin, were tried on all three kernels.  With one 64-thread workgroup per pixel the forward went 67 -> 39 us
// (64x128 map) — but so it did in plain order once a workgroup took 256 / (C/4) pixels (41 us): the kernel had been bound
// by the workgroup dispatch rate, not by L2 misses.  On the fat workgroups the order changes nothing (forward 41.1 / 41.6,
// pass A 85 / 86, pass B 51 / 51 us), and contiguous eighths made the 128x256 map 50% slower in pass A.
// a group of C/4 threads per output pixel m = (n, ho, wo), 256 / (C/4) pixels per workgroup (one when C >= 1024 or C/4 does
// not divide 256); loop over taps; lanes over channels.  One 64-thread workgroup per pixel (C = 128: 65 536 of them, half
// their lanes idle) was bound by the workgroup dispatch rate: 137 us for 302 MB.
__global__ __launch_bounds__(256) void deform_sample_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ mask,
                                                                float* __restrict__ cols, DeformGeom g) {
  const int lanes = g.C / 4;
  const bool shared = lanes < (int)blockDim.x && (int)blockDim.x % lanes == 0;    // several pixels per workgroup
  const int ppb = shared ? (int)blockDim.x / lanes : 1;
  const int t = shared ? (int)threadIdx.x % lanes : (int)threadIdx.x;
  const int tstep = shared ? lanes : (int)blockDim.x;
  const int m = (int)blockIdx.x * ppb + (shared ? (int)threadIdx.x / lanes : 0);
  if (m >= g.N * g.Ho * g.Wo) return;
  const int wo = m % g.Wo;
  const int ho = (m / g.Wo) % g.Ho;
  const int n = m / (g.Wo * g.Ho);
  const int T = g.KH * g.KW;
  const int cpg = g.C / g.dg;
  const float* __restrict__ img = x + (size_t)n * g.H * g.W * g.C;
  const float* __restrict__ off_m = offset + (size_t)m * g.off_ld;
  const float* __restrict__ msk_m = mask ? mask + (size_t)m * g.mask_ld : nullptr;
  float* __restrict__ col_m = cols + (size_t)m * T * g.C;
  for (int tap = 0; tap < T; ++tap) {
    const int i = tap / g.KW, j = tap - i * g.KW;
    for (int c = t * 4; c < g.C; c += tstep * 4) {
      const int grp = c / cpg;
      const float oh = off_m[grp * 2 * T + 2 * tap], ow = off_m[grp * 2 * T + 2 * tap + 1];
      const float mk = msk_m ? modulation(msk_m, grp * T + tap, g.mask_sigmoid) : 1.f;
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
  const float* __restrict__ off_m = offset + (size_t)m * g.off_ld;
  const float* __restrict__ msk_m = mask ? mask + (size_t)m * g.mask_ld : nullptr;
  const float* __restrict__ gcol_m = gcols + (size_t)m * T * g.C;
  float* __restrict__ goff_m = goffset + (size_t)m * g.goff_ld;
  float* __restrict__ gmsk_m = gmask ? gmask + (size_t)m * g.gmask_ld : nullptr;
  const bool wave_uniform_group = (cpg % 256) == 0 || g.dg == 1;  // all 64 lanes of a wave in one group
  for (int tap = 0; tap < T; ++tap) {
    const int i = tap / g.KW, j = tap - i * g.KW;
    for (int c0 = 0; c0 < g.C; c0 += blockDim.x * 4) {
      const int c = c0 + threadIdx.x * 4;
      const bool active = c < g.C;
      const int cc = active ? c : 0;
      const int grp = cc / cpg;
      const float oh = off_m[grp * 2 * T + 2 * tap], ow = off_m[grp * 2 * T + 2 * tap + 1];
      const float mk = msk_m ? modulation(msk_m, grp * T + tap, g.mask_sigmoid) : 1.f;
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
        if (g.mask_sigmoid) d_m = d_m * mk * (1.f - mk);
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

// backward of the sampling with the input-gradient accumulated in LDS.  A bilinear sample scatters into 4 input cells
// and a 3x3 deformable conv takes 9 samples per output pixel: 36 global float atomics per output pixel and channel in
// the kernel above (~70 G atomics/s on this part: 1.8 ms for a 64x128x256 map, 27x the forward).  Offsets are a few
// pixels, so the scatter of an 8x8 tile of output pixels lands almost entirely in a 16x16 window of the input: the
// workgroup accumulates that window for 64 channels in LDS (ds_add_f32, lanes = consecutive channels: conflict
// free), sends only out-of-window hits to global memory directly, and flushes the window once — ~9x fewer global atomics.
constexpr int kDTile = 8, kDHalo = 3, kDWin = 16, kDChunk = 64;

__global__ __launch_bounds__(256) void deform_sample_bwd_lds_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ offset,
                                                                    const float* __restrict__ mask,
                                                                    const float* __restrict__ gcols,
                                                                    float* __restrict__ gx,
                                                                    float* __restrict__ goffset,
                                                                    float* __restrict__ gmask, DeformGeom g,
                                                                    int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* win = reinterpret_cast<float*>(smem);   // [kDWin * kDWin][kDChunk]
  const int tile = blockIdx.x;
  const int n = tile / (tiles_x * tiles_y);
  const int ty = (tile / tiles_x) % tiles_y, tx = tile % tiles_x;
  const int y0 = ty * kDTile, x0 = tx * kDTile;                  // first output pixel of the tile (stride 1)
  const int oy = y0 - g.pad - kDHalo, ox = x0 - g.pad - kDHalo;  // input coordinates of window cell (0, 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * kDChunk + lane;
  const int T = g.KH * g.KW;
  const int cpg = g.C / g.dg;
  const int grp = c / cpg;
  const bool uniform_group = (cpg % kDChunk) == 0;   // the 64 channels of a chunk share one deformable group
  for (int i = threadIdx.x; i < kDWin * kDWin * kDChunk; i += 256) win[i] = 0.f;
  __syncthreads();
  const float* __restrict__ img = x + (size_t)n * g.H * g.W * g.C;
  float* __restrict__ gimg = gx ? gx + (size_t)n * g.H * g.W * g.C : nullptr;
  for (int p = wave; p < kDTile * kDTile; p += 4) {
    const int ho = y0 + p / kDTile, wo = x0 + p % kDTile;
    if (ho >= g.Ho || wo >= g.Wo) continue;      // wave-uniform
    const size_t m = ((size_t)n * g.Ho + ho) * g.Wo + wo;
    const float* __restrict__ off_m = offset + m * g.off_ld;
    const float* __restrict__ msk_m = mask ? mask + m * g.mask_ld : nullptr;
    const float* __restrict__ gcol_m = gcols + m * T * g.C;
    float* __restrict__ goff_m = goffset + m * g.goff_ld;
    float* __restrict__ gmsk_m = gmask ? gmask + m * g.gmask_ld : nullptr;
    for (int tap = 0; tap < T; ++tap) {
      const int i = tap / g.KW, j = tap - i * g.KW;
      const float oh = off_m[grp * 2 * T + 2 * tap], ow = off_m[grp * 2 * T + 2 * tap + 1];
      const float mk = msk_m ? modulation(msk_m, grp * T + tap, g.mask_sigmoid) : 1.f;
      const float h_im = (float)(ho - g.pad + i * g.dil) + oh;
      const float w_im = (float)(wo - g.pad + j * g.dil) + ow;
      const Corner k = corner_of(h_im, w_im, g.H, g.W);
      const float gv = gcol_m[(size_t)tap * g.C + c];
      const size_t p1 = ((size_t)k.hl * g.W + k.wl) * g.C + c, p2 = p1 + g.C;
      const size_t p3 = p1 + (size_t)g.W * g.C, p4 = p3 + g.C;
      const float a1 = k.in1 ? img[p1] : 0.f, a2 = k.in2 ? img[p2] : 0.f;
      const float a3 = k.in3 ? img[p3] : 0.f, a4 = k.in4 ? img[p4] : 0.f;
      const float hh = 1.f - k.lh, hw = 1.f - k.lw;
      const float w1 = hh * hw, w2 = hh * k.lw, w3 = k.lh * hw, w4 = k.lh * k.lw;
      const float ge = gv * mk;   // gradient wrt the unmasked sample
      if (gimg) {
        const int cy = k.hl - oy, cx = k.wl - ox;   // window cell of the top-left corner
        const bool r0 = (unsigned)cy < (unsigned)kDWin, r1 = (unsigned)(cy + 1) < (unsigned)kDWin;
        const bool q0 = (unsigned)cx < (unsigned)kDWin, q1 = (unsigned)(cx + 1) < (unsigned)kDWin;
        float* wl = win + ((size_t)cy * kDWin + cx) * kDChunk + lane;
        if (k.in1) { if (r0 && q0) unsafeAtomicAdd(wl, w1 * ge); else unsafeAtomicAdd(gimg + p1, w1 * ge); }
        if (k.in2) { if (r0 && q1) unsafeAtomicAdd(wl + kDChunk, w2 * ge); else unsafeAtomicAdd(gimg + p2, w2 * ge); }
        if (k.in3) { if (r1 && q0) unsafeAtomicAdd(wl + kDWin * kDChunk, w3 * ge); else unsafeAtomicAdd(gimg + p3, w3 * ge); }
        if (k.in4) { if (r1 && q1) unsafeAtomicAdd(wl + (kDWin + 1) * kDChunk, w4 * ge); else unsafeAtomicAdd(gimg + p4, w4 * ge); }
      }
      // get_coordinate_weight (deform_conv_kernel_cuda.cu:153-195): d sample / d h, d sample / d w
      float d_h = ge * (hw * (a3 - a1) + k.lw * (a4 - a2));
      float d_w = ge * (hh * (a2 - a1) + k.lh * (a4 - a3));
      float d_m = gv * (w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4);
      if (!k.valid) d_h = d_w = 0.f;
      if (g.mask_sigmoid) d_m = d_m * mk * (1.f - mk);
      if (uniform_group) {
        d_h = wave_sum_f(d_h);
        d_w = wave_sum_f(d_w);
        d_m = wave_sum_f(d_m);
        if (lane == 0) {
          atomicAdd(goff_m + grp * 2 * T + 2 * tap, d_h);
          atomicAdd(goff_m + grp * 2 * T + 2 * tap + 1, d_w);
          if (gmsk_m) atomicAdd(gmsk_m + grp * T + tap, d_m);
        }
      } else {
        atomicAdd(goff_m + grp * 2 * T + 2 * tap, d_h);
        atomicAdd(goff_m + grp * 2 * T + 2 * tap + 1, d_w);
        if (gmsk_m) atomicAdd(gmsk_m + grp * T + tap, d_m);
      }
    }
  }
  __syncthreads();
  if (gimg) {
    for (int cell = wave; cell < kDWin * kDWin; cell += 4) {
      const int gy = oy + cell / kDWin, gxx = ox + cell % kDWin;
      if ((unsigned)gy >= (unsigned)g.H || (unsigned)gxx >= (unsigned)g.W) continue;
      const float v = win[(size_t)cell * kDChunk + lane];
      if (v != 0.f) unsafeAtomicAdd(gimg + ((size_t)gy * g.W + gxx) * g.C + c, v);
    }
  }
}

// ---- backward of the sampling without float atomics on the input gradient (round 3) --------------------------------
// Measured (tools/deform_bwd_bench.py, res4 shape 2 x 64 x 128 x 256): the LDS-window kernel above spends 0.73 of its
// 0.85 ms in the window's ds_add_f32 — LDS float atomics retire at ~0.4 lanes per clock and CU on this part — and the
// reference's global-atomic form takes 1.9 ms.  The scatter is therefore turned into a GATHER through per-cell lists, in
// two launches:
//
//  pass A (deform_sample_bwd_coord_kernel), sample-centric.  A wavefront takes 4 consecutive output pixels (16 lanes x 4
//    channels each) and walks groups -> taps (three per step, their 15 sixteen-byte buffer loads in flight together) ->
//    64-channel chunks: the gradients w.r.t. offsets / modulation are reductions over ALL channels of a (pixel, tap, group)
//    sample, finished by one 4-step reduction over the 16 lanes and written with a plain store (one writer per address: no
//    atomics, no zero fill, deterministic).  It also appends the sample to the list of the position its bilinear footprint
//    starts at — an INTEGER atomic per sample, 9 per output pixel instead of 36 float atomics per output pixel AND CHANNEL.
//  pass B (deform_gx_gather_kernel), cell-centric: a quarter-wavefront per (input cell, 64-channel chunk) walks the four
//    lists that cover the cell — eight entries per round, their gcols rows loaded together — and accumulates weight * gcols
//    in registers, then adds the sum to gx with one plain 16-byte read-modify-write (it owns the address).  Offsets of any
//    size are handled alike; gcols is read ~4 times (once per corner), rows of C * 4 bytes.
// A list belongs to a TOP-LEFT position (hl, wl) in [-1, H-1] x [-1, W-1] and a deformable group, and holds one 16-byte entry
// {sample, lh, lw, modulation} per (pixel, tap) whose bilinear footprint starts there: ONE integer atomic per sample (9 per
// output pixel) — the first version of this scheme kept one list per CELL with an entry per corner, 36 atomics per pixel, and
// the stage ablation (DADET_DEFORM_ABLATE, `tools/probes/deform_prof.sh`; 64x128x256 / 128x256x128 / 32x64x512 maps) showed
// them to cost as much as the whole channel loop: pass A 84 / 298 / 50 us, without the appends 60 / 162 / 36, appends alone
// 52 / 195 / 25.  A cell gathers from the four lists whose footprint covers it — top-left positions (y, x), (y, x-1), (y-1, x),
// (y-1, x-1), as corner 0 .. 3 — and rebuilds its corner weight from (lh, lw) with the forward's own expression.
// Lists hold kListCap entries (a regular 3x3 pattern puts 9 on a position); a sample that finds its list full is added to gx
// directly with float atomics by pass A (gx is zero-filled by the caller).  The order of a list — hence the last bits of gx
// — depends on the arrival order of the appends.
constexpr int kListCap = 32;

struct ListEntry {
  unsigned sample;   // (m * T + tap): row of gcols
  float lh, lw;      // fractional parts of the sampling position
  float mk;          // modulation
};

__device__ inline size_t tl_list(const DeformGeom& g, int n, int hl, int wl, int grp) {
  return (((size_t)n * (g.H + 1) + (hl + 1)) * (g.W + 1) + (wl + 1)) * g.dg + grp;
}

__global__ __launch_bounds__(256) void deform_sample_bwd_coord_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ offset,
                                                                      const float* __restrict__ mask,
                                                                      const float* __restrict__ gcols,
                                                                      float* __restrict__ gx,
                                                                      float* __restrict__ goffset,
                                                                      float* __restrict__ gmask, int* __restrict__ counts,
                                                                      ListEntry* __restrict__ entries, DeformGeom g,
                                                                      unsigned* __restrict__ amax_g) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, l16 = lane & 15;           // pixel of the wavefront's group of four, channel quad
  const int T = g.KH * g.KW;
  const int cpg = g.C / g.dg;
  const int64_t grp4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int wo4 = (g.Wo + 3) / 4;
  float amx = 0.f;      // max|.| of the offset / modulation gradients this lane stores (amax_g: the offset conv's weight
                        // gradient reads them as a GEMM operand of contraction mode 4)
  if (grp4 < (int64_t)g.N * g.Ho * wo4) {
  const int n = (int)(grp4 / ((int64_t)g.Ho * wo4));
  const int ho = (int)((grp4 / wo4) % g.Ho);
  const int wo = (int)(grp4 % wo4) * 4 + sub;
  const bool live = wo < g.Wo;
  const size_t m = ((size_t)n * g.Ho + ho) * g.Wo + (live ? wo : 0);
  const float* __restrict__ img = x + (size_t)n * g.H * g.W * g.C;
  float* __restrict__ gimg = gx ? gx + (size_t)n * g.H * g.W * g.C : nullptr;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(img, (unsigned)((size_t)g.H * g.W * g.C * 4));
  const __amdgpu_buffer_rsrc_t gr = make_rsrc(gcols, (unsigned)((size_t)g.N * g.Ho * g.Wo * T * g.C * 4));
  const float* __restrict__ off_m = offset + m * g.off_ld;
  const float* __restrict__ msk_m = mask ? mask + m * g.mask_ld : nullptr;
  float* __restrict__ goff_m = goffset + m * g.goff_ld;
  float* __restrict__ gmsk_m = gmask ? gmask + m * g.gmask_ld : nullptr;
  for (int grp = 0; grp < g.dg; ++grp) {
    for (int tap0 = 0; tap0 < T; tap0 += 3) {
      Corner k[3];
      float mk[3], cw[3][4];
      unsigned full[3];       // bit q: corner q found its list full -> float atomics on gx (all lanes of the pixel)
      unsigned b1o[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int tap = tap0 + u;
        const bool on = live && tap < T;
        const int tt = on ? tap : 0;
        const int i = tt / g.KW, j = tt - i * g.KW;
        const float oh = off_m[grp * 2 * T + 2 * tt], ow = off_m[grp * 2 * T + 2 * tt + 1];
        mk[u] = msk_m ? modulation(msk_m, grp * T + tt, g.mask_sigmoid) : 1.f;
        const float h_im = (float)(ho * g.stride - g.pad + i * g.dil) + oh;
        const float w_im = (float)(wo * g.stride - g.pad + j * g.dil) + ow;
        k[u] = corner_of(h_im, w_im, g.H, g.W);
        if (!on) k[u].valid = k[u].in1 = k[u].in2 = k[u].in3 = k[u].in4 = false;
        const float hh = 1.f - k[u].lh, hw = 1.f - k[u].lw;
        cw[u][0] = hh * hw; cw[u][1] = hh * k[u].lw; cw[u][2] = k[u].lh * hw; cw[u][3] = k[u].lh * k[u].lw;
        b1o[u] = (unsigned)(((size_t)k[u].hl * g.W + k[u].wl) * g.C) * 4u;
        full[u] = 0;
        // one lane per pixel appends the sample to the list of its top-left position.  (MEASURED on the per-corner lists:
        // the 12 (tap, corner) pairs of a step appended from 12 different lanes at once — one atomic round trip per step
        // instead of 12 — made the kernel SLOWER, 86 -> 117 us on the 64x128 map: neighbouring pixels hit the same
        // counters, and it is that contention, not the latency of one lane's chain, that the kernel waits for.)
        if (gimg && l16 == 0 && g.ablate != 1 && k[u].valid) {
          const size_t list = tl_list(g, n, k[u].hl, k[u].wl, grp);
          const int slot = atomicAdd(counts + list, 1);
          if (slot < kListCap) {
            ListEntry en;
            en.sample = (unsigned)(m * T + tt);
            en.lh = k[u].lh;
            en.lw = k[u].lw;
            en.mk = mk[u];
            entries[list * kListCap + slot] = en;
          } else {          // list full: the corners inside the map go to gx directly (all lanes of the pixel, below)
            full[u] = (k[u].in1 ? 1u : 0u) | (k[u].in2 ? 2u : 0u) | (k[u].in3 ? 4u : 0u) | (k[u].in4 ? 8u : 0u);
          }
        }
        full[u] = __shfl(full[u], lane & 48, 64);
      }
      float d_h[3] = {0.f, 0.f, 0.f}, d_w[3] = {0.f, 0.f, 0.f}, d_m[3] = {0.f, 0.f, 0.f};
      for (int c = grp * cpg + l16 * 4; c < (g.ablate == 2 ? 0 : (grp + 1) * cpg); c += kDChunk) {
        float4 gv[3], a1[3], a2[3], a3[3], a4[3];
        const unsigned rowb = (unsigned)(g.W * g.C) * 4u, pixb = (unsigned)g.C * 4u, cb = (unsigned)c * 4u;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          // buffer loads: a corner outside the map / a tap beyond the last one is an out-of-range offset that returns
          // 0 — no branches, all 15 loads of the three taps are in flight together (conv_common.h)
          const bool on = live && tap0 + u < T;
          gv[u] = buf_load4(gr, on ? (unsigned)((m * T + tap0 + u) * g.C + c) * 4u : kOOB);
          a1[u] = buf_load4(xr, k[u].in1 ? b1o[u] + cb : kOOB);
          a2[u] = buf_load4(xr, k[u].in2 ? b1o[u] + pixb + cb : kOOB);
          a3[u] = buf_load4(xr, k[u].in3 ? b1o[u] + rowb + cb : kOOB);
          a4[u] = buf_load4(xr, k[u].in4 ? b1o[u] + rowb + pixb + cb : kOOB);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float hh = 1.f - k[u].lh, hw = 1.f - k[u].lw;
          const float g4[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
          const float b1[4] = {a1[u].x, a1[u].y, a1[u].z, a1[u].w}, b2[4] = {a2[u].x, a2[u].y, a2[u].z, a2[u].w};
          const float b3[4] = {a3[u].x, a3[u].y, a3[u].z, a3[u].w}, b4[4] = {a4[u].x, a4[u].y, a4[u].z, a4[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ge = g4[e] * mk[u];   // gradient wrt the unmasked sample
            // get_coordinate_weight (deform_conv_kernel_cuda.cu:153-195): d sample / d h, d sample / d w
            d_h[u] += ge * (hw * (b3[e] - b1[e]) + k[u].lw * (b4[e] - b2[e]));
            d_w[u] += ge * (hh * (b2[e] - b1[e]) + k[u].lh * (b4[e] - b3[e]));
            d_m[u] += g4[e] * (cw[u][0] * b1[e] + cw[u][1] * b2[e] + cw[u][2] * b3[e] + cw[u][3] * b4[e]);
          }
          if (full[u]) {      // overflowed lists (rare): this pixel's corners go to gx directly
            float* base = gimg + ((size_t)k[u].hl * g.W + k[u].wl) * g.C + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ge = g4[e] * mk[u];
              if (full[u] & 1u) unsafeAtomicAdd(base + e, cw[u][0] * ge);
              if (full[u] & 2u) unsafeAtomicAdd(base + g.C + e, cw[u][1] * ge);
              if (full[u] & 4u) unsafeAtomicAdd(base + (size_t)g.W * g.C + e, cw[u][2] * ge);
              if (full[u] & 8u) unsafeAtomicAdd(base + (size_t)g.W * g.C + g.C + e, cw[u][3] * ge);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int tap = tap0 + u;
        if (!k[u].valid) d_h[u] = d_w[u] = 0.f;
        if (g.mask_sigmoid) d_m[u] = d_m[u] * mk[u] * (1.f - mk[u]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {      // sum over the 16 lanes of the pixel
          d_h[u] += __shfl_xor(d_h[u], o, 64);
          d_w[u] += __shfl_xor(d_w[u], o, 64);
          d_m[u] += __shfl_xor(d_m[u], o, 64);
        }
        if (l16 == 0 && live && tap < T) {     // the only writer of these addresses
          goff_m[grp * 2 * T + 2 * tap] = d_h[u];
          goff_m[grp * 2 * T + 2 * tap + 1] = d_w[u];
          amx = fmaxf(amx, fmaxf(fabsf(d_h[u]), fabsf(d_w[u])));
          if (gmsk_m) {
            gmsk_m[grp * T + tap] = d_m[u];
            amx = fmaxf(amx, fabsf(d_m[u]));
          }
        }
      }
    }
  }
  }
  if (amax_g) amax_publish(amax_g, amx);      // (every thread of the workgroup: it contains barriers)
}

// pass B: gx[cell][ch] += sum over the four lists covering the cell of weight * gcols[sample][ch]; a quarter-wavefront per
// (cell, chunk)
__global__ __launch_bounds__(256) void deform_gx_gather_kernel(const float* __restrict__ gcols, float* __restrict__ gx,
                                                               const int* __restrict__ counts,
                                                               const ListEntry* __restrict__ entries, DeformGeom g) {
  const int l16 = threadIdx.x & 15;
  const int chunks = g.C / kDChunk;
  const int cpg = g.C / g.dg;
  const int64_t quarter = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t cells = (int64_t)g.N * g.H * g.W;
  const int chunk = (int)(quarter % chunks);
  const int64_t cell = quarter / chunks;
  if (cell >= cells) return;
  const int c = chunk * kDChunk + l16 * 4;
  const int grp = c / cpg;
  const int x = (int)(cell % g.W), y = (int)((cell / g.W) % g.H), n = (int)(cell / ((int64_t)g.W * g.H));
  const int T = g.KH * g.KW;
  const __amdgpu_buffer_rsrc_t gr = make_rsrc(gcols, (unsigned)((size_t)g.N * g.Ho * g.Wo * T * g.C * 4));
  // corner q of a sample whose footprint starts at (hl, wl) is the cell (hl + (q >> 1), wl + (q & 1))
  size_t lst[4];
  int end[4];                       // running ends of the four lists laid end to end
  int total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    lst[q] = tl_list(g, n, y - (q >> 1), x - (q & 1), grp);
    total += min(counts[lst[q]], kListCap);
    end[q] = total;
  }
  float4 acc = zero4();
  const int lane = threadIdx.x & 63;
  // The 16 lanes of the quarter share the entries: lane u < 8 decodes entry e0 + u (which list, the corner weight) once, the
  // others pick it up by a lane shuffle — decoding all eight in every lane made this kernel VALU-bound (51 -> 86 us on the
  // 64x128 map when the lists moved from cells to top-left positions).  The next round's entries are fetched and decoded
  // while this round's gcols rows are in flight.
  auto decode = [&](const int e0, unsigned& smp_out, float& wgt_out) {
    smp_out = 0xFFFFFFFFu;
    wgt_out = 0.f;
    const int e = e0 + (l16 & 7);
    if (e < total) {
      const int q = (e >= end[0]) + (e >= end[1]) + (e >= end[2]);
      const int first = q == 0 ? 0 : (q == 1 ? end[0] : (q == 2 ? end[1] : end[2]));
      const size_t l = q == 0 ? lst[0] : (q == 1 ? lst[1] : (q == 2 ? lst[2] : lst[3]));
      const ListEntry en = entries[l * kListCap + (e - first)];
      // the forward's corner weights (hh * hw, hh * lw, lh * hw, lh * lw), then the modulation — same expression order
      const float hh = 1.f - en.lh, hw = 1.f - en.lw;
      const float a = (q & 2) ? en.lh : hh, b = (q & 1) ? en.lw : hw;
      smp_out = en.sample;
      wgt_out = (a * b) * en.mk;
    }
  };
  unsigned my_smp;
  float my_wgt;
  decode(0, my_smp, my_wgt);
  for (int e0 = 0; e0 < total; e0 += 8) {
    unsigned smp[8];
    float wgt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      smp[u] = (unsigned)__shfl((int)my_smp, (lane & 48) | u, 64);
      wgt[u] = __shfl(my_wgt, (lane & 48) | u, 64);
    }
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = buf_load4(gr, smp[u] != 0xFFFFFFFFu ? (unsigned)((size_t)smp[u] * g.C + c) * 4u : kOOB);
    decode(e0 + 8, my_smp, my_wgt);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += wgt[u] * v[u].x; acc.y += wgt[u] * v[u].y;
      acc.z += wgt[u] * v[u].z; acc.w += wgt[u] * v[u].w;
    }
  }
  float4* dst = reinterpret_cast<float4*>(gx + (size_t)cell * g.C + c);
  float4 o = *dst;          // the caller's zero fill + whatever overflowed lists sent here with atomics in pass A
  o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
  *dst = o;
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

static int deform_sample_forward_impl(const float* x, const float* offset, const float* mask, float* cols, DeformGeom g,
                                      void* stream) {
  const int lanes = g.C / 4;
  const int threads = (lanes < 256 && 256 % lanes == 0) ? 256 : ((lanes >= 256) ? 256 : ((lanes + 63) / 64) * 64);
  const int ppb = (lanes < threads && threads % lanes == 0) ? threads / lanes : 1;      // output pixels per workgroup
  int rc = deform_check("deform_sample_forward", g.N, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.dil, g.dg, g.Ho, g.Wo);
  if (rc) return rc;
  if (g.N == 0) return DADET_OK;
  DADET_REQUIRE(x && offset && cols && al16d(x) && al16d(cols), "deform_sample_forward: bad pointers");
  const int T = g.KH * g.KW;
  DADET_REQUIRE(g.off_ld >= g.dg * 2 * T && (!mask || g.mask_ld >= g.dg * T), "deform_sample_forward: row stride too small");
  hipLaunchKernelGGL(deform_sample_fwd_kernel, dim3((unsigned)ceil_div64((int64_t)g.N * g.Ho * g.Wo, ppb)),
                     dim3(threads), 0,
                     as_stream(stream), x, offset, mask, cols, g);
  return check_launch("deform_sample_forward");
}

extern "C" int dadet_deform_sample_forward(const float* x, const float* offset, const float* mask, float* cols,
                                           int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                                           int dil, int deformable_groups, int Ho, int Wo, void* stream) {
  const int T = KH * KW, dg = deformable_groups;
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, dg, Ho, Wo, dg * 2 * T, dg * T, dg * 2 * T, dg * T, 0};
  return deform_sample_forward_impl(x, offset, mask, cols, g, stream);
}

extern "C" int dadet_deform_sample_forward_ld(const float* x, const float* offset, int offset_ld, const float* mask,
                                              int mask_ld, int mask_is_logit, float* cols, int N, int H, int W, int C,
                                              int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                                              int Ho, int Wo, void* stream) {
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo, offset_ld, mask_ld, 0, 0, mask_is_logit};
  return deform_sample_forward_impl(x, offset, mask, cols, g, stream);
}

static int deform_sample_backward_impl(const float* x, const float* offset, const float* mask, const float* gcols,
                                       float* gx, float* goffset, float* gmask, DeformGeom g, void* workspace,
                                       size_t workspace_bytes, void* stream, unsigned* amax_g = nullptr,
                                       bool* amax_done = nullptr) {
  static const int ablate = getenv("DADET_DEFORM_ABLATE") ? atoi(getenv("DADET_DEFORM_ABLATE")) : 0;
  g.ablate = ablate;
  int rc = deform_check("deform_sample_backward", g.N, g.H, g.W, g.C, g.KH, g.KW, g.stride, g.pad, g.dil, g.dg, g.Ho, g.Wo);
  if (rc) return rc;
  if (g.N == 0) return DADET_OK;
  DADET_REQUIRE(x && offset && gcols && goffset && al16d(x) && al16d(gcols), "deform_sample_backward: bad pointers");
  DADET_REQUIRE(!mask == !gmask || !gmask, "deform_sample_backward: gmask needs mask");
  const int T = g.KH * g.KW;
  DADET_REQUIRE(g.off_ld >= g.dg * 2 * T && g.goff_ld >= g.dg * 2 * T && (!mask || g.mask_ld >= g.dg * T) &&
                    (!gmask || g.gmask_ld >= g.dg * T),
                "deform_sample_backward: row stride too small");
  static const bool allow_lds = !(getenv("DADET_DEFORM_BWD_LDS") && getenv("DADET_DEFORM_BWD_LDS")[0] == '0');
  if (allow_lds && g.stride == 1 && g.dil * (g.KH - 1) <= 2 && g.dil * (g.KW - 1) <= 2 && g.C % kDChunk == 0) {
    const int tiles_x = ceil_div(g.Wo, kDTile), tiles_y = ceil_div(g.Ho, kDTile);
    const size_t lds = sizeof(float) * kDWin * kDWin * kDChunk;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deform_sample_bwd_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("deform_sample_backward: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DADET_ELAUNCH;
      }
      attr_set = true;
    }
    static const bool window = getenv("DADET_DEFORM_BWD_LDS") && getenv("DADET_DEFORM_BWD_LDS")[0] == '1';
    const size_t lists = (size_t)g.N * (g.H + 1) * (g.W + 1) * g.dg;
    const size_t need = lists * sizeof(int) + 256 + lists * kListCap * sizeof(ListEntry);
    if (!window && workspace && workspace_bytes >= need && ((g.C / g.dg) % kDChunk) == 0 &&
        (((uintptr_t)x | (uintptr_t)gcols | (uintptr_t)gx | (uintptr_t)workspace) & 15) == 0 &&
        (uint64_t)g.N * g.Ho * g.Wo * T * g.C * 4 < 0xFFFFFFF0ull && (uint64_t)g.N * g.Ho * g.Wo * T < 0xFFFFFFFFull) {
      int* counts = static_cast<int*>(workspace);
      ListEntry* entries = reinterpret_cast<ListEntry*>(static_cast<char*>(workspace) + ((lists * sizeof(int) + 255) & ~(size_t)255));
      if (gx) (void)hipMemsetAsync(counts, 0, lists * sizeof(int), as_stream(stream));
      // pass A: offset / modulation gradients + the per-cell lists, one wavefront per 4 pixels
      const int64_t waves_a = (int64_t)g.N * g.Ho * ((g.Wo + 3) / 4);
      hipLaunchKernelGGL(deform_sample_bwd_coord_kernel, dim3((unsigned)ceil_div64(waves_a, 4)), dim3(256), 0,
                         as_stream(stream), x, offset, mask, gcols, gx, goffset, gmask, counts, entries, g, amax_g);
      if (amax_done) *amax_done = amax_g != nullptr;
      if (gx) {   // pass B: the input gradient, gathered per cell (behind pass A in stream order)
        const int64_t quarters = (int64_t)g.N * g.H * g.W * (g.C / kDChunk);
        hipLaunchKernelGGL(deform_gx_gather_kernel, dim3((unsigned)ceil_div64(quarters, 16)), dim3(256), 0,
                           as_stream(stream), gcols, gx, counts, entries, g);
      }
      return check_launch("deform_sample_backward(gather)");
    }
    hipLaunchKernelGGL(deform_sample_bwd_lds_kernel, dim3((unsigned)(g.N * tiles_x * tiles_y), (unsigned)(g.C / kDChunk)),
                       dim3(256), lds, as_stream(stream), x, offset, mask, gcols, gx, goffset, gmask, g, tiles_x,
                       tiles_y);
    return check_launch("deform_sample_backward(lds)");
  }
  const int threads = (g.C / 4 >= 256) ? 256 : ((g.C / 4 + 63) / 64) * 64;
  hipLaunchKernelGGL(deform_sample_bwd_kernel, dim3((unsigned)(g.N * g.Ho * g.Wo)), dim3(threads), 0,
                     as_stream(stream), x, offset, mask, gcols, gx, goffset, gmask, g);
  return check_launch("deform_sample_backward");
}

extern "C" int dadet_deform_sample_backward(const float* x, const float* offset, const float* mask,
                                            const float* gcols, float* gx, float* goffset, float* gmask, int N,
                                            int H, int W, int C, int KH, int KW, int stride, int pad, int dil,
                                            int deformable_groups, int Ho, int Wo, void* stream) {
  const int T = KH * KW, dg = deformable_groups;
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, dg, Ho, Wo, dg * 2 * T, dg * T, dg * 2 * T, dg * T, 0};
  return deform_sample_backward_impl(x, offset, mask, gcols, gx, goffset, gmask, g, nullptr, 0, stream);
}

extern "C" int dadet_deform_sample_backward_workspace_bytes(int N, int H, int W, int deformable_groups, size_t* bytes_out) {
  DADET_REQUIRE(N >= 0 && H > 0 && W > 0 && deformable_groups > 0 && bytes_out, "deform_sample_backward_workspace_bytes: bad args");
  const size_t lists = (size_t)N * (H + 1) * (W + 1) * deformable_groups;
  *bytes_out = lists * sizeof(int) + 256 + lists * kListCap * sizeof(ListEntry);
  return DADET_OK;
}

extern "C" int dadet_deform_sample_backward_ld(const float* x, const float* offset, int offset_ld, const float* mask,
                                               int mask_ld, int mask_is_logit, const float* gcols, float* gx,
                                               float* goffset, int goffset_ld, float* gmask, int gmask_ld, int N, int H,
                                               int W, int C, int KH, int KW, int stride, int pad, int dil,
                                               int deformable_groups, int Ho, int Wo, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo, offset_ld, mask_ld, goffset_ld, gmask_ld,
               mask_is_logit};
  return deform_sample_backward_impl(x, offset, mask, gcols, gx, goffset, gmask, g, workspace, workspace_bytes, stream);
}

// The same with the largest magnitude of the offset-conv output gradient left in `amax_gom` (a zero-initialised slot of
// dadet_amax's layout): goffset and gmask are two column ranges of ONE [N*Ho*Wo][goffset_ld] tensor whose other columns the
// caller zero-filled (gom of DFConv2d's offset branch), `gom_floats` its size.  The gather form's first pass leaves the
// maximum as it stores the gradients; the other forms are followed by one dadet_amax pass over the tensor.
extern "C" int dadet_deform_sample_backward_ld_m(const float* x, const float* offset, int offset_ld, const float* mask,
                                                 int mask_ld, int mask_is_logit, const float* gcols, float* gx,
                                                 float* goffset, int goffset_ld, float* gmask, int gmask_ld, int N, int H,
                                                 int W, int C, int KH, int KW, int stride, int pad, int dil,
                                                 int deformable_groups, int Ho, int Wo, void* workspace,
                                                 size_t workspace_bytes, float* amax_gom, long long gom_floats,
                                                 void* stream) {
  DeformGeom g{N, H, W, C, KH, KW, stride, pad, dil, deformable_groups, Ho, Wo, offset_ld, mask_ld, goffset_ld, gmask_ld,
               mask_is_logit};
  bool done = false;
  const int rc = deform_sample_backward_impl(x, offset, mask, gcols, gx, goffset, gmask, g, workspace, workspace_bytes,
                                             stream, reinterpret_cast<unsigned*>(amax_gom), &done);
  if (rc || !amax_gom || done || N == 0) return rc;
  return dadet_amax(goffset, gom_floats, amax_gom, stream);
}

// ------------------------------------------------------------------------------------------------
// Deformable position-sensitive ROI pooling (reference: tools/cityscapes/maskrcnn_benchmark/csrc/cuda/
// deform_pool_kernel_cuda.cu:30-141 forward, :143-264 backward).  NHWC data [B][H][W][C], C = output_dim * gs * gs;
// output / count [R][PH][PW][output_dim]; trans [R][num_classes*2][part][part] (reference layout, tiny).
// One thread per output element with the output channel fastest, so a wavefront covers consecutive channels of
// one bin: all lanes share the bin's sample coordinates.
namespace dadet {

struct PsGeom {
  int B, H, W, C, R, PH, PW, out_dim, gs, part, spp, no_trans, num_classes, ch_each_class;
  float scale, trans_std;
};

struct PsBin {
  int batch, part_h, part_w, class_id, gh, gw;
  float wstart, hstart, sub_w, sub_h, roi_w, roi_h;
};

__device__ inline PsBin ps_bin(const PsGeom& g, const float* __restrict__ rois, const float* __restrict__ trans,
                               int n, int ctop, int ph, int pw) {
  PsBin b;
  const float* r = rois + (size_t)n * 5;
  b.batch = (int)r[0];
  const float sw = roundf(r[1]) * g.scale - 0.5f, sh = roundf(r[2]) * g.scale - 0.5f;
  const float ew = (roundf(r[3]) + 1.f) * g.scale - 0.5f, eh = (roundf(r[4]) + 1.f) * g.scale - 0.5f;
  b.roi_w = fmaxf(ew - sw, 0.1f);
  b.roi_h = fmaxf(eh - sh, 0.1f);
  const float bin_h = b.roi_h / (float)g.PH, bin_w = b.roi_w / (float)g.PW;
  b.sub_h = bin_h / (float)g.spp;
  b.sub_w = bin_w / (float)g.spp;
  b.part_h = (int)floorf((float)ph / (float)g.PH * (float)g.part);
  b.part_w = (int)floorf((float)pw / (float)g.PW * (float)g.part);
  b.class_id = ctop / g.ch_each_class;
  float tx = 0.f, ty = 0.f;
  if (!g.no_trans) {
    const size_t base = ((size_t)n * g.num_classes + b.class_id) * 2;
    tx = trans[((base)*g.part + b.part_h) * g.part + b.part_w] * g.trans_std;
    ty = trans[((base + 1) * g.part + b.part_h) * g.part + b.part_w] * g.trans_std;
  }
  b.wstart = (float)pw * bin_w + sw + tx * b.roi_w;
  b.hstart = (float)ph * bin_h + sh + ty * b.roi_h;
  int gw = (int)floorf((float)pw * (float)g.gs / (float)g.PW);
  int gh = (int)floorf((float)ph * (float)g.gs / (float)g.PH);
  b.gw = min(max(gw, 0), g.gs - 1);
  b.gh = min(max(gh, 0), g.gs - 1);
  return b;
}

__global__ void deform_psroi_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois,
                                        const float* __restrict__ trans, float* __restrict__ out,
                                        float* __restrict__ cnt, PsGeom g) {
  const int64_t total = (int64_t)g.R * g.PH * g.PW * g.out_dim;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ctop = (int)(idx % g.out_dim);
    const int pw = (int)((idx / g.out_dim) % g.PW);
    const int ph = (int)((idx / g.out_dim / g.PW) % g.PH);
    const int n = (int)(idx / g.out_dim / g.PW / g.PH);
    const PsBin b = ps_bin(g, rois, trans, n, ctop, ph, pw);
    const float* img = data + (size_t)b.batch * g.H * g.W * g.C;
    const int c = (ctop * g.gs + b.gh) * g.gs + b.gw;
    float sum = 0.f;
    int count = 0;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w = b.wstart + (float)iw * b.sub_w, h = b.hstart + (float)ih * b.sub_h;
        if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
        const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        const float dx = w - (float)x1, dy = h - (float)y1;
        const float v11 = img[((size_t)y1 * g.W + x1) * g.C + c], v12 = img[((size_t)y2 * g.W + x1) * g.C + c];
        const float v21 = img[((size_t)y1 * g.W + x2) * g.C + c], v22 = img[((size_t)y2 * g.W + x2) * g.C + c];
        sum += (1.f - dx) * (1.f - dy) * v11 + (1.f - dx) * dy * v12 + dx * (1.f - dy) * v21 + dx * dy * v22;
        ++count;
      }
    out[idx] = count == 0 ? 0.f : sum / (float)count;
    cnt[idx] = (float)count;
  }
}

__global__ void deform_psroi_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ cnt,
                                        const float* __restrict__ data, const float* __restrict__ rois,
                                        const float* __restrict__ trans, float* __restrict__ gdata,
                                        float* __restrict__ gtrans, PsGeom g) {
  const int64_t total = (int64_t)g.R * g.PH * g.PW * g.out_dim;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (cnt[idx] <= 0.f) continue;
    const int ctop = (int)(idx % g.out_dim);
    const int pw = (int)((idx / g.out_dim) % g.PW);
    const int ph = (int)((idx / g.out_dim / g.PW) % g.PH);
    const int n = (int)(idx / g.out_dim / g.PW / g.PH);
    const PsBin b = ps_bin(g, rois, trans, n, ctop, ph, pw);
    const float diff = gout[idx] / cnt[idx];
    const float* img = data + (size_t)b.batch * g.H * g.W * g.C;
    float* gimg = gdata + (size_t)b.batch * g.H * g.W * g.C;
    const int c = (ctop * g.gs + b.gh) * g.gs + b.gw;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w = b.wstart + (float)iw * b.sub_w, h = b.hstart + (float)ih * b.sub_h;
        if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        const size_t p00 = ((size_t)y0 * g.W + x0) * g.C + c, p01 = ((size_t)y1 * g.W + x0) * g.C + c;
        const size_t p10 = ((size_t)y0 * g.W + x1) * g.C + c, p11 = ((size_t)y1 * g.W + x1) * g.C + c;
        unsafeAtomicAdd(gimg + p00, (1.f - dx) * (1.f - dy) * diff);
        unsafeAtomicAdd(gimg + p01, (1.f - dx) * dy * diff);
        unsafeAtomicAdd(gimg + p10, dx * (1.f - dy) * diff);
        unsafeAtomicAdd(gimg + p11, dx * dy * diff);
        if (g.no_trans) continue;
        const float U00 = img[p00], U01 = img[p01], U10 = img[p10], U11 = img[p11];
        float gx = (U11 * dy + U10 * (1.f - dy) - U01 * dy - U00 * (1.f - dy)) * g.trans_std * diff;
        gx *= b.roi_w;
        float gy = (U11 * dx + U01 * (1.f - dx) - U10 * dx - U00 * (1.f - dx)) * g.trans_std * diff;
        gy *= b.roi_h;
        const size_t base = ((size_t)n * g.num_classes + b.class_id) * 2;
        atomicAdd(gtrans + ((base)*g.part + b.part_h) * g.part + b.part_w, gx);
        atomicAdd(gtrans + ((base + 1) * g.part + b.part_h) * g.part + b.part_w, gy);
      }
  }
}

static int ps_geom(PsGeom* g, const char* who, int B, int H, int W, int C, int R, int out_size, int out_dim,
                   int group_size, int part_size, int sample_per_part, int no_trans, int num_classes,
                   float scale, float trans_std) {
  DADET_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && R >= 0 && out_size > 0 && out_dim > 0 && group_size > 0 &&
                    part_size > 0 && sample_per_part > 0 && num_classes > 0,
                "%s: bad dims", who);
  DADET_REQUIRE(C == out_dim * group_size * group_size, "%s: C must equal output_dim * group_size^2", who);
  DADET_REQUIRE(out_dim % num_classes == 0, "%s: output_dim must be a multiple of num_classes", who);
  *g = PsGeom{B, H, W, C, R, out_size, out_size, out_dim, group_size, part_size, sample_per_part, no_trans,
              num_classes, out_dim / num_classes, scale, trans_std};
  return DADET_OK;
}

}  // namespace dadet

extern "C" int dadet_deform_psroi_pool_forward(const float* data, const float* rois, const float* trans,
                                               float* out, float* top_count, int B, int H, int W, int C, int R,
                                               int no_trans, float spatial_scale, int output_dim, int group_size,
                                               int pooled_size, int part_size, int sample_per_part,
                                               float trans_std, int num_classes, void* stream) {
  dadet::PsGeom g;
  int rc = dadet::ps_geom(&g, "deform_psroi_pool_forward", B, H, W, C, R, pooled_size, output_dim, group_size,
                          part_size, sample_per_part, no_trans, no_trans ? 1 : num_classes, spatial_scale, trans_std);
  if (rc) return rc;
  if (R == 0) return DADET_OK;
  DADET_REQUIRE(data && rois && out && top_count && (no_trans || trans), "deform_psroi_pool_forward: null pointer");
  const int64_t total = (int64_t)R * pooled_size * pooled_size * output_dim;
  int64_t blocks = dadet::ceil_div64(total, 256);
  if (blocks > dadet::kMaxStreamBlocks) blocks = dadet::kMaxStreamBlocks;
  hipLaunchKernelGGL(dadet::deform_psroi_fwd_kernel, dim3((int)blocks), dim3(256), 0, dadet::as_stream(stream), data,
                     rois, trans, out, top_count, g);
  return dadet::check_launch("deform_psroi_pool_forward");
}

extern "C" int dadet_deform_psroi_pool_backward(const float* grad_out, const float* top_count, const float* data,
                                                const float* rois, const float* trans, float* grad_data,
                                                float* grad_trans, int B, int H, int W, int C, int R, int no_trans,
                                                float spatial_scale, int output_dim, int group_size,
                                                int pooled_size, int part_size, int sample_per_part,
                                                float trans_std, int num_classes, void* stream) {
  dadet::PsGeom g;
  int rc = dadet::ps_geom(&g, "deform_psroi_pool_backward", B, H, W, C, R, pooled_size, output_dim, group_size,
                          part_size, sample_per_part, no_trans, no_trans ? 1 : num_classes, spatial_scale, trans_std);
  if (rc) return rc;
  if (R == 0) return DADET_OK;
  DADET_REQUIRE(grad_out && top_count && data && rois && grad_data && (no_trans || (trans && grad_trans)),
                "deform_psroi_pool_backward: null pointer");
  const int64_t total = (int64_t)R * pooled_size * pooled_size * output_dim;
  int64_t blocks = dadet::ceil_div64(total, 256);
  if (blocks > dadet::kMaxStreamBlocks) blocks = dadet::kMaxStreamBlocks;
  hipLaunchKernelGGL(dadet::deform_psroi_bwd_kernel, dim3((int)blocks), dim3(256), 0, dadet::as_stream(stream),
                     grad_out, top_count, data, rois, trans, grad_data, grad_trans, g);
  return dadet::check_launch("deform_psroi_pool_backward");
}
