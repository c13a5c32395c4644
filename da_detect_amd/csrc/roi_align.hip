// ROIAlign forward / backward for NHWC fp32 feature maps on gfx950.
//
// Replaces _C.roi_align_forward / _C.roi_align_backward of the reference
// (maskrcnn_benchmark/csrc/ROIAlign.h:11-45; kernels csrc/cpu/ROIAlign_cpu.cpp:18-219 and
// csrc/cuda/ROIAlign_cuda.cu:16-254).  The reference maps one thread to one (n,c,ph,pw) output element
// of an NCHW tensor, so the four bilinear taps of a wavefront are scattered 4-byte reads.  Here the map
// is NHWC: one workgroup owns one (roi, ph) row of bins, its lanes run along the CHANNEL axis with
// 16-byte accesses, so every bilinear tap is one fully coalesced C*4-byte row and the per-sample
// coordinate/weight math is computed once per sample instead of once per output element.
//
// Numerics: the sample coordinates, weights and the accumulation are evaluated in exactly the
// reference's operation order (ROIAlign_cpu.cpp:34-104,198-209) and this translation unit is built with
// -ffp-contract=off, so the forward is bit-comparable with the reference CPU kernel.
#include "common.h"

namespace dadet {

struct RoiGeom {
  int batch;
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w;
  float count;
};

// reference: ROIAlign_cpu.cpp:140-172 / ROIAlign_cuda.cu:78-104
__device__ inline RoiGeom roi_geometry(const float* __restrict__ roi, float scale, int pooled_h,
                                       int pooled_w, int sampling_ratio) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.start_w = roi[1] * scale;
  g.start_h = roi[2] * scale;
  const float end_w = roi[3] * scale;
  const float end_h = roi[4] * scale;
  const float roi_w = fmaxf(end_w - g.start_w, 1.f);
  const float roi_h = fmaxf(end_h - g.start_h, 1.f);
  g.bin_h = roi_h / (float)pooled_h;
  g.bin_w = roi_w / (float)pooled_w;
  g.grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / (float)pooled_h);
  g.grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / (float)pooled_w);
  g.count = (float)(g.grid_h * g.grid_w);
  return g;
}

struct Tap {
  int p1, p2, p3, p4;  // pixel indices y*W+x of the four neighbours, -1 when the sample is skipped
  float w1, w2, w3, w4;
};

// reference: ROIAlign_cpu.cpp:46-104 (pre_calc_for_bilinear_interpolate) / ROIAlign_cuda.cu:16-62
__device__ inline Tap bilinear_tap(float y, float x, int H, int W) {
  Tap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.p1 = t.p2 = t.p3 = t.p4 = -1;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  const float ly = y - (float)y_low, lx = x - (float)x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  t.w1 = hy * hx;
  t.w2 = hy * lx;
  t.w3 = ly * hx;
  t.w4 = ly * lx;
  t.p1 = y_low * W + x_low;
  t.p2 = y_low * W + x_high;
  t.p3 = y_high * W + x_low;
  t.p4 = y_high * W + x_high;
  return t;
}

__device__ inline float sample_coord(float start, int p, float bin, int i, int grid) {
  // roi_start + p*bin + (i + .5f) * bin / grid        (ROIAlign_cpu.cpp:34-41)
  return start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;
}

// one workgroup per (roi, ph); lanes stride over channel quads; loop over pw.
template <int VEC>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(
    const float* __restrict__ input, const float* __restrict__ rois, float* __restrict__ output, int C,
    int H, int W, int pooled_h, int pooled_w, float scale, int sampling_ratio, const int* __restrict__ order,
    int bin_stride, const int64_t* __restrict__ levels, int level) {
  // XCD-aware order: the 14 bin rows of one ROI read overlapping feature rows; hardware deals consecutive workgroup
  // ids to the 8 XCDs round-robin, which made every XCD's L2 fetch the same rows again (PMC: 1.95 GB through the
  // fabric for 0.48 GB of algorithmic traffic).  The remap gives each XCD a contiguous range of (roi, ph).
  // `order` (roi_order_kernel): the ROIs in Z-order of their centres, image by image.  Proposals arrive in score order,
  // i.e. spatially random: every XCD then pulls the whole feature map through its 4 MB L2 (PMC r01: 1.33 GB fetched +
  // written per launch for 0.48 GB algorithmic).  With spatially sorted ROIs the contiguous range of an XCD covers a
  // compact region and neighbouring ROIs share the rows already in L2.  Outputs stay at their original row r.
  // bin_stride s > 1: only the bins (ph, pw) with ph % s == 0 and pw % s == 0 of the pooled_h x pooled_w grid are
  // evaluated, into a compact [R][ceil(pooled_h / s)][ceil(pooled_w / s)][C] output (dadet_roi_align_forward_sub)
  const int out_h = (pooled_h + bin_stride - 1) / bin_stride, out_w = (pooled_w + bin_stride - 1) / bin_stride;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int r = order ? order[wg / out_h] : wg / out_h;
  const int oph = wg % out_h;
  const int ph = oph * bin_stride;
  // feature pyramids (dadet_roi_align_forward_level): one launch per level over ALL ROIs, every ROI pooled by the launch
  // of its own level into its own output row — no per-level index lists, no host round trip to size them
  if (levels && levels[r] != (int64_t)level) return;
  const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, pooled_h, pooled_w, sampling_ratio);
  const float* __restrict__ img = input + (size_t)g.batch * H * W * C;
  float* __restrict__ out_row = output + ((size_t)r * out_h + oph) * out_w * C;

  for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
    for (int opw = 0; opw < out_w; ++opw) {
      const int pw = opw * bin_stride;
      float acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
          const Tap t = bilinear_tap(y, x, H, W);
          if (t.p1 < 0) continue;  // contributes exactly +0 in the reference
          float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
          if constexpr (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(img + (size_t)t.p1 * C + c);
            const float4 b = *reinterpret_cast<const float4*>(img + (size_t)t.p2 * C + c);
            const float4 d = *reinterpret_cast<const float4*>(img + (size_t)t.p3 * C + c);
            const float4 e = *reinterpret_cast<const float4*>(img + (size_t)t.p4 * C + c);
            v1[0] = a.x; v1[1] = a.y; v1[2] = a.z; v1[3] = a.w;
            v2[0] = b.x; v2[1] = b.y; v2[2] = b.z; v2[3] = b.w;
            v3[0] = d.x; v3[1] = d.y; v3[2] = d.z; v3[3] = d.w;
            v4[0] = e.x; v4[1] = e.y; v4[2] = e.z; v4[3] = e.w;
          } else {
            v1[0] = img[(size_t)t.p1 * C + c];
            v2[0] = img[(size_t)t.p2 * C + c];
            v3[0] = img[(size_t)t.p3 * C + c];
            v4[0] = img[(size_t)t.p4 * C + c];
          }
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            // output_val += w1*d1 + w2*d2 + w3*d3 + w4*d4   (ROIAlign_cpu.cpp:201-204)
            const float s = ((t.w1 * v1[v] + t.w2 * v2[v]) + t.w3 * v3[v]) + t.w4 * v4[v];
            acc[v] = acc[v] + s;
          }
        }
      }
      if constexpr (VEC == 4) {
        float4 o;
        o.x = acc[0] / g.count; o.y = acc[1] / g.count; o.z = acc[2] / g.count; o.w = acc[3] / g.count;
        *reinterpret_cast<float4*>(out_row + (size_t)opw * C + c) = o;
      } else {
        out_row[(size_t)opw * C + c] = acc[0] / g.count;
      }
    }
  }
}

// ROI processing order of the forward kernel: ascending (image, Morton code of the ROI centre in feature pixels), ties by
// ROI index.  One workgroup; bitonic sort of (key << 32 | r) in LDS.
__device__ inline unsigned spread_bits(unsigned v) {   // abcdefgh -> 0a0b0c0d0e0f0g0h (up to 16 bits)
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ __launch_bounds__(1024) void roi_order_kernel(const float* __restrict__ rois, int R, int N, float scale, int H,
                                                         int W, int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);   // [N], N = pow2 >= R
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < R) {
      const float* q = rois + (size_t)i * 5;
      const float cx = 0.5f * (q[1] + q[3]) * scale, cy = 0.5f * (q[2] + q[4]) * scale;
      const unsigned ux = (unsigned)fminf(fmaxf(cx, 0.f), (float)(W - 1));
      const unsigned uy = (unsigned)fminf(fmaxf(cy, 0.f), (float)(H - 1));
      const unsigned b = (unsigned)fmaxf(q[0], 0.f);
      k = ((unsigned long long)((b << 24) | (spread_bits(uy >> 1) << 1) | spread_bits(ux >> 1)) << 32) | (unsigned)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int kk = 2; kk <= N; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = keys[i], b = keys[p];
          if ((a > b) == ((i & kk) == 0)) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < R; i += blockDim.x) order[i] = (int)(unsigned)keys[i];
}

// conservative "ROI r can contribute to the 2x2 pixel tile at (y0, x0)" test of the gather backward
__device__ inline bool roi_touches_tile(const RoiGeom& g, int pooled_h, int pooled_w, int y0, int x0) {
  const float roi_h = g.bin_h * (float)pooled_h, roi_w = g.bin_w * (float)pooled_w;
  return g.start_h <= (float)y0 + 2.f && g.start_h + roi_h >= (float)y0 - 1.f && g.start_w <= (float)x0 + 2.f &&
         g.start_w + roi_w >= (float)x0 - 1.f;
}

// backward: scatter g * w / count to the four neighbours with hardware fp32 atomics
// (reference: ROIAlign_cuda.cu:178-254; g1 = top_diff * w1 / count at :236-239).
template <int VEC>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float* __restrict__ grad_in,
    int C, int H, int W, int pooled_h, int pooled_w, float scale, int sampling_ratio) {
  const int r = blockIdx.x / pooled_h;
  const int ph = blockIdx.x % pooled_h;
  const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, pooled_h, pooled_w, sampling_ratio);
  float* __restrict__ img = grad_in + (size_t)g.batch * H * W * C;
  const float* __restrict__ go_row = grad_out + ((size_t)r * pooled_h + ph) * pooled_w * C;

  for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
    for (int pw = 0; pw < pooled_w; ++pw) {
      float go[VEC];
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(go_row + (size_t)pw * C + c);
        go[0] = q.x; go[1] = q.y; go[2] = q.z; go[3] = q.w;
      } else {
        go[0] = go_row[(size_t)pw * C + c];
      }
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
          const Tap t = bilinear_tap(y, x, H, W);
          if (t.p1 < 0) continue;
          float* q1 = img + (size_t)t.p1 * C + c;
          float* q2 = img + (size_t)t.p2 * C + c;
          float* q3 = img + (size_t)t.p3 * C + c;
          float* q4 = img + (size_t)t.p4 * C + c;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            unsafeAtomicAdd(q1 + v, go[v] * t.w1 / g.count);
            unsafeAtomicAdd(q2 + v, go[v] * t.w2 / g.count);
            unsafeAtomicAdd(q3 + v, go[v] * t.w3 / g.count);
            unsafeAtomicAdd(q4 + v, go[v] * t.w4 / g.count);
          }
        }
      }
    }
  }
}


// ---- backward, gather form (no atomics, deterministic) -------------------------------------------------
// One workgroup per input-gradient pixel (b, y, x), lanes along channels.  The pixel receives
// g[r,ph,pw,c] * wy * wx / count from every sample of every ROI of image b whose bilinear footprint touches
// it; ROIAlign weights are separable (w1..w4 = {hy,ly} x {hx,lx}), so per ROI the kernel walks the few
// sample rows / columns that can reach (y, x) and accumulates in a fixed order.  Cost is reads of the
// (L2-resident) output gradient rows instead of ~R*PH*PW*samples*4*C global atomics.
__device__ inline float axis_weight(float coord, int limit, int pixel) {
  // 1-D version of bilinear_tap(): weight with which a sample at `coord` lands on `pixel` (0 = not at all)
  if (coord < -1.0f || coord > (float)limit) return 0.f;
  if (coord <= 0.f) coord = 0.f;
  int lo = (int)coord, hi;
  if (lo >= limit - 1) {
    hi = lo = limit - 1;
    coord = (float)lo;
  } else {
    hi = lo + 1;
  }
  const float l = coord - (float)lo, h = 1.f - l;
  float w = 0.f;
  if (lo == pixel) w += h;
  if (hi == pixel) w += l;
  return w;
}

__device__ inline void candidate_range(float start, float bin, int grid, int pooled, int pixel, int* lo,
                                       int* hi) {
  // sample s (0 <= s < pooled*grid) sits near start + (s + .5) * bin / grid; keep those within (pixel-1, pixel+1)
  // with one sample of slack either side — axis_weight() decides exactly.
  const float step = bin / (float)grid;
  int a = (int)floorf(((float)pixel - 1.f - start) / step - 0.5f) - 1;
  int b = (int)ceilf(((float)pixel + 1.f - start) / step - 0.5f) + 1;
  *lo = a < 0 ? 0 : a;
  *hi = b > pooled * grid - 1 ? pooled * grid - 1 : b;
}

template <int VEC>
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float* __restrict__ grad_in, int C,
    int H, int W, int R, int pooled_h, int pooled_w, float scale, int sampling_ratio) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned char* touches = reinterpret_cast<unsigned char*>(smem);  // [R]
  const int pix = blockIdx.x;
  const int b = pix / (H * W);
  const int y = (pix / W) % H;
  const int x = pix % W;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, pooled_h, pooled_w, sampling_ratio);
    const float roi_h = g.bin_h * (float)pooled_h, roi_w = g.bin_w * (float)pooled_w;
    const bool hit = g.batch == b && g.start_h <= (float)y + 1.f && g.start_h + roi_h >= (float)y - 1.f &&
                     g.start_w <= (float)x + 1.f && g.start_w + roi_w >= (float)x - 1.f;
    touches[r] = hit ? 1 : 0;
  }
  __syncthreads();
  for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int r = 0; r < R; ++r) {
      if (!touches[r]) continue;  // uniform
      const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, pooled_h, pooled_w, sampling_ratio);
      int sy0, sy1, sx0, sx1;
      candidate_range(g.start_h, g.bin_h, g.grid_h, pooled_h, y, &sy0, &sy1);
      candidate_range(g.start_w, g.bin_w, g.grid_w, pooled_w, x, &sx0, &sx1);
      const float* __restrict__ go_roi = grad_out + (size_t)r * pooled_h * pooled_w * C + c;
      for (int sy = sy0; sy <= sy1; ++sy) {
        const int ph = sy / g.grid_h, iy = sy - ph * g.grid_h;
        const float wy = axis_weight(sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h), H, y);
        if (wy == 0.f) continue;
        for (int sx = sx0; sx <= sx1; ++sx) {
          const int pw = sx / g.grid_w, ix = sx - pw * g.grid_w;
          const float wx = axis_weight(sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w), W, x);
          if (wx == 0.f) continue;
          const float w = wy * wx;
          const float* q = go_roi + (size_t)(ph * pooled_w + pw) * C;
          if constexpr (VEC == 4) {
            const float4 gq = *reinterpret_cast<const float4*>(q);
            acc[0] += gq.x * w / g.count; acc[1] += gq.y * w / g.count;
            acc[2] += gq.z * w / g.count; acc[3] += gq.w * w / g.count;
          } else {
            acc[0] += q[0] * w / g.count;
          }
        }
      }
    }
    float* dst = grad_in + (size_t)pix * C + c;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      dst[0] = acc[0];
    }
  }
}

// ---- backward, gather form with contribution lists over 2x2 pixel tiles ----------------------------------------
// Same mathematics as roi_align_bwd_gather_kernel (every pixel sums its contributions in the order ROI ascending,
// sample row, sample column).  Two things change the cost:
//  * a bilinear sample lands on a 2x2 block of pixels, so a one-pixel workgroup re-reads every output-gradient row
//    up to four times (PMC: 1.5 GB fetched for 411 MB of gradient).  Here a workgroup owns a 2x2 pixel tile and a
//    row is loaded once for all the tile pixels it reaches;
//  * the wave-uniform coordinate / weight arithmetic is not executed by all 256 lanes for every ROI: per range of
//    256 ROIs the lanes test one ROI each, the touching ones are compacted in order, ONE LANE PER ROI lists that
//    ROI's non-zero contributions into LDS, and all lanes then stream over the list.
struct Contribution {
  int row;        // (r * pooled_h + ph) * pooled_w + pw
  float count;    // samples per bin of that ROI
  float w[4];     // wy * wx for the tile pixels (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)
  float inv;      // 1 / count
  int pow2;       // count is a power of two: (g * w) * inv == (g * w) / count bit for bit, and 16 IEEE divisions per
                  // entry and lane (~10 VALU instructions each) become multiplications — the kernel was VALU bound on them
};
constexpr int kListCap = 1024;   // entries per round (32 KB of LDS)

// bs > 1 (dadet_roi_align_backward_sub): only bins with ph % bs == 0 and pw % bs == 0 carry a gradient; their rows live
// in a compact [R][ceil(pooled_h / bs)][ceil(pooled_w / bs)] grid
__device__ inline int roi_contributions(const RoiGeom& g, int r, int H, int W, int pooled_h, int pooled_w, int y0,
                                        int x0, Contribution* out, int bs = 1) {
  const int out_h = (pooled_h + bs - 1) / bs, out_w = (pooled_w + bs - 1) / bs;
  int sy0, sy1, sx0, sx1, lo, hi, n = 0;
  candidate_range(g.start_h, g.bin_h, g.grid_h, pooled_h, y0, &sy0, &hi);
  candidate_range(g.start_h, g.bin_h, g.grid_h, pooled_h, y0 + 1, &lo, &sy1);
  candidate_range(g.start_w, g.bin_w, g.grid_w, pooled_w, x0, &sx0, &hi);
  candidate_range(g.start_w, g.bin_w, g.grid_w, pooled_w, x0 + 1, &lo, &sx1);
  for (int sy = sy0; sy <= sy1; ++sy) {
    const int ph = sy / g.grid_h, iy = sy - ph * g.grid_h;
    if (bs > 1 && ph % bs) continue;
    const float cy = sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
    const float wy0 = axis_weight(cy, H, y0), wy1 = axis_weight(cy, H, y0 + 1);
    if (wy0 == 0.f && wy1 == 0.f) continue;
    for (int sx = sx0; sx <= sx1; ++sx) {
      const int pw = sx / g.grid_w, ix = sx - pw * g.grid_w;
      if (bs > 1 && pw % bs) continue;
      const float cx = sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
      const float wx0 = axis_weight(cx, W, x0), wx1 = axis_weight(cx, W, x0 + 1);
      if (wx0 == 0.f && wx1 == 0.f) continue;
      if (out) {
        Contribution c;
        c.row = (r * out_h + ph / bs) * out_w + pw / bs;
        c.count = g.count;
        c.w[0] = wy0 * wx0; c.w[1] = wy0 * wx1; c.w[2] = wy1 * wx0; c.w[3] = wy1 * wx1;
        c.inv = 1.f / g.count;
        const int ci = (int)g.count;
        c.pow2 = (ci & (ci - 1)) == 0;
        out[n] = c;
      }
      ++n;
    }
  }
  return n;
}

// (MEASURED, round 3, R-101-FPN-DCN — three launches per step on maps of up to 2 x 256 x 512 pixels, 0.64 ms together: a
// cover map of the 8 x 8-pixel cells any ROI can reach, so that tiles of untouched cells zero-fill without walking the
// ROI ranges, changed nothing, 60.2 / 60.4 vs 60.3 / 60.7 ms per step: the empty tiles are not where the time goes.)
template <int VEC, int MAXC>   // MAXC channel groups per lane: C <= 256 * VEC * MAXC
__global__ __launch_bounds__(256) void roi_align_bwd_list_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float* __restrict__ grad_in, int B, int C,
    int H, int W, int R, int pooled_h, int pooled_w, float scale, int sampling_ratio, int bin_stride,
    const int64_t* __restrict__ levels, int level) {
  __shared__ Contribution s_list[kListCap];
  __shared__ int s_ids[256];        // touching ROIs of the current range, ascending
  __shared__ int s_wave_n[4];
  __shared__ int s_total;
  const int tiles_x = (W + 1) / 2, tiles_y = (H + 1) / 2;
  // Neighbouring tiles read the same gradient rows (a bin's four samples reach over up to 3 x 3 pixels).  Hardware
  // deals consecutive workgroup ids to the 8 XCDs round-robin: the four tiles of a 2 x 2 super-tile are given to ONE
  // XCD (one L2 fetch of the shared rows), super-tiles go round-robin over the XCDs so that regions crowded with ROIs
  // spread evenly (a contiguous band per XCD measured slower: the ROIs are not spread evenly over the image).
  const int sx = (tiles_x + 1) / 2, sy = (tiles_y + 1) / 2;
  const int xcd = blockIdx.x % kNumXCD, idx = blockIdx.x / kNumXCD;
  const int super = (idx >> 2) * kNumXCD + xcd, q = idx & 3;
  const int b = super / (sx * sy);
  const int ty = ((super / sx) % sy) * 2 + (q >> 1), tx = (super % sx) * 2 + (q & 1);
  if (b >= B || ty >= tiles_y || tx >= tiles_x) return;
  const int y0 = ty * 2;
  const int x0 = tx * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[MAXC][4][VEC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[k][p][v] = 0.f;

  auto accumulate = [&](int n_entries) {
    constexpr int U = MAXC == 1 ? 8 : 4;   // entries in flight per lane (one dependent 16-byte load at a time otherwise); box head, 256 ROIs: 4 -> 182 us, 8 -> 157, 16 -> 164
#pragma unroll 1
    for (int e0 = 0; e0 < n_entries; e0 += U) {
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int c = (threadIdx.x + k * 256) * VEC;
        if (c >= C) break;
        float gq[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = min(e0 + u, n_entries - 1);     // the tail re-reads the last row; its weights are skipped
          const float* src = grad_out + (size_t)s_list[e].row * C + c;
          if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            gq[u][0] = t.x; gq[u][1] = t.y; gq[u][2] = t.z; gq[u][3] = t.w;
          } else {
            gq[u][0] = src[0];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (e0 + u >= n_entries) break;
          const Contribution q = s_list[e0 + u];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if (q.w[p] == 0.f) continue;   // wave-uniform: keeps every pixel's sum free of +0 terms
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const float t = gq[u][v] * q.w[p];
              acc[k][p][v] += q.pow2 ? t * q.inv : t / q.count;   // wave-uniform choice
            }
          }
        }
      }
    }
  };

  for (int base = 0; base < R; base += 256) {
    // 1. which ROIs of this range touch the tile; ordered compaction (wave ballots, wave order = ROI order)
    const int r = base + (int)threadIdx.x;
    bool hit = false;
    if (r < R) {
      const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, pooled_h, pooled_w, sampling_ratio);
      hit = g.batch == b && (!levels || levels[r] == (int64_t)level) && roi_touches_tile(g, pooled_h, pooled_w, y0, x0);
    }
    const unsigned long long ballot = __ballot(hit);
    if (lane == 0) s_wave_n[wave] = __popcll(ballot);
    __syncthreads();
    int before = 0;
    for (int wv = 0; wv < wave; ++wv) before += s_wave_n[wv];
    const int ntouch = s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
    if (hit) s_ids[before + __popcll(ballot & ((1ULL << lane) - 1ULL))] = r;
    __syncthreads();
    // 2. rounds of up to 64 touching ROIs: one lane per ROI lists its contributions, then everybody accumulates
    for (int start = 0; start < ntouch; start += 64) {
      const int nround = min(64, ntouch - start);
      if (wave == 0) {
        int my_n = 0, rr = -1;
        RoiGeom g;
        if (lane < nround) {
          rr = s_ids[start + lane];
          g = roi_geometry(rois + (size_t)rr * 5, scale, pooled_h, pooled_w, sampling_ratio);
          my_n = roi_contributions(g, rr, H, W, pooled_h, pooled_w, y0, x0, nullptr, bin_stride);
        }
        int incl = my_n;   // inclusive wave scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(incl, off, 64);
          if (lane >= off) incl += t;
        }
        const int total = __shfl(incl, 63, 64);
        if (lane == 0) s_total = total;
        if (total <= kListCap && rr >= 0)
          roi_contributions(g, rr, H, W, pooled_h, pooled_w, y0, x0, s_list + (incl - my_n), bin_stride);
      }
      __syncthreads();
      const int total = s_total;
      if (total <= kListCap) {
        accumulate(total);
      } else {
        // pathological pile-up of tiny ROIs on one tile: list them one ROI at a time (a 14x14-bin ROI offers at most
        // 32 x 32 candidate samples to a tile, <= kListCap)
        for (int i = 0; i < nround; ++i) {
          __syncthreads();
          if (threadIdx.x == 0) {
            const int r1 = s_ids[start + i];
            const RoiGeom g1 = roi_geometry(rois + (size_t)r1 * 5, scale, pooled_h, pooled_w, sampling_ratio);
            const int n1 = roi_contributions(g1, r1, H, W, pooled_h, pooled_w, y0, x0, nullptr, bin_stride);
            if (n1 <= kListCap) roi_contributions(g1, r1, H, W, pooled_h, pooled_w, y0, x0, s_list, bin_stride);
            s_total = n1 <= kListCap ? n1 : -1;
          }
          __syncthreads();
          if (s_total < 0) __builtin_trap();   // cannot happen for pooled sizes <= 14 x 14 with sampling_ratio <= 2
          accumulate(s_total);
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int y = y0 + (p >> 1), x = x0 + (p & 1);
    if (y >= H || x >= W) continue;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = (threadIdx.x + k * 256) * VEC;
      if (c >= C) break;
      float* dst = grad_in + ((size_t)(b * H + y) * W + x) * C + c;
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[k][p][0], acc[k][p][1], acc[k][p][2], acc[k][p][3]);
      } else {
        dst[0] = acc[k][p][0];
      }
    }
  }
}

}  // namespace dadet

using namespace dadet;

static int roi_args_ok(const void* a, const void* b, const void* c, int B, int C, int H, int W, int R,
                       int ph, int pw) {
  DADET_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && ph > 0 && pw > 0,
                "roi_align: bad dims B=%d C=%d H=%d W=%d R=%d ph=%d pw=%d", B, C, H, W, R, ph, pw);
  if (R > 0) DADET_REQUIRE(a && b && c, "roi_align: null pointer");
  return DADET_OK;
}

extern "C" int dadet_roi_align_workspace_bytes(int B, int H, int W, int R, size_t* bytes) {
  DADET_REQUIRE(bytes && B > 0 && H > 0 && W > 0 && R >= 0, "roi_align_workspace_bytes: bad arguments");
  *bytes = sizeof(int) * (size_t)R + 16;
  return DADET_OK;
}

static int roi_align_forward_impl(const float* input, const float* rois, float* output, int B, int C, int H, int W, int R,
                                  int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, void* workspace,
                                  size_t workspace_bytes, void* stream, int bin_stride = 1,
                                  const int64_t* levels = nullptr, int level = 0) {
  int rc = roi_args_ok(input, rois, output, B, C, H, W, R, pooled_h, pooled_w);
  if (rc) return rc;
  DADET_REQUIRE(bin_stride >= 1 && bin_stride <= pooled_h && bin_stride <= pooled_w, "roi_align_forward: bin_stride=%d",
                bin_stride);
  if (R == 0) return DADET_OK;
  int* order = nullptr;
  if (workspace && R > 64 && R <= 4096) {     // spatial processing order (see roi_align_fwd_kernel)
    DADET_REQUIRE(workspace_bytes >= sizeof(int) * (size_t)R, "roi_align_forward: workspace of %zu bytes is too small",
                  workspace_bytes);
    order = static_cast<int*>(workspace);
    int N = 2;
    while (N < R) N <<= 1;
    hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(1024), sizeof(unsigned long long) * (size_t)N, as_stream(stream),
                       rois, R, N, spatial_scale, H, W, order);
  }
  const dim3 grid((unsigned)(R * ((pooled_h + bin_stride - 1) / bin_stride)));
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(input) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(output) & 15) == 0);
  if (vec) {
    const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_fwd_kernel<4>, grid, dim3(threads), 0, as_stream(stream), input, rois,
                       output, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, order, bin_stride, levels, level);
  } else {
    const int threads = (C >= 256) ? 256 : ((C + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_fwd_kernel<1>, grid, dim3(threads), 0, as_stream(stream), input, rois,
                       output, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, order, bin_stride, levels, level);
  }
  return check_launch("roi_align_forward");
}

extern "C" int dadet_roi_align_forward_sub(const float* input, const float* rois, float* output, int B, int C, int H,
                                           int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                           int sampling_ratio, int bin_stride, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  return roi_align_forward_impl(input, rois, output, B, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                workspace, workspace_bytes, stream, bin_stride);
}

extern "C" int dadet_roi_align_forward(const float* input, const float* rois, float* output, int B, int C, int H, int W,
                                       int R, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                       void* stream) {
  return roi_align_forward_impl(input, rois, output, B, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                nullptr, 0, stream);
}

extern "C" int dadet_roi_align_forward_ws(const float* input, const float* rois, float* output, int B, int C, int H,
                                          int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                          int sampling_ratio, void* workspace, size_t workspace_bytes, void* stream) {
  return roi_align_forward_impl(input, rois, output, B, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                workspace, workspace_bytes, stream);
}

extern "C" int dadet_roi_align_backward_atomic(const float* grad_output, const float* rois, float* grad_input,
                                        int B, int C, int H, int W, int R, int pooled_h, int pooled_w,
                                        float spatial_scale, int sampling_ratio, void* stream) {
  int rc = roi_args_ok(grad_output, rois, grad_input, B, C, H, W, R, pooled_h, pooled_w);
  if (rc) return rc;
  if (R == 0) return DADET_OK;
  const dim3 grid((unsigned)(R * pooled_h));
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_output) & 15) == 0);
  if (vec) {
    const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_bwd_kernel<4>, grid, dim3(threads), 0, as_stream(stream), grad_output,
                       rois, grad_input, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  } else {
    const int threads = (C >= 256) ? 256 : ((C + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_bwd_kernel<1>, grid, dim3(threads), 0, as_stream(stream), grad_output,
                       rois, grad_input, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  }
  return check_launch("roi_align_backward_atomic");
}

static int roi_align_backward_impl(const float* grad_output, const float* rois, float* grad_input, int B, int C, int H,
                                   int W, int R, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                   int bin_stride, void* stream, const int64_t* levels = nullptr, int level = 0) {
  int rc = roi_args_ok(grad_output, rois, grad_input, B, C, H, W, R, pooled_h, pooled_w);
  if (rc) return rc;
  DADET_REQUIRE(bin_stride >= 1 && bin_stride <= pooled_h && bin_stride <= pooled_w, "roi_align_backward: bin_stride=%d",
                bin_stride);
  hipStream_t st = as_stream(stream);
  if (R == 0) {
    (void)hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * C * H * W, st);
    return check_launch("roi_align_backward(empty)");
  }
  DADET_REQUIRE(R <= 60000, "roi_align_backward: R=%d exceeds the per-workgroup ROI table", R);
  const dim3 grid((unsigned)(B * H * W));
  const size_t lds = ((size_t)R + 15) & ~(size_t)15;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_output) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(grad_input) & 15) == 0);
  static const bool use_list = !(getenv("DADET_ROI_BWD_LIST") && getenv("DADET_ROI_BWD_LIST")[0] == '0');
  if (vec && use_list && C <= 256 * 4 * 4 && pooled_h <= 14 && pooled_w <= 14) {
    // 2 x 2 super-tiles of 2 x 2-pixel tiles, padded to whole rounds of the 8 XCDs; grid.y only carries B
    const int supers = B * (((W + 1) / 2 + 1) / 2) * (((H + 1) / 2 + 1) / 2);
    const dim3 tgrid((unsigned)(((supers + kNumXCD - 1) / kNumXCD) * kNumXCD * 4), 1, 1);
    if (C <= 1024)
      hipLaunchKernelGGL((roi_align_bwd_list_kernel<4, 1>), tgrid, dim3(256), 0, st, grad_output, rois, grad_input, B, C,
                         H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio, bin_stride, levels, level);
    else
      hipLaunchKernelGGL((roi_align_bwd_list_kernel<4, 4>), tgrid, dim3(256), 0, st, grad_output, rois, grad_input, B, C,
                         H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio, bin_stride, levels, level);
  } else if (levels) {
    set_error("roi_align_backward_level: needs C %% 4 == 0 (C=%d), 16-byte aligned buffers and a pooled grid of at most "
              "14 x 14 (%d x %d)", C, pooled_h, pooled_w);
    return DADET_EUNSUPPORTED;
  } else if (bin_stride != 1) {
    set_error("roi_align_backward_sub: needs C %% 4 == 0 (C=%d), 16-byte aligned buffers and a pooled grid of at most "
              "14 x 14 (%d x %d)", C, pooled_h, pooled_w);
    return DADET_EUNSUPPORTED;
  } else if (vec) {
    const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel<4>, grid, dim3(threads), lds, st, grad_output, rois,
                       grad_input, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  } else {
    const int threads = (C >= 256) ? 256 : ((C + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel<1>, grid, dim3(threads), lds, st, grad_output, rois,
                       grad_input, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  }
  return check_launch("roi_align_backward");
}

extern "C" int dadet_roi_align_backward(const float* grad_output, const float* rois, float* grad_input, int B, int C,
                                        int H, int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                        int sampling_ratio, void* stream) {
  return roi_align_backward_impl(grad_output, rois, grad_input, B, C, H, W, R, pooled_h, pooled_w, spatial_scale,
                                 sampling_ratio, 1, stream);
}

extern "C" int dadet_roi_align_backward_sub(const float* grad_output, const float* rois, float* grad_input, int B, int C,
                                            int H, int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                            int sampling_ratio, int bin_stride, void* stream) {
  return roi_align_backward_impl(grad_output, rois, grad_input, B, C, H, W, R, pooled_h, pooled_w, spatial_scale,
                                 sampling_ratio, bin_stride, stream);
}

// Feature pyramids (reference: modeling/poolers.py:91-121 splits the ROIs by level with nonzero and scatters the per-level
// results back with an index_put): `levels[r]` (int64, the LevelMapper's result on the device) names the level of ROI r; the
// launch for `level` pools exactly those ROIs into THEIR rows of the shared [R][ph][pw][C] output and leaves the others
// alone.  One call per level with the same rois / levels / output; every ROI has exactly one level, so the output is
// complete after the last call.
extern "C" int dadet_roi_align_forward_level(const float* input, const float* rois, const int64_t* levels, int level,
                                             float* output, int B, int C, int H, int W, int R, int pooled_h, int pooled_w,
                                             float spatial_scale, int sampling_ratio, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  DADET_REQUIRE(levels || R == 0, "roi_align_forward_level: null levels");
  return roi_align_forward_impl(input, rois, output, B, C, H, W, R, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                workspace, workspace_bytes, stream, 1, levels, level);
}

// gradient of the level's map: the gather sweeps its pixel tiles and takes the ROIs of this level only
extern "C" int dadet_roi_align_backward_level(const float* grad_output, const float* rois, const int64_t* levels, int level,
                                              float* grad_input, int B, int C, int H, int W, int R, int pooled_h,
                                              int pooled_w, float spatial_scale, int sampling_ratio, void* stream) {
  DADET_REQUIRE(levels || R == 0, "roi_align_backward_level: null levels");
  return roi_align_backward_impl(grad_output, rois, grad_input, B, C, H, W, R, pooled_h, pooled_w, spatial_scale,
                                 sampling_ratio, 1, stream, levels, level);
}
