// ROIPool (max pooling over quantised ROI bins) forward / backward for NHWC fp32 on gfx950.
//
// Reference being restated: maskrcnn_benchmark/csrc/cuda/ROIPool_cuda.cu:16-75 (RoIPoolFForward: round(roi*scale),
// malformed ROIs forced to 1x1, floor/ceil bin edges clipped to the map, empty bin -> 0 with argmax -1, first
// maximum wins in (h, w) scan order) and :77-108 (RoIPoolFBackward: grad routed to the argmax position).
// Layout here: input [B][H][W][C], output / argmax [R][PH][PW][C]; argmax holds h*W + w (the reference's
// per-plane index), so it is layout independent.  One wavefront lane owns 4 consecutive channels of one bin: every
// visited position is one coalesced 16 B-per-lane row read.  Backward scatters with atomics like the reference
// (overlapping ROIs hit the same cell).
#include <cfloat>

#include "common.h"

namespace dadet {

struct PoolBin {
  int batch, hstart, hend, wstart, wend;
};

__device__ inline PoolBin pool_bin(const float* __restrict__ roi, float scale, int H, int W, int PH, int PW, int ph,
                                   int pw) {
  PoolBin b;
  b.batch = (int)roi[0];
  const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
  const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
  const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
  const float bin_h = (float)rh / (float)PH, bin_w = (float)rw / (float)PW;
  b.hstart = min(max((int)floorf((float)ph * bin_h) + sh, 0), H);
  b.hend = min(max((int)ceilf((float)(ph + 1) * bin_h) + sh, 0), H);
  b.wstart = min(max((int)floorf((float)pw * bin_w) + sw, 0), W);
  b.wend = min(max((int)ceilf((float)(pw + 1) * bin_w) + sw, 0), W);
  return b;
}

template <int VEC>
__global__ void roi_pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ rois,
                                    float* __restrict__ out, int* __restrict__ argmax, int C, int H, int W, int PH,
                                    int PW, float scale) {
  const int bin = blockIdx.x;  // (n, ph, pw)
  const int pw = bin % PW, ph = (bin / PW) % PH, n = bin / (PW * PH);
  const PoolBin b = pool_bin(rois + (size_t)n * 5, scale, H, W, PH, PW, ph, pw);
  const bool empty = b.hend <= b.hstart || b.wend <= b.wstart;
  const float* __restrict__ img = in + (size_t)b.batch * H * W * C;
  for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
    float best[VEC];
    int idx[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      best[v] = empty ? 0.f : -FLT_MAX;
      idx[v] = -1;
    }
    for (int h = b.hstart; h < b.hend; ++h)
      for (int w = b.wstart; w < b.wend; ++w) {
        float val[VEC];
        const float* p = img + ((size_t)h * W + w) * C + c;
        if (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          val[0] = q.x; val[1 % VEC] = q.y; val[2 % VEC] = q.z; val[3 % VEC] = q.w;
        } else {
          val[0] = p[0];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (val[v] > best[v]) {
            best[v] = val[v];
            idx[v] = h * W + w;
          }
      }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      out[(size_t)bin * C + c + v] = best[v];
      argmax[(size_t)bin * C + c + v] = idx[v];
    }
  }
}

__global__ void roi_pool_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ argmax,
                                    const float* __restrict__ rois, float* __restrict__ gin, int C, int H, int W,
                                    int PH, int PW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = argmax[i];
    if (a < 0) continue;
    const int c = (int)(i % C);
    const int n = (int)(i / ((int64_t)C * PW * PH));
    const int batch = (int)rois[(size_t)n * 5];
    unsafeAtomicAdd(gin + ((size_t)batch * H * W + a) * C + c, gout[i]);
  }
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_roi_pool_forward(const float* input, const float* rois, float* output, int* argmax, int B, int C,
                                      int H, int W, int R, int pooled_h, int pooled_w, float spatial_scale,
                                      void* stream) {
  DADET_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_pool_forward: bad dims");
  if (R == 0) return DADET_OK;
  DADET_REQUIRE(input && rois && output && argmax, "roi_pool_forward: null pointer");
  DADET_REQUIRE((int64_t)H * W < INT32_MAX, "roi_pool_forward: map too large for int32 argmax");
  const dim3 grid((unsigned)(R * pooled_h * pooled_w));
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(input) & 15) == 0);
  if (vec) {
    const int threads = (C / 4 >= 256) ? 256 : ((C / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_pool_fwd_kernel<4>, grid, dim3(threads), 0, as_stream(stream), input, rois, output, argmax,
                       C, H, W, pooled_h, pooled_w, spatial_scale);
  } else {
    const int threads = (C >= 256) ? 256 : ((C + 63) / 64) * 64;
    hipLaunchKernelGGL(roi_pool_fwd_kernel<1>, grid, dim3(threads), 0, as_stream(stream), input, rois, output, argmax,
                       C, H, W, pooled_h, pooled_w, spatial_scale);
  }
  return check_launch("roi_pool_forward");
}

extern "C" int dadet_roi_pool_backward(const float* grad_output, const int* argmax, const float* rois,
                                       float* grad_input, int B, int C, int H, int W, int R, int pooled_h,
                                       int pooled_w, void* stream) {
  DADET_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && pooled_h > 0 && pooled_w > 0,
                "roi_pool_backward: bad dims");
  DADET_REQUIRE(grad_input, "roi_pool_backward: null pointer");
  hipStream_t st = as_stream(stream);
  (void)hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * C * H * W, st);
  if (R == 0) return check_launch("roi_pool_backward(empty)");
  DADET_REQUIRE(grad_output && argmax && rois, "roi_pool_backward: null pointer");
  const int64_t total = (int64_t)R * pooled_h * pooled_w * C;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3((int)blocks), dim3(256), 0, st, grad_output, argmax, rois, grad_input,
                     C, H, W, pooled_h, pooled_w, total);
  return check_launch("roi_pool_backward");
}
