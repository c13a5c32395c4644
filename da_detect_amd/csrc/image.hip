// Input pipeline on the device: Pillow-exact bilinear resize of a decoded RGB image, horizontal flip, BGR-255
// conversion and mean / std normalisation, written straight into the padded batch tensor (gfx950).
//
// What it replaces (all host-side in the reference, per image and per worker process):
//   maskrcnn_benchmark/data/transforms/transforms.py:32-62  Resize -> torchvision F.resize -> PIL Image.resize(BILINEAR)
//   transforms.py:65-74  RandomHorizontalFlip (the coin is tossed by the caller)       :77-79  ToTensor (u8 / 255)
//   transforms.py:82-97  Normalize with to_bgr255 (channel swap, * 255, - PIXEL_MEAN, / PIXEL_STD)
//   structures/image_list.py:49-91  to_image_list zero padding to SIZE_DIVISIBILITY
// Pillow's resample is integer arithmetic (8-bit pixels, 22-bit fixed-point triangle-filter coefficients, horizontal
// pass rounded to 8 bits, then vertical pass): it is reproduced bit for bit — the coefficient tables are computed
// on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (double precision), the two passes
// below apply them.  HBM-bound byte work: lanes run along the output row, 3 channels per lane.
#include "common.h"

namespace dadet {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ inline unsigned char clip8(int acc) {
  const int v = acc >> kPrecisionBits;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][xx][c] = clip8(2^21 + sum_k in[y][xmin+k][c] * coeff[xx][k])
__global__ void resample_h_kernel(const unsigned char* __restrict__ in, int H, int W, const int* __restrict__ bounds,
                                  const int* __restrict__ coeffs, int ksize, int out_w,
                                  unsigned char* __restrict__ out) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xx >= out_w) return;
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = coeffs + (size_t)xx * ksize;
  const unsigned char* row = in + ((size_t)y * W + xmin) * 3;
  int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
  for (int i = 0; i < n; ++i) {
    const int c = k[i];
    a0 += (int)row[3 * i] * c;
    a1 += (int)row[3 * i + 1] * c;
    a2 += (int)row[3 * i + 2] * c;
  }
  unsigned char* o = out + ((size_t)y * out_w + xx) * 3;
  o[0] = clip8(a0);
  o[1] = clip8(a1);
  o[2] = clip8(a2);
}

// vertical pass + flip + (BGR, * 255) + normalise; out is fp32 [out_h rows][row_stride pixels][3] (NHWC of a
// 3-channel image whose padded width is row_stride); bounds == nullptr: no vertical resampling
__global__ void resample_v_normalize_kernel(const unsigned char* __restrict__ in, int in_h, int w,
                                            const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                            int ksize, int out_h, int flip, int to_bgr255, float m0, float m1,
                                            float m2, float s0, float s1, float s2, float* __restrict__ out,
                                            int row_stride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y;
  if (x >= w) return;
  unsigned char r, g, b;
  if (bounds) {
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = coeffs + (size_t)yy * ksize;
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < n; ++i) {
      const unsigned char* p = in + ((size_t)(ymin + i) * w + x) * 3;
      const int c = k[i];
      a0 += (int)p[0] * c;
      a1 += (int)p[1] * c;
      a2 += (int)p[2] * c;
    }
    r = clip8(a0);
    g = clip8(a1);
    b = clip8(a2);
  } else {
    const unsigned char* p = in + ((size_t)yy * w + x) * 3;
    r = p[0];
    g = p[1];
    b = p[2];
  }
  // ToTensor: u8 / 255; Normalize(to_bgr255): [2,1,0] * 255, then (v - mean) / std — the reference's fp32 op order
  float c0 = (float)r / 255.f, c1 = (float)g / 255.f, c2 = (float)b / 255.f;
  if (to_bgr255) {
    const float t = c0;
    c0 = c2 * 255.f;
    c1 = c1 * 255.f;
    c2 = t * 255.f;
  }
  const int xo = flip ? (w - 1 - x) : x;
  float* o = out + ((size_t)yy * row_stride + xo) * 3;
  o[0] = (c0 - m0) / s0;
  o[1] = (c1 - m1) / s1;
  o[2] = (c2 - m2) / s2;
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_image_resample_h(const unsigned char* image_hwc, int H, int W, const int* bounds, const int* coeffs,
                                      int ksize, int out_w, unsigned char* out_hwc, void* stream) {
  DADET_REQUIRE(H > 0 && W > 0 && out_w > 0 && ksize > 0, "image_resample_h: bad dims");
  DADET_REQUIRE(image_hwc && bounds && coeffs && out_hwc, "image_resample_h: null pointer");
  hipLaunchKernelGGL(resample_h_kernel, dim3(ceil_div(out_w, 256), H), dim3(256), 0, as_stream(stream), image_hwc, H, W,
                     bounds, coeffs, ksize, out_w, out_hwc);
  return check_launch("image_resample_h");
}

extern "C" int dadet_image_resample_v_normalize(const unsigned char* image_hwc, int in_h, int w, const int* bounds,
                                                const int* coeffs, int ksize, int out_h, int flip, int to_bgr255,
                                                const float* mean3, const float* std3, float* out_hw3,
                                                int out_row_stride, void* stream) {
  DADET_REQUIRE(in_h > 0 && w > 0 && out_h > 0 && out_row_stride >= w, "image_resample_v_normalize: bad dims");
  DADET_REQUIRE(image_hwc && mean3 && std3 && out_hw3, "image_resample_v_normalize: null pointer");
  DADET_REQUIRE(bounds || out_h == in_h, "image_resample_v_normalize: no coefficient table but out_h != in_h");
  DADET_REQUIRE(!bounds || (coeffs && ksize > 0), "image_resample_v_normalize: missing coefficients");
  hipLaunchKernelGGL(resample_v_normalize_kernel, dim3(ceil_div(w, 256), out_h), dim3(256), 0, as_stream(stream),
                     image_hwc, in_h, w, bounds, coeffs, ksize, out_h, flip ? 1 : 0, to_bgr255 ? 1 : 0, mean3[0],
                     mean3[1], mean3[2], std3[0], std3[1], std3[2], out_hw3, out_row_stride);
  return check_launch("image_resample_v_normalize");
}
