// EXPERIMENT (tools/planes_probe.py): the split-bf16 forward GEMM fed with operands that were split into bf16 term
// planes ONCE in HBM (by split_planes_kernel) instead of being re-split by every workgroup that stages them — an
// activation element of the RPN 3x3 conv is otherwise split 9 taps x 8 column tiles = 72 times.  Measures what a
// "producer writes the planes" design could gain; not on the product path.
#include "conv_common.h"

namespace dadet {

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
constexpr int PSTRIDE = 40;   // bf16 per staged row (80 bytes), as conv_split.hip

__device__ inline unsigned pack2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// fp32 [n] -> three bf16 planes [3][n]: x = p0 + p1 + p2 exactly (round-to-nearest residual chain)
__global__ void split_planes_kernel(const float4* __restrict__ x, uint2* __restrict__ planes, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = x[i];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const unsigned a = pack2(v.x, v.y), b = pack2(v.z, v.w);
      planes[(int64_t)t * n4 + i] = make_uint2(a, b);
      v.x -= __builtin_bit_cast(float, a << 16);
      v.y -= __builtin_bit_cast(float, a & 0xFFFF0000u);
      v.z -= __builtin_bit_cast(float, b << 16);
      v.w -= __builtin_bit_cast(float, b & 0xFFFF0000u);
    }
  }
}

__device__ inline uint2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}

// 128x128x32 tiles, 4 waves, 3 term planes, 6 MFMAs per product; x / w are plane tensors (element offsets as in the
// fp32 tensors, 2 bytes per element, plane p at + p * plane_bytes)
__global__ __launch_bounds__(256, 2) void conv_fwd_planes_kernel(const ConvArgs a, unsigned x_plane_bytes,
                                                                 unsigned w_plane_bytes) {
  constexpr int TM = 2, TN = 2, TERMS = 3, BM = 128, BN = 128, A_LOADS = 4, B_LOADS = 4;
  constexpr int A_PLANE = BM * PSTRIDE, B_PLANE = BN * PSTRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);
  __bf16* Bs = As + TERMS * A_PLANE;
  const int nwg = a.tiles_m * a.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int bm0 = (tile / a.tiles_n) * BM, bn0 = (tile % a.tiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int lcol = t & 7, lgrp = t >> 3;
  const int lrow = (lgrp >> 3) * 8 + (lgrp & 1) * 4 + ((lgrp >> 1) & 3);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, 3u * x_plane_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, 3u * w_plane_bytes);
  int pixbase[A_LOADS], hi0[A_LOADS], wi0[A_LOADS];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int m = bm0 + lrow + 32 * i;
    if (m < a.M) {
      const int img = m / HoWo, rem = m - img * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      pixbase[i] = img * a.H * a.W;
      hi0[i] = ho * a.stride - a.pad;
      wi0[i] = wo * a.stride - a.pad;
    } else {
      pixbase[i] = 0;
      hi0[i] = -(1 << 28);
      wi0[i] = 0;
    }
  }
  unsigned wrow[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int n = bn0 + lrow + 32 * i;
    wrow[i] = n < a.Cout ? (unsigned)n * (unsigned)a.K * 2u : kOOB;
  }
  uint2 pa_[A_LOADS][TERMS], pb_[B_LOADS][TERMS];
  int kk = lcol * 4;
  int tap = kk / a.Cin, kc = kk - tap * a.Cin, kr = tap / a.KW, ks = tap - kr * a.KW;
  auto load_a = [&]() {
    const bool kvalid = kk < a.K;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = hi0[i] + kr, wi = wi0[i] + ks;
      const bool ok = kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const unsigned off = ((unsigned)(pixbase[i] + hi * a.W + wi) * (unsigned)a.Cin + (unsigned)kc) * 2u;
#pragma unroll
      for (int p = 0; p < TERMS; ++p) pa_[i][p] = buf_load2(xr, ok ? off + (unsigned)p * x_plane_bytes : kOOB);
    }
  };
  auto load_b = [&]() {
    const bool kvalid = kk < a.K;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        pb_[i][p] = buf_load2(wr, (kvalid && wrow[i] != kOOB) ? wrow[i] + (unsigned)kk * 2u + (unsigned)p * w_plane_bytes
                                                              : kOOB);
  };
  auto advance = [&]() {
    kk += BK;
    kc += BK;
    while (kc >= a.Cin) {
      kc -= a.Cin;
      if (++ks == a.KW) {
        ks = 0;
        ++kr;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        *reinterpret_cast<uint2*>(As + p * A_PLANE + (lrow + 32 * i) * PSTRIDE + lcol * 4) = pa_[i][p];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        *reinterpret_cast<uint2*>(Bs + p * B_PLANE + (lrow + 32 * i) * PSTRIDE + lcol * 4) = pb_[i][p];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = (a.K + BK - 1) / BK;
  load_a();
  load_b();
  advance();
  store_tile();
  __syncthreads();
  const int frag_row = lane & 31, frag_k = (lane >> 5) * 8;
  const __bf16* Ab = As + (wm * TM * 32 + frag_row) * PSTRIDE + frag_k;
  const __bf16* Bb = Bs + (wn * TN * 32 + frag_row) * PSTRIDE + frag_k;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    load_a();     // tile kt + 1 straight into the staging registers (they were stored before the last barrier)
    load_b();
    advance();
#pragma unroll
    for (int step = 0; step < BK / 16; ++step) {
      bf16x8p fa[TERMS][TM], fb[TERMS][TN];
#pragma unroll
      for (int p = 0; p < TERMS; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[p][i] = *reinterpret_cast<const bf16x8p*>(Ab + p * A_PLANE + i * 32 * PSTRIDE + step * 16);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          fb[p][i] = *reinterpret_cast<const bf16x8p*>(Bb + p * B_PLANE + i * 32 * PSTRIDE + step * 16);
      }
#pragma unroll
      for (int order = 2 * (TERMS - 1); order >= 0; --order)
#pragma unroll
        for (int pa = 0; pa < TERMS; ++pa) {
          const int pb = order - pa;
          if (pb < 0 || pb >= TERMS || pa + pb > TERMS - 1) continue;
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][im], fb[pb][in], acc[im][in], 0, 0, 0);
        }
    }
    if (more) {
      __syncthreads();
      store_tile();
      __syncthreads();
    }
  }
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y, a.y_bytes);
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int n = bn0 + wn * TN * 32 + in * 32 + col_in;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = bm0 + wm * TM * 32 + im * 32 + (reg & 3) + 8 * (reg >> 2) + row_hi;
        buf_store1(yr, (n < a.Cout && m < a.M) ? ((unsigned)m * (unsigned)a.Cout + (unsigned)n) * 4u : kOOB,
                   acc[im][in][reg]);
      }
  }
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_split_planes(const float* x, void* planes_bf16, int64_t n, void* stream) {
  DADET_REQUIRE(n >= 0 && n % 4 == 0, "split_planes: n must be a multiple of 4");
  if (n == 0) return DADET_OK;
  DADET_REQUIRE(x && planes_bf16, "split_planes: null pointer");
  int64_t blocks = ceil_div64(n / 4, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(split_planes_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<uint2*>(planes_bf16), n / 4);
  return check_launch("split_planes");
}

// probe only: stride-1 "same" convolution (or 1x1), plain store epilogue
extern "C" int dadet_conv_forward_planes_probe(const dadet_conv_desc* d, const void* x_planes, const void* w_planes,
                                               float* y, void* stream) {
  DADET_REQUIRE(d && x_planes && w_planes && y, "conv_forward_planes_probe: null pointer");
  DADET_REQUIRE(d->Cin % 4 == 0, "conv_forward_planes_probe: Cin % 4");
  ConvArgs a{};
  a.x = reinterpret_cast<const float*>(x_planes);
  a.w = reinterpret_cast<const float*>(w_planes);
  a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
  a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo; a.OutH = d->Ho; a.OutW = d->Wo; a.os = 1;
  a.M = d->N * d->Ho * d->Wo;
  a.K = d->KH * d->KW * d->Cin;
  a.y_bytes = (unsigned)((size_t)a.M * a.Cout * 4);
  a.tiles_m = ceil_div(a.M, 128);
  a.tiles_n = ceil_div(a.Cout, 128);
  const unsigned xpb = (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2);
  const unsigned wpb = (unsigned)((size_t)d->Cout * a.K * 2);
  const size_t lds = sizeof(__bf16) * 3 * 256 * PSTRIDE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_planes_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_fwd_planes_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, as_stream(stream), a, xpb,
                     wpb);
  return check_launch("conv_forward_planes_probe");
}
