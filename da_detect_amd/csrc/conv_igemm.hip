// Implicit-GEMM convolution (forward / data-gradient / weight-gradient) on the gfx950 fp32 matrix
// pipe (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain — guide §3), NHWC activations,
// [Cout][KH][KW][Cin] weights.
//
// Stands in for the ATen conv2d / linear calls of the reference hot path and fuses what the reference
// runs as separate elementwise passes around them:
//   forward : conv -> FrozenBatchNorm2d affine (layers/batch_norm.py:19-24) -> (+ residual) -> relu_
//             (modeling/backbone/resnet.py:294-314, :331-336), conv + bias + relu (rpn/rpn.py:39-46,
//             da_heads/da_heads.py:32-37), nn.Linear (+relu) (da_heads.py:61-68, roi_box_predictors.py:28-33)
//   dgrad   : the same kernel run on the output gradient with the flipped/transposed weights, with the
//             upstream ReLU gating and the residual-gradient add fused into the epilogue
//   wgrad   : dW = gY^T * im2col(X), split over the (huge) N*Ho*Wo reduction axis, deterministic two-pass.
//
// GEMM view (forward): C[m][n] = sum_k A[m][k] * B[n][k],  m = (img, ho, wo), n = cout, k = (r, s, cin).
// Both operands are K-contiguous in HBM (NHWC rows / KRSC rows), so a K-tile of 32 is one 128-byte run
// per row: each lane moves 16 B, a wavefront covers 8 rows x 128 B.  Tiles are staged through LDS with a
// +4-float row pad (row stride 36 floats): the MFMA fragment reads are ds_read_b128 (4 consecutive k per
// lane, the k-slot order is permuted identically for A and B so the product is unchanged) and are
// bank-conflict free for the 16-lane service groups of ds_read_b128 (36*i mod 64 is a permutation of the
// 16 quad-slots); the staging writes are ds_write_b128 of 8 contiguous lanes per row.
// A 256-thread workgroup = 4 wavefronts as 2x2, each wavefront owns a (TM*32)x(TN*32) block of the
// output tile, accumulators live in registers (16 fp32 per 32x32 block).  K loop: the global loads of
// tile t+1 are issued into registers before the MFMAs of tile t and written to the (single) LDS buffer
// after them; at ~36 KB of LDS and 144 registers three workgroups are resident per CU (12 waves), and it is
// this cross-workgroup overlap that keeps the matrix pipe fed across the two barriers of a K-tile (PMC on
// the double-buffered / 2-workgroup variant: 38% of wave cycles parked in s_waitcnt/s_barrier).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of m-tiles.
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include "conv_common.h"
#include <atomic>
#include <stdlib.h>

namespace dadet {

template <int TM, int TN>
__global__ __launch_bounds__(256, 3) void conv_fwd_kernel(const ConvArgs a) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;  // float4 loads per thread per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);                 // [BM][LDS_STRIDE]
  float* Bs = As + BM * LDS_STRIDE;                           // [BN][LDS_STRIDE]

  const int nwg = a.tiles_m * a.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int bm0 = (tile / a.tiles_n) * BM;
  const int bn0 = (tile % a.tiles_n) * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lcol = t & 7;    // float4 column of the K-tile this thread stages
  const int lrow = t >> 3;   // first staged row (0..31)

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, a.w_bytes);

  // per staged A row: image pixel base and top-left input coordinate
  int pixbase[A_LOADS], hi0[A_LOADS], wi0[A_LOADS];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int m = bm0 + lrow + 32 * i;
    if (m < a.M) {
      const int img = m / HoWo;
      const int rem = m - img * HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      pixbase[i] = img * a.H * a.W;
      hi0[i] = ho * a.stride - a.pad;
      wi0[i] = wo * a.stride - a.pad;
    } else {
      pixbase[i] = 0;
      hi0[i] = -(1 << 28);  // fails every bounds check
      wi0[i] = 0;
    }
  }
  // weight rows of this thread (byte offsets of column 0), out-of-range rows read as zero
  unsigned wrow[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int n = bn0 + lrow + 32 * i;
    wrow[i] = n < a.Cout ? (unsigned)n * (unsigned)a.K * 4u : kOOB;
  }

  float4 ra[A_LOADS], rb[B_LOADS];

  // (r, s, c) of this thread's float4 column, advanced incrementally from K-tile to K-tile
  int kk = lcol * 4;
  int tap = kk / a.Cin;
  int kc = kk - tap * a.Cin;
  int kr = tap / a.KW;
  int ks = tap - kr * a.KW;

  auto load_tile = [&]() {
    const bool kvalid = kk < a.K;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = hi0[i] + kr, wi = wi0[i] + ks;
      const bool ok = kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const unsigned off = ((unsigned)(pixbase[i] + hi * a.W + wi) * (unsigned)a.Cin + (unsigned)kc) * 4u;
      ra[i] = buf_load4(xr, ok ? off : kOOB);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      rb[i] = buf_load4(wr, (kvalid && wrow[i] != kOOB) ? wrow[i] + (unsigned)kk * 4u : kOOB);
    // advance to the next K-tile
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
      *reinterpret_cast<float4*>(As + (lrow + 32 * i) * LDS_STRIDE + lcol * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      *reinterpret_cast<float4*>(Bs + (lrow + 32 * i) * LDS_STRIDE + lcol * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  load_tile();
  store_tile();
  __syncthreads();

  const int frag_row = lane & 31;        // row of the 32-row fragment this lane feeds
  const int frag_k = (lane >> 5) * 4;    // which 4-float half of each 8-float k-group
  const float* Ab = As + (wm * TM * 32 + frag_row) * LDS_STRIDE + frag_k;
  const float* Bb = Bs + (wn * TN * 32 + frag_row) * LDS_STRIDE + frag_k;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile();  // buffer loads in flight during the MFMAs below
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_STRIDE + j * 8);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        fb[i] = *reinterpret_cast<const float4*>(Bb + i * 32 * LDS_STRIDE + j * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            const float av = e == 0 ? fa[im].x : e == 1 ? fa[im].y : e == 2 ? fa[im].z : fa[im].w;
            const float bv = e == 0 ? fb[in].x : e == 1 ? fb[in].y : e == 2 ? fb[in].z : fb[in].w;
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[im][in], 0, 0, 0);
          }
      }
    }
    // one LDS buffer (36 KB for the 128x128 tile): three workgroups share a CU, so another workgroup's MFMAs
    // cover this one's staging; the price is a second barrier per K-tile
    if (kt + 1 < nk) {
      __syncthreads();
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: D layout of the 32x32 MFMA — col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  // Ragged edges are handled by out-of-range buffer offsets (loads give 0, stores are dropped).
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y, a.y_bytes);
  const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.addend ? a.addend : a.y, a.addend ? a.y_bytes : 0u);
  const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_ref ? a.mask_ref : a.y, a.mask_ref ? a.y_bytes : 0u);
  const int col_in = lane & 31;
  const int row_hi = 4 * (lane >> 5);
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int n = bn0 + wn * TN * 32 + in * 32 + col_in;
    const bool nvalid = n < a.Cout;
    const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
    const float bi = (a.bias && nvalid) ? a.bias[n] : 0.f;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
      // four rows (one accumulator row-group) at a time keeps the epilogue's live registers small; the
      // scheduling barrier stops hipcc from interleaving all 16 groups (which cost 200+ VGPRs and occupancy)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        unsigned offs[4];
        float add[4], msk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = bm0 + wm * TM * 32 + im * 32 + q + 8 * g + row_hi;
          unsigned orow = (unsigned)m;
          if (a.os != 1) {
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            orow = (unsigned)((img * a.OutH + ho * a.os) * a.OutW + wo * a.os);
          }
          offs[q] = (nvalid && m < a.M) ? (orow * (unsigned)a.Cout + (unsigned)n) * 4u : kOOB;
        }
        if (a.addend) {
#pragma unroll
          for (int q = 0; q < 4; ++q) add[q] = buf_load1(ar, offs[q]);
        }
        if (a.relu_mode == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) msk[q] = buf_load1(mr, offs[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[im][in][g * 4 + q];
          if (a.scale) v = v * sc;
          if (a.bias) v = v + bi;
          if (a.addend) v = v + add[q];
          if (a.relu_mode == 1) v = fmaxf(v, 0.f);
          else if (a.relu_mode == 2) v = (msk[q] > 0.f) ? v : 0.f;
          buf_store1(yr, offs[q], v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient.  GEMM view: D[co][kc] = sum_m gY[m][co] * Xg[m][kc], kc = (r, s, ci).
// Both operands are staged as [32 m-rows][128 columns] (their natural HBM orientation, 16 B per lane
// along the channel axis); MFMA fragments are ds_read_b32 down the columns — consecutive lanes hit
// consecutive banks, so no padding is needed.  grid = (co tiles * kc tiles, m splits).

__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TILE = 128, RK = 32;  // output tile 128x128, 32 m-rows per step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Gs = reinterpret_cast<float*>(smem);   // [RK][TILE]  gY rows
  float* Xs = Gs + RK * TILE;                   // [RK][TILE]  gathered X rows

  const int tile = xcd_remap(blockIdx.x, a.tiles_co * a.tiles_kc);
  const int co0 = (tile / a.tiles_kc) * TILE;
  const int kc0 = (tile % a.tiles_kc) * TILE;
  const int split = blockIdx.y;
  const int m_begin = split * a.rows_per_split;
  int m_end = m_begin + a.rows_per_split;
  if (m_end > a.M) m_end = a.M;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lcol = t & 31;   // float4 column (0..31) of the 128-wide row
  const int lrow = t >> 5;   // 0..7 ; rows lrow + 8*i

  // this thread's X column -> (tap, channel) is fixed for the whole kernel
  const int kk = kc0 + lcol * 4;
  const bool kvalid = kk < a.K;
  const int tap = kk / a.Cin;
  const int ci = kk - tap * a.Cin;
  const int r = tap / a.KW;
  const int s = tap - r * a.KW;
  const int co = co0 + lcol * 4;
  const bool covalid = co < a.Cout;  // Cout % 4 == 0 is required by the host wrapper
  const int HoWo = a.Ho * a.Wo;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t gr = make_rsrc(a.gy, a.gy_bytes);
  float4 rg[4], rx[4];
  const unsigned co_off = covalid ? (unsigned)co * 4u : kOOB;
  // (img, ho, wo) of this thread's four rows, advanced by RK rows per step
  int r_img[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_begin + lrow + 8 * i;
    r_img[i] = m / HoWo;
    const int rem = m - r_img[i] * HoWo;
    r_ho[i] = rem / a.Wo;
    r_wo[i] = rem - r_ho[i] * a.Wo;
  }
  const int d_img = RK / HoWo, d_ho = (RK - d_img * HoWo) / a.Wo, d_wo = RK - d_img * HoWo - d_ho * a.Wo;
  int m_cur = m_begin;
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_cur + lrow + 8 * i;
      const bool mv = m < m_end;
      rg[i] = buf_load4(gr, (mv && covalid) ? (unsigned)m * (unsigned)a.gy_ld * 4u + co_off : kOOB);
      const int hi = r_ho[i] * a.stride - a.pad + r;
      const int wi = r_wo[i] * a.stride - a.pad + s;
      const bool ok = mv && kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const unsigned off = ((unsigned)((r_img[i] * a.H + hi) * a.W + wi) * (unsigned)a.Cin + (unsigned)ci) * 4u;
      rx[i] = buf_load4(xr, ok ? off : kOOB);
      // advance this row by RK output pixels
      int wo = r_wo[i] + d_wo;
      const int cw = wo >= a.Wo;
      wo -= cw ? a.Wo : 0;
      int ho = r_ho[i] + d_ho + cw;
      const int ch = ho >= a.Ho;
      ho -= ch ? a.Ho : 0;
      r_wo[i] = wo;
      r_ho[i] = ho;
      r_img[i] += d_img + ch;
    }
    m_cur += RK;
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(Gs + (lrow + 8 * i) * TILE + lcol * 4) = rg[i];
      *reinterpret_cast<float4*>(Xs + (lrow + 8 * i) * TILE + lcol * 4) = rx[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nsteps = (m_end - m_begin + RK - 1) / RK;
  if (nsteps > 0) {
    load_tile();
    store_tile();
  }
  __syncthreads();
  const int fcol = lane & 31, fk = lane >> 5;
  const float* Gb = Gs + wm * 64 + fcol;
  const float* Xb = Xs + wn * 64 + fcol;
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) load_tile();
#pragma unroll
    for (int k2 = 0; k2 < RK / 2; ++k2) {
      const int row = 2 * k2 + fk;
      const float g0 = Gb[row * TILE], g1 = Gb[row * TILE + 32];
      const float x0 = Xb[row * TILE], x1 = Xb[row * TILE + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x1, acc[1][1], 0, 0, 0);
    }
    if (st + 1 < nsteps) {
      __syncthreads();
      store_tile();
      __syncthreads();
    }
  }

  float* out = a.direct ? a.out : a.out + (size_t)split * a.Cout * a.K;
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
#pragma unroll
  for (int in = 0; in < 2; ++in) {
    const int kc = kc0 + wn * 64 + in * 32 + col_in;
    if (kc >= a.K) continue;
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int c = co0 + wm * 64 + im * 32 + (reg & 3) + 8 * (reg >> 2) + row_hi;
        if (c >= a.Cout) continue;
        const size_t off = (size_t)c * a.K + kc;
        float v = acc[im][in][reg];
        if (a.direct) {
          if (a.out_scale) v = v * a.out_scale[c];
          if (a.accumulate) v = v + out[off];
        }
        out[off] = v;
      }
  }
}

__global__ void wgrad_reduce_kernel(const float4* __restrict__ part, const float* __restrict__ out_scale,
                                    float4* __restrict__ dw, int64_t total4, int K4, int splits,
                                    int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 s = part[i];
    for (int p = 1; p < splits; ++p) {
      const float4 v = part[(int64_t)p * total4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (out_scale) {
      const float sc = out_scale[i / K4];
      s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
    }
    if (accumulate) {
      const float4 o = dw[i];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    dw[i] = s;
  }
}

// The same reduction for up to kReduceBatch (32: a step's deferred passes go out in two launches) weight gradients in ONE launch (dadet_conv_wgrad_reduce_batch): a residual
// block's backward produces 3 - 4 split weight gradients of 0.3 - 9 MB each; one reduction pass per tensor is a
// 10 - 40 us launch that runs at ~1.5 TB/s because it is over before it fills the chip (45 launches, 1.2 ms per step).
// Block b belongs to the item whose block range contains it; within an item the arithmetic is wgrad_reduce_kernel's.
constexpr int kReduceBatch = 32;
struct ReduceItem {
  const float4* part;
  const float* out_scale;
  float4* dw;
  long long total4;
  int K4, splits, accumulate, first_block;
};
struct ReduceBatch {
  ReduceItem item[kReduceBatch];
  int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch batch) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < kReduceBatch; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.item[i].first_block) k = i;
  const ReduceItem& it = batch.item[k];
  const int nblocks = (k + 1 < batch.n ? batch.item[k + 1].first_block : (int)gridDim.x) - it.first_block;
  for (int64_t i = (int64_t)((int)blockIdx.x - it.first_block) * 256 + threadIdx.x; i < it.total4;
       i += (int64_t)nblocks * 256) {
    float4 s = it.part[i];
    for (int p = 1; p < it.splits; ++p) {
      const float4 v = it.part[(int64_t)p * it.total4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (it.out_scale) {
      const float sc = it.out_scale[i / it.K4];
      s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
    }
    if (it.accumulate) {
      const float4 o = it.dw[i];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    it.dw[i] = s;
  }
}

// wt[ci][KH-1-r][KW-1-s][co] = w[co][r][s][ci] * scale[co]
// 32x32 LDS tile transpose between the co axis and the ci axis for one (r,s) tap.
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w,
                                                               const float* __restrict__ scale,
                                                               float* __restrict__ wt, int Cout, int KH,
                                                               int KW, int Cin, int CoutPad) {
  // CoutPad >= Cout: the output rows are CoutPad wide, the columns co >= Cout are zeros (a weight whose output channels the
  // forward pads to a multiple of four — the offset branch of a deformable block)
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int r = tap / KW, s = tap % KW;
  const int tapT = (KH - 1 - r) * KW + (KW - 1 - s);
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t K = (int64_t)KH * KW * Cin, Kt = (int64_t)KH * KW * CoutPad;
  for (int j = ty; j < 32; j += 8) {
    const int co = co0 + j, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      v = w[(int64_t)co * K + (int64_t)tap * Cin + ci];
      if (scale) v = v * scale[co];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int ci = ci0 + j, co = co0 + tx;
    if (co < CoutPad && ci < Cin) wt[(int64_t)ci * Kt + (int64_t)tapT * CoutPad + co] = tile[tx][j];
  }
}

// Every registered weight in ONE launch (dadet_conv_weight_transpose_batch): the backward pass of a step needs the
// transposed, FrozenBN-folded form of ~44 convolution weights, and producing each right in front of its data-gradient GEMM
// put 42 launches of ~5 us (plus their dispatch gaps) into the serial GEMM chain (rocprofv3 timeline of round 3: 0.22 ms
// per step with nothing else running).  The table lives in device memory; block b serves the item whose block range holds it.
struct TransposeItem {
  const float* w;
  const float* scale;
  float* wt;
  int Cout, KH, KW, Cin;
  int first_block, blocks_ci, blocks_co;
  int cout_pad;     // width of the output rows (>= Cout; 0: Cout), see weight_transpose_kernel
};

__global__ __launch_bounds__(256) void weight_transpose_batch_kernel(const TransposeItem* __restrict__ items, int n) {
  __shared__ float tile[32][33];
  __shared__ int s_item;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;                 // last item whose first_block <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].first_block <= (int)blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    s_item = lo;
  }
  __syncthreads();
  const TransposeItem it = items[s_item];
  int b = (int)blockIdx.x - it.first_block;
  const int bx = b % it.blocks_ci;
  b /= it.blocks_ci;
  const int by = b % it.blocks_co, tap = b / it.blocks_co;
  const int r = tap / it.KW, sidx = tap % it.KW;
  const int tapT = (it.KH - 1 - r) * it.KW + (it.KW - 1 - sidx);
  const int ci0 = bx * 32, co0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int cout_pad = it.cout_pad > it.Cout ? it.cout_pad : it.Cout;
  const int64_t K = (int64_t)it.KH * it.KW * it.Cin, Kt = (int64_t)it.KH * it.KW * cout_pad;
  for (int j = ty; j < 32; j += 8) {
    const int co = co0 + j, ci = ci0 + tx;
    float v = 0.f;
    if (co < it.Cout && ci < it.Cin) {
      v = it.w[(int64_t)co * K + (int64_t)tap * it.Cin + ci];
      if (it.scale) v = v * it.scale[co];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int ci = ci0 + j, co = co0 + tx;
    if (co < cout_pad && ci < it.Cin) it.wt[(int64_t)ci * Kt + (int64_t)tapT * cout_pad + co] = tile[tx][j];
  }
}

static int conv_desc_check(const dadet_conv_desc* d, const char* who) {
  DADET_REQUIRE(d, "%s: null descriptor", who);
  DADET_REQUIRE(d->N >= 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                    d->stride > 0 && d->pad >= 0 && d->Ho > 0 && d->Wo > 0,
                "%s: bad dims", who);
  DADET_REQUIRE(d->Cin % 4 == 0, "%s: Cin=%d must be a multiple of 4 (pad the channel axis)", who, d->Cin);
  DADET_REQUIRE((int64_t)d->N * d->H * d->W * d->Cin < (1LL << 31) &&
                    (int64_t)d->N * d->Ho * d->Wo < (1LL << 31),
                "%s: tensor too large for 32-bit pixel indexing", who);
  return DADET_OK;
}

}  // namespace dadet

using namespace dadet;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int TM, int TN>
static int launch_fwd(ConvArgs& a, hipStream_t st) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cout, BN);
  const size_t lds = sizeof(float) * (BM + BN) * LDS_STRIDE;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_kernel<TM, TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_fwd_kernel<TM, TN>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, st, a);
  return check_launch("conv_forward");
}

// Tile variant of the split-bf16 forward / data-gradient GEMM.  Short reductions are HBM bound, and a workgroup's load /
// compute / store phases overlap only across the workgroups that share a CU: 64x64 tiles (30 KB of LDS, 5 workgroups per
// CU instead of 2) run the K <= 256 layers 3 - 38% faster (tools/gemm_table.py: res2 1x1 64->256 0.211 -> 0.153 ms, res3
// 1x1 128->512 0.156 -> 0.130).  Only whole K-tiles (the K = 76 RPN data gradient got 44% slower) and only where the
// 128x128 tile would be chosen.  DADET_FWD_VARIANT forces a variant (tools/fwd_sweep.py; read per call).
static int split_fwd_variant(int M, int Cout, int K) {
  int variant = fwd_variant(M, Cout);
  static const int kmax = getenv("DADET_SHORTK_MAX") ? atoi(getenv("DADET_SHORTK_MAX")) : 256;
  if (variant == 0 && K <= kmax && K % BK == 0) variant = 2;
  if (const char* e = getenv("DADET_FWD_VARIANT")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 2) variant = v;
  }
  return variant;
}

// ---- split-K for GEMMs whose tile grid cannot fill the chip (the M = 512 linear layers of the box / instance heads:
// 4 x 16 workgroups walking K = 2048 alone took 40 - 70 us each, one after the other in the loss turn-around).
// The K range is cut over blockIdx.y, partial sums go to a per-stream scratch buffer owned by the library (grown on
// demand, reused in stream order), and one pass sums them in split order and applies the epilogue.
__global__ void splitk_reduce_kernel(const float4* __restrict__ partial, int splits, size_t stride4,
                                     const float4* __restrict__ scale, const float4* __restrict__ bias,
                                     const float4* __restrict__ addend, const float4* __restrict__ mask,
                                     float4* __restrict__ y, int64_t total4, int C4, int relu_mode,
                                     unsigned* __restrict__ amax_y) {
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = partial[i];
    for (int p = 1; p < splits; ++p) {
      const float4 q = partial[(size_t)p * stride4 + i];
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    const int c = (int)(i % C4);
    if (scale) { const float4 q = scale[c]; v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w; }
    if (bias) { const float4 q = bias[c]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (addend) { const float4 q = addend[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (relu_mode == 1) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (relu_mode == 2) {
      const float4 q = mask[i];
      v.x = q.x > 0.f ? v.x : 0.f; v.y = q.y > 0.f ? v.y : 0.f;
      v.z = q.z > 0.f ? v.z : 0.f; v.w = q.w > 0.f ? v.w : 0.f;
    }
    y[i] = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (amax_y) amax_publish(amax_y, mx);
}

// max|x| over a tensor, merged into *slot (mode 4: a GEMM operand whose producer left no maximum).  Bits of non-negative
// floats order like unsigned integers; amax_publish: at most one atomic per workgroup, sharded by XCD.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ slot) {
  float mx = 0.f;
  const int64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) mx = fmaxf(mx, fabsf(x[n4 * 4 + threadIdx.x]));
  amax_publish(slot, mx);
}

// the same for many tensors in one launch (the weights of a model once per optimizer step): item i owns the workgroups
// [first_block, first_block + blocks)
__global__ __launch_bounds__(256) void amax_batch_kernel(const dadet_amax_item* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {          // last item whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dadet_amax_item it = items[lo];
  const int b = (int)blockIdx.x - it.first_block;
  const float* x = reinterpret_cast<const float*>(it.x);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const int64_t n4 = it.n / 4;
  float mx = 0.f;
  // four loads in flight per lane and round (one dependent load per round ran this launch at 1.6 TB/s)
  const int64_t stride = (int64_t)it.blocks * 256;
  int64_t i = (int64_t)b * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                         fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                         fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (b == 0 && threadIdx.x < (unsigned)(it.n - n4 * 4)) mx = fmaxf(mx, fabsf(x[n4 * 4 + threadIdx.x]));
  amax_publish(reinterpret_cast<unsigned*>(it.slot), mx);
}

// ---- non-finite guard: device words + a ring of recent launch records (conv_common.h: nf_check) -------------------------
__device__ unsigned g_nf_words[2];
__device__ unsigned g_nf_taken[2];
// read-and-clear in one atomic step per word: a record set while the host polls is either in this poll or in the next
__global__ void nf_take_kernel() {
  g_nf_taken[1] = atomicExch(&g_nf_words[1], 0u);
  g_nf_taken[0] = atomicExch(&g_nf_words[0], 0u);
}
namespace {
struct NfRecord { unsigned id; char kind[24]; int M, N, K, KH; };
constexpr int kNfRing = 8192;
NfRecord g_nf_ring[kNfRing];
std::atomic<unsigned> g_nf_next{0};
}  // namespace
namespace dadet {
unsigned nf_next_launch(const char* kind, int M, int N, int K, int KH) {
  const unsigned id = g_nf_next.fetch_add(1);
  NfRecord& r = g_nf_ring[id % kNfRing];
  r.id = id;
  snprintf(r.kind, sizeof(r.kind), "%s", kind);
  r.M = M; r.N = N; r.K = K; r.KH = KH;
  return id;
}
unsigned* nf_flag_ptr() {
  static unsigned* p = [] {
    void* q = nullptr;
    return hipGetSymbolAddress(&q, HIP_SYMBOL(g_nf_words)) == hipSuccess ? static_cast<unsigned*>(q) : nullptr;
  }();
  static const bool off = getenv("DADET_NONFINITE_GUARD") && getenv("DADET_NONFINITE_GUARD")[0] == '0';
  return off ? nullptr : p;
}
}  // namespace dadet

namespace {
struct Scratch { void* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mutex;
std::unordered_map<hipStream_t, Scratch> g_scratch;

// scratch of `bytes` for work queued on `st`; contents are only valid in stream order
void* stream_scratch(hipStream_t st, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  Scratch& s = g_scratch[st];
  if (s.bytes < bytes) {
    if (s.p) {
      (void)hipStreamSynchronize(st);   // the old buffer may still be read by queued work
      (void)hipFree(s.p);
      s.p = nullptr;
      s.bytes = 0;
    }
    const size_t want = bytes < (size_t)(8u << 20) ? (size_t)(8u << 20) : bytes * 2;
    if (hipMalloc(&s.p, want) != hipSuccess) {
      s.p = nullptr;
      return nullptr;
    }
    s.bytes = want;
  }
  return s.p;
}

// arrival counters of the stream-K tail: one persistent zero-initialised buffer per stream (work queued on a stream is
// ordered, so one launch owns it at a time); the workgroup that completes a tile resets that tile's counter
constexpr int kSkCounters = 4096 + 4 * 2048;     // + the symmetric two-part meeting of conv_big.hip: four words per tile
int* stream_counters(hipStream_t st) {
  static std::mutex m;
  static std::unordered_map<hipStream_t, int*> table;
  std::lock_guard<std::mutex> lock(m);
  int*& p = table[st];
  if (!p) {
    if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(int) * kSkCounters) != hipSuccess) {
      p = nullptr;
      return nullptr;
    }
    if (hipMemset(p, 0, sizeof(int) * kSkCounters) != hipSuccess) return nullptr;
  }
  return p;
}

// mode 4 through the plain entry points (no maxima handed in): two slots per stream that the library fills itself
unsigned* stream_amax_slots(hipStream_t st) {
  static std::mutex m;
  static std::unordered_map<hipStream_t, unsigned*> table;
  std::lock_guard<std::mutex> lock(m);
  unsigned*& p = table[st];
  if (!p && hipMalloc(reinterpret_cast<void**>(&p), sizeof(unsigned) * (7 * (size_t)kAmaxStride + 4)) != hipSuccess)
    p = nullptr;
  return p;
}
// zero the eight shards of `n` adjacent slots
hipError_t zero_slots(unsigned* first, int n, hipStream_t st) {
  return hipMemset2DAsync(first, sizeof(unsigned) * kAmaxStride, 0, sizeof(unsigned) * n, 8, st);
}

int launch_amax(const float* x, int64_t n, unsigned* slot, hipStream_t st) {
  int64_t blocks = ceil_div64(n / 4 > 0 ? n / 4 : 1, 256 * 4);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(amax_kernel, dim3((int)blocks), dim3(256), 0, st, x, n, slot);
  return check_launch("amax");
}

// Stream-K tail plan for the 128x128 split kernel (conv_fwd_split_sk_kernel).  Returns false when the plain grid is at
// least as good: the last pass of the tile grid over the 2 x 256 workgroup slots is (nearly) full, or K is so short that
// parking / summing partial tiles would cost more than the idle slots.
struct SkPlan { int dp_tiles, sk_tiles, units, iters, max_parts; };
bool streamk_plan(const ConvArgs& a, int variant, SkPlan* p) {
  const char* env = getenv("DADET_STREAMK");      // read per call: tests and A/B runs flip it at run time
  if ((env && env[0] == '0') || variant != 0 || a.ablate) return false;
  const int slots = 2 * kNumCU;
  const int tiles = ceil_div(a.M, 128) * ceil_div(a.Cout, 128);
  const int nk = ceil_div(a.K, BK);
  const int tail = tiles % slots;
  // a grid below one pass is not stream-K'd: cutting 256 tiles into 512 halves measured 7 - 22% SLOWER (the partial-tile
  // round trip costs more than the second workgroup per CU gains; tools/streamk_bench.py)
  if (tiles > kNumCU && tiles < slots && nk >= 32) {
    // between one workgroup per CU and two (e.g. the 392 tiles of the res5 GEMMs over 256 ROIs): all workgroups are
    // resident at once, but 136 CUs run two of them and 120 run one — the launch lasts as long as the pairs.  All tiles
    // become stream-K tiles: 512 equal ranges, every CU gets two.  DADET_STREAMK_SMALL=0 switches this case off.
    // Mode 4 (three MFMAs per K=16): a tile's K loop is short enough that parking / summing the partial tiles and the
    // operand panels the ranges no longer share cost more than the uneven CUs — `img_only` 15.27 -> 14.81 ms, R-101-FPN-DCN
    // 49.1 -> 46.6 ms with this case off (three alternating runs each on one box); it stays on for the six-MFMA mode 3,
    // where it was measured (+18% on the res5 GEMMs).  DADET_STREAMK_SMALL = 0 / 1 forces it.
    const char* small = getenv("DADET_STREAMK_SMALL");
    if (small ? small[0] == '0' : gemm_mode() == 4) return false;
    p->dp_tiles = 0;
    p->sk_tiles = tiles;
    p->iters = ceil_div(tiles * nk, slots);
    p->units = ceil_div(tiles * nk, p->iters);
    p->max_parts = ceil_div(nk, p->iters) + 1;
    return true;
  }
  if (tail == 0 || nk < 16 || tiles < slots || tail > kSkCounters) return false;
  if (tail > slots * 7 / 8) return false;                 // the last pass is full enough
  if (tiles > 6 * slots && tail > slots / 2) return false;  // many passes: the idle share is small
  p->dp_tiles = tiles - tail;
  p->sk_tiles = tail;
  // ranges of at least 8 K-tiles: a short tail (e.g. 32 tiles of 64 K-tiles behind three full passes) is spread over
  // fewer workgroups rather than cut into slivers
  p->iters = ceil_div(tail * nk, slots);
  if (p->iters < 8) p->iters = 8;
  p->units = ceil_div(tail * nk, p->iters);
  p->max_parts = ceil_div(nk, p->iters) + 1;
  return true;
}

// number of K elements per split (multiple of BK), or 0 when the launch should not be split
int splitk_plan(const ConvArgs& a, int variant) {
  static const bool enabled = !(getenv("DADET_SPLITK") && getenv("DADET_SPLITK")[0] == '0');
  if (!enabled || a.os != 1 || a.Cout % 4 != 0 || a.K < 256) return 0;
  const int bm = variant == 2 ? 64 : 128, bn = variant == 0 ? 128 : 64;
  const int tiles = ceil_div(a.M, bm) * ceil_div(a.Cout, bn);
  if (tiles > kNumCU / 2) return 0;
  int want = ceil_div(2 * kNumCU, tiles);
  if (want > a.K / 128) want = a.K / 128;     // at least four K-tiles per workgroup
  if (want < 2) return 0;
  const int ksplit = ceil_div(ceil_div(a.K, want), BK) * BK;
  return ceil_div(a.K, ksplit) >= 2 ? ksplit : 0;
}
}  // namespace

extern "C" int dadet_amax(const float* x, long long n, float* slot, void* stream) {
  DADET_REQUIRE(n >= 0 && slot && (n == 0 || (x && al16(x))), "amax: bad arguments");
  if (n == 0) return DADET_OK;
  return launch_amax(x, n, reinterpret_cast<unsigned*>(slot), as_stream(stream));
}

extern "C" int dadet_amax_batch(const dadet_amax_item* items_dev, int n, int total_blocks, void* stream) {
  DADET_REQUIRE(n >= 0 && (n == 0 || (items_dev && total_blocks > 0)), "amax_batch: bad arguments");
  if (n == 0) return DADET_OK;
  hipLaunchKernelGGL(amax_batch_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream), items_dev, n);
  return check_launch("amax_batch");
}

static int conv_forward_impl(const dadet_conv_desc* d, const float* x, const float* w,
                             const float* scale, const float* bias, const float* addend,
                             const float* mask_ref, float* y, const float* amax_x, const float* amax_w,
                             float* amax_y, void* stream) {
  int rc = conv_desc_check(d, "conv_forward");
  if (rc) return rc;
  if (d->N == 0) return DADET_OK;
  DADET_REQUIRE(x && w && y && al16(x) && al16(w), "conv_forward: x / w must be non-null and 16-byte aligned");
  DADET_REQUIRE(d->relu_mode >= 0 && d->relu_mode <= 2, "conv_forward: relu_mode");
  DADET_REQUIRE(d->relu_mode != 2 || mask_ref, "conv_forward: relu_mode 2 needs mask_ref");
  const int os = d->out_spatial_stride > 0 ? d->out_spatial_stride : 1;
  DADET_REQUIRE(os == 1 ? (d->OutH == d->Ho && d->OutW == d->Wo)
                        : ((d->Ho - 1) * os < d->OutH && (d->Wo - 1) * os < d->OutW),
                "conv_forward: OutH/OutW inconsistent with Ho/Wo and out_spatial_stride");
  ConvArgs a;
  a.x = x; a.w = w; a.scale = scale; a.bias = bias; a.addend = addend; a.mask_ref = mask_ref; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
  a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo; a.OutH = d->OutH; a.OutW = d->OutW;
  a.os = os; a.relu_mode = d->relu_mode;
  a.M = d->N * d->Ho * d->Wo;
  a.K = d->KH * d->KW * d->Cin;
  const uint64_t xb = (uint64_t)d->N * d->H * d->W * d->Cin * 4, wb = (uint64_t)d->Cout * a.K * 4,
                 yb = (uint64_t)d->N * d->OutH * d->OutW * d->Cout * 4;
  DADET_REQUIRE(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull && yb < 0xFFFFFFF0ull,
                "conv_forward: tensors of 4 GB or more are not addressable through one buffer descriptor");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
  {
    static const int ablate = getenv("DADET_ABLATE") ? atoi(getenv("DADET_ABLATE")) : 0;
    a.ablate = ablate;
    a.ws_panel0 = 0;
  }
  hipStream_t st = as_stream(stream);
  {
    const char* env = getenv("DADET_EPILOGUE_V4");      // 0: the 4-byte epilogue (A/B runs, bit-identity test)
    a.epi_v4 = !(env && env[0] == '0') && os == 1 && d->Cout % 4 == 0 && al16(y) && (!addend || al16(addend)) &&
               (!mask_ref || al16(mask_ref)) && (!scale || al16(scale)) && (!bias || al16(bias));
  }
  a.ksplit = 0;
  a.split_stride = 0;
  a.sk_dp_tiles = a.sk_tiles = a.sk_units = a.sk_iters = a.sk_max_parts = 0;
  a.sk_ws = nullptr;
  a.sk_counters = nullptr;
  a.amax_x = a.amax_w = nullptr;
  a.amax_y = nullptr;
  a.nf_flag = nullptr;
  a.launch_id = 0;
  a.big_splits = 0;
  if (gemm_mode() == 4) {
    // operand maxima: the caller's slots, or (plain entry point) two per-stream slots filled here
    if (!amax_x || !amax_w) {
      unsigned* own = stream_amax_slots(st);
      if (!own) { set_error("conv_forward: could not allocate the operand-maximum slots"); return DADET_ELAUNCH; }
      if (zero_slots(own, 2, st) != hipSuccess) return check_launch("conv_forward(amax memset)");
      if (!amax_x) {
        rc = launch_amax(x, (int64_t)(xb / 4), own, st);
        if (rc) return rc;
        amax_x = reinterpret_cast<const float*>(own);
      }
      if (!amax_w) {
        rc = launch_amax(w, (int64_t)(wb / 4), own + 1, st);
        if (rc) return rc;
        amax_w = reinterpret_cast<const float*>(own + 1);
      }
    }
    a.amax_x = amax_x; a.amax_w = amax_w;
    a.amax_y = reinterpret_cast<unsigned*>(amax_y);
    a.nf_flag = nf_flag_ptr();
    a.launch_id = a.nf_flag ? nf_next_launch("conv_forward", a.M, a.Cout, a.K, a.KH) : 0;
  }   // (the other modes neither read nor leave maxima)
  if (gemm_mode() >= 3 && ws_eligible(a)) return launch_fwd_ws(a, gemm_mode(), st);   // weight-stationary 1x1, K <= 256
  a.big_splits = 0;
  if (gemm_mode() == 4 && big_eligible(a)) {                                           // 256 x 256 tiles, long K
    const size_t ws_bytes = big_workspace_bytes(a);
    float* ws = ws_bytes ? static_cast<float*>(stream_scratch(st, ws_bytes)) : nullptr;
    int* counters = ws_bytes ? stream_counters(st) : nullptr;
    if (ws_bytes && (!ws || !counters)) {
      set_error("conv_forward: could not allocate %zu bytes of split-reduction scratch", ws_bytes);
      return DADET_ELAUNCH;
    }
    return launch_fwd_big(a, st, ws, counters);
  }
  if (gemm_mode() != 0) {
    const int variant = split_fwd_variant(a.M, a.Cout, a.K);
    SkPlan sk;
    if (streamk_plan(a, variant, &sk)) {
      // partial tiles in the per-stream scratch (reused in stream order), arrival counters in their own buffer
      const size_t ws_bytes = sizeof(float) * (size_t)sk.sk_tiles * sk.max_parts * 128 * 128;
      a.sk_ws = static_cast<float*>(stream_scratch(st, ws_bytes));
      a.sk_counters = stream_counters(st);
      if (!a.sk_ws || !a.sk_counters) {
        set_error("conv_forward: could not allocate %zu bytes of stream-K scratch", ws_bytes);
        return DADET_ELAUNCH;
      }
      a.sk_dp_tiles = sk.dp_tiles; a.sk_tiles = sk.sk_tiles; a.sk_units = sk.units; a.sk_iters = sk.iters;
      a.sk_max_parts = sk.max_parts;
      return launch_fwd_split_sk(a, gemm_mode(), st);
    }
    const int ksplit = splitk_plan(a, variant);
    if (ksplit && al16(y) && (!addend || al16(addend)) && (!mask_ref || al16(mask_ref)) &&
        (!scale || al16(scale)) && (!bias || al16(bias))) {
      const int splits = ceil_div(a.K, ksplit);
      const size_t per = (size_t)a.M * a.Cout;
      float* ws = static_cast<float*>(stream_scratch(st, sizeof(float) * per * splits));
      if (!ws) {
        set_error("conv_forward: could not allocate %zu bytes of split-K scratch", sizeof(float) * per * splits);
        return DADET_ELAUNCH;
      }
      ConvArgs p = a;
      p.scale = p.bias = p.addend = p.mask_ref = nullptr;
      p.amax_y = nullptr;          // the reduce pass sees the final values
      p.relu_mode = 0;
      p.y = ws;
      p.ksplit = ksplit;
      p.split_stride = (unsigned)per;
      rc = launch_fwd_split(p, variant, gemm_mode(), st);
      if (rc) return rc;
      const int64_t total4 = (int64_t)per / 4;
      int64_t blocks = ceil_div64(total4, 256);
      if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, st,
                         reinterpret_cast<const float4*>(ws), splits, per / 4,
                         reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(bias),
                         reinterpret_cast<const float4*>(addend), reinterpret_cast<const float4*>(mask_ref),
                         reinterpret_cast<float4*>(y), total4, a.Cout / 4, a.relu_mode, a.amax_y);
      return check_launch("conv_forward(split-K reduce)");
    }
    return launch_fwd_split(a, variant, gemm_mode(), st);
  }
  switch (fwd_variant(a.M, a.Cout)) {
    case 0: return launch_fwd<2, 2>(a, st);
    case 1: return launch_fwd<2, 1>(a, st);
    default: return launch_fwd<1, 1>(a, st);
  }
}

extern "C" int dadet_conv_forward(const dadet_conv_desc* d, const float* x, const float* w,
                                  const float* scale, const float* bias, const float* addend,
                                  const float* mask_ref, float* y, void* stream) {
  return conv_forward_impl(d, x, w, scale, bias, addend, mask_ref, y, nullptr, nullptr, nullptr, stream);
}

extern "C" int dadet_conv_forward_scaled(const dadet_conv_desc* d, const float* x, const float* w,
                                         const float* scale, const float* bias, const float* addend,
                                         const float* mask_ref, float* y, const float* amax_x, const float* amax_w,
                                         float* amax_y, void* stream) {
  return conv_forward_impl(d, x, w, scale, bias, addend, mask_ref, y, amax_x, amax_w, amax_y, stream);
}

extern "C" int dadet_conv_forward_variant(const dadet_conv_desc* d) {
  if (!d) return -1;
  const int M = d->N * d->Ho * d->Wo;
  if (gemm_mode() >= 3) {      // 3: the weight-stationary 1x1 kernel (assuming 16-byte aligned tensors, as torch allocates them)
    ConvArgs a;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.os = d->out_spatial_stride > 0 ? d->out_spatial_stride : 1;
    a.ksplit = 0; a.K = d->KH * d->KW * d->Cin; a.Cout = d->Cout; a.M = M;
    a.epi_v4 = a.os == 1 && d->Cout % 4 == 0;
    a.x_bytes = (unsigned)((uint64_t)d->N * d->H * d->W * d->Cin * 4 > 0x7FFFFFFFull ? 0x80000000u : (uint64_t)d->N * d->H * d->W * d->Cin * 4);
    a.w_bytes = 0;
    if (ws_eligible(a)) return 3;
    if (gemm_mode() == 4) {    // 4: the 256 x 256-tile kernel (conv_big.hip)
      a.Cin = d->Cin;
      const uint64_t xb = (uint64_t)d->N * d->H * d->W * d->Cin * 4, wb = (uint64_t)d->Cout * a.K * 4,
                     yb = (uint64_t)d->N * d->OutH * d->OutW * d->Cout * 4;
      a.x_bytes = xb > 0x7FFFFFFFull ? 0x80000000u : (unsigned)xb;
      a.w_bytes = wb > 0x7FFFFFFFull ? 0x80000000u : (unsigned)wb;
      a.y_bytes = yb > 0x7FFFFFFFull ? 0x80000000u : (unsigned)yb;
      if (const int bv = big_variant(a)) return 3 + bv;     // 4: 256 x 256 tile, 5: 256 x 128 tile
    }
  }
  return gemm_mode() != 0 ? split_fwd_variant(M, d->Cout, d->KH * d->KW * d->Cin) : fwd_variant(M, d->Cout);
}

extern "C" int dadet_nonfinite_poll(char* msg, int cap) {
  unsigned w[2] = {0, 0};
  unsigned* dev = nf_flag_ptr();
  if (!dev) { if (msg && cap > 0) msg[0] = 0; return 0; }
  // every stream of the process first (PyTorch's side streams and the weight-gradient lane are non-blocking: the null
  // stream does not order against them), then one exchange kernel, then its two words
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  hipLaunchKernelGGL(nf_take_kernel, dim3(1), dim3(1), 0, 0);
  if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_nf_taken), sizeof(w), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (w[1] == 0) { if (msg && cap > 0) msg[0] = 0; return 0; }
  if (msg && cap > 0) {
    const unsigned id = w[0] - 1u;
    const NfRecord& r = g_nf_ring[id % kNfRing];
    if (w[0] != 0 && r.id == id)
      snprintf(msg, cap, "%s launch #%u (M=%d N=%d K=%d, %dx%d taps): non-finite sums in %u wavefront(s) — an operand's "
               "largest-magnitude slot below its data overflows the fp16 split (contraction mode 4)", r.kind, id, r.M, r.N,
               r.K, r.KH, r.KH, w[1]);
    else
      snprintf(msg, cap, "GEMM launch #%u: non-finite sums in %u wavefront(s) (launch record no longer in the ring)", id, w[1]);
  }
  return (int)w[1];
}

extern "C" int dadet_conv_wgrad_variant(const dadet_conv_desc* d) {
  if (!d) return -1;
  int tco, tkc, splits, rps;
  return wgrad_big_plan(d, &tco, &tkc, &splits, &rps) ? 1 : 0;
}

// Split plan of the weight gradient: the (co tile, kc tile) grid is small (4 ... 576 tiles), so the reduction over the
// M = N*Ho*Wo rows is cut into `splits` ranges to fill the 2 x 256 workgroup slots of the chip.  The number of
// workgroups matters in steps of 512: one more than a multiple of 512 costs a whole extra pass of mostly idle CUs
// (tools/wgrad_sweep.py: res5 3x3, 144 tiles: 3 splits = 432 workgroups 0.80 ms, 4 splits = 576 workgroups 0.99 ms,
// 7 splits = 1008 workgroups 0.73 ms).  The plan minimises a small cost model fitted to that sweep, in microseconds:
// passes x (fixed + K-steps x step time) + the reduction pass over the partial results.
static void wgrad_plan(const dadet_conv_desc* d, int* tiles_co, int* tiles_kc, int* splits, int* rps) {
  const int M = d->N * d->Ho * d->Wo, K = d->KH * d->KW * d->Cin;
  *tiles_co = ceil_div(d->Cout, 128);
  *tiles_kc = ceil_div(K, 128);
  const int tiles = (*tiles_co) * (*tiles_kc);
  int min_rows = 128;                           // at least 4 K-steps per split
  if (const char* e = getenv("DADET_WGRAD_MIN_ROWS")) { int v = atoi(e); if (v >= 32) min_rows = v; }
  const int max_splits = ceil_div(M, min_rows);
  const int slots = 2 * kNumCU;                 // two workgroups per CU
  const double kStep2 = 3.0, kStep1 = 2.0;      // one 32-row K-step with two / one workgroup(s) on the CU
  const double kFixed = 9.0;                    // prologue + epilogue of a workgroup
  const double dw_bytes = 4.0 * d->Cout * (double)K;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= max_splits && (s == 1 || (long)tiles * s <= 8 * slots); ++s) {
    int rows = ceil_div(ceil_div(M, s), 32) * 32;
    if (ceil_div(M, rows) != s) continue;       // same plan as a smaller s
    const double steps = rows / 32.0;
    const long wgs = (long)tiles * s;
    const long full = wgs / slots, rem = wgs % slots;
    double cost = full * (kFixed + steps * kStep2);
    if (rem > 0) {
      const double step = rem <= slots / 2 ? kStep1 : kStep1 + (kStep2 - kStep1) * (rem - slots / 2) / (slots / 2);
      cost += kFixed + steps * step;
    }
    if (s > 1) cost += 5.0 + (s + 1) * dw_bytes / 3.0e6;   // reduction pass: launch + (s reads + 1 write) at 3 TB/s
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  if (const char* e = getenv("DADET_WGRAD_SPLITS")) { int v = atoi(e); if (v > 0) best = v < max_splits ? v : max_splits; }
  int rows = ceil_div(ceil_div(M, best), 32) * 32;
  *rps = rows;
  *splits = ceil_div(M, rows);
}

extern "C" int dadet_conv_wgrad_workspace_bytes(const dadet_conv_desc* d, size_t* bytes_out) {
  int rc = conv_desc_check(d, "conv_wgrad_workspace_bytes");
  if (rc) return rc;
  DADET_REQUIRE(bytes_out, "conv_wgrad_workspace_bytes: null out");
  if (d->N == 0) { *bytes_out = 0; return DADET_OK; }
  // The query does not know gy's row pitch, and conv_wgrad_impl leaves the 256 x 256 kernel for padded rows (gy_ld !=
  // Cout): the answer is the LARGER of the two plans' needs, so that whichever kernel runs finds its space
  int tco, tkc, splits, rps;
  size_t big = 0;
  if (wgrad_big_plan(d, &tco, &tkc, &splits, &rps))       // 256 x 256 tiles: dense [splits][Cout][K] partial sums
    big = splits > 1 ? sizeof(float) * (size_t)splits * d->Cout * d->KH * d->KW * d->Cin : 0;
  wgrad_plan(d, &tco, &tkc, &splits, &rps);
  // (rounded up to whole 128 x 128 tiles)
  const size_t small = splits > 1 ? sizeof(float) * (size_t)splits * tco * tkc * 128 * 128 : 0;
  *bytes_out = big > small ? big : small;
  return DADET_OK;
}

static int conv_wgrad_impl(const dadet_conv_desc* d, const float* x, const float* gy, const float* out_scale, float* dw,
                           int accumulate, void* workspace, size_t workspace_bytes, dadet_wgrad_pending* pending,
                           void* stream, int gy_ld = 0, const float* amax_x = nullptr, const float* amax_gy = nullptr) {
  if (pending) pending->splits = 0;
  int rc = conv_desc_check(d, "conv_wgrad");
  if (rc) return rc;
  DADET_REQUIRE(dw, "conv_wgrad: null dw");
  hipStream_t st = as_stream(stream);
  const int K = d->KH * d->KW * d->Cin;
  if (d->N == 0) {
    if (!accumulate) (void)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->Cout * K, st);
    return check_launch("conv_wgrad(empty)");
  }
  DADET_REQUIRE(x && gy && al16(x) && al16(gy) && al16(dw), "conv_wgrad: pointers must be 16-byte aligned");
  // gy_ld: rows of gy padded to a multiple of four channels (the offset branch of a deformable block: 18 / 27 channels in
  // rows of 20 / 28) — the padding columns are read with the last channel quad and never stored
  if (gy_ld == 0) gy_ld = d->Cout;
  DADET_REQUIRE(gy_ld % 4 == 0 && gy_ld >= d->Cout && gy_ld - d->Cout < 4,
                "conv_wgrad: gy rows of %d floats for Cout=%d (need a multiple of 4, less than 4 above Cout)", gy_ld, d->Cout);
  DADET_REQUIRE(K % 4 == 0, "conv_wgrad: KH*KW*Cin=%d must be a multiple of 4", K);
  WgradArgs a;
  a.gy_ld = gy_ld;
  a.x = x; a.gy = gy; a.out_scale = out_scale;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
  a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
  a.M = d->N * d->Ho * d->Wo; a.K = K;
  const uint64_t xb = (uint64_t)d->N * d->H * d->W * d->Cin * 4, gb = (uint64_t)a.M * gy_ld * 4;
  DADET_REQUIRE(xb < 0xFFFFFFF0ull && gb < 0xFFFFFFF0ull,
                "conv_wgrad: tensors of 4 GB or more are not addressable through one buffer descriptor");
  a.x_bytes = (unsigned)xb; a.gy_bytes = (unsigned)gb;
  const bool big = gy_ld == d->Cout && wgrad_big_plan(d, &a.tiles_co, &a.tiles_kc, &a.splits, &a.rows_per_split);
  if (!big) wgrad_plan(d, &a.tiles_co, &a.tiles_kc, &a.splits, &a.rows_per_split);
  a.accumulate = accumulate;
  a.amax_x = a.amax_gy = nullptr;
  a.nf_flag = nullptr;
  a.launch_id = 0;
  if (gemm_mode() == 4) {
    if (!amax_x || !amax_gy) {
      unsigned* own = stream_amax_slots(st);
      if (!own) { set_error("conv_wgrad: could not allocate the operand-maximum slots"); return DADET_ELAUNCH; }
      if (zero_slots(own + 2, 2, st) != hipSuccess) return check_launch("conv_wgrad(amax memset)");
      if (!amax_x) {
        rc = launch_amax(x, (int64_t)(xb / 4), own + 2, st);
        if (rc) return rc;
        amax_x = reinterpret_cast<const float*>(own + 2);
      }
      if (!amax_gy) {
        rc = launch_amax(gy, (int64_t)(gb / 4), own + 3, st);
        if (rc) return rc;
        amax_gy = reinterpret_cast<const float*>(own + 3);
      }
    }
    a.amax_x = amax_x; a.amax_gy = amax_gy;
    a.nf_flag = nf_flag_ptr();
    a.launch_id = a.nf_flag ? nf_next_launch("conv_wgrad", a.M, a.Cout, a.K, a.KH) : 0;
  }
  if (a.splits == 1) {
    a.direct = 1;
    a.out = dw;
  } else {
    const size_t need = big ? sizeof(float) * (size_t)a.splits * d->Cout * K
                            : sizeof(float) * (size_t)a.splits * a.tiles_co * a.tiles_kc * 128 * 128;
    if (!workspace || workspace_bytes < need) {
      set_error("conv_wgrad: workspace %zu < required %zu", workspace_bytes, need);
      return DADET_EWORKSPACE;
    }
    a.direct = 0;
    a.out = static_cast<float*>(workspace);
  }
  // (measured and removed, round 2: the last-arriving split of a tile summing the partials inside the GEMM kernel — 32.1 ms
  // against 28.8 ms per step: every one of the ~800 workgroups of a launch had to publish its 64 KB tile write-through, where
  // the separate pass reads partials that mostly still sit in L2 / MALL)
  const size_t lds = sizeof(float) * 2 * 32 * 128;  // 32 KB: three workgroups per CU
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  if (big) {
    rc = launch_wgrad_big(a, st);
  } else if (gemm_mode() != 0) {
    rc = launch_wgrad_split(a, gemm_mode(), st);
  } else {
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(a.tiles_co * a.tiles_kc, a.splits), dim3(256), lds, st, a);
    rc = check_launch("conv_wgrad");
  }
  if (rc) return rc;
  if (a.splits > 1 && pending) {      // the caller batches the reduction passes
    pending->partials = static_cast<const float*>(workspace);
    pending->out_scale = out_scale;
    pending->dw = dw;
    pending->count = (long long)d->Cout * K;
    pending->K = K;
    pending->splits = a.splits;
    pending->accumulate = accumulate;
    return DADET_OK;
  }
  if (a.splits > 1) {
    const int64_t total4 = (int64_t)d->Cout * K / 4;
    int64_t blocks = ceil_div64(total4, 256);
    if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(workspace), out_scale, reinterpret_cast<float4*>(dw),
                       total4, K / 4, a.splits, accumulate);
    rc = check_launch("conv_wgrad(reduce)");
  }
  return rc;
}

extern "C" int dadet_conv_wgrad(const dadet_conv_desc* d, const float* x, const float* gy,
                                const float* out_scale, float* dw, int accumulate, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return conv_wgrad_impl(d, x, gy, out_scale, dw, accumulate, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int dadet_conv_wgrad_partials(const dadet_conv_desc* d, const float* x, const float* gy,
                                         const float* out_scale, float* dw, int accumulate, void* workspace,
                                         size_t workspace_bytes, dadet_wgrad_pending* pending_out, void* stream) {
  DADET_REQUIRE(pending_out, "conv_wgrad_partials: null pending_out");
  return conv_wgrad_impl(d, x, gy, out_scale, dw, accumulate, workspace, workspace_bytes, pending_out, stream);
}

extern "C" int dadet_conv_wgrad_partials_ld(const dadet_conv_desc* d, const float* x, const float* gy, int gy_ld,
                                            const float* out_scale, float* dw, int accumulate, void* workspace,
                                            size_t workspace_bytes, dadet_wgrad_pending* pending_out, void* stream) {
  DADET_REQUIRE(pending_out, "conv_wgrad_partials_ld: null pending_out");
  return conv_wgrad_impl(d, x, gy, out_scale, dw, accumulate, workspace, workspace_bytes, pending_out, stream, gy_ld);
}

extern "C" int dadet_conv_wgrad_scaled(const dadet_conv_desc* d, const float* x, const float* gy, int gy_ld,
                                       const float* out_scale, float* dw, int accumulate, void* workspace,
                                       size_t workspace_bytes, dadet_wgrad_pending* pending_out, const float* amax_x,
                                       const float* amax_gy, void* stream) {
  return conv_wgrad_impl(d, x, gy, out_scale, dw, accumulate, workspace, workspace_bytes, pending_out, stream, gy_ld,
                         amax_x, amax_gy);
}

// ---- several weight gradients in one launch (conv_big.hip: conv_wgrad_big_group_kernel; conv_split.hip: its 128 x 128 form)
// which kernel serves the whole group: 256 (every problem qualifies for the 256 x 256 tile), 128 (contraction mode 4, every
// problem on the 128 x 128 kernel's ordinary path and on the same side of its small-map switch), 0 (no grouped launch)
static int wgrad_group_kind(const dadet_conv_desc* descs, const int n) {
  if (n < 1 || n > kWgradGroupMax || gemm_mode() != 4) return 0;
  bool big = true, small = true;
  for (int i = 0; i < n; ++i) {
    const dadet_conv_desc& d = descs[i];
    if (conv_desc_check(&d, "conv_wgrad_group") || d.N == 0) return 0;
    const int M = d.N * d.Ho * d.Wo, K = d.KH * d.KW * d.Cin;
    if (K % 4 != 0 || d.Cout % 4 != 0 || M < 128) return 0;
    if ((uint64_t)d.N * d.H * d.W * d.Cin * 4 >= 0x7FFFFF00ull || (uint64_t)M * d.Cout * 4 >= 0x7FFFFF00ull) return 0;
    big = big && wgrad_group_member(&d);
    small = small && (d.Wo < 32) == (descs[0].Wo < 32);
  }
  static const bool small_on = !(getenv("DADET_WGRAD_GROUP_128") && getenv("DADET_WGRAD_GROUP_128")[0] == '0');
  return big ? 256 : (small && small_on ? 128 : 0);
}

extern "C" int dadet_conv_wgrad_group_plan(const dadet_conv_desc* descs, int n, int* splits_out, size_t* workspace_bytes_out) {
  DADET_REQUIRE(descs && n >= 1 && splits_out && workspace_bytes_out, "conv_wgrad_group_plan: bad arguments");
  const int kind = wgrad_group_kind(descs, n);
  if (!kind) return 0;
  int tco[kWgradGroupMax], tkc[kWgradGroupMax], rows;
  wgrad_group_plan(n, descs, kind, tco, tkc, splits_out, &rows);
  for (int i = 0; i < n; ++i)
    workspace_bytes_out[i] = splits_out[i] > 1 ? sizeof(float) * (size_t)splits_out[i] * descs[i].Cout * descs[i].KH *
                                                      descs[i].KW * descs[i].Cin
                                                : 0;
  return kind;
}

extern "C" int dadet_conv_wgrad_group(const dadet_conv_desc* descs, int n, const float* const* x, const float* const* gy,
                                      const float* const* out_scale, float* const* dw, const int* accumulate,
                                      void* const* workspace, const size_t* workspace_bytes,
                                      dadet_wgrad_pending* pending_out, const float* const* amax_x,
                                      const float* const* amax_gy, void* stream) {
  DADET_REQUIRE(descs && n >= 1 && n <= kWgradGroupMax && x && gy && dw && accumulate && workspace && workspace_bytes &&
                    pending_out && amax_x && amax_gy, "conv_wgrad_group: bad arguments (1 - 4 problems, every array non-null)");
  DADET_REQUIRE(gemm_mode() == 4, "conv_wgrad_group: contraction mode 4 only (mode %d is set)", gemm_mode());
  hipStream_t st = as_stream(stream);
  const int kind = wgrad_group_kind(descs, n);
  DADET_REQUIRE(kind != 0, "conv_wgrad_group: these problems do not form a grouped launch (dadet_conv_wgrad_group_plan)");
  int tco[kWgradGroupMax], tkc[kWgradGroupMax], splits[kWgradGroupMax], rows;
  for (int i = 0; i < n; ++i) {
    DADET_REQUIRE(x[i] && gy[i] && dw[i] && amax_x[i] && amax_gy[i] && al16(x[i]) && al16(gy[i]) && al16(dw[i]),
                  "conv_wgrad_group: problem %d: null or misaligned pointer", i);
    for (int j = 0; j < i; ++j)
      DADET_REQUIRE(dw[i] != dw[j], "conv_wgrad_group: problems %d and %d write the same dw", j, i);
  }
  wgrad_group_plan(n, descs, kind, tco, tkc, splits, &rows);
  WgradArgs a[kWgradGroupMax];
  for (int i = 0; i < n; ++i) {
    const dadet_conv_desc* d = &descs[i];
    WgradArgs& w = a[i];
    w.gy_ld = d->Cout;
    w.x = x[i]; w.gy = gy[i]; w.out_scale = out_scale ? out_scale[i] : nullptr;
    w.N = d->N; w.H = d->H; w.W = d->W; w.Cin = d->Cin; w.Cout = d->Cout; w.KH = d->KH; w.KW = d->KW;
    w.stride = d->stride; w.pad = d->pad; w.Ho = d->Ho; w.Wo = d->Wo;
    w.M = d->N * d->Ho * d->Wo; w.K = d->KH * d->KW * d->Cin;
    w.x_bytes = (unsigned)((uint64_t)d->N * d->H * d->W * d->Cin * 4);
    w.gy_bytes = (unsigned)((uint64_t)w.M * d->Cout * 4);
    w.tiles_co = tco[i]; w.tiles_kc = tkc[i]; w.splits = splits[i]; w.rows_per_split = rows;
    w.accumulate = accumulate[i];
    w.amax_x = amax_x[i]; w.amax_gy = amax_gy[i];
    w.nf_flag = nf_flag_ptr();
    w.launch_id = w.nf_flag ? nf_next_launch("conv_wgrad_group", w.M, w.Cout, w.K, w.KH) : 0;
    pending_out[i].splits = 0;
    if (splits[i] == 1) {
      w.direct = 1;
      w.out = dw[i];
    } else {
      const size_t need = sizeof(float) * (size_t)splits[i] * d->Cout * w.K;
      if (!workspace[i] || workspace_bytes[i] < need) {
        set_error("conv_wgrad_group: problem %d: workspace %zu < required %zu", i, workspace_bytes[i], need);
        return DADET_EWORKSPACE;
      }
      w.direct = 0;
      w.out = static_cast<float*>(workspace[i]);
    }
  }
  int rc = kind == 256 ? launch_wgrad_big_group(a, n, st) : launch_wgrad_split_group(a, n, st);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    if (splits[i] <= 1) continue;
    pending_out[i].partials = static_cast<const float*>(workspace[i]);
    pending_out[i].out_scale = out_scale ? out_scale[i] : nullptr;
    pending_out[i].dw = dw[i];
    pending_out[i].count = (long long)descs[i].Cout * a[i].K;
    pending_out[i].K = a[i].K;
    pending_out[i].splits = splits[i];
    pending_out[i].accumulate = accumulate[i];
  }
  return DADET_OK;
}

extern "C" int dadet_conv_wgrad_reduce_batch(const dadet_wgrad_pending* items, int n, void* stream) {
  DADET_REQUIRE(n >= 0 && (n == 0 || items), "conv_wgrad_reduce_batch: bad arguments");
  hipStream_t st = as_stream(stream);
  for (int base = 0; base < n; base += kReduceBatch) {
    ReduceBatch b;
    b.n = 0;
    int blocks_total = 0;
    for (int i = base; i < n && b.n < kReduceBatch; ++i) {
      const dadet_wgrad_pending& p = items[i];
      if (p.splits <= 1) continue;       // nothing pending for this one (splits == 1 wrote dw itself)
      DADET_REQUIRE(p.partials && p.dw && p.count > 0 && p.count % 4 == 0 && p.K > 0 && p.K % 4 == 0,
                    "conv_wgrad_reduce_batch: item %d is malformed", i);
      ReduceItem& it = b.item[b.n++];
      it.part = reinterpret_cast<const float4*>(p.partials);
      it.out_scale = p.out_scale;
      it.dw = reinterpret_cast<float4*>(p.dw);
      it.total4 = p.count / 4;
      it.K4 = p.K / 4;
      it.splits = p.splits;
      it.accumulate = p.accumulate;
      it.first_block = blocks_total;
      int64_t blocks = ceil_div64(it.total4, 256);
      if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
      blocks_total += (int)blocks;
    }
    if (b.n == 0) continue;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks_total), dim3(256), 0, st, b);
    int rc = check_launch("conv_wgrad_reduce_batch");
    if (rc) return rc;
  }
  return DADET_OK;
}

static int weight_transpose_impl(const float* w, const float* scale, float* wt, int Cout, int KH, int KW, int Cin,
                                 int cout_pad, void* stream) {
  DADET_REQUIRE(w && wt && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "conv_weight_transpose: bad args");
  DADET_REQUIRE(KH * KW <= 65535, "conv_weight_transpose: kernel too large");
  DADET_REQUIRE(cout_pad >= Cout, "conv_weight_transpose: cout_pad=%d < Cout=%d", cout_pad, Cout);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(ceil_div(Cin, 32), ceil_div(cout_pad, 32), KH * KW),
                     dim3(256), 0, as_stream(stream), w, scale, wt, Cout, KH, KW, Cin, cout_pad);
  return check_launch("conv_weight_transpose");
}

extern "C" int dadet_conv_weight_transpose(const float* w, const float* scale, float* wt, int Cout, int KH,
                                           int KW, int Cin, void* stream) {
  return weight_transpose_impl(w, scale, wt, Cout, KH, KW, Cin, Cout, stream);
}

extern "C" int dadet_conv_weight_transpose_padded(const float* w, const float* scale, float* wt, int Cout, int KH,
                                                  int KW, int Cin, int cout_pad, void* stream) {
  return weight_transpose_impl(w, scale, wt, Cout, KH, KW, Cin, cout_pad, stream);
}

extern "C" int dadet_conv_weight_transpose_batch(const dadet_transpose_item* items_dev, int n, int total_blocks,
                                                 void* stream) {
  static_assert(sizeof(dadet_transpose_item) == sizeof(dadet::TransposeItem), "transpose item layout");
  DADET_REQUIRE(n >= 0 && total_blocks >= 0, "conv_weight_transpose_batch: bad args");
  if (n == 0 || total_blocks == 0) return DADET_OK;
  DADET_REQUIRE(items_dev, "conv_weight_transpose_batch: null table");
  hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const dadet::TransposeItem*>(items_dev), n);
  return check_launch("conv_weight_transpose_batch");
}
