// Implicit-GEMM convolution with fp32 operands contracted on the bf16 matrix pipe through an exact
// operand split (gfx950: v_mfma_f32_32x32x16_bf16 is 16x the rate of v_mfma_f32_32x32x2_f32).
//
//   x = x0 + x1 (+ x2) + eps,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (round to nearest even)
// Each term carries 8 significand bits and the residuals are formed exactly in fp32, so the 2-term split
// represents x to 2^-17 relative and the 3-term split to 2^-25 — below one fp32 ulp.  Products of bf16 values are
// exact in the MFMA's fp32 accumulator, so
//   TERMS = 3:  a*b ~= a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0      6 MFMAs per K=16, error ~2^-24 |ab|  (fp32 class)
//   TERMS = 2:  a*b ~= a0b0 + a0b1 + a1b0                           3 MFMAs per K=16, error ~2^-16 |ab|
// against 8 fp32 MFMAs (512 cycles) for the same K=16 on the fp32 pipe: 192 resp. 96 cycles of matrix time.
// The split is done once per staged element (when the K-tile goes from registers to LDS), i.e. it is amortised
// over the 128-wide reuse of the tile; the LDS image is one [rows][32 + 8 pad] bf16 plane per term (80-byte
// rows: ds_read_b128 fragment reads and ds_write_b64 staging writes are both bank-conflict free).
// Everything else — buffer-descriptor loads, tap/channel indexing, XCD-aware tile order, fused epilogue — is the
// fp32 kernel's (conv_igemm.hip).  The 3-term split is the library's DEFAULT contraction (a C-ABI consumer gets it
// without any call); dadet_set_gemm_mode(0) selects the exact-fp32 MFMA kernels instead.
#include "conv_common.h"
#include <stdlib.h>

// build-time A/B switch (a second library through DADET_LIB): 0 = fragment reads per k16 group also in mode 4
#ifndef DADET_FRAG2
#define DADET_FRAG2 1
#endif

namespace dadet {

static int g_gemm_mode = 4;
int gemm_mode() { return g_gemm_mode; }

// AB: stage-ablation mask for profiling experiments (tools/ablate.py).  It is a COMPILE-TIME parameter: as run-time
// branches the checks cut the K loop into a dozen basic blocks and the scheduler could no longer interleave the
// MFMAs with the split / LDS traffic across them.  Production launches use AB = 0.
// Vectorised form of the fused epilogue (round 2).  An MFMA 32x32 accumulator gives a lane 16 values of ONE output column
// (4 x 4 consecutive rows), so the scalar epilogue below moves 4 bytes per lane and instruction: 64 stores plus up to
// 128 loads (residual, ReLU mask) per lane for a 64 x 64 wave tile.  The short-K layers (res2 / res3 1x1 convs: K = 64 ..
// 256, two to eight K-tiles) are nothing but prologue and epilogue and ran at 3.6 - 3.9 TB/s.  Here every 32 x 32 block
// is turned through the wave's slice of the (by then idle) operand LDS — 16 ds_write_b32 into [32][40] floats (rows r
// and r + 4 of the two half-waves fall on disjoint bank halves), 4 ds_read_b128 back — so that a lane owns 4 consecutive
// COLUMNS of 4 rows and y / addend / mask move 16 bytes per lane.  Same arithmetic per element, in the same order:
// results are bit-identical to the scalar epilogue (tests/test_ops_gpu.py).  Needs Cout % 4 == 0, 16-byte aligned
// tensors, no output stride; otherwise the scalar form runs.

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_v4(const ConvArgs& a, f32x16 (&acc)[TM][TN], char* smem, const int bm0,
                                                 const int bn0, const int wm, const int wn, const int lane,
                                                 const int wave, const unsigned split_y) {
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * EPI_STRIDE);
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y + (size_t)split_y * a.split_stride, a.y_bytes);
  const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.addend ? a.addend : a.y, a.addend ? a.y_bytes : 0u);
  const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_ref ? a.mask_ref : a.y, a.mask_ref ? a.y_bytes : 0u);
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;      // read side: 8 lanes x 16 B = one 32-column row
  // Round 4: EVERY residual / gate load of the tile is issued here, before the first block is turned through LDS (the
  // K loop's staging registers are dead by now: up to 2 x 64 VGPRs for a 128 x 128 tile).  Before, each 32 x 32 block
  // issued its own loads behind its LDS writes and waited for them — four dependent HBM round trips per workgroup on every
  // data-gradient launch (they all gate), with the matrix pipe idle chip-wide on single-pass grids.
  unsigned offs[TM][TN][4];
  float4 ad[TM][TN][4], mk[TM][TN][4];
  float mx = 0.f;                                        // max|y| of what this lane stores (mode 4: a.amax_y)
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int n = bn0 + wn * TN * 32 + in * 32 + c4;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int m = bm0 + wm * TM * 32 + im * 32 + pass * 8 + rrow;
        offs[im][in][pass] = (n < a.Cout && m < a.M) ? ((unsigned)m * (unsigned)a.Cout + (unsigned)n) * 4u : kOOB;
      }
  }
  if (a.addend) {
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) ad[im][in][pass] = buf_load4(ar, offs[im][in][pass]);
  }
  if (a.relu_mode == 2) {
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) mk[im][in][pass] = buf_load4(mr, offs[im][in][pass]);
  }
  __syncthreads();                                       // every wave is done with the operand planes
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int n = bn0 + wn * TN * 32 + in * 32 + c4;
    const bool nvalid = n < a.Cout;                      // Cout % 4 == 0: the four columns are valid together
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale && nvalid) sc = *reinterpret_cast<const float4*>(a.scale + n);
    if (a.bias && nvalid) bi = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
    for (int im = 0; im < TM; ++im) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[(q + 8 * g + row_hi) * EPI_STRIDE + col_in] = acc[im][in][g * 4 + q];
      // a wave's own data only: no workgroup barrier, the LDS traffic of one wave is ordered
      __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + rrow;
        const float4 v4 = *reinterpret_cast<const float4*>(tile + row * EPI_STRIDE + c4);
        const unsigned off = offs[im][in][pass];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {bi.x, bi.y, bi.z, bi.w};
        float adv[4] = {0.f, 0.f, 0.f, 0.f}, mkv[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.addend) {
          const float4 t = ad[im][in][pass];
          adv[0] = t.x; adv[1] = t.y; adv[2] = t.z; adv[3] = t.w;
        }
        if (a.relu_mode == 2) {
          const float4 t = mk[im][in][pass];
          mkv[0] = t.x; mkv[1] = t.y; mkv[2] = t.z; mkv[3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = v[e];
          if (a.scale) x = x * s4[e];
          if (a.bias) x = x + b4[e];
          if (a.addend) x = x + adv[e];
          if (a.relu_mode == 1) x = fmaxf(x, 0.f);
          else if (a.relu_mode == 2) x = (mkv[e] > 0.f) ? x : 0.f;
          v[e] = x;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), yr,
                                               (int)off, 0, 0);
        if (a.amax_y && off != kOOB)
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
      __builtin_amdgcn_wave_barrier();                   // the tile is rewritten by the next block
    }
  }
  if (a.amax_y) amax_publish(a.amax_y, mx);
}

// fused epilogue of the forward / data-gradient GEMM (same as conv_igemm.hip): y = gate(acc * scale + bias + addend);
// a split-K launch stores raw partial sums (no scale / bias / addend / gate)
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], const int bm0, const int bn0,
                                              const int wm, const int wn, const int lane, const unsigned split_y) {
  const int HoWo = a.Ho * a.Wo;
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y + (size_t)split_y * a.split_stride, a.y_bytes);
  const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.addend ? a.addend : a.y, a.addend ? a.y_bytes : 0u);
  const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_ref ? a.mask_ref : a.y, a.mask_ref ? a.y_bytes : 0u);
  const int col_in = lane & 31;
  const int row_hi = 4 * (lane >> 5);
  float mx = 0.f;
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int n = bn0 + wn * TN * 32 + in * 32 + col_in;
    const bool nvalid = n < a.Cout;
    const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
    const float bi = (a.bias && nvalid) ? a.bias[n] : 0.f;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
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
          if (a.amax_y && offs[q] != kOOB) mx = fmaxf(mx, fabsf(v));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (a.amax_y) amax_publish(a.amax_y, mx);
}

// One output tile over the K range [k_lo, k_hi) (whole K-tiles).  sk = nullptr: the result goes through the epilogue.
// sk != nullptr (stream-K segment, see conv_fwd_split_sk_kernel): the raw partial sums are parked in the tile's workspace
// slot; the workgroup that parks a tile's LAST missing part sums all parts in part order and runs the epilogue.
struct SkPart {
  float* tile_ws;      // [parts][128 x 128] partial sums of this tile
  int* counter;        // arrivals of this tile
  int part, parts;
};

template <int TM, int TN, int FMT, int AB>
__device__ __forceinline__ void conv_fwd_split_body(const ConvArgs& a, char* smem, const int tile, const int k_lo,
                                                    const int k_hi, const unsigned split_y, const SkPart* sk) {
  constexpr int TERMS = Fmt<FMT>::terms;
  constexpr bool F16 = Fmt<FMT>::f16;
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  // mode 4: per-tensor power-of-two scales from the operands' maxima (scalar loads; see conv_common.h)
  int ea = 0, eb = 0;
  if (F16) {
    ea = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
    eb = a.amax_w ? fmt4_exp(amax_read(a.amax_w)) : 0;
  }
  const float sa = pow2f(ea), sb = pow2f(eb);
  constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;
  constexpr int A_PLANE = BM * PLANE_STRIDE, B_PLANE = BN * PLANE_STRIDE;  // bf16 elements
  __bf16* As = reinterpret_cast<__bf16*>(smem);   // [TERMS][BM][PLANE_STRIDE]
  __bf16* Bs = As + TERMS * A_PLANE;              // [TERMS][BN][PLANE_STRIDE]

  const int bm0 = (tile / a.tiles_n) * BM;
  const int bn0 = (tile % a.tiles_n) * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lcol = t & 7;
  // staged row of this thread.  Eight lanes write one row's 64 bytes; a ds_write_b64 is serviced in 16-lane
  // groups over 32 banks, and with 80-byte rows two rows are bank-disjoint exactly when they are 4 (mod 8) apart,
  // so consecutive 8-lane groups take rows r and r + 4 (PMC: 33% of LDS cycles were conflicts with r, r + 1).
  const int lgrp = t >> 3;
  const int lrow = (lgrp >> 3) * 8 + (lgrp & 1) * 4 + ((lgrp >> 1) & 3);

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, a.w_bytes);
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
      hi0[i] = -(1 << 28);
      wi0[i] = 0;
    }
  }
  unsigned wrow[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int n = bn0 + lrow + 32 * i;
    wrow[i] = n < a.Cout ? (unsigned)n * (unsigned)a.K * 4u : kOOB;
  }
  float4 ra[A_LOADS], rb[B_LOADS];
  int kk = k_lo + lcol * 4;
  int tap = kk / a.Cin;
  int kc = kk - tap * a.Cin;
  int kr = tap / a.KW;
  int ks = tap - kr * a.KW;

  // one K-tile of raw fp32 operands in registers.  A and B are fetched by separate calls so that each can be issued
  // right after ITS registers were consumed by the split (see the K loop): a fetch then has a whole K-tile of MFMAs
  // to land before it is needed.
  auto load_a = [&]() {
    const bool kvalid = kk < k_hi;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = hi0[i] + kr, wi = wi0[i] + ks;
      const bool ok = kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const unsigned off = ((unsigned)(pixbase[i] + hi * a.W + wi) * (unsigned)a.Cin + (unsigned)kc) * 4u;
      ra[i] = buf_load4(xr, ok ? off : kOOB);
    }
  };
  auto load_b = [&]() {
    const bool kvalid = kk < k_hi;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      rb[i] = buf_load4(wr, (kvalid && wrow[i] != kOOB) ? wrow[i] + (unsigned)kk * 4u : kOOB);
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
  // staged tile as packed bf16 terms: produced from ra / rb while the MFMAs of the current tile are in flight
  // (plain VALU work that hipcc interleaves with the matrix instructions), so that between the two barriers of
  // a K-tile only the ds_writes remain
  uint2 pa_[A_LOADS][TERMS], pb_[B_LOADS][TERMS];
  auto split_a = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      if constexpr (F16) split4h(ra[i], sa, pa_[i]);
      else split4<TERMS>(ra[i], pa_[i]);
    }
  };
  auto split_b = [&]() {
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      if constexpr (F16) split4h(rb[i], sb, pb_[i]);
      else split4<TERMS>(rb[i], pb_[i]);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        *reinterpret_cast<uint2*>(As + p * A_PLANE + (lrow + 32 * i) * PLANE_STRIDE + lcol * 4) = pa_[i][p];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        *reinterpret_cast<uint2*>(Bs + p * B_PLANE + (lrow + 32 * i) * PLANE_STRIDE + lcol * 4) = pb_[i][p];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (k_hi - k_lo + BK - 1) / BK;
  load_a();
  load_b();
  advance();
  split_a();
  split_b();
  store_tile();
  __syncthreads();
  if (!(AB & 4)) {   // tile 1 is in flight while tile 0 is multiplied
    load_a();
    load_b();
  }
  advance();

  // MFMA 32x32x16 bf16 fragments: lane l feeds row (l & 31), k = 8*(l >> 5) .. +7 of each 16-wide k step
  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 8;
  const __bf16* Ab = As + (wm * TM * 32 + frag_row) * PLANE_STRIDE + frag_k;
  const __bf16* Bb = Bs + (wn * TN * 32 + frag_row) * PLANE_STRIDE + frag_k;
  constexpr int ab = AB;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    // LDS holds tile kt, the registers hold tile kt + 1 (fetched one iteration ago).  Per k16 group: split one
    // operand of tile kt + 1 in the shadow of the MFMAs (each 32x32x16 occupies the matrix pipe for 32 cycles = 8 issue
    // slots), then fetch that operand of tile kt + 2 into the registers just freed.  Everything is unconditional on
    // purpose: past the last K-tile every offset is out of range (the loads return 0) and the split works on dead
    // registers — loads, MFMAs and split stay in ONE basic block for the scheduler.
    // FRAG2 (mode 4): the fragments of BOTH k16 steps of the tile are read up front.  With 12 MFMAs per group instead of
    // 24, the LDS round trip in front of each group (reads issued, then waited for) is twice as large a share of it; the
    // second group's reads now land under the first group's MFMAs (32 more VGPRs).
    constexpr bool FRAG2 = F16 && AB == 0 && DADET_FRAG2;
    bf16x8 fa_[BK / 16][TERMS][TM], fb_[BK / 16][TERMS][TN];
    auto read_frags = [&](const int step) {
#pragma unroll
      for (int p = 0; p < TERMS; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa_[step][p][i] = *reinterpret_cast<const bf16x8*>(Ab + p * A_PLANE + i * 32 * PLANE_STRIDE + step * 16);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          fb_[step][p][i] = *reinterpret_cast<const bf16x8*>(Bb + p * B_PLANE + i * 32 * PLANE_STRIDE + step * 16);
      }
    };
    if (FRAG2) {
#pragma unroll
      for (int step = 0; step < BK / 16; ++step) read_frags(step);
    }
#pragma unroll
    for (int step = 0; step < BK / 16; ++step) {
      auto& fa = fa_[step];
      auto& fb = fb_[step];
      if (!FRAG2 && (!(ab & 16) || kt == 0)) read_frags(step);
      // the next tile's operands have landed by now: split one operand per k16 group of MFMAs
      if (!(ab & 1)) {
        if (step == 0) split_a();
        else if (!(ab & 32)) split_b();
      }
      // smallest cross terms first, the leading a0*b0 last
#pragma unroll
      for (int order = 2 * (TERMS - 1); order >= 0; --order) {
#pragma unroll
        for (int pa = 0; pa < TERMS; ++pa) {
          const int pb = order - pa;
          if (pb < 0 || pb >= TERMS) continue;
          if (pa + pb > TERMS - 1) continue;  // dropped: below the split's own residual
          if (!(ab & 8))
#pragma unroll
          for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
              acc[im][in] = mfma_32x32x16<F16>(fa[pa][im], fb[pb][in], acc[im][in]);
        }
      }
      if (AB == 0) {
        // desired issue order for this k16 group: fragment reads up front, then every MFMA followed by the VALU
        // instructions that fit in its shadow
        constexpr int kMfma = TM * TN * (TERMS == 3 ? 6 : 3);
        constexpr int kValuPerMfma = (TM == 2 ? A_LOADS : B_LOADS) * (TERMS == 3 ? 26 : 14) / kMfma + 1;
        if (!FRAG2) __builtin_amdgcn_sched_group_barrier(0x100, TERMS * (TM + TN), 0);                  // DS reads
        else if (step == 0) __builtin_amdgcn_sched_group_barrier(0x100, (BK / 16) * TERMS * (TM + TN), 0);
#pragma unroll
        for (int i = 0; i < kMfma; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);      // VALU in its shadow
        }
      }
      // registers of the operand just split are free: fetch that operand of tile kt + 2
      if (!(ab & 4)) {
        if (step == 0) load_a();
        else load_b();
      }
    }
    advance();
    if (more) {
      if (!(ab & 64)) __syncthreads();
      if (!(ab & 2)) store_tile();
      if (!(ab & 64)) __syncthreads();
    }
  }

  if (F16) {
    // undo the operand scales (exact: powers of two, in two steps so that no intermediate leaves fp32's range
    // unless the result does) — partial sums of split-K / stream-K parts are parked in true units
    const int t = -(ea + eb);
    const float u1 = pow2f(t / 2), u2 = pow2f(t - t / 2);
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[im][in][e] = acc[im][in][e] * u1 * u2;
    nf_check<TM, TN>(acc, a.nf_flag, a.launch_id);
  }

  if (sk) {
    // park the partial sums: lane-linear, 16 bytes per lane and store, written through to memory
    __shared__ int s_ticket;
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(sk->tile_ws, (unsigned)(sk->parts * BM * BN * 4));
    const unsigned mine = (unsigned)sk->part * (BM * BN * 4) + (unsigned)t * 16u;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          buf_store4_wt(pr, mine + ((im * TN + in) * 4 + g) * 4096u,
                        make_float4(acc[im][in][g * 4], acc[im][in][g * 4 + 1], acc[im][in][g * 4 + 2],
                                    acc[im][in][g * 4 + 3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave drains its stores ...
    __syncthreads();                                      // ... before one lane announces the part
    if (t == 0) s_ticket = __hip_atomic_fetch_add(sk->counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != sk->parts - 1) return;
    if (t == 0) *sk->counter = 0;      // every part has arrived: ready for the next launch (no memset per launch)
    // all parts are in memory: sum them in part order (the same order whichever workgroup arrives last)
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[im][in][e] = 0.f;
    for (int p = 0; p < sk->parts; ++p) {
      const unsigned src = (unsigned)p * (BM * BN * 4) + (unsigned)t * 16u;
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = buf_load4_sc1(pr, src + ((im * TN + in) * 4 + g) * 4096u);
            acc[im][in][g * 4] += v.x; acc[im][in][g * 4 + 1] += v.y;
            acc[im][in][g * 4 + 2] += v.z; acc[im][in][g * 4 + 3] += v.w;
          }
    }
  }

  if (a.epi_v4) conv_epilogue_v4<TM, TN>(a, acc, smem, bm0, bn0, wm, wn, lane, wave, split_y);
  else conv_epilogue<TM, TN>(a, acc, bm0, bn0, wm, wn, lane, split_y);
}

// (64x64 tiles: 97 VGPRs as compiled — one above the 96 that let FIVE wavefronts share a SIMD; its 30 KB of LDS allow five
// workgroups per CU too.  The bound asks for that occupancy: the short-K layers this variant serves are all prologue and
// epilogue, and a fifth resident workgroup is one more to cover them.  DADET_FWD_OCC5=0 at BUILD time restores (256, 2).)
#ifndef DADET_FWD_OCC5
#define DADET_FWD_OCC5 1
#endif
template <int TM, int TN, int FMT, int AB = 0>
__global__ __launch_bounds__(256, (DADET_FWD_OCC5 && TM * TN == 1 && FMT >= 3 && AB == 0) ? 5 : 2)
void conv_fwd_split_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k_lo = a.ksplit ? (int)blockIdx.y * a.ksplit : 0;
  const int k_hi = a.ksplit ? min(a.K, k_lo + a.ksplit) : a.K;
  conv_fwd_split_body<TM, TN, FMT, AB>(a, smem, xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n), k_lo, k_hi, blockIdx.y,
                                         nullptr);
}

// Stream-K tail.  A grid of T output tiles runs on 512 workgroup slots (2 per CU) in ceil(T / 512) passes; the last pass
// of e.g. the res5 3x3 convs (784 tiles) leaves 240 slots idle for the length of a whole tile (tools/gemm_table.py: ~10%
// of those launches).  Here the first `sk_dp_tiles` tiles (a multiple of the slot count) are computed one per workgroup
// as before; the remaining sk_tiles * nk K-tile iterations are cut into `sk_units` equal contiguous ranges, one per
// extra workgroup, so the tail ends (T mod 512) / 512 of a pass after the full passes instead of a whole one.  A range
// covers the end of one tile and the start of the next; the parts of a tile meet in the workspace (conv_fwd_split_body).
template <int FMT>
__global__ __launch_bounds__(256, 2) void conv_fwd_split_sk_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < a.sk_dp_tiles) {
    conv_fwd_split_body<2, 2, FMT, 0>(a, smem, xcd_remap(blockIdx.x, a.sk_dp_tiles), 0, a.K, 0, nullptr);
    return;
  }
  const int nk = (a.K + BK - 1) / BK;
  // XCD-aware order of the ranges, like the tiles': hardware deals workgroup b to XCD b % 8 (sk_dp_tiles is a multiple of
  // 8), so the ranges of one XCD are made contiguous — they walk neighbouring tiles, which share operand panels in L2
  // (PMC before: 1.0 GB per launch through the fabric against ~0.3 GB algorithmic)
  const int unit = xcd_remap((int)blockIdx.x - a.sk_dp_tiles, a.sk_units);
  int it = unit * a.sk_iters;
  const int end = min(it + a.sk_iters, a.sk_tiles * nk);
  bool first = true;
  while (it < end) {
    const int tl = it / nk, k0 = it - tl * nk;
    const int k1 = min(nk, k0 + (end - it));
    if (!first) __syncthreads();      // the previous segment's fragment reads vs this segment's first LDS stores
    first = false;
    const int k_lo = k0 * BK, k_hi = min(a.K, k1 * BK);
    if (k0 == 0 && k1 == nk) {
      conv_fwd_split_body<2, 2, FMT, 0>(a, smem, a.sk_dp_tiles + tl, k_lo, k_hi, 0, nullptr);
    } else {
      const int first_unit = (tl * nk) / a.sk_iters, last_unit = ((tl + 1) * nk - 1) / a.sk_iters;
      SkPart part;
      part.tile_ws = a.sk_ws + (size_t)tl * a.sk_max_parts * (128 * 128);
      part.counter = a.sk_counters + tl;
      part.part = unit - first_unit;
      part.parts = last_unit - first_unit + 1;
      conv_fwd_split_body<2, 2, FMT, 0>(a, smem, a.sk_dp_tiles + tl, k_lo, k_hi, 0, &part);
    }
    it += k1 - k0;
  }
}

// (measured and removed, rounds 1 - 2: a double-buffered variant with K-steps of 16 — +1.7% on the RPN conv, +6% on res5
// 3x3, -2 .. -4% on the 256-tile layers; a wave-specialised producer / consumer variant with 8 wavefronts — the same as the
// 4-wave kernel on long-K layers, slower on short-K ones; a static wave priority by hardware wave slot — no effect.
// DESIGN.md section 6.)

template <int TM, int TN, int FMT, int AB = 0>
static int launch_split(ConvArgs& a, hipStream_t st) {
  constexpr int TERMS = Fmt<FMT>::terms;
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cout, BN);
  const size_t lds = sizeof(__bf16) * TERMS * (BM + BN) * PLANE_STRIDE;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_split_kernel<TM, TN, FMT, AB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_forward(split): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  const int ksplits = a.ksplit ? ceil_div(a.K, a.ksplit) : 1;
  hipLaunchKernelGGL((conv_fwd_split_kernel<TM, TN, FMT, AB>), dim3(a.tiles_m * a.tiles_n, ksplits), dim3(256), lds,
                     st, a);
  return check_launch("conv_forward(split)");
}

template <int FMT>
static int launch_split_sk(ConvArgs& a, hipStream_t st) {
  constexpr int TERMS = Fmt<FMT>::terms;
  const size_t lds = sizeof(__bf16) * TERMS * 256 * PLANE_STRIDE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_split_sk_kernel<FMT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_forward(split, stream-K): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_fwd_split_sk_kernel<FMT>), dim3(a.sk_dp_tiles + a.sk_units), dim3(256), lds, st, a);
  return check_launch("conv_forward(split, stream-K)");
}

int launch_fwd_split_sk(ConvArgs& a, int fmt, hipStream_t st) {
  a.tiles_m = ceil_div(a.M, 128);
  a.tiles_n = ceil_div(a.Cout, 128);
  return fmt == 4 ? launch_split_sk<4>(a, st) : fmt == 2 ? launch_split_sk<2>(a, st) : launch_split_sk<3>(a, st);
}

int launch_fwd_split(ConvArgs& a, int variant, int fmt, hipStream_t st) {
  const int terms = fmt;
  if (fmt == 4) {
    switch (variant) {
      case 0: return launch_split<2, 2, 4>(a, st);
      case 1: return launch_split<2, 1, 4>(a, st);
      default: return launch_split<1, 1, 4>(a, st);
    }
  }
  if (a.ablate && variant == 0 && terms == 3) {   // profiling experiments only (DADET_ABLATE)
    switch (a.ablate) {
      case 1: return launch_split<2, 2, 3, 1>(a, st);
      case 2: return launch_split<2, 2, 3, 2>(a, st);
      case 3: return launch_split<2, 2, 3, 3>(a, st);
      case 8: return launch_split<2, 2, 3, 8>(a, st);
      case 16: return launch_split<2, 2, 3, 16>(a, st);
      case 19: return launch_split<2, 2, 3, 19>(a, st);
      case 23: return launch_split<2, 2, 3, 23>(a, st);
      case 87: return launch_split<2, 2, 3, 87>(a, st);    // MFMAs only, no barriers either
      default: set_error("conv_forward(split): no kernel compiled for ablation mask %d", a.ablate); return DADET_EINVAL;
    }
  }
  if (terms == 2) {
    switch (variant) {
      case 0: return launch_split<2, 2, 2>(a, st);
      case 1: return launch_split<2, 1, 2>(a, st);
      default: return launch_split<1, 1, 2>(a, st);
    }
  }
  switch (variant) {
    case 0: return launch_split<2, 2, 3>(a, st);
    case 1: return launch_split<2, 1, 3>(a, st);
    default: return launch_split<1, 1, 3>(a, st);
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient on the split-bf16 pipe.  D[co][kc] = sum_m gY[m][co] * Xg[m][kc].  The bf16 MFMA wants each
// lane's 8 k-values (= 8 consecutive m) contiguous, but m is the slow axis of both operands in HBM, so the
// staging transposes in registers: a thread loads a 4(m) x 4(channel) block (four 16-byte row loads), splits,
// and writes for each channel the four m-values as one 8-byte ds_write into a [channel][m] plane
// ([128][32 + 8 pad] bf16 per term).  Lane -> (m-group = t % 8, channel-quad = t / 8) keeps the global loads as
// 128-byte row segments and the LDS writes bank-conflict free.
// (`bid` / `split`: tile index and part of the reduction inside the workgroup's OWN problem — blockIdx.x / blockIdx.y, or
// derived from the index inside the problem in a grouped launch)
template <int FMT, bool SMALL_MAP>
__device__ __forceinline__ void wgrad_split_body(const WgradArgs& a, const int bid, const int split,
                                                 const bool tile_is_bid = false) {
  constexpr int TERMS = Fmt<FMT>::terms;
  constexpr bool F16 = Fmt<FMT>::f16;
  constexpr int TILE = 128, RK = 32;
  int eg = 0, ex = 0;   // mode 4: per-tensor power-of-two scales of gy and x
  if (F16) {
    eg = a.amax_gy ? fmt4_exp(amax_read(a.amax_gy)) : 0;
    ex = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
  }
  const float sg = pow2f(eg), sx = pow2f(ex);
  constexpr int PLANE = TILE * PLANE_STRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* Gs = reinterpret_cast<__bf16*>(smem);  // [TERMS][128 co][PLANE_STRIDE]
  __bf16* Xs = Gs + TERMS * PLANE;               // [TERMS][128 kc][PLANE_STRIDE]

  const int tile = tile_is_bid ? bid : xcd_remap(bid, a.tiles_co * a.tiles_kc);
  const int co0 = (tile / a.tiles_kc) * TILE;
  const int kc0 = (tile % a.tiles_kc) * TILE;
  const int m_begin = split * a.rows_per_split;
  int m_end = m_begin + a.rows_per_split;
  if (m_end > a.M) m_end = a.M;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mg = t & 7;    // rows mg*4 .. mg*4+3 of the 32-row step
  const int cq = t >> 3;   // channels cq*4 .. cq*4+3 of the 128-wide tile

  const int kk = kc0 + cq * 4;
  const bool kvalid = kk < a.K;
  const int tap = kk / a.Cin;
  const int ci = kk - tap * a.Cin;
  const int r = tap / a.KW;
  const int s = tap - r * a.KW;
  const int co = co0 + cq * 4;
  const bool covalid = co < a.Cout;
  const int HoWo = a.Ho * a.Wo;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t gr = make_rsrc(a.gy, a.gy_bytes);
  const unsigned co_off = covalid ? (unsigned)co * 4u : kOOB;

  float4 rg[4], rx[4];
  int r_img[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m_begin + mg * 4 + j;
    r_img[j] = m / HoWo;
    const int rem = m - r_img[j] * HoWo;
    r_ho[j] = rem / a.Wo;
    r_wo[j] = rem - r_ho[j] * a.Wo;
  }
  const int d_img = RK / HoWo, d_ho = (RK - d_img * HoWo) / a.Wo, d_wo = RK - d_img * HoWo - d_ho * a.Wo;
  int m_cur = m_begin;
  // the two operands are fetched by separate calls so that each is issued right after ITS registers were split
  auto load_g = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m_cur + mg * 4 + j;
      rg[j] = buf_load4(gr, (m < m_end && covalid) ? (unsigned)m * (unsigned)a.gy_ld * 4u + co_off : kOOB);
    }
  };
  auto load_x = [&]() {   // also advances to the next 32-row step
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m_cur + mg * 4 + j;
      const int hi = r_ho[j] * a.stride - a.pad + r;
      const int wi = r_wo[j] * a.stride - a.pad + s;
      const bool ok = m < m_end && kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const unsigned off = ((unsigned)((r_img[j] * a.H + hi) * a.W + wi) * (unsigned)a.Cin + (unsigned)ci) * 4u;
      rx[j] = buf_load4(xr, ok ? off : kOOB);
      // advance the row by RK output pixels.  Maps narrower than RK (the 7x7 maps of the box head: 32 pixels are 4.6
      // rows): branch-free mixed-radix add of (d_img, d_ho, d_wo) — a loop over the wrap-arounds costs 16% there
      if (SMALL_MAP) {
        int wo = r_wo[j] + d_wo;
        const int cw = wo >= a.Wo;
        wo -= cw ? a.Wo : 0;
        int ho = r_ho[j] + d_ho + cw;
        const int ch = ho >= a.Ho;
        ho -= ch ? a.Ho : 0;
        r_wo[j] = wo;
        r_ho[j] = ho;
        r_img[j] += d_img + ch;
      } else {   // Wo >= RK: at most one wrap, rarely taken
        r_wo[j] += RK;
        if (r_wo[j] >= a.Wo) {
          r_wo[j] -= a.Wo;
          if (++r_ho[j] == a.Ho) {
            r_ho[j] = 0;
            ++r_img[j];
          }
        }
      }
    }
    m_cur += RK;
  };
  uint2 pg_[4][TERMS], px_[4][TERMS];
  auto split_block = [&](const float4 (&v)[4], uint2 (&out)[4][TERMS], const float sc) {
    // 4x4 register transpose: channel c of the quad gets (row0[c], row1[c], row2[c], row3[c])
    const float4 cols[4] = {make_float4(v[0].x, v[1].x, v[2].x, v[3].x), make_float4(v[0].y, v[1].y, v[2].y, v[3].y),
                            make_float4(v[0].z, v[1].z, v[2].z, v[3].z), make_float4(v[0].w, v[1].w, v[2].w, v[3].w)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (F16) split4h(cols[c], sc, out[c]);
      else split4<TERMS>(cols[c], out[c]);
    }
  };
  auto store_plane = [&](__bf16* base, const uint2 (&parts)[4][TERMS]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
        *reinterpret_cast<uint2*>(base + p * PLANE + (cq * 4 + c) * PLANE_STRIDE + mg * 4) = parts[c][p];
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
    load_g();
    load_x();
    split_block(rg, pg_, sg);
    split_block(rx, px_, sx);
    store_plane(Gs, pg_);
    store_plane(Xs, px_);
    load_g();   // step 1 is in flight while step 0 is multiplied
    load_x();
  }
  __syncthreads();
  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 8;
  const __bf16* Gb = Gs + (wm * 64 + frag_row) * PLANE_STRIDE + frag_k;
  const __bf16* Xb = Xs + (wn * 64 + frag_row) * PLANE_STRIDE + frag_k;
  for (int st = 0; st < nsteps; ++st) {
    const bool more = st + 1 < nsteps;
    // LDS holds step st, the registers hold step st + 1.  Per k16 group: split (and 4x4-transpose) one operand of
    // step st + 1 in the shadow of the MFMAs, then fetch that operand of step st + 2 into the freed registers.  All
    // unconditional (rows past m_end load zeros): loads, MFMAs and split stay in one basic block for the scheduler.
    // (mode 4: the fragments of both k16 groups are read up front — see FRAG2 in conv_fwd_split_body)
    constexpr bool FRAG2 = F16 && DADET_FRAG2;
    bf16x8 fg_[RK / 16][TERMS][2], fx_[RK / 16][TERMS][2];
    auto read_frags = [&](const int step) {
#pragma unroll
      for (int p = 0; p < TERMS; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fg_[step][p][i] = *reinterpret_cast<const bf16x8*>(Gb + p * PLANE + i * 32 * PLANE_STRIDE + step * 16);
          fx_[step][p][i] = *reinterpret_cast<const bf16x8*>(Xb + p * PLANE + i * 32 * PLANE_STRIDE + step * 16);
        }
    };
    if (FRAG2) {
#pragma unroll
      for (int step = 0; step < RK / 16; ++step) read_frags(step);
    }
#pragma unroll
    for (int step = 0; step < RK / 16; ++step) {
      auto& fg = fg_[step];
      auto& fx = fx_[step];
      if (!FRAG2) read_frags(step);
      if (step == 0) split_block(rg, pg_, sg);
      else split_block(rx, px_, sx);
#pragma unroll
      for (int order = 2 * (TERMS - 1); order >= 0; --order) {
#pragma unroll
        for (int pa = 0; pa < TERMS; ++pa) {
          const int pb = order - pa;
          if (pb < 0 || pb >= TERMS) continue;
          if (pa + pb > TERMS - 1) continue;
#pragma unroll
          for (int im = 0; im < 2; ++im)
#pragma unroll
            for (int in = 0; in < 2; ++in)
              acc[im][in] = mfma_32x32x16<F16>(fg[pa][im], fx[pb][in], acc[im][in]);
        }
      }
      {
        constexpr int kMfma = 4 * (TERMS == 3 ? 6 : 3);
        constexpr int kValuPerMfma = 4 * (TERMS == 3 ? 26 : 14) / kMfma + 1;
        if (!FRAG2) __builtin_amdgcn_sched_group_barrier(0x100, TERMS * 4, 0);          // DS reads
        else if (step == 0) __builtin_amdgcn_sched_group_barrier(0x100, (RK / 16) * TERMS * 4, 0);
#pragma unroll
        for (int i = 0; i < kMfma; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);     // VALU in its shadow
        }
      }
      if (step == 0) load_g();
      else load_x();
    }
    if (more) {
      __syncthreads();
      store_plane(Gs, pg_);
      store_plane(Xs, px_);
      __syncthreads();
    }
  }

  if (F16) {   // undo the operand scales (exact; two steps, see conv_fwd_split_body)
    const int t = -(eg + ex);
    const float u1 = pow2f(t / 2), u2 = pow2f(t - t / 2);
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[im][in][e] = acc[im][in][e] * u1 * u2;
    nf_check<2, 2>(acc, a.nf_flag, a.launch_id);
  }
  const bool final_out = a.direct;              // this workgroup writes dw itself (scale / accumulate applied here)
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
        if (final_out) {
          if (a.out_scale) v = v * a.out_scale[c];
          if (a.accumulate) v = v + out[off];
        }
        out[off] = v;
      }
  }
}

template <int FMT, bool SMALL_MAP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(const WgradArgs a) {
  wgrad_split_body<FMT, SMALL_MAP>(a, blockIdx.x, blockIdx.y);
}

// several weight gradients in one launch (conv_common.h: WgradGroup): the 2 x 256 workgroup slots shared among the
// problems, every part of every problem the same number of rows
template <int FMT, bool SMALL_MAP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_group_kernel(const WgradGroup g) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < kWgradGroupMax; ++k)
    if (k < g.n && (int)blockIdx.x >= g.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
  const int local = (int)blockIdx.x - g.first[i];
  // one XCD: a contiguous range of (part, tile) pairs — the tiles of a part read the same operand rows (conv_wgrad_big_kernel's
  // order); the body's own remap of the tile index is then the identity's job
  const int T = g.a[i].tiles_co * g.a[i].tiles_kc;
  const int lid = g.by_rows ? xcd_remap(local, T * g.a[i].splits) : local;
  const int split = lid / T;
  wgrad_split_body<FMT, SMALL_MAP>(g.a[i], lid - split * T, split, g.by_rows != 0);
}

template <int FMT, bool SMALL_MAP>
static int launch_wgrad_group_terms(const WgradGroup& g, hipStream_t st) {
  constexpr int TERMS = Fmt<FMT>::terms;
  const size_t lds = sizeof(__bf16) * TERMS * 2 * 128 * PLANE_STRIDE;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_split_group_kernel<FMT, SMALL_MAP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_wgrad_group(split): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_split_group_kernel<FMT, SMALL_MAP>), dim3(g.first[kWgradGroupMax]), dim3(256), lds, st, g);
  return check_launch("conv_wgrad_group(split)");
}

// contraction mode 4 only; every problem on the same side of the small-map switch (the caller checks)
int launch_wgrad_split_group(const WgradArgs* a, const int n, hipStream_t st) {
  WgradGroup g;
  g.n = n;
  int at = 0;
  for (int i = 0; i < kWgradGroupMax; ++i) {
    g.a[i] = a[i < n ? i : n - 1];
    g.first[i] = at;
    if (i < n) at += a[i].tiles_co * a[i].tiles_kc * a[i].splits;
  }
  g.first[kWgradGroupMax] = at;
  static const int by_rows = !(getenv("DADET_WGRAD_GROUP_BY_ROWS") && getenv("DADET_WGRAD_GROUP_BY_ROWS")[0] == '0');
  g.by_rows = by_rows;
  return a[0].Wo < 32 ? launch_wgrad_group_terms<4, true>(g, st) : launch_wgrad_group_terms<4, false>(g, st);
}

template <int FMT, bool SMALL_MAP>
static int launch_wgrad_terms(WgradArgs& a, hipStream_t st) {
  constexpr int TERMS = Fmt<FMT>::terms;
  const size_t lds = sizeof(__bf16) * TERMS * 2 * 128 * PLANE_STRIDE;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_split_kernel<FMT, SMALL_MAP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_wgrad(split): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_split_kernel<FMT, SMALL_MAP>), dim3(a.tiles_co * a.tiles_kc, a.splits), dim3(256),
                     lds, st, a);
  return check_launch("conv_wgrad(split)");
}

int launch_wgrad_split(WgradArgs& a, int terms, hipStream_t st) {
  const bool small_map = a.Wo < 32;   // narrower than one K-step of rows
  if (terms == 4) return small_map ? launch_wgrad_terms<4, true>(a, st) : launch_wgrad_terms<4, false>(a, st);
  if (terms == 2) return small_map ? launch_wgrad_terms<2, true>(a, st) : launch_wgrad_terms<2, false>(a, st);
  return small_map ? launch_wgrad_terms<3, true>(a, st) : launch_wgrad_terms<3, false>(a, st);
}

}  // namespace dadet

// 4 = 2-term fp16 split under per-tensor power-of-two scales (3 MFMAs / K=16, fp32-class accuracy); 3 = 3-term bf16 split
// (6 MFMAs / K=16, fp32-class accuracy, no scales); 0 = exact fp32 MFMA; 2 = 2-term bf16 split (3 MFMAs / K=16, ~2^-16)
extern "C" int dadet_set_gemm_mode(int mode) {
  if (mode != 0 && mode != 2 && mode != 3 && mode != 4) {
    dadet::set_error("set_gemm_mode: mode must be 0, 2, 3 or 4");
    return DADET_EINVAL;
  }
  dadet::g_gemm_mode = mode;
  return DADET_OK;
}
extern "C" int dadet_get_gemm_mode(void) { return dadet::g_gemm_mode; }
