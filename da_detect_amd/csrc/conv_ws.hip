// Weight-stationary 1x1 convolution for the short-K layers (K = Cin in {64, 128, 256}; round 4).
//
// What these layers are: the expanding 1x1 of every bottleneck (conv3: mid -> 4 mid, + residual + ReLU) and its mirror in
// backward (the data gradient of conv1: mid -> 4 mid, + residual gradient, ReLU gate) — M x K operands, M x 4K results,
// i.e. little arithmetic per byte: ~2.2 ms per `img_only` step at 2.4 - 3.8 TB/s and 54 - 115 TF/s on the tiled kernel
// (profiles/r03_gemm_table_per_shape.txt).  There the 64 x 64 tiles of conv_fwd_split_kernel<1,1,3> pay, per tile, the
// split of BOTH operands (as much VALU time as MFMA time), two barriers per 32-deep K-tile around 12 MFMAs per wave, and
// a serial load -> contract -> store life that only overlaps across the five workgroups of a CU.
//
// Here a workgroup (8 waves, one per CU) owns a PANEL of BN output channels: the BN x K weights are split into their
// three bf16 term planes ONCE and stay in LDS (101 - 110 KB) for the workgroup's life.  Each wave then streams 32-row
// slabs of the activation matrix straight from global memory into MFMA A-fragments in registers — no LDS staging, no
// barrier after the prologue, each activation element split exactly once per panel — and multiplies them against the
// resident panel (B-fragments by ds_read_b128).  Waves run unsynchronised: while one waits in its epilogue (residual /
// gate loads, 16-byte stores) the other wave of the SIMD owns the matrix pipe, so the store stream of one slab overlaps
// the contraction of the next by construction.
//
// K order: an MFMA consumes, per lane (row r = lane & 31, half h = lane >> 5), 8 consecutive k-slots 8h .. 8h+7 of a
// 16-deep step.  Any permutation of k is legal as long as both operands use it, so the slots of the two steps of a
// 32-deep chunk c are mapped to PHYSICAL k = 32c + 16h + 8s + j: a lane then reads 64 contiguous bytes of its row per
// chunk (four buffer_load_b128 with immediate offsets), the two halves of a wave cover whole 128-byte lines, and the
// weight fragments are ordinary 16-byte reads of the [n][k] planes at k = 32c + 16h + 8s.
//
// Arithmetic per output element is the tiled kernel's (same six products per k16 step, fp32 accumulation, same
// epilogue expression); the k order within the accumulation differs, so results agree to fp32 rounding, not bit for bit.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace dadet {

constexpr unsigned kOOBws = 0x80000000u;   // beyond every buffer this kernel accepts (< 2 GB), survives immediate offsets
constexpr int WS_ROWS = 256;               // rows per workgroup pass: 8 waves x 32

template <int N, int I = 0, class F>
__device__ __forceinline__ void ws_unroll(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ws_unroll<N, I + 1>(f);
  }
}

// FMT 3: three bf16 term planes, six MFMAs per k16 step; FMT 4: two fp16 term planes under the per-tensor scales, three
// MFMAs (conv_common.h) — the panel is a third smaller, an activation element costs 3 VALU operations instead of ~4.8
// EROWS: rows of a wave's private epilogue slice — 32 (a whole 32 x 32 accumulator block at a time) or 16 (two halves: the
// K = 256 / BN = 128 form, whose 132 KB of weight planes leave 20 KB for the eight slices)
template <int K, int BN, int FMT, int EROWS = 32>
__global__ __launch_bounds__(512, 1) void conv1x1_ws_kernel(const ConvArgs a) {
  constexpr int T = Fmt<FMT>::terms;
  constexpr bool F16 = Fmt<FMT>::f16;
  constexpr int NB = BN / 32;               // 32-column blocks per wave
  constexpr int STRIDE = K + 8;             // bf16 per weight row in LDS: (2K + 16) bytes = 4 banks mod 64 -> conflict-free b128 reads
  constexpr int PLANE = BN * STRIDE;
  constexpr int CH = K / 32;                // 32-deep chunks per slab
  // Activation registers: a RING of four chunk slots (16 floats of this lane's row each, 64 VGPRs in all).  Chunk q of the
  // flat chunk sequence (slab after slab) lives in slot q & 3; once a slot's chunk has been split, chunk q + 4 is loaded
  // into it — three chunks of MFMA time (3 x 768 NB cycles with two waves on a SIMD, >= 1.9 us) for the load to land.
  constexpr int SUPER = CH < 4 ? 4 : CH;    // chunks per loop iteration: whole slabs, a multiple of the ring
  constexpr int SL = SUPER / CH;            // slabs per loop iteration (2 for K = 64)
  // column blocks whose residual / gate loads are in flight together (register budget: 256 with two waves per SIMD)
  constexpr int EPB = 2;
  constexpr bool EARLY = NB == 2;           // residual / gate loads of a slab issued in the middle of its contraction

  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* Bs = reinterpret_cast<__bf16*>(smem);                      // [T][BN][STRIDE]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  float* tile = reinterpret_cast<float*>(smem + T * PLANE * 2) + wave * (EROWS * EPI_STRIDE);
  float* s_affine = reinterpret_cast<float*>(smem + T * PLANE * 2) + 8 * (EROWS * EPI_STRIDE);   // [BN] scale, [BN] bias
  int ea = 0, eb = 0;                       // FMT 4: exponents of the operands' power-of-two scales
  if (F16) {
    ea = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
    eb = a.amax_w ? fmt4_exp(amax_read(a.amax_w)) : 0;
  }
  const float sa = pow2f(ea), sb = pow2f(eb);
  const float u1 = pow2f(-(ea + eb) / 2), u2 = pow2f(-(ea + eb) - (-(ea + eb)) / 2);   // undo both, in two exact steps
  float mx = 0.f;                           // max|y| of what this lane stores (a.amax_y)

  // hardware deals workgroup b to XCD b % 8: the `panels` workgroups of one row group sit on ONE XCD, so the activation
  // rows they all read come through one L2
  const int panels = a.tiles_n;
  const int G = (int)gridDim.x / panels;
  const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
  const int panel = j % panels, group = (j / panels) * 8 + xcd;
  const int n0 = (a.ws_panel0 + panel) * BN;
  const int tiles = a.tiles_m;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, a.w_bytes);
  const int frow = lane & 31, fh = lane >> 5;
  const int HoWo = a.Ho * a.Wo;
  const bool dense = a.stride == 1;         // 1x1, pad 0, stride 1: output row m reads input pixel m

  auto row_off = [&](int tile_idx) __attribute__((always_inline)) -> unsigned {
    const int m = tile_idx * WS_ROWS + wave * 32 + frow;
    if (tile_idx >= tiles || m >= a.M) return kOOBws;
    int pix = m;
    if (!dense) {
      const int img = m / HoWo;
      const int rem = m - img * HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      pix = (img * a.H + ho * a.stride) * a.W + wo * a.stride;
    }
    return (unsigned)pix * (unsigned)(K * 4) + (unsigned)fh * 64u;
  };

  float4 raw[4][4];                         // the ring
  // ---- weight panel: fp32 [BN][K] -> three bf16 planes in LDS, once per workgroup
  {
    constexpr int Q = K / 4;                // float4 per weight row
    constexpr int PER = (BN * Q) / 512;
    float4 wv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = t + i * 512;
      const int n = idx / Q, kq = idx - n * Q;
      wv[i] = buf_load4(wr, (n0 + n < a.Cout) ? (unsigned)(((n0 + n) * K + kq * 4) * 4) : kOOBws);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = t + i * 512;
      const int n = idx / Q, kq = idx - n * Q;
      uint2 p[T];
      if constexpr (F16) split4h(wv[i], sb, p);
      else split4<T>(wv[i], p);
#pragma unroll
      for (int pl = 0; pl < T; ++pl) *reinterpret_cast<uint2*>(Bs + pl * PLANE + n * STRIDE + kq * 4) = p[pl];
    }
  }
  if (t < BN) {
    const int n = n0 + t;
    s_affine[t] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
    s_affine[BN + t] = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
  }
  __syncthreads();                          // the only workgroup barrier of the kernel
  // the first four chunks go in flight (after the panel: its staging registers are free again)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned o = row_off(group + (q / CH) * G);
#pragma unroll
    for (int v = 0; v < 4; ++v) raw[q][v] = buf_load4(xr, o + (unsigned)((q % CH) * 128 + v * 16));
  }
  __builtin_amdgcn_sched_barrier(0);

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;

  // term planes of one k16 step of this lane's row: 8 floats -> T x 8 bf16 / fp16
  auto split8 = [&](const float4 lo, const float4 hi, bf16x8 (&out)[T]) __attribute__((always_inline)) {
    uint2 pl[T], ph[T];
    if constexpr (F16) {
      split4h(lo, sa, pl);
      split4h(hi, sa, ph);
    } else {
      split4<T>(lo, pl);
      split4<T>(hi, ph);
    }
#pragma unroll
    for (int p = 0; p < T; ++p) out[p] = __builtin_bit_cast(bf16x8, make_uint4(pl[p].x, pl[p].y, ph[p].x, ph[p].y));
  };
  const __bf16* Bb = Bs + frow * STRIDE + fh * 16;
  bf16x8 pa[T];                             // planes of the NEXT step to be multiplied (carried across buffers / slabs)

  // One k16 step: chunk position q of the iteration (slot q & 3), half s.  The planes of the NEXT step are formed (VALU)
  // in the shadow of this step's MFMAs; in the second half of a chunk the slot is free (both its halves are split) and the
  // chunk four positions ahead is fetched into it.  `ahead[k]` = row offset of the slab k slabs after the iteration's first.
  auto step = [&](auto Q, auto S, const unsigned (&ahead)[4]) __attribute__((always_inline)) {
    constexpr int q = decltype(Q)::value, sidx = decltype(S)::value;
    constexpr int c = q % CH, slot = q & 3;
    if (sidx == 1) {
      constexpr int qa = q + 4;             // position of the chunk to fetch, relative to the iteration start
      const unsigned o = ahead[qa / CH];
#pragma unroll
      for (int v = 0; v < 4; ++v) raw[slot][v] = buf_load4(xr, o + (unsigned)((qa % CH) * 128 + v * 16));
    }
    bf16x8 fb[T][NB];
#pragma unroll
    for (int p = 0; p < T; ++p)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        fb[p][nb] = *reinterpret_cast<const bf16x8*>(Bb + p * PLANE + nb * 32 * STRIDE + c * 32 + sidx * 8);
    bf16x8 pn[T];
    if (sidx == 0) split8(raw[slot][2], raw[slot][3], pn);
    else split8(raw[(q + 1) & 3][0], raw[(q + 1) & 3][1], pn);
    // smallest cross terms first, the leading a0*b0 last (the tiled kernel's order)
#pragma unroll
    for (int order = 2 * (T - 1); order >= 0; --order)
#pragma unroll
      for (int x = 0; x < T; ++x) {
        const int y = order - x;
        if (y < 0 || y >= T || x + y > T - 1) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_32x32x16<F16>(pa[x], fb[y][nb], acc[nb]);
      }
    {
      constexpr int kMfma = (T == 3 ? 6 : 3) * NB;
      constexpr int kValuPerMfma = (T == 3 ? 56 : 28) / kMfma + 1;
      if (sidx == 1) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // the slot's refill first
      __builtin_amdgcn_sched_group_barrier(0x100, T * NB, 0);              // DS reads
#pragma unroll
      for (int m = 0; m < kMfma; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);      // VALU in its shadow
      }
    }
#pragma unroll
    for (int p = 0; p < T; ++p) pa[p] = pn[p];
    // a step is one scheduling region: without this fence the slot refills of the second half of a slab (same base
    // register: the NEXT slab's row) were clustered behind the slab's last MFMA, i.e. not prefetched at all
    __builtin_amdgcn_sched_barrier(0);
  };

  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y, a.y_bytes);
  const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.addend ? a.addend : a.y, a.addend ? a.y_bytes : 0u);
  const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_ref ? a.mask_ref : a.y, a.mask_ref ? a.y_bytes : 0u);
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;

  // fused epilogue of one slab (conv_epilogue_v4's: every 32 x 32 accumulator block turned through the wave's private
  // LDS slice so that a lane owns 4 consecutive columns; y / residual / gate move 16 bytes per lane), then acc = 0.
  // Two phases: epi_issue computes the addresses of EPB column blocks and puts their residual / gate loads in flight,
  // epi_finish consumes them.  With registers to spare (EARLY: the K = 256 / BN = 64 form, 170 VGPRs without them) the
  // loads of a slab go out in the MIDDLE of its contraction: a wave then spends its epilogue storing, not waiting — with
  // two waves per SIMD a wave that waits leaves the matrix pipe to ONE wave for the ~2 us of an HBM round trip, which is
  // 40% of a slab's 5 us.
  unsigned e_offs[EPB][4];                  // EARLY only: alive from the middle of a slab to its epilogue
  float4 e_ad[EPB][4], e_mk[EPB][4];
  auto epi_issue = [&](int tile_idx, int nb0, unsigned (&offs)[EPB][4], float4 (&ad)[EPB][4], float4 (&mk)[EPB][4])
      __attribute__((always_inline)) {
    const int mbase = tile_idx * WS_ROWS + wave * 32;
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
      const int n = n0 + (nb0 + e) * 32 + c4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int m = mbase + pass * 8 + rrow;
        offs[e][pass] = (n < a.Cout && tile_idx < tiles && m < a.M) ? ((unsigned)m * (unsigned)a.Cout + (unsigned)n) * 4u : kOOB;
      }
    }
    if (a.addend) {
#pragma unroll
      for (int e = 0; e < EPB; ++e)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) ad[e][pass] = buf_load4(ar, offs[e][pass]);
    }
    if (a.relu_mode == 2) {
#pragma unroll
      for (int e = 0; e < EPB; ++e)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) mk[e][pass] = buf_load4(mr, offs[e][pass]);
    }
  };
  auto epi_finish = [&](int nb0, const unsigned (&offs)[EPB][4], const float4 (&ad)[EPB][4], const float4 (&mk)[EPB][4])
      __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
      const float4 sc = *reinterpret_cast<const float4*>(s_affine + (nb0 + e) * 32 + c4);
      const float4 bi = *reinterpret_cast<const float4*>(s_affine + BN + (nb0 + e) * 32 + c4);
      constexpr int GH = EROWS / 8;                      // 8-row groups per trip through the slice
#pragma unroll
      for (int hh = 0; hh < 4 / GH; ++hh) {
#pragma unroll
      for (int g = hh * GH; g < (hh + 1) * GH; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          tile[(q + 8 * (g - hh * GH) + row_hi) * EPI_STRIDE + col_in] =
              F16 ? acc[nb0 + e][g * 4 + q] * u1 * u2 : acc[nb0 + e][g * 4 + q];
      __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): a wave's own LDS traffic is ordered
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = hh * GH; pass < (hh + 1) * GH; ++pass) {
        const float4 v4 = *reinterpret_cast<const float4*>(tile + ((pass - hh * GH) * 8 + rrow) * EPI_STRIDE + c4);
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {bi.x, bi.y, bi.z, bi.w};
        float adv[4] = {0.f, 0.f, 0.f, 0.f}, mkv[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.addend) {
          const float4 q4 = ad[e][pass];
          adv[0] = q4.x; adv[1] = q4.y; adv[2] = q4.z; adv[3] = q4.w;
        }
        if (a.relu_mode == 2) {
          const float4 q4 = mk[e][pass];
          mkv[0] = q4.x; mkv[1] = q4.y; mkv[2] = q4.z; mkv[3] = q4.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float x = v[k];
          if (a.scale) x = x * s4[k];
          if (a.bias) x = x + b4[k];
          if (a.addend) x = x + adv[k];
          if (a.relu_mode == 1) x = fmaxf(x, 0.f);
          else if (a.relu_mode == 2) x = (mkv[k] > 0.f) ? x : 0.f;
          v[k] = x;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), yr,
                                               (int)offs[e][pass], 0, 0);
        if (F16 && a.amax_y && offs[e][pass] != kOOB)
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
      __builtin_amdgcn_wave_barrier();                   // the slice is rewritten by the next half / block
      }
    }
  };
  auto epilogue = [&](int tile_idx) __attribute__((always_inline)) {
    if (F16 && a.nf_flag) {      // non-finite guard (conv_common.h: nf_check), on the scaled sums: the same class
      bool bad = false;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) bad |= (__builtin_bit_cast(unsigned, acc[nb][e]) & 0x7f800000u) == 0x7f800000u;
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) {
        atomicCAS(a.nf_flag, 0u, a.launch_id + 1u);
        atomicAdd(a.nf_flag + 1, 1u);
      }
    }
#pragma unroll
    for (int nb0 = 0; nb0 < NB; nb0 += EPB) {
      if (EARLY && nb0 == 0) {
        epi_finish(nb0, e_offs, e_ad, e_mk);
      } else {
        unsigned offs[EPB][4];
        float4 ad[EPB][4], mk[EPB][4];
        epi_issue(tile_idx, nb0, offs, ad, mk);
        epi_finish(nb0, offs, ad, mk);
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
  };

  split8(raw[0][0], raw[0][1], pa);
  for (int tl = group; tl < tiles; tl += SL * G) {
    // row offsets of the slabs this iteration fetches from: (q + 4) / CH slabs ahead, q = 0 .. SUPER - 1
    unsigned ahead[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ahead[k] = (k <= (SUPER + 3) / CH) ? row_off(tl + k * G) : kOOBws;
    ws_unroll<SUPER>([&](auto Q) __attribute__((always_inline)) {
      constexpr int q = decltype(Q)::value;
      if (EARLY && (q % CH) == CH / 2) epi_issue(tl + (q / CH) * G, 0, e_offs, e_ad, e_mk);
      step(Q, std::integral_constant<int, 0>{}, ahead);
      step(Q, std::integral_constant<int, 1>{}, ahead);
      if ((q + 1) % CH == 0) epilogue(tl + (q / CH) * G);   // a slab past the end (K = 64, odd slab count) stores nothing
    });
  }
  if (F16 && a.amax_y) amax_publish(a.amax_y, mx);          // once per workgroup: it walked all of its slabs
}

// panel width per K: the three (FMT 4: two) planes of BN x (K + 8) 16-bit terms plus the epilogue slices (40 KB) must
// fit 160 KB: 128 columns for K = 64 / 128, 64 for K = 256

bool ws_eligible(const ConvArgs& a) {
  const char* env = getenv("DADET_WS_1X1");           // read per call: the tests and A/B runs flip it at run time
  if (env && env[0] == '0') return false;
  if (a.KH != 1 || a.KW != 1 || a.pad != 0 || a.os != 1 || a.ksplit) return false;
  if (a.K != 64 && a.K != 128 && a.K != 256) return false;
  if (!a.epi_v4) return false;                              // 16-byte epilogue: Cout % 4 == 0, aligned tensors
  if (a.x_bytes >= kOOBws || a.w_bytes >= kOOBws) return false;
  if (a.Cout < 128 || a.Cout % 32) return false;            // narrow layers stay on the 128 x 64 tiles
  if (a.M < 64 * WS_ROWS) return false;                     // too few slabs to fill the chip's 256 workgroups
  return true;
}

template <int K, int BN, int FMT, int EROWS = 32>
static int launch_ws(ConvArgs& a, hipStream_t st) {
  const size_t lds = (size_t)Fmt<FMT>::terms * BN * (K + 8) * 2 + 8 * EROWS * EPI_STRIDE * 4 + 2 * BN * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<K, BN, FMT, EROWS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_forward(weight-stationary 1x1): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  const int panels_all = ceil_div(a.Cout, BN);
  a.tiles_m = ceil_div(a.M, WS_ROWS);
  // one workgroup per CU: `panels` x G with G a multiple of 8 (the XCD mapping above) and at most one slab group per slab
  const int need = ceil_div(a.tiles_m, 8) * 8;
  auto groups = [&](const int panels) {
    int G = (kNumCU / panels) / 8 * 8;
    if (G < 8) G = 8;
    return G > need ? need : G;
  };
  // A panel count that does not divide the chip (18 panels of the deformable blocks' data gradient, 2304 columns: 18 x 8 =
  // 144 workgroups walk 8 slabs each) is cut into 2 or 3 launches over column ranges when that shortens the walk: launches
  // x slabs per workgroup is the launch's length in slab times (9 panels x 24 groups: 2 x 3 instead of 8).
  int parts = 1, best = ceil_div(a.tiles_m, groups(panels_all));
  for (int p = 2; p <= 3 && p <= panels_all; ++p) {
    const int cost = p * ceil_div(a.tiles_m, groups(ceil_div(panels_all, p)));
    if (cost * 8 < best * 7) { best = cost; parts = p; }      // (at least an eighth shorter: every launch loads its panels anew)
  }
  const int per = ceil_div(panels_all, parts);
  for (int p0 = 0; p0 < panels_all; p0 += per) {
    a.ws_panel0 = p0;
    a.tiles_n = panels_all - p0 < per ? panels_all - p0 : per;
    hipLaunchKernelGGL((conv1x1_ws_kernel<K, BN, FMT, EROWS>), dim3(a.tiles_n * groups(a.tiles_n)), dim3(512), lds, st, a);
  }
  return check_launch("conv_forward(weight-stationary 1x1)");
}

int launch_fwd_ws(ConvArgs& a, int fmt, hipStream_t st) {
  if (fmt == 4) {
    switch (a.K) {
      case 64: return launch_ws<64, 128, 4>(a, st);
      case 128: return launch_ws<128, 128, 4>(a, st);
      default: {
        // K = 256: 128-column panels where the layer has at least 256 columns (res4 conv3 and its mirror, 1024 columns:
        // a workgroup then stores 512 contiguous bytes per row instead of 256, and the activations are read by 8 panels
        // instead of 16) — the weight planes take 132 KB, the epilogue slices are halved to fit.  DADET_WS_K256_BN=64: off
        const char* e = getenv("DADET_WS_K256_BN");      // read per call (A/B runs, tests)
        const bool wide = !(e && atoi(e) == 64);
        if (wide && a.Cout >= 256) return launch_ws<256, 128, 4, 16>(a, st);
        return launch_ws<256, 64, 4>(a, st);
      }
    }
  }
  switch (a.K) {
    case 64: return launch_ws<64, 128, 3>(a, st);
    case 128: return launch_ws<128, 128, 3>(a, st);
    default: return launch_ws<256, 64, 3>(a, st);
  }
}

}  // namespace dadet
