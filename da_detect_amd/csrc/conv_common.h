// Shared declarations of the implicit-GEMM convolution kernels (conv_igemm.hip: exact fp32 MFMA;
// conv_split.hip: split-bf16 MFMA).
#pragma once
#include <cstdlib>
#include "common.h"

namespace dadet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Buffer-descriptor access (guide T8): a raw buffer load whose byte offset is >= num_records returns 0 and
// a raw buffer store there is dropped, so padding taps, ragged tile edges and the K tail need NO branches
// and NO selects — every load of a K-tile is issued back to back and waited for once, right before the LDS
// write.  (The first version used `ok ? *p : 0`: hipcc lowered it to exec-masked flat_loads each followed by
// s_waitcnt vmcnt(0), i.e. eight serialised memory round trips per K-tile in front of the MFMAs.)
constexpr unsigned kOOB = 0xFFFFFFF0u;  // 16-byte aligned, beyond any supported buffer
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ inline float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ inline float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ inline void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, 0);
}

// Write-through (sc1) 16-byte stores and sc1 loads for tiles handed from one workgroup to another INSIDE a launch (the
// stream-K tail).  The XCDs' L2s are not coherent with each other:
// a plain store may sit dirty in the producer's L2.  An agent-scope release fence writes the whole L2 back (~2 - 6 us
// each: with one per workgroup the fused weight-gradient reduction made the step 2x SLOWER); write-through stores drained
// with s_waitcnt vmcnt(0) before the arrival counter is bumped, and sc1 loads on the consumer side, need no fence
// (MI355X_MICROARCH.md, hand-off table: "publish-large").
__device__ inline void buf_store4_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 16);
}
__device__ inline float4 buf_load4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


__device__ inline unsigned pack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));   // pure: free to be scheduled between MFMAs
  return r;
}
__device__ inline float lo_as_float(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ inline float hi_as_float(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// split four consecutive-k floats into TERMS planes of 4 bf16 (8 bytes each)
template <int TERMS>
__device__ inline void split4(const float4 v, uint2 (&out)[TERMS]) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
    const unsigned p0 = pack_bf16(a, b), p1 = pack_bf16(c, d);
    out[t] = make_uint2(p0, p1);
    if (t + 1 < TERMS) {
      a -= lo_as_float(p0); b -= hi_as_float(p0);
      c -= lo_as_float(p1); d -= hi_as_float(p1);
    }
  }
}

// ---- fp16 two-term split (contraction mode 4, round 4) -----------------------------------------------------------
// x * s = h + l + eps with h = f16(x * s), l = f16(x * s - h): 11 + 11 significand bits and the sign of l leave
// |eps| <= 2^-23 |x s| — one bit short of fp32 — and the one dropped product l_a * l_b is below 2^-24 |a b|, so
//   a * b ~= h_a h_b + h_a l_b + l_a h_b        3 MFMAs per K=16 (v_mfma_f32_32x32x16_f16), fp32 accumulation
// has the accuracy class of the six-product bf16 split at HALF the matrix work and two operand planes instead of three.
// What bf16 gave for free is range: fp16 spans 2^-24 .. 65504.  `s` is a power of two per TENSOR (exact to apply and to
// undo) that brings the tensor's largest magnitude into [2^14, 2^15): nothing overflows, an element above 2^-16 of the
// tensor's maximum keeps the full 2^-23 relative accuracy, and smaller ones (l falls into fp16's subnormals) an absolute
// error below 2^-40 of the maximum.  The maximum comes from a device-side slot (amax_read: scalar loads):
// every producer in this library that can feed a GEMM leaves max|y| there (fused epilogues, dadet_amax otherwise).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// exponent e of the power-of-two scale 2^e for a tensor whose largest magnitude is `amax` (>= 0; inf / nan / 0 -> e = 0)
__device__ inline int fmt4_exp(const float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
  if (be == 0 || be == 255) return 0;
  int e = 14 - (be - 127);
  return e > 126 ? 126 : (e < -126 ? -126 : e);
}
__device__ inline float pow2f(const int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// split four consecutive-k floats, scaled by s, into the two fp16 planes (8 bytes each).  Inline assembly for two reasons:
// v_fma_mix_f32 forms the residual x * s - h straight from the packed half (one instruction instead of a conversion and
// a subtraction; x * s is exact, so the single rounding is the subtraction's), and written as plain conversions the
// whole split was SUNK by hipcc out of the K loop's MFMA blocks into the block between the two barriers (seen in the
// ISA: 64 conversions serialised behind the MFMAs, the GEMM 40% slower on all-zero operands than the bf16 two-term form).
__device__ inline unsigned pack_f16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ inline float resid_lo(float x, float s, unsigned h) {   // x * s - (low half of h)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
  return r;
}
__device__ inline float resid_hi(float x, float s, unsigned h) {   // x * s - (high half of h)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
  return r;
}
__device__ inline void split4h(const float4 v, const float s, uint2 (&out)[2]) {
  const unsigned h0 = pack_f16(v.x * s, v.y * s), h1 = pack_f16(v.z * s, v.w * s);
  out[0] = make_uint2(h0, h1);
  out[1] = make_uint2(pack_f16(resid_lo(v.x, s, h0), resid_hi(v.y, s, h0)),
                      pack_f16(resid_lo(v.z, s, h1), resid_hi(v.w, s, h1)));
}

// FMT (template parameter of the split kernels): 3 / 2 = three / two bf16 terms, 4 = two fp16 terms
template <int FMT> struct Fmt { static constexpr int terms = FMT == 3 ? 3 : 2; static constexpr bool f16 = FMT == 4; };

template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// A slot is EIGHT words, DADET_AMAX_STRIDE floats apart (include/dadet.h): workgroup b merges into shard b % 8 (hardware
// deals workgroups round-robin to the 8 XCDs), a reader takes the maximum of the eight.  One word takes ~12 ns per
// device-scope atomic (MI355X_MICROARCH.md, "fanin"): the first version — one atomic per wavefront on one word — cost
// the 128x128 GEMM 20 us per launch (2048 waves).  Here: one LDS reduction per workgroup, one check per workgroup of
// the shard's current value (an sc1 load; the atomics drop the line from L2, so the next load sees them), and an atomic
// only from a workgroup that would raise it — a few per shard and launch.
constexpr int kAmaxStride = DADET_AMAX_STRIDE;
__device__ inline float amax_read(const float* slot) {
  float m = slot[0];
#pragma unroll
  for (int s = 1; s < 8; ++s) m = fmaxf(m, slot[s * kAmaxStride]);
  return m;
}
// every thread of the workgroup calls this once (it contains barriers); m >= 0: max|v| of what the thread stored.
// Bits of non-negative floats order like unsigned integers.
__device__ inline void amax_publish(unsigned* slot, float m) {
  __shared__ unsigned s_amax;
  unsigned* shard = slot + (size_t)((blockIdx.x + blockIdx.y) & 7) * kAmaxStride;
  unsigned seen = 0;
  if (threadIdx.x == 0) {
    s_amax = 0;
    seen = __hip_atomic_load(shard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // lands while the others reduce
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&s_amax, __builtin_bit_cast(unsigned, m));
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned b = s_amax;
    if (b > seen) __hip_atomic_fetch_max(shard, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Non-finite guard of contraction mode 4.  A per-tensor scale taken from a stale or too small maximum overflows fp16 in
// the operand split: the leading term becomes inf, its products with the zeros of a ReLU'd operand NaN.  Every GEMM looks at
// its accumulators once, right behind the K loop (one class test per register, ~0.1% of a tile's time), and a workgroup
// that finds a non-finite value records the launch: the first one's id by compare-and-swap, a count by an atomic add —
// nothing is written on clean data.  The host reads the two words at its logging period (dadet_nonfinite_poll) and names
// the launch from a ring of recent launch records instead of reporting "loss is NaN" hundreds of kernels later.
template <int TM, int TN>
__device__ __forceinline__ void nf_check(const f32x16 (&acc)[TM][TN], unsigned* flag, const unsigned id) {
  if (!flag) return;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        bad |= (__builtin_bit_cast(unsigned, acc[i][j][e]) & 0x7f800000u) == 0x7f800000u;
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) {      // one lane per wavefront that saw any
    atomicCAS(flag, 0u, id + 1u);
    atomicAdd(flag + 1, 1u);
  }
}
// host side (conv_igemm.hip): id of the launch being prepared (recorded in the ring) and the device words
unsigned nf_next_launch(const char* kind, int M, int N, int K, int KH);
unsigned* nf_flag_ptr();

constexpr int PLANE_STRIDE = 40;  // bf16 per staged row: 32 + 8 pad = 80 bytes
constexpr int EPI_STRIDE = 40;    // floats per transposed row of the 16-byte epilogue

constexpr int BK = 32;          // K-tile
constexpr int LDS_STRIDE = 36;  // floats per staged row (32 + 4 pad, keeps 16-byte alignment)

struct ConvArgs {
  const float* x;
  const float* w;
  const float* scale;
  const float* bias;
  const float* addend;
  const float* mask_ref;
  float* y;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, OutH, OutW, os, relu_mode;
  int M, K;        // GEMM rows, reduction length
  int tiles_m, tiles_n;
  unsigned x_bytes, w_bytes, y_bytes;  // buffer extents (< 4 GB each)
  int ablate;                          // profiling only (DADET_ABLATE): bit mask of pipeline stages to skip
  int ws_panel0;                       // weight-stationary 1x1 kernel: first column panel of this launch (conv_ws.hip)
  int epi_v4;                          // 16-byte epilogue through an LDS transpose (conv_epilogue_v4); 0: 4-byte form
  // split-K (small grids only, conv_split.hip): blockIdx.y handles K range [y * ksplit, (y + 1) * ksplit) and writes
  // its raw partial sums to y + blockIdx.y * split_stride; 0 / 0 = the whole reduction in one workgroup
  int ksplit;
  unsigned split_stride;               // floats
  // stream-K tail (conv_fwd_split_sk_kernel): tiles [0, sk_dp_tiles) one per workgroup, the K-tile iterations of the
  // last sk_tiles tiles in sk_units contiguous ranges of sk_iters; partial tiles meet in sk_ws ([sk_tiles][sk_max_parts]
  // [128 x 128] floats) under the arrival counters sk_counters[sk_tiles] (zeroed before the launch)
  int sk_dp_tiles, sk_tiles, sk_units, sk_iters, sk_max_parts;
  float* sk_ws;
  int* sk_counters;
  // large-tile kernel (conv_big.hip): number of K ranges a tile's reduction is cut into (0: another kernel runs); the
  // parts meet in sk_ws ([tile][part][256 x 256] floats) under sk_counters[tile] (arrivals) / [2048 + tile] (parked)
  int big_splits;
  // two parts meet symmetrically under sk_counters[4096 + 4 * tile ..] (tickets, flag of part 0, flag of part 1, started;
  // never reset); big_asym != 0 (DADET_BIG_ASYM=1, A/B runs): the round-5 hand-over (part 0 parks everything, part 1 finishes)
  int big_asym;
  // > 0: only the tiles from big_body on (the partly filled last round of a grid of more tiles than CUs) are cut into
  // big_splits parts; tiles [0, big_body) run their whole reduction in one workgroup (conv_big.hip: big_tail_plan)
  int big_body;
  // non-finite guard (mode 4): device words {first offending launch id + 1, count} and this launch's id; null = off
  unsigned* nf_flag;
  unsigned launch_id;
  // mode 4 (fp16 two-term split): max|x|, max|w| of the operands (device floats; null = scale 1) and the slot that
  // receives max|y| of what this launch stores (null = not wanted; zero or an earlier launch's maximum before)
  const float* amax_x;
  const float* amax_w;
  unsigned* amax_y;
};

struct WgradArgs {
  const float* x;
  const float* gy;
  const float* out_scale;
  float* out;       // dw (splits == 1) or workspace [splits][Cout][K]
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int M, K;
  int gy_ld;        // floats between consecutive rows of gy (>= Cout, a multiple of 4: rows padded to 16 bytes)
  int tiles_co, tiles_kc, splits, rows_per_split;  // rows_per_split is a multiple of 32
  int direct;       // 1: write dw with scale / accumulate applied here
  int accumulate;
  unsigned x_bytes, gy_bytes;
  const float* amax_x;   // mode 4: max|x|, max|gy| (device floats; null = scale 1)
  const float* amax_gy;
  unsigned* nf_flag;     // non-finite guard, see ConvArgs
  unsigned launch_id;
};

// several weight gradients in one launch (dadet_conv_wgrad_group): problem i owns workgroups [first[i], first[i + 1])
constexpr int kWgradGroupMax = 4;
struct WgradGroup {
  WgradArgs a[kWgradGroupMax];
  int first[kWgradGroupMax + 1];
  int n;
  int by_rows;      // 128 x 128 form: workgroups of one XCD take neighbouring tiles of the same rows
};

// tile variant chosen for a forward / dgrad GEMM of M rows and Cout columns
//   0: 128x128 (TM=2,TN=2)   1: 128x64 (TM=2,TN=1)   2: 64x64 (TM=1,TN=1)
inline int fwd_variant(int M, int Cout) {
  const int64_t t128 = (int64_t)ceil_div(M, 128) * ceil_div(Cout, 128);
  // 128x128 tiles from one workgroup per CU on: tools/fwd_sweep.py, res4 3x3 256->256 0.115 ms against 0.128 ms with
  // 128x64 tiles, res4 1x1 1024->256 0.061 against 0.064 (with two GEMM streams the step did not notice; with one: +0.3%)
  static const int min_tiles = getenv("DADET_FWD_MIN_TILES128") ? atoi(getenv("DADET_FWD_MIN_TILES128")) : kNumCU;
  if (Cout > 64 && t128 >= min_tiles) return 0;
  if (Cout > 32) {
    const int64_t t64 = (int64_t)ceil_div(M, 128) * ceil_div(Cout, 64);
    if (t64 >= kNumCU || M <= 64 * 64) return 1;
    return 2;
  }
  return 1;
}

// 4 = products from a 2-term fp16 split of both operands under per-tensor power-of-two scales; 3 / 2 = from a
// 3- / 2-term bf16 split; 0 = exact fp32 MFMA
int gemm_mode();
int launch_fwd_split(ConvArgs& a, int variant, int fmt, hipStream_t st);
int launch_fwd_split_sk(ConvArgs& a, int fmt, hipStream_t st);
int launch_wgrad_split(WgradArgs& a, int fmt, hipStream_t st);
// 256 x 256-tile kernel of mode 4 for the long-K layers (conv_big.hip); ws / counters: split-reduction workspace of
// big_workspace_bytes(a) bytes and the stream's zeroed counters (may be null when that is 0)
bool big_eligible(const ConvArgs& a);
int big_variant(const ConvArgs& a);      // 0 none, 1 the 256 x 256 tile, 2 the 256 x 128 tile
size_t big_workspace_bytes(const ConvArgs& a);
int launch_fwd_big(ConvArgs& a, hipStream_t st, float* ws, int* counters);
// weight gradient on 256 x 256 tiles (conv_big.hip): the plan (false: the 128 x 128 kernel runs) and the launch; partial sums
// in the 128 x 128 kernel's [splits][Cout][K] layout
bool wgrad_big_plan(const dadet_conv_desc* d, int* tiles_co, int* tiles_kc, int* splits, int* rps);
int launch_wgrad_big(WgradArgs& a, hipStream_t st);
// several weight gradients in one launch of that kernel (dadet_conv_wgrad_group): membership, the common rows per part, launch
bool wgrad_group_member(const dadet_conv_desc* d);
void wgrad_group_plan(int n, const dadet_conv_desc* d, int tile, int* tiles_co, int* tiles_kc, int* splits, int* rows);
int launch_wgrad_big_group(const WgradArgs* a, int n, hipStream_t st);
int launch_wgrad_split_group(const WgradArgs* a, int n, hipStream_t st);      // the 128 x 128 kernel's grouped form (conv_split.hip)
// weight-stationary 1x1 kernel for K = 64 / 128 / 256 (conv_ws.hip)
bool ws_eligible(const ConvArgs& a);
int launch_fwd_ws(ConvArgs& a, int fmt, hipStream_t st);

}  // namespace dadet
