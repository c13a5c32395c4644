// Large-tile implicit-GEMM convolution of contraction mode 4 (fp32 operands as two fp16 terms under per-tensor
// power-of-two scales, 3 x v_mfma_f32_32x32x16_f16 per K = 16; conv_common.h) — the forward / data-gradient kernel for
// the long-K layers since round 5.
//
// Why another kernel.  conv_fwd_split_kernel<2,2,4> (128 x 128 tile, 4 waves of 64 x 64, one set of operand planes, two
// barriers per K-tile, two workgroups per CU) issues 7.65 VALU and 1.0 LDS instructions per MFMA and sits at 0.28 - 0.36 of
// the contraction's ceiling; tools/native/gemm_lab.hip measures what the tile shape and the schedule are worth on RANDOM
// operands (profiles/r05_gemm_lab.txt): the same arithmetic in a 256 x 256 tile with 8 waves of 128 x 64 runs 35% faster
// (0.35 -> 0.46 of 833 TF/s), and with the two waves of every SIMD taking TURNS on the matrix pipe 42% faster (0.49) — 75%
// of what a register-only MFMA stream sustains on such operand bits under this part's power management (1625 of the
// nominal 2500 TF/s).  Per MFMA the loop carries 1.5 VALU, 0.5 ds_read_b128 and 0.33 ds_write_b64.
//
// Structure.  One workgroup = 512 threads = 8 waves = a 256 x BN output tile (BN = 256: 2 x 4 waves of 128 x 64), one
// workgroup per CU, two waves per SIMD.  Waves 0-3 (group 0) and 4-7 (group 1) — one wave of each group on every SIMD —
// run COMPLEMENTARY segments separated by workgroup barriers (the guide's 8-phase idea, T3 / T4):
//     multiply segment   24 MFMAs of one k16 step from registers, nothing else
//     load segment       read the 12 fragments of the wave's next k16 step, wait for the global loads issued one load
//                        segment ago, split them (2 VALU per element: v_fma_mix{lo,hi}_f16) and store a quarter of a future
//                        K-tile's planes, issue the next global loads
// so the matrix pipe of a SIMD always has one wave multiplying while its partner does everything else, fragments are read
// when the wave does not multiply (ONE fragment register set), and the load segment may branch (tap changes) without
// cutting an MFMA schedule.  Segment s: group 0 multiplies step s / 2 for even s, group 1 step (s - 1) / 2 for odd s.
// Group 0 stages the activation rows (the implicit-GEMM gather), group 1 the weight rows; K-tile T + 1 is written during
// segments 4T - 1 .. 4T + 2 into the slot K-tile T - 1 was last read from in segment 4T - 2 (two slots of 64 KB).
//
// LDS image of a K-tile: planes A_h | A_l | B_h | B_l of [rows][32 k] fp16 (64-byte rows); the four 16-byte chunks of a
// row are XOR-swizzled with (row >> 2) & 3, which makes the ds_read_b128 fragment reads (lane groups of 16: rows {0-3,
// 12-15, 20-27} / {4-11, 16-19, 28-31} of one chunk column) and the 8-byte staging stores conflict-free without padding.
//
// Small grids.  A 256 x 256 tile leaves layers such as res4 3x3 (64 tiles) or the RPN conv (128) short of the 256 CUs:
// the reduction is cut into S contiguous K ranges (tiles x S <= 256 workgroups).  The parts of a tile meet in a workspace:
// a part takes an arrival ticket; all but the last park their sums (lane-linear 16-byte write-through stores, drained, then
// one "parked" count) and leave; the last one waits for the parked count (the others are resident and past their K loop: a
// bounded wait, no dependence on workgroups that may not have been dispatched), adds the parked parts to its registers in
// part order — its own at its index, so the sum does not depend on who arrives last — and runs the epilogue.
//
// Epilogue: conv_split.hip's 16-byte form (every 32 x 32 accumulator block turned through LDS so that a lane owns 4
// consecutive columns), two blocks per round of residual / gate loads.
#include "conv_common.h"

// DADET_BIG_TIMING (build-time, probes only): workgroup 0 .. 255 of conv_big_kernel leave s_memtime stamps at the phase
// boundaries (start, prologue done, K loop done, parts met, end) in a device array that dadet_big_timing_read copies out
#ifndef DADET_BIG_TIMING
#define DADET_BIG_TIMING 0
#endif
#if DADET_BIG_TIMING
__device__ unsigned long long g_big_stamps[256 * 8];
#define BIG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_big_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dadet_big_timing_read(unsigned long long* host, int n) {
  // (copies the stamps out and clears them: stamp 4 marks a workgroup that wrote its tile, stamp 5 one that parked a part)
  const int e = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_big_stamps), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_big_stamps)) == hipSuccess) (void)hipMemset(dev, 0, sizeof(g_big_stamps));
  return e;
}
#else
#define BIG_STAMP(i) do { } while (0)
#endif

namespace dadet {

namespace {
constexpr unsigned kBigOOB = 0x80000000u;   // an invalid row: beyond any buffer the dispatcher admits (< 2 GB), survives + soffset

// x s = h + l, h = f16(x s), l = f16(x s - h) for four consecutive-k floats.  v_fma_mix{lo,hi}_f16 form the scaled value /
// the residual in fp32 (both exact: s is a power of two, the residual of a rounding is representable) and round once to
// fp16 into one half of the destination — the same bits as conv_common.h's split4h at 2 instead of 3 VALU per element.
__device__ __forceinline__ void split4m(const float4 v, const float s, uint2& h, uint2& l) {
  unsigned h0, h1, l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(v.x), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(v.y), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h1) : "v"(v.z), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h1) : "v"(v.w), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(s), "v"(h0));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(s), "v"(h0));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(s), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(s), "v"(h1));
  h = make_uint2(h0, h1);
  l = make_uint2(l0, l1);
}

// m / d for 0 <= m < 2^24 (d > 0, rd = 1 / d): a float estimate and one correction each way
__device__ __forceinline__ int div_small(const int m, const int d, const float rd, int& rem) {
  int q = (int)((float)m * rd);
  int r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}
}  // namespace

// Which (tile, K part) this workgroup computes and into how many parts its tile's reduction is cut.  Uniform cut
// (big_body == 0): T * S workgroups, XCD-aware order over (part, tile).  Tail cut (big_body > 0, a multiple of 8): the first
// big_body workgroups take one whole tile each, the rest share the tiles of the partly filled last round, S parts each —
// the two ranges are placed on the XCDs separately, so that every XCD gets its share of both.
__device__ __forceinline__ void big_place(const ConvArgs& a, const int T, int& S, int& part, int& tile) {
  const int body = a.big_body;
  if (body == 0) {
    S = a.big_splits;
    const int lid = xcd_remap(blockIdx.x, T * S);
    part = lid / T;
    tile = lid - part * T;
  } else if ((int)blockIdx.x < body) {
    S = 1;
    part = 0;
    tile = xcd_remap(blockIdx.x, body);
  } else {
    S = a.big_splits;
    const int tt = T - body;
    const int r = xcd_remap((int)blockIdx.x - body, tt * S);
    part = r / tt;
    tile = body + (r - part * tt);
  }
}

// ---- pieces of what follows a tile's K loop, over the accumulator row blocks [LO, HI) of a wave ------------------------
// park: lane-linear, 16 bytes per lane and store, written through to memory: row block im as SLOT im - LO + SLOT0 (slot j,
// column block in, quarter g at ((j * TN + in) * 4 + g) * 8192 behind the lane's 16 bytes of the part's area)
template <int TM, int TN, int LO, int HI, int SLOT0>
__device__ __forceinline__ void big_park(const __amdgpu_buffer_rsrc_t pr, const unsigned mine, const f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int im = LO; im < HI; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        buf_store4_wt(pr, mine + (((im - LO + SLOT0) * TN + in) * 4 + g) * 8192u,
                      make_float4(acc[im][in][g * 4], acc[im][in][g * 4 + 1], acc[im][in][g * 4 + 2], acc[im][in][g * 4 + 3]));
}

// add another part's parked slots SLOT0 .. to the row blocks [LO, HI): up to sixteen 16-byte loads per lane in flight (four
// blocks: one round trip through the fabric, ~2 us)
template <int TM, int TN, int LO, int HI, int SLOT0>
__device__ __forceinline__ void big_add_parked(const __amdgpu_buffer_rsrc_t pr, const unsigned src, f32x16 (&acc)[TM][TN]) {
  constexpr int NB = (HI - LO) * TN, B = NB < 4 ? NB : 4;
  static_assert(NB % B == 0, "blocks per batch");
#pragma unroll
  for (int b0 = 0; b0 < NB; b0 += B) {
    float4 v[B][4];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) v[b][g] = buf_load4_sc1(pr, src + ((SLOT0 * TN + b0 + b) * 4 + g) * 8192u);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int im = LO + (b0 + b) / TN, in = (b0 + b) % TN;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[im][in][g * 4] += v[b][g].x; acc[im][in][g * 4 + 1] += v[b][g].y;
        acc[im][in][g * 4 + 2] += v[b][g].z; acc[im][in][g * 4 + 3] += v[b][g].w;
      }
    }
  }
}

// epilogue: y = gate(acc * scale + bias + addend), 16 bytes per lane through an LDS transpose (conv_split.hip:
// conv_epilogue_v4), the same arithmetic per element in the same order; two blocks per round of residual / gate loads — the
// two column blocks of one row block (256 contiguous bytes per output row and round).  Row blocks [0, TM / 2) start at output
// row row_a, row blocks [TM / 2, TM) at row_b; the second half only when `both`.  -> max |y| over what was stored
template <int TM, int TN>
__device__ __forceinline__ float big_epilogue(const ConvArgs& a, const f32x16 (&acc)[TM][TN], char* smem, const int row_a,
                                              const int row_b, const int col0, const bool both) {
  static_assert(TN == 2, "a round is the two column blocks of a row block");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile_f = reinterpret_cast<float*>(smem) + wave * (2 * 32 * EPI_STRIDE);
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(a.y, a.y_bytes);
  const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.addend ? a.addend : a.y, a.addend ? a.y_bytes : 0u);
  const __amdgpu_buffer_rsrc_t mr = make_rsrc(a.mask_ref ? a.mask_ref : a.y, a.mask_ref ? a.y_bytes : 0u);
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
  float mx = 0.f;
  float4 scv[2], biv[2];
  bool nvalid[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = col0 + b * 32 + c4;
    nvalid[b] = n < a.Cout;                              // Cout % 4 == 0: the four columns are valid together
    scv[b] = make_float4(1.f, 1.f, 1.f, 1.f);
    biv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale && nvalid[b]) scv[b] = *reinterpret_cast<const float4*>(a.scale + n);
    if (a.bias && nvalid[b]) biv[b] = *reinterpret_cast<const float4*>(a.bias + n);
  }
#pragma unroll
  for (int im = 0; im < TM; ++im) {
    if (im >= TM / 2 && !both) break;
    const int rbase = im < TM / 2 ? row_a + im * 32 : row_b + (im - TM / 2) * 32;
    unsigned offs[2][4];
    float4 ad[2][4], mk[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int m = rbase + pass * 8 + rrow;
        offs[b][pass] = (nvalid[b] && m < a.M) ? ((unsigned)m * (unsigned)a.Cout + (unsigned)(col0 + b * 32 + c4)) * 4u : kBigOOB;
      }
    if (a.addend) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) ad[b][pass] = buf_load4(ar, offs[b][pass]);
    }
    if (a.relu_mode == 2) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) mk[b][pass] = buf_load4(mr, offs[b][pass]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* tl = tile_f + b * (32 * EPI_STRIDE);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) tl[(e + 8 * g + row_hi) * EPI_STRIDE + col_in] = acc[im][b][g * 4 + e];
    }
    // a wave's own data only: no workgroup barrier, the LDS traffic of one wave is ordered
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float* tl = tile_f + b * (32 * EPI_STRIDE);
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + rrow;
        const float4 v4 = *reinterpret_cast<const float4*>(tl + row * EPI_STRIDE + c4);
        const unsigned off = offs[b][pass];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const float s4[4] = {scv[b].x, scv[b].y, scv[b].z, scv[b].w}, b4[4] = {biv[b].x, biv[b].y, biv[b].z, biv[b].w};
        float adv[4] = {0.f, 0.f, 0.f, 0.f}, mkv[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.addend) {
          const float4 tv = ad[b][pass];
          adv[0] = tv.x; adv[1] = tv.y; adv[2] = tv.z; adv[3] = tv.w;
        }
        if (a.relu_mode == 2) {
          const float4 tv = mk[b][pass];
          mkv[0] = tv.x; mkv[1] = tv.y; mkv[2] = tv.z; mkv[3] = tv.w;
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
        if (a.amax_y && off != kBigOOB)
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
    __builtin_amdgcn_wave_barrier();                   // the transpose space is rewritten by the next round
  }
  return mx;
}

// What follows a tile's K loop, shared by the tile shapes: undo the operand scales, meet the other K parts of the tile (if
// the reduction was cut), run the fused epilogue.  row0 / col0: first output row / column of this WAVE's TM x TN blocks.
//
// Two parts (the usual cut) meet SYMMETRICALLY (round 6): part p owns the row blocks [p TM/2, (p + 1) TM/2) of every wave;
// it parks only the blocks it does not own, adds the other part's contribution to the ones it does, and runs the epilogue on
// those alone — half the parked bytes, half the epilogue per workgroup, neither partner idle (before: part 0 parked its
// whole tile and left, part 1 waited, re-read it and stored everything: ~19 - 36% of a launch).  a + b has no order: the
// sums do not depend on which part owns a block.  Part 1 first SWAPS its two register halves (selects on a uniform
// condition), so that "own" is row blocks [0, TM/2) and "foreign" [TM/2, TM) in both parts and every register index below
// is static without a second copy of the code (a park over a run-time choice of registers made the compiler spill the K
// loop's staging registers: the loop runs at the register limit).
// Waiting for the partner must not depend on a workgroup that has not been dispatched (two such launches on two streams
// could hold each other's CUs): a part waits only for a partner that is known to be RESIDENT.  Every workgroup of a two-part
// launch announces itself in the tile's `started` word as its first action (big_announce: one fetch-add whose result, the
// launch's epoch, waits in LDS for the end of the K loop — no atomic round trip is left behind the loop).  A part that
// finds its partner started parks its foreign half, publishes flag = epoch and waits for the partner's flag; one whose
// partner has not even been dispatched parks its own half too, publishes epoch | FULL and leaves — the partner then reads
// that from the flag, adds the whole tile and stores it, as before round 6.  (Both cannot leave: a part that has finished
// has started.)  Nothing is ever reset: per launch and tile `started` moves by two and the flags carry the epoch
// (started / 2 + 1); launches that share the words are ordered on their stream, also when a captured graph replays them.
// sk_counters + 4096 + 4 * tile: started | flag of part 0 | flag of part 1 | unused.
__device__ __forceinline__ unsigned big_announce(const ConvArgs& a, const int tile, const int S) {
  unsigned prev = 0;
  if (S == 2 && !a.big_asym && threadIdx.x == 0)
    prev = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a.sk_counters) + 4096 + 4 * tile, 1u, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
  return prev;
}

template <int TM, int TN, int BN>
__device__ __forceinline__ void big_finish(const ConvArgs& a, const f32x16 (&acc_t)[TM][TN], char* smem, const int tile,
                                           const int part, const int S, const int row0, const int col0, const int ea,
                                           const int eb, const unsigned* meet_word) {
  const int wtile = tile - a.big_body;      // the tile's place in the parked-part workspace (only cut tiles have one)
  const int t = threadIdx.x;
  constexpr int H = TM / 2;
  // From here on the sums live in NEW register tuples: the (vector) multiplication that undoes the operand scales defines
  // them, and nothing below writes into the K loop's tuples again.  The loop runs at the register limit (128 accumulator +
  // 48 fragment + 32 staging registers); a tuple that stays live through everything behind the loop is, to the register
  // allocator, a long and therefore cheap range — with the symmetric meeting's extra code behind the loop it took two
  // accumulator blocks out of the loop's registers (reloaded and stored again around every multiply segment: the step went
  // from 12 to 20 ms).  Ending the loop's tuples here keeps them short and hot.  (Tuples, not 128 scalars: a park stores
  // four consecutive registers per instruction, and scalars scattered by the allocator cost three times the park's time.)
  f32x16 acc[TM][TN];
  {
    // undo the operand scales (exact: powers of two, in two steps so that no intermediate leaves fp32's range unless the
    // result does): the parts of a split reduction are parked in true units
    const int tt = -(ea + eb);
    const float u1 = pow2f(tt / 2), u2 = pow2f(tt - tt / 2);
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in) acc[im][in] = (acc_t[im][in] * u1) * u2;
  }

  nf_check<TM, TN>(acc, a.nf_flag, a.launch_id);
  BIG_STAMP(2);
  int* s_word = reinterpret_cast<int*>(smem);        // the operand planes are dead (every wave passed the last barrier)
  int row_a = row0, row_b = row0 + H * 32;           // output rows of the register halves [0, H) and [H, TM)
  bool both = true;                                  // this workgroup stores both halves
  auto drain_and_meet = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave drains its stores ...
    __syncthreads();                                    // ... before one lane announces them
  };
  if (S == 2 && !a.big_asym) {
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(a.sk_ws + (size_t)wtile * 2 * (256 * BN), (unsigned)(2 * 256 * BN * 4));
    unsigned* cnt = reinterpret_cast<unsigned*>(a.sk_counters) + 4096 + 4 * tile;   // started | flag 0 | flag 1
    const unsigned mine = (unsigned)part * (256 * BN * 4) + (unsigned)t * 16u;
    const unsigned theirs = (unsigned)(part ^ 1) * (256 * BN * 4) + (unsigned)t * 16u;
    constexpr unsigned kFull = 0x80000000u;
    const unsigned epoch = *meet_word / 2u + 1u;
    if (part != 0) {      // own half first: registers [0, H) <-> [H, TM)
#pragma unroll
      for (int im = 0; im < H; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          const f32x16 lo = acc[im][in];
          acc[im][in] = acc[im + H][in];
          acc[im + H][in] = lo;
        }
      row_a = row0 + H * 32;
      row_b = row0;
    }
    big_park<TM, TN, H, TM, 0>(pr, mine, acc);              // the foreign half: slots [0, H) of this part's area
    // (the look at `started` travels with the stores)
    if (t == 0) s_word[0] = (int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_and_meet();
    if ((unsigned)s_word[0] != 2u * epoch) {
      // the partner has not been dispatched: leave the own half as well (slots [H, TM)) and go
      big_park<TM, TN, 0, H, H>(pr, mine, acc);
      drain_and_meet();
      if (t == 0) __hip_atomic_store(cnt + 1 + part, epoch | kFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      BIG_STAMP(5);
      return;
    }
    if (t == 0) {
      __hip_atomic_store(cnt + 1 + part, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the partner is resident: in its K loop or past it
      unsigned f;
      int spins = 0;
      while (((f = __hip_atomic_load(cnt + 1 + (part ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~kFull) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins == (1 << 22)) {     // seconds, where microseconds are expected: a defect — reported by the guard's poll
          if (a.nf_flag) {              // (dadet_nonfinite_poll names this launch) instead of a hung queue
            atomicCAS(a.nf_flag, 0u, a.launch_id + 1u);
            atomicAdd(a.nf_flag + 1, 1u);
          }
          f = epoch;
          break;
        }
      }
      s_word[1] = (int)f;
    }
    __syncthreads();
    if (((unsigned)s_word[1] & kFull) == 0u) {
      big_add_parked<TM, TN, 0, H, 0>(pr, theirs, acc);     // the partner's foreign half is this part's own
      both = false;
    } else {
      // the partner saw nobody when it finished and left its whole tile: its foreign half (slots [0, H)) is this part's
      // own, its own half (slots [H, TM)) this part's foreign one
      big_add_parked<TM, TN, 0, TM, 0>(pr, theirs, acc);
    }
    __syncthreads();      // s_word is about to be reused as transpose space
  } else if (S > 1) {
    const __amdgpu_buffer_rsrc_t pr = make_rsrc(a.sk_ws + (size_t)wtile * S * (256 * BN), (unsigned)(S * 256 * BN * 4));
    int* arrive = a.sk_counters + tile;
    int* parked = a.sk_counters + 2048 + tile;
    if (t == 0) s_word[0] = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = s_word[0];
    if (ticket != S - 1) {
      big_park<TM, TN, 0, TM, 0>(pr, (unsigned)part * (256 * BN * 4) + (unsigned)t * 16u, acc);
      drain_and_meet();
      if (t == 0) __hip_atomic_fetch_add(parked, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      BIG_STAMP(5);
      return;
    }
    // the last part to arrive: the others have left their K loops (they hold tickets) and only finish their stores
    if (t == 0) {
      while (__hip_atomic_load(parked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S - 1) __builtin_amdgcn_s_sleep(8);
      *arrive = 0;        // ready for the next launch on this stream (no memset per launch)
      *parked = 0;
    }
    __syncthreads();
    // sum in part order, this part's registers at its own index: the result does not depend on who came last
    if (S == 2) {
      big_add_parked<TM, TN, 0, TM, 0>(pr, (unsigned)(part ^ 1) * (256 * BN * 4) + (unsigned)t * 16u, acc);
    } else
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        float4 own4[4], sum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          own4[g] = make_float4(acc[im][in][g * 4], acc[im][in][g * 4 + 1], acc[im][in][g * 4 + 2], acc[im][in][g * 4 + 3]);
          sum[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int p = 0; p < S; ++p) {
          float4 v[4];
          if (p != part) {
            const unsigned src = (unsigned)p * (256 * BN * 4) + (unsigned)t * 16u + ((im * TN + in) * 4) * 8192u;
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = buf_load4_sc1(pr, src + g * 8192u);
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = own4[g];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            sum[g].x += v[g].x; sum[g].y += v[g].y; sum[g].z += v[g].z; sum[g].w += v[g].w;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[im][in][g * 4] = sum[g].x; acc[im][in][g * 4 + 1] = sum[g].y;
          acc[im][in][g * 4 + 2] = sum[g].z; acc[im][in][g * 4 + 3] = sum[g].w;
        }
      }
    __syncthreads();      // s_word is about to be reused as transpose space
  }

  BIG_STAMP(3);
  const float mx = big_epilogue<TM, TN>(a, acc, smem, row_a, row_b, col0, both);
  if (a.amax_y) amax_publish(a.amax_y, mx);
  BIG_STAMP(4);
}

template <int BN>
__global__ __launch_bounds__(512, 2) void conv_big_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(BN == 256, "wave layout below is 2 x 4 waves of 128 x 64");
  BIG_STAMP(0);
  constexpr int TM = 4, TN = 2;
  constexpr int kPlane = 256 * 64;          // one fp16 plane of 256 rows x 32 k
  constexpr int kStage = 4 * kPlane;        // A_h | A_l | B_h | B_l
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: everything derived from it stays scalar
  const int grp = wave >> 2, tg = t & 255;
  const int wm = wave >> 2, wn = wave & 3;
  const int T = a.tiles_m * a.tiles_n;
  // workgroup -> (K part, tile): hardware deals workgroup b to XCD b % 8; the remap gives every XCD a contiguous range of
  // (part, tile) pairs, i.e. neighbouring tiles of one K range, which share operand panels in that XCD's L2
  int S, part, tile;
  big_place(a, T, S, part, tile);
  const int bm0 = (tile / a.tiles_n) * 256, bn0 = (tile % a.tiles_n) * BN;
  const unsigned announced = big_announce(a, tile, S);
  unsigned* meet_word = reinterpret_cast<unsigned*>(smem + 2 * kStage + 16 * 256 * sizeof(int));   // behind the row descriptors
  const int nk = a.K / 32;
  const int kt_lo = (int)(((long long)nk * part) / S), kt_hi = (int)(((long long)nk * (part + 1)) / S);
  const int nT = kt_hi - kt_lo;

  const int ea = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
  const int eb = a.amax_w ? fmt4_exp(amax_read(a.amax_w)) : 0;
  const float sc = pow2f(grp ? eb : ea);

  // ---- staging state: 8 lanes per 128-byte row, 32 rows per pass, 4 passes = one QUARTER of a K-tile (half of the
  // group's operand: 128 rows).  Quarter q of a group = (K-tile q / 2 of the part, half q % 2).
  const int lcol = tg & 7, lrow = tg >> 3;
  const __amdgpu_buffer_rsrc_t rr = grp ? make_rsrc(a.w, a.w_bytes) : make_rsrc(a.x, a.x_bytes);
  unsigned goff[2][4];          // byte offset of this thread's 16 bytes in row (half, pass) at the load stream's tap
  // group 0's row descriptors (first pixel of the row's image, hi0 << 16 | wi0 & 0xffff) live in LDS behind the two
  // K-tile slots: they are needed only when the load stream crosses into the next filter tap
  int* desc = reinterpret_cast<int*>(smem + 2 * kStage) + tg;      // [16][256] ints, element k at desc[k * 256]
  // load stream: K-tile index (absolute), channel offset inside the tap, tap row / column
  int ld_kt = kt_lo, ld_kc, ld_kr, ld_ks;
  {
    const int k0 = kt_lo * 32;
    const int tap = k0 / a.Cin;
    ld_kc = k0 - tap * a.Cin;
    ld_kr = tap / a.KW;
    ld_ks = tap - ld_kr * a.KW;
  }
  auto recompute = [&]() {      // group 0: offsets of the eight rows at tap (ld_kr, ld_ks)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int px = desc[(h * 4 + i) * 256], hw = desc[(8 + h * 4 + i) * 256];
        const int hi = (hw >> 16) + ld_kr, wi = (int)(short)(hw & 0xffff) + ld_ks;
        const bool ok = ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);   // no short circuit: no branches
        goff[h][i] = ok ? ((unsigned)(px + hi * a.W + wi) * (unsigned)a.Cin + (unsigned)(lcol * 4)) * 4u : kBigOOB;
      }
  };
  if (grp == 0) {
    const int HoWo = a.Ho * a.Wo;
    const float r_howo = 1.0f / (float)HoWo, r_wo = 1.0f / (float)a.Wo;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = bm0 + h * 128 + lrow + 32 * i;
        int px = 0, hw = (int)0xC0000000;     // hi0 = -16384: no tap brings an invalid row inside the map
        if (m < a.M) {
          int rem, wo;
          const int img = div_small(m, HoWo, r_howo, rem);
          const int ho = div_small(rem, a.Wo, r_wo, wo);
          px = img * a.H * a.W;
          hw = (int)(((unsigned)(ho * a.stride - a.pad) << 16) | ((unsigned)(wo * a.stride - a.pad) & 0xffffu));
        }
        desc[(h * 4 + i) * 256] = px;
        desc[(8 + h * 4 + i) * 256] = hw;
      }
    recompute();                // (a thread reads back its own words only: no barrier needed)
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = bn0 + h * 128 + lrow + 32 * i;
        goff[h][i] = n < a.Cout ? ((unsigned)n * (unsigned)a.K + (unsigned)(lcol * 4)) * 4u : kBigOOB;
      }
  }
  // this thread's 8-byte piece inside a plane, pass i of half h at + (h * 128 + 32 * i) * 64
  const unsigned wofs = (grp ? 2 * kPlane : 0) + lrow * 64 + ((((lcol >> 1) ^ ((lrow >> 2) & 3))) << 4) + (lcol & 1) * 8;
  const int fr = lane & 31;
  const unsigned fo = fr * 64 + ((((lane >> 5)) ^ ((fr >> 2) & 3)) << 4);
  const unsigned fa_base = (wm * 128) * 64 + fo;
  const unsigned fb_base = 2 * kPlane + (wn * 64) * 64 + fo;

  // two staging buffers: a load segment first issues the global loads of the quarter after next into one of them, then
  // splits and stores the other (fetched one load segment — a multiply segment and a half — earlier)
  float4 raw[2][4];
  f16x8 fa[2][TM], fb[2][TN];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fetch the load stream's next quarter; after a second half the stream moves on by one K-tile.  Past the part's last
  // K-tile the stream keeps walking (the data lands in a slot nobody reads again); offsets beyond a buffer return zeros.
  auto loads = [&](float4 (&dst)[4], const int half) {
    const int soff = grp ? ld_kt * 128 : ld_kc * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(half ? goff[1][i] : goff[0][i]), soff, 0));
    if (half) {
      ++ld_kt;
      ld_kc += 32;
      if (ld_kc >= a.Cin) {
        ld_kc = 0;
        if (++ld_ks == a.KW) { ld_ks = 0; ++ld_kr; }
        if (grp == 0) recompute();
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // the loads go out FIRST in their segment
  };
  auto stage = [&](const float4 (&v)[4], const int slot, const int half) {
    char* st = smem + slot * kStage + wofs + half * (128 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 h, l;
      split4m(v[i], sc, h, l);
      *reinterpret_cast<uint2*>(st + i * 2048) = h;
      *reinterpret_cast<uint2*>(st + kPlane + i * 2048) = l;
    }
  };
  auto read_frags = [&](const int slot, const int step) {
    const char* cur = smem + slot * kStage;
    // in the order the MFMAs want them: l_a h_b first
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const f16x8*>(cur + ((fb_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[1][i] = *reinterpret_cast<const f16x8*>(cur + kPlane + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f16x8*>(cur + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[1][i] = *reinterpret_cast<const f16x8*>(cur + kPlane + ((fb_base + i * 2048) ^ (step * 32)));
  };
  // multiply segment: the 24 MFMAs of one k16 step and NOTHING else (smallest cross terms first: l_a h_b, h_a l_b, h_a h_b).
  // (Measured, round 5: the split's VALU placed behind these MFMAs instead of in the load segment — 0.464 -> 0.445 of the
  // ceiling on 16384 x 4096 x 4096: a wave's own VALU between its MFMAs costs the pipe more than the partner's.)
  auto mfma_seg = [&]() {
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa][im], fb[pb][in], acc[im][in], 0, 0, 0);
    }
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  // Schedule (quarters q of a group's own operand; K-tile T of the part in slot T & 1):
  //   group 0 multiplies in even segments; its load segment 2v + 1 stores quarter v + 3 and fetches quarter v + 4
  //   group 1 multiplies in odd segments;  its load segment 2v     stores quarter v + 2 and fetches quarter v + 3
  // prologue: quarters 0 .. 2 (group 0) / 0 .. 1 (group 1) stored synchronously, the next one in flight
  // (all of them fetched up front — one trip to memory instead of three; the accumulators are not live yet)
  {
    float4 q0[4], q1[4];
    loads(q0, 0);
    loads(q1, 1);
    loads(raw[0], 0);
    if (grp == 0) loads(raw[1], 1);
    stage(q0, 0, 0);
    stage(q1, 0, 1);
    if (grp == 0) stage(raw[0], 1, 0);
  }
  if (t == 0) *meet_word = announced;
  bar();
  BIG_STAMP(1);
  if (grp == 0) {
    read_frags(0, 0);
    for (int Tt = 0; Tt < nT; ++Tt) {
      const int cur = Tt & 1;
      mfma_seg();                                                      // segment 4T: step 2T
      bar();
      loads(raw[0], 0); read_frags(cur, 1); stage(raw[1], cur ^ 1, 1);  // segment 4T + 1: second half of K-tile T + 1
      bar();
      mfma_seg();                                                      // segment 4T + 2: step 2T + 1
      bar();
      loads(raw[1], 1); read_frags(cur ^ 1, 0); stage(raw[0], cur, 0);  // segment 4T + 3: first half of K-tile T + 2
      bar();
    }
  } else {
    for (int Tt = 0; Tt < nT; ++Tt) {
      const int cur = Tt & 1;
      loads(raw[1], 1); read_frags(cur, 0); stage(raw[0], cur ^ 1, 0);  // segment 4T: first half of K-tile T + 1
      bar();
      mfma_seg();                                                      // segment 4T + 1: step 2T
      bar();
      loads(raw[0], 0); read_frags(cur, 1); stage(raw[1], cur ^ 1, 1);  // segment 4T + 2: second half of K-tile T + 1
      bar();
      mfma_seg();                                                      // segment 4T + 3: step 2T + 1
      bar();
    }
  }

  big_finish<TM, TN, BN>(a, acc, smem, tile, part, S, bm0 + wm * (TM * 32), bn0 + wn * (TN * 32), ea, eb, meet_word);
}

// ---- 256 x 128 tile: the layers with 128 or 256 output channels (res3 / res4 3x3 and their 1x1 neighbours) --------------
// The same idea with another geometry: 8 waves as 4 x 2 of 64 x 64 (64 accumulator registers), and a multiply segment
// covers a WHOLE K-tile (both k16 steps: 24 MFMAs, their 16 fragments in 64 registers), so a K-tile costs two barriers
// instead of four.  Segment s: group 0 multiplies K-tile s / 2 for even s, group 1 K-tile (s - 1) / 2 for odd s.  The
// 256 + 128 operand rows of a K-tile are staged in six passes of 32 rows per group and load segment: group 0 takes
// activation rows 0 .. 191, group 1 activation rows 192 .. 255 and the 128 weight rows.  Group 0's load segment 2v + 1
// stores K-tile v + 2 (into the slot K-tile v was last read from in segment 2v) and fetches K-tile v + 3; group 1's load
// segment 2v stores K-tile v + 1 and fetches K-tile v + 2.
__global__ __launch_bounds__(512, 2) void conv_big128_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 128, TM = 2, TN = 2;
  constexpr int kPlaneA = 256 * 64, kPlaneB = 128 * 64;
  constexpr int kStage = 2 * kPlaneA + 2 * kPlaneB;       // A_h | A_l | B_h | B_l = 48 KB
  BIG_STAMP(0);
#if DADET_BIG_TIMING
  const int abl = a.ablate;      // probe builds: DADET_ABLATE bit 1 no global loads, 2 no split + store, 4 no fragment reads, 8 no MFMA
#else
  constexpr int abl = 0;
#endif
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, tg = t & 255;
  const int wm = wave >> 1, wn = wave & 1;
  const int T = a.tiles_m * a.tiles_n;
  int S, part, tile;
  big_place(a, T, S, part, tile);
  const int bm0 = (tile / a.tiles_n) * 256, bn0 = (tile % a.tiles_n) * BN;
  const unsigned announced = big_announce(a, tile, S);
  unsigned* meet_word = reinterpret_cast<unsigned*>(smem + 2 * kStage + 12 * 512 * sizeof(int));   // behind the row descriptors
  const int nk = a.K / 32;
  const int kt_lo = (int)(((long long)nk * part) / S), kt_hi = (int)(((long long)nk * (part + 1)) / S);
  const int nT = kt_hi - kt_lo;
  const int ea = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
  const int eb = a.amax_w ? fmt4_exp(amax_read(a.amax_w)) : 0;
  const float sa = pow2f(ea), sb = pow2f(eb);

  // six row slots per thread: operand row rho = 192 * grp + 32 * i + lrow of the 384 (activation rows first)
  const int lcol = tg & 7, lrow = tg >> 3;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes), wr = make_rsrc(a.w, a.w_bytes);
  unsigned goff[6];
  int* desc = reinterpret_cast<int*>(smem + 2 * kStage) + t;       // [12][512] ints: first pixel, hi0 << 16 | wi0
  // (indexed by the THREAD, not by its index inside the group: both groups own activation rows here)
  int ld_kt = kt_lo, ld_kc, ld_kr, ld_ks;
  {
    const int k0 = kt_lo * 32;
    const int tap = k0 / a.Cin;
    ld_kc = k0 - tap * a.Cin;
    ld_kr = tap / a.KW;
    ld_ks = tap - ld_kr * a.KW;
  }
  auto is_act = [&](const int i) { return grp == 0 || i < 2; };     // wave-uniform
  auto recompute = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (is_act(i)) {
        const int px = desc[i * 512], hw = desc[(6 + i) * 512];
        const int hi = (hw >> 16) + ld_kr, wi = (int)(short)(hw & 0xffff) + ld_ks;
        const bool ok = ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
        goff[i] = ok ? ((unsigned)(px + hi * a.W + wi) * (unsigned)a.Cin + (unsigned)(lcol * 4)) * 4u : kBigOOB;
      }
  };
  {
    const int HoWo = a.Ho * a.Wo;
    const float r_howo = 1.0f / (float)HoWo, r_wo = 1.0f / (float)a.Wo;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (is_act(i)) {
        const int m = bm0 + 192 * grp + 32 * i + lrow;
        int px = 0, hw = (int)0xC0000000;
        if (m < a.M) {
          int rem, wo;
          const int img = div_small(m, HoWo, r_howo, rem);
          const int ho = div_small(rem, a.Wo, r_wo, wo);
          px = img * a.H * a.W;
          hw = (int)(((unsigned)(ho * a.stride - a.pad) << 16) | ((unsigned)(wo * a.stride - a.pad) & 0xffffu));
        }
        desc[i * 512] = px;
        desc[(6 + i) * 512] = hw;
      } else {
        const int n = bn0 + 32 * (i - 2) + lrow;
        goff[i] = n < a.Cout ? ((unsigned)n * (unsigned)a.K + (unsigned)(lcol * 4)) * 4u : kBigOOB;
      }
    }
    recompute();
  }
  const unsigned swz = ((((lcol >> 1) ^ ((lrow >> 2) & 3))) << 4) + (lcol & 1) * 8;
  const int fr = lane & 31;
  const unsigned fo = fr * 64 + ((((lane >> 5)) ^ ((fr >> 2) & 3)) << 4);
  const unsigned fa_base = (wm * 64) * 64 + fo;
  const unsigned fb_base = 2 * kPlaneA + (wn * 64) * 64 + fo;

  float4 raw[2][6];
  f16x8 fa[2][2][TM], fb[2][2][TN];     // [k16 step][plane][block]
  f32x16 acc[TM][TN];
#if DADET_BIG_TIMING
  for (int i = 0; i < 12; ++i) (&raw[0][0])[i] = make_float4(1.f, 2.f, 3.f, 4.f);     // (defined values for the ablated stages)
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) { (&fa[0][0][0])[i][e] = (_Float16)1; (&fb[0][0][0])[i][e] = (_Float16)1; }
#endif
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto loads = [&](float4 (&dst)[6]) {        // one whole K-tile of this group's rows, then the stream moves on
    if (!(abl & 1))
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bool act = is_act(i);
      dst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(act ? xr : wr, (int)goff[i],
                                                                                 act ? ld_kc * 4 : ld_kt * 128, 0));
    }
    ++ld_kt;
    ld_kc += 32;
    if (ld_kc >= a.Cin) {
      ld_kc = 0;
      if (++ld_ks == a.KW) { ld_ks = 0; ++ld_kr; }
      recompute();
    }
    __builtin_amdgcn_sched_barrier(0);      // the loads go out FIRST in their segment
  };
  auto stage = [&](const float4 (&v)[6], const int slot) {
    if (!(abl & 2))
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bool act = is_act(i);
      const int row = act ? 192 * grp + 32 * i + lrow : 32 * (i - 2) + lrow;
      char* st = smem + slot * kStage + (act ? 0 : 2 * kPlaneA) + row * 64 + swz;
      uint2 h, l;
      split4m(v[i], act ? sa : sb, h, l);
      *reinterpret_cast<uint2*>(st) = h;
      *reinterpret_cast<uint2*>(st + (act ? kPlaneA : kPlaneB)) = l;
    }
  };
  auto read_frags = [&](const int slot) {
    const char* cur = smem + slot * kStage;
    if (!(abl & 4))
#pragma unroll
    for (int step = 0; step < 2; ++step) {
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[step][0][i] = *reinterpret_cast<const f16x8*>(cur + ((fb_base + i * 2048) ^ (step * 32)));
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[step][1][i] = *reinterpret_cast<const f16x8*>(cur + kPlaneA + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[step][0][i] = *reinterpret_cast<const f16x8*>(cur + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[step][1][i] = *reinterpret_cast<const f16x8*>(cur + kPlaneB + ((fb_base + i * 2048) ^ (step * 32)));
    }
  };
  auto mfma_seg = [&]() {
    if (!(abl & 8))
#pragma unroll
    for (int step = 0; step < 2; ++step)
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in = 0; in < TN; ++in)
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[step][pa][im], fb[step][pb][in], acc[im][in], 0, 0, 0);
      }
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: K-tile 0 stored by both groups, K-tile 1 by group 0 (its segment "-1"); group 0 has K-tile 2 on its way
  // (raw[0]), group 1 K-tile 1 (raw[1])
  // (all fetched up front: one trip to memory)
  {
    float4 q0[6];
    loads(q0);
    loads(raw[1]);
    if (grp == 0) loads(raw[0]);
    stage(q0, 0);
    if (grp == 0) stage(raw[1], 1);
  }
  if (t == 0) *meet_word = announced;
  bar();
  BIG_STAMP(1);
  if (grp == 0) {
    read_frags(0);
    for (int Tt = 0; Tt < nT; Tt += 2) {
      mfma_seg();                                               // segment 2T: K-tile T (even T)
      bar();
      loads(raw[1]); read_frags(1); stage(raw[0], 0);           // segment 2T + 1: K-tile T + 2 stored, T + 3 fetched
      bar();
      if (Tt + 1 < nT) {
        mfma_seg();                                             // K-tile T + 1
        bar();
        loads(raw[0]); read_frags(0); stage(raw[1], 1);
        bar();
      }
    }
  } else {
    for (int Tt = 0; Tt < nT; Tt += 2) {
      loads(raw[0]); read_frags(0); stage(raw[1], 1);           // segment 2T: K-tile T + 1 stored, T + 2 fetched (even T)
      bar();
      mfma_seg();                                               // segment 2T + 1: K-tile T
      bar();
      if (Tt + 1 < nT) {
        loads(raw[1]); read_frags(1); stage(raw[0], 0);
        bar();
        mfma_seg();
        bar();
      }
    }
  }
  big_finish<TM, TN, BN>(a, acc, smem, tile, part, S, bm0 + wm * (TM * 32), bn0 + wn * (TN * 32), ea, eb, meet_word);
}

// ---- weight gradient, 256 x 256 tile ------------------------------------------------------------------------------------
// dW[co][kk] = sum_m gY[m][co] * Xg[m][kk], kk = (r, s, ci): the same pipeline with the reduction over the pixels m.  Both
// operands arrive m-major (a row of gY is Cout contiguous floats, a gathered activation row Cin contiguous floats), the
// MFMA wants each lane's 8 reduction indices contiguous: a thread fetches a 4 (m) x 4 (channel) block — four 16-byte row
// loads, 32 lanes side by side cover 512 contiguous bytes of a row — and writes, per channel, the four m values as ONE 8-byte
// piece of that channel's plane row: the planes are [channel][32 m] fp16, the image the forward kernel reads its
// fragments from.  Group 0 stages gY (rows m, channels co0 ..), group 1 the gathered activations (channels kk0 .. map to a
// filter tap and an input channel once per thread; the four rows' pixel coordinates advance by 32 rows per K-tile).  A
// quarter is (K-tile of 32 rows, 128 channels).  The reduction over M is cut into `splits` ranges of whole K-tiles; every
// part writes its tile of [splits][Cout][K] partial sums (or dW itself when there is one part) — the layout
// wgrad_reduce_batch_kernel sums, with the FrozenBN scale, in its deterministic split order.
// (`bid`: the workgroup's index inside ITS problem — blockIdx.x, or that minus the problem's first workgroup in a grouped launch)
__device__ __forceinline__ void wgrad_big_body(const WgradArgs& a, const int bid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = 4, TN = 2;
  constexpr int kPlane = 256 * 64;
  constexpr int kStage = 4 * kPlane;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, tg = t & 255;
  const int wm = wave >> 2, wn = wave & 3;
  const int T = a.tiles_co * a.tiles_kc;
  const int lid = xcd_remap(bid, T * a.splits);
  const int part = lid / T, tile = lid - part * T;      // one XCD: one range of rows, neighbouring tiles
  const int co0 = (tile / a.tiles_kc) * 256, kk0 = (tile % a.tiles_kc) * 256;
  const int m_begin = part * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int nT = (m_end - m_begin + 31) / 32;
  const int eg = a.amax_gy ? fmt4_exp(amax_read(a.amax_gy)) : 0;
  const int ex = a.amax_x ? fmt4_exp(amax_read(a.amax_x)) : 0;
  const float sc = pow2f(grp ? ex : eg);

  const int mg = tg & 7, cq = tg >> 3;       // rows 4 mg .. 4 mg + 3 of the K-tile, channels 4 cq .. 4 cq + 3 of the half
  const __amdgpu_buffer_rsrc_t rr = grp ? make_rsrc(a.x, a.x_bytes) : make_rsrc(a.gy, a.gy_bytes);
  // group 0: byte offset of (row 4 mg + j of K-tile 0 of the part, this thread's channel quad of half h)
  unsigned goff[2][4];
  // group 1: per half the tap (r << 8 | s) and the byte offset of the input channel, per row the image's first pixel, the
  // output coordinates (ho << 16 | wo) and the input coordinates of tap (0, 0) (hi0 << 16 | wi0 & 0xffff)
  int tap[2];
  unsigned cioff[2];
  // (both in LDS behind the two K-tile slots, as two int4 per thread: they are live only inside a load segment, and the
  // multiply segments need every register — with them in VGPRs the loop spilled six values per K-tile)
  int4* rowst = reinterpret_cast<int4*>(smem + 2 * kStage) + tg;      // [2][256] int4: pix[0..3] | hw0[0..3]
  int ld_m = m_begin;                        // first row of the load stream's K-tile
  const int HoWo = a.Ho * a.Wo;
  // 32 output pixels further, in input coordinates of tap (0, 0): (d_img images, d_hs rows, d_ws columns) with carries at
  // wi0 >= w_lim (wo >= Wo) and hi0 >= h_lim (ho >= Ho)
  const int d_img = 32 / HoWo, d_ho = (32 - d_img * HoWo) / a.Wo, d_wo = 32 - d_img * HoWo - d_ho * a.Wo;
  const int d_hs = d_ho * a.stride, d_ws = d_wo * a.stride, w_span = a.Wo * a.stride, h_span = a.Ho * a.stride;
  const int w_lim = w_span - a.pad, h_lim = h_span - a.pad;
  if (grp == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = co0 + 128 * h + 4 * cq;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        goff[h][j] = c < a.Cout ? ((unsigned)(4 * mg + j) * (unsigned)a.gy_ld + (unsigned)c) * 4u : kBigOOB;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = kk0 + 128 * h + 4 * cq;
      const int tp = kk / a.Cin;
      const int r = tp / a.KW;
      tap[h] = kk < a.K ? ((r << 8) | (tp - r * a.KW)) : (1 << 20);        // an invalid column: a tap far outside
      cioff[h] = (unsigned)(kk - tp * a.Cin) * 4u;
    }
    const float r_howo = 1.0f / (float)HoWo, r_wo = 1.0f / (float)a.Wo;
    int pix[4], hw0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m_begin + 4 * mg + j;      // < 2^24 (checked by the dispatcher)
      int rem, wo;
      const int img = div_small(m, HoWo, r_howo, rem);
      const int ho = div_small(rem, a.Wo, r_wo, wo);
      pix[j] = img * a.H * a.W;
      hw0[j] = (int)(((unsigned)(ho * a.stride - a.pad) << 16) | ((unsigned)(wo * a.stride - a.pad) & 0xffffu));
    }
    rowst[0] = make_int4(pix[0], pix[1], pix[2], pix[3]);
    rowst[256] = make_int4(hw0[0], hw0[1], hw0[2], hw0[3]);
  }
  const int fr = lane & 31;
  const unsigned fo = fr * 64 + ((((lane >> 5)) ^ ((fr >> 2) & 3)) << 4);
  const unsigned fa_base = (wm * 128) * 64 + fo;
  const unsigned fb_base = 2 * kPlane + (wn * 64) * 64 + fo;
  // plane row of channel c' of the quad: 128 h + 4 cq + c'; its 16-byte chunks are swizzled with (row >> 2) & 3 = cq & 3
  const unsigned wofs = (grp ? 2 * kPlane : 0) + (4 * cq) * 64 + ((((mg >> 1) ^ (cq & 3))) << 4) + (mg & 1) * 8;

  float4 raw[2][4];
  f16x8 fa[2][TM], fb[2][TN];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fetch half h (128 channels) of the load stream's K-tile; after the second half the stream moves on by 32 rows.  Rows
  // at or beyond M lie beyond the buffers (zeros); rows of the NEXT part are fetched only past this part's last K-tile,
  // into a slot nobody reads again.
  auto loads = [&](float4 (&dst)[4], const int h) {
    if (grp == 0) {
      const int soff = (ld_m - m_begin) * a.gy_ld * 4 + m_begin * a.gy_ld * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(h ? goff[1][j] : goff[0][j]), soff, 0));
    } else {
      const int tp = h ? tap[1] : tap[0];
      const unsigned cb = h ? cioff[1] : cioff[0];
      const int r = tp >> 8, s2 = tp & 0xff;
      const int4 pv = rowst[0], hv = rowst[256];
      int pix[4] = {pv.x, pv.y, pv.z, pv.w}, hw0[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hi = (hw0[j] >> 16) + r, wi = (int)(short)(hw0[j] & 0xffff) + s2;
        const bool ok = ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
        const unsigned off = (unsigned)(pix[j] + hi * a.W + wi) * (unsigned)(a.Cin * 4) + cb;
        dst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(ok ? off : kBigOOB), 0, 0));
      }
      if (h) {      // the four rows move on by 32 output pixels: branch-free mixed-radix add of (d_img, d_ho, d_wo)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int wi = (int)(short)(hw0[j] & 0xffff) + d_ws, hi = (hw0[j] >> 16) + d_hs;
          const bool cw = wi >= w_lim;
          wi -= cw ? w_span : 0;
          hi += cw ? a.stride : 0;
          const bool ch = hi >= h_lim;
          hi -= ch ? h_span : 0;
          pix[j] += (d_img + (ch ? 1 : 0)) * a.H * a.W;
          hw0[j] = (int)(((unsigned)hi << 16) | ((unsigned)wi & 0xffffu));
        }
        rowst[0] = make_int4(pix[0], pix[1], pix[2], pix[3]);
        rowst[256] = make_int4(hw0[0], hw0[1], hw0[2], hw0[3]);
      }
    }
    if (h) ld_m += 32;
    __builtin_amdgcn_sched_barrier(0);      // the loads go out FIRST in their segment
  };
  // 4 x 4 register transpose (a renaming), split, one 8-byte store per channel and plane.  (The two rows a 16-lane store
  // group touches are 4 apart — the same half of the store path's 128-byte bank window, a two-way conflict on 8 stores per
  // segment: ~16 cycles; steering odd channel quads through the orders 1 0 3 2 costs 16 selects, more than it returns.)
  auto stage = [&](const float4 (&v)[4], const int slot, const int h) {
    char* st = smem + slot * kStage + wofs + h * (128 * 64);
    const float4 cols[4] = {make_float4(v[0].x, v[1].x, v[2].x, v[3].x), make_float4(v[0].y, v[1].y, v[2].y, v[3].y),
                            make_float4(v[0].z, v[1].z, v[2].z, v[3].z), make_float4(v[0].w, v[1].w, v[2].w, v[3].w)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint2 hh, ll;
      split4m(cols[c], sc, hh, ll);
      *reinterpret_cast<uint2*>(st + c * 64) = hh;
      *reinterpret_cast<uint2*>(st + kPlane + c * 64) = ll;
    }
  };
  auto read_frags = [&](const int slot, const int step) {
    const char* cur = smem + slot * kStage;
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const f16x8*>(cur + ((fb_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[1][i] = *reinterpret_cast<const f16x8*>(cur + kPlane + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f16x8*>(cur + ((fa_base + i * 2048) ^ (step * 32)));
#pragma unroll
    for (int i = 0; i < TN; ++i) fb[1][i] = *reinterpret_cast<const f16x8*>(cur + kPlane + ((fb_base + i * 2048) ^ (step * 32)));
  };
  auto mfma_seg = [&]() {
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa][im], fb[pb][in], acc[im][in], 0, 0, 0);
    }
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  // the schedule of conv_big_kernel (quarters = halves of the group's 256 channels)
  {
    float4 q0[4], q1[4];      // all fetched up front: one trip to memory
    loads(q0, 0);
    loads(q1, 1);
    loads(raw[0], 0);
    if (grp == 0) loads(raw[1], 1);
    stage(q0, 0, 0);
    stage(q1, 0, 1);
    if (grp == 0) stage(raw[0], 1, 0);
  }
  bar();
  if (grp == 0) {
    read_frags(0, 0);
    for (int Tt = 0; Tt < nT; ++Tt) {
      const int cur = Tt & 1;
      mfma_seg();
      bar();
      loads(raw[0], 0); read_frags(cur, 1); stage(raw[1], cur ^ 1, 1);
      bar();
      mfma_seg();
      bar();
      loads(raw[1], 1); read_frags(cur ^ 1, 0); stage(raw[0], cur, 0);
      bar();
    }
  } else {
    for (int Tt = 0; Tt < nT; ++Tt) {
      const int cur = Tt & 1;
      loads(raw[1], 1); read_frags(cur, 0); stage(raw[0], cur ^ 1, 0);
      bar();
      mfma_seg();
      bar();
      loads(raw[0], 0); read_frags(cur, 1); stage(raw[1], cur ^ 1, 1);
      bar();
      mfma_seg();
      bar();
    }
  }

  // ---- epilogue: undo the scales, turn every 32 x 32 block through LDS, 16-byte stores along kk.  The scaled sums are
  // tuples of their own (big_finish: the K loop's accumulator tuples end at this multiplication, so that the register
  // allocator sees them as short, hot ranges of the loop and keeps its spills out of it)
  f32x16 res[TM][TN];
  {
    const int tt = -(eg + ex);
    const float u1 = pow2f(tt / 2), u2 = pow2f(tt - tt / 2);
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in) res[im][in] = (acc[im][in] * u1) * u2;
  }
  nf_check<TM, TN>(res, a.nf_flag, a.launch_id);
  float* out = a.direct ? a.out : a.out + (size_t)part * a.Cout * a.K;
  const bool final_out = a.direct;
  float* tile_f = reinterpret_cast<float*>(smem) + wave * (32 * EPI_STRIDE);
  const int col_in = lane & 31, row_hi = 4 * (lane >> 5);
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int kk = kk0 + wn * 64 + in * 32 + c4;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile_f[(e + 8 * g + row_hi) * EPI_STRIDE + col_in] = res[im][in][g * 4 + e];
      __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): a wave's own data only
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int c = co0 + wm * 128 + im * 32 + pass * 8 + rrow;
        float4 v = *reinterpret_cast<const float4*>(tile_f + (pass * 8 + rrow) * EPI_STRIDE + c4);
        if (c < a.Cout && kk < a.K) {
          float4* dst = reinterpret_cast<float4*>(out + (size_t)c * a.K + kk);
          if (final_out) {
            if (a.out_scale) { const float sv = a.out_scale[c]; v.x *= sv; v.y *= sv; v.z *= sv; v.w *= sv; }
            if (a.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          }
          *dst = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ __launch_bounds__(512, 2) void conv_wgrad_big_kernel(const WgradArgs a) { wgrad_big_body(a, blockIdx.x); }

// Several weight gradients in ONE launch (the 3 - 4 of a bottleneck block's backward): the chip's 256 workgroup slots are
// shared among the problems instead of being filled once per problem.  Every workgroup leaves a 256 KB tile of partial sums
// whatever its problem, so a launch parks 64 MB: per PROBLEM that is a third to a quarter of what its own full-chip launch
// parks, its K loops are 3 - 4 x as long against the same prologue / epilogue, and a weight of four tiles (res4's 1x1
// layers), too small for a launch of its own, rides along.  Problem i owns workgroups [first[i], first[i + 1]).  (Not
// padded to multiples of 8: the XCD remap of a problem takes the index inside the problem — the XCDs' identities are
// rotated by first[i] % 8, their chunks of neighbouring tiles stay chunks — and padding pushed 255 workgroups to a grid of
// 264, 33 per XCD of 32 CUs: four XCDs ran a second round, the launch took twice as long.)
__global__ __launch_bounds__(512, 2) void conv_wgrad_big_group_kernel(const WgradGroup g) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < kWgradGroupMax; ++k)
    if (k < g.n && (int)blockIdx.x >= g.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
  const int local = (int)blockIdx.x - g.first[i];
  wgrad_big_body(g.a[i], local);
}

// ---- host side -------------------------------------------------------------------------------------------------------
// 0: never, 1: where the plan below expects a gain, 2: wherever the kernel is applicable (tests).  DADET_BIG_GEMM sets the
// start-up value (A/B runs); dadet_set_big_gemm changes it at run time
static int g_big_mode = [] {
  const char* e = getenv("DADET_BIG_GEMM");
  const int v = e ? atoi(e) : 1;
  return v >= 0 && v <= 2 ? v : 1;
}();

// Number of K ranges a tile's reduction is cut into.  Reductions of K >= 8192 (the RPN 3x3 conv) are ALWAYS cut in two: the
// hand-over (the last part reads one parked tile, ~4 us) is small against 288 K-tiles per tile, and the result of that
// layer then does not depend on how many rows the launch happens to have — the overlapped training schedule runs the RPN
// head on the labelled images only, the plain one on all of them, and tests/test_full_size_gpu.py asks both for the same
// sampled ROIs.  Shorter reductions are cut in two only when the grid leaves half of the CUs without a tile.  (The bound
// was K >= 4096 until the box head of the recipes that pool BOTH images showed what that costs: res5 3x3 on 512 ROIs is 196
// tiles — two parts are 392 workgroups, two rounds on 256 CUs, where one part per tile is one round: `da` 17.77 -> 17.53,
// `triplet` 21.81 -> 21.60 ms per step.  The box head sees the same rows in every schedule.)
static int big_split_plan(const int tiles, const int nk) {
  if (const char* e = getenv("DADET_BIG_SPLITS")) {     // read per call: tests and A/B runs force the part count
    const int v = atoi(e);
    if (v >= 1 && v <= 8 && nk / v >= 1) return v;
  }
  if (nk >= 256) return 2;
  return (tiles <= kNumCU / 2 && nk >= 16) ? 2 : 1;
}

// which large-tile kernel serves this problem: 0 none, 1 the 256 x 256 tile, 2 the 256 x 128 tile
int big_variant(const ConvArgs& a) {
  if (g_big_mode == 0) return 0;
  if (a.os != 1 || !a.epi_v4 || a.Cin % 32 != 0 || a.K % 32 != 0 || a.Cout % 4 != 0) return 0;
  if (a.x_bytes >= 0x7FFFFF00u || a.w_bytes >= 0x7FFFFF00u || a.y_bytes >= 0x7FFFFF00u) return 0;
  if (a.M >= (1 << 24)) return 0;
  if (g_big_mode == 2) {                       // tests: wherever applicable; DADET_BIG_TILE_N picks the tile width
    const char* e = getenv("DADET_BIG_TILE_N");
    return (e && atoi(e) == 128) ? 2 : 1;
  }
  // plan (tools/native/gemm_lab.hip, profiles/r05_gemm_lab.txt): +29 .. +35% against the 128 x 128 kernel where the grid fills
  // the chip with at most two K parts per tile; 64 tiles of 256 x 256 in four parts — res4 3x3 — only equal it (the last
  // part reads 768 KB), so layers of up to 256 output channels take the 256 x 128 tile
  if (a.K < 512 || a.M < 4096 || a.Cout < 128) return 0;
  const int tm = ceil_div(a.M, 256);
  // 129 .. 256 output channels over at least 96 row tiles (the pyramid's 256-channel 3x3 / lateral layers on P2 / P3,
  // M = 262144 / 65536): one column of 256 x 256 tiles fills the chip without any K split, on the kernel whose loop holds
  // the matrix pipe 84% of the time instead of 61% (round 6; DADET_BIG_N256_WIDE=0: the 256 x 128 tile as before)
  static const bool wide256 = !(getenv("DADET_BIG_N256_WIDE") && getenv("DADET_BIG_N256_WIDE")[0] == '0');
  if (wide256 && a.Cout > 128 && a.Cout <= 256 && tm >= 96) return 1;
  if (a.Cout <= 256) return tm * ceil_div(a.Cout, 128) >= 96 ? 2 : 0;
  return tm * ceil_div(a.Cout, 256) >= 96 ? 1 : 0;
}
bool big_eligible(const ConvArgs& a) { return big_variant(a) != 0; }

static int big_tiles(const ConvArgs& a, const int variant) {
  return ceil_div(a.M, 256) * ceil_div(a.Cout, variant == 2 ? 128 : 256);
}

// Tail cut: a grid of a few tiles more than a multiple of the CU count (the res5 head on 512 ROIs: 98 x 8 = 784 tiles of
// 256 x 256, 3.06 rounds) runs its last, nearly empty round for a whole tile's time — a fifth of `M=25088 N=2048 K=512`'s
// 265 us.  The tiles of that round (at most a quarter of the chip) are cut into two parts instead (the symmetric meeting):
// twice the workgroups, half the round.  -> number of whole-tile workgroups (0: no tail cut).  DADET_BIG_TAIL=0: never.
static int big_tail_plan(const int tiles, const int nk) {
  const char* e = getenv("DADET_BIG_TAIL");
  if (e && e[0] == '0') return 0;
  if (getenv("DADET_BIG_SPLITS")) return 0;       // a forced uniform cut (tests, A/B runs)
  const int tail = tiles % kNumCU;
  if (tiles <= kNumCU || tiles > 2048 || tail == 0 || tail > kNumCU / 4 || nk < 4) return 0;
  return tiles - tail;
}

size_t big_workspace_bytes(const ConvArgs& a) {
  const int variant = big_variant(a);
  if (!variant) return 0;
  const int tiles = big_tiles(a, variant);
  const size_t tile_bytes = (size_t)256 * (variant == 2 ? 128 : 256) * sizeof(float);
  const int s = tiles <= 2048 ? big_split_plan(tiles, a.K / 32) : 1;
  if (s > 1) return (size_t)tiles * s * tile_bytes;
  const int body = big_tail_plan(tiles, a.K / 32);
  return body ? (size_t)(tiles - body) * 2 * tile_bytes : 0;
}

int launch_fwd_big(ConvArgs& a, hipStream_t st, float* ws, int* counters) {
  const int variant = big_variant(a);
  const int bn = variant == 2 ? 128 : 256;
  a.tiles_m = ceil_div(a.M, 256);
  a.tiles_n = ceil_div(a.Cout, bn);
  const int tiles = a.tiles_m * a.tiles_n;
  a.big_splits = (ws && counters && tiles <= 2048) ? big_split_plan(tiles, a.K / 32) : 1;
  a.big_body = 0;
  if (a.big_splits == 1 && ws && counters) {
    a.big_body = big_tail_plan(tiles, a.K / 32);
    if (a.big_body) a.big_splits = 2;
  }
  const int grid = a.big_body ? a.big_body + (tiles - a.big_body) * a.big_splits : tiles * a.big_splits;
  a.sk_ws = ws;
  a.sk_counters = counters;
  {
    const char* e = getenv("DADET_BIG_ASYM");       // read per call: A/B runs and tests
    a.big_asym = (e && e[0] == '1') ? 1 : 0;
  }
  // two K-tile slots + the staging groups' row descriptors
  // (+ 16 bytes: the launch's epoch of the two-part meeting, big_announce)
  const size_t lds = 16 + (variant == 2 ? 2 * (2 * 256 * 64 + 2 * 128 * 64) + 12 * 512 * sizeof(int)
                                        : 2 * 4 * 256 * 64 + 16 * 256 * sizeof(int));
  static bool attr_set[3] = {false, false, false};
  if (!attr_set[variant]) {
    const void* fn = variant == 2 ? reinterpret_cast<const void*>(conv_big128_kernel)
                                  : reinterpret_cast<const void*>(conv_big_kernel<256>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_forward(big): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set[variant] = true;
  }
  if (variant == 2) hipLaunchKernelGGL(conv_big128_kernel, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL((conv_big_kernel<256>), dim3(grid), dim3(512), lds, st, a);
  return check_launch("conv_forward(big)");
}

// ---- weight gradient: plan and launch -----------------------------------------------------------------------------
// The grid is small (Cout x K in tiles of 256 x 256: 4 .. 72 tiles), so the reduction over the M rows is cut into `splits`
// ranges of whole K-tiles that fill the 256 CUs once.  Returns false when the 128 x 128 kernel should run.
bool wgrad_big_plan(const dadet_conv_desc* d, int* tiles_co, int* tiles_kc, int* splits, int* rps) {
  static const bool enabled = !(getenv("DADET_WGRAD_BIG") && getenv("DADET_WGRAD_BIG")[0] == '0');   // A/B runs
  if (!enabled || g_big_mode == 0 || gemm_mode() != 4) return false;
  const int M = d->N * d->Ho * d->Wo, K = d->KH * d->KW * d->Cin;
  if (d->Cin % 4 != 0 || K % 4 != 0 || M >= (1 << 24)) return false;
  if ((uint64_t)d->N * d->H * d->W * d->Cin * 4 >= 0x7FFFFF00ull || (uint64_t)M * ((d->Cout + 3) / 4 * 4) * 4 >= 0x7FFFFF00ull)
    return false;
  if (g_big_mode != 2 && (d->Cout < 256 || K < 256 || M < 2048)) return false;
  *tiles_co = ceil_div(d->Cout, 256);
  *tiles_kc = ceil_div(K, 256);
  const int tiles = (*tiles_co) * (*tiles_kc);
  // Every workgroup leaves a 256 KB tile of partial sums: 64 MB per launch once the chip is full, whatever the layer.  A
  // weight of four tiles (res4's 1x1 layers: 1 MB) would be cut into 64 parts — 64 MB written and read again for 41 us of
  // GEMM, where the 128 x 128 kernel's 16 tiles need a quarter of that traffic for 47 us: below eight tiles it keeps the
  // layer (profiles/r05_step_timeline_img_only.txt: the step-end reduction passes read what these launches park)
  static const int min_tiles = getenv("DADET_WGRAD_BIG_MIN_TILES") ? atoi(getenv("DADET_WGRAD_BIG_MIN_TILES")) : 8;
  if (g_big_mode != 2 && tiles < min_tiles) return false;
  int s = kNumCU / tiles;
  if (const char* e = getenv("DADET_WGRAD_BIG_SPLITS")) { const int v = atoi(e); if (v > 0) s = v; }
  if (s < 1) s = 1;
  // a part is ONE fp32 accumulator chain over its rows: beyond ~4096 rows the chain's own rounding shows against the
  // exact-fp32 kernel, whose plan always cuts (tests/test_ops_gpu.py::test_split_bf16_accuracy_at_production_k: the RPN
  // conv's dense gradient, 8192 rows x 144 tiles in one part, RMS 1.27e-6 against 1.01e-6) — at least ceil(M / 4096) parts
  // (the step has no such launch: its 144-tile weight is the RPN conv, whose gradient runs on the <= 256 sampled rows)
  if (!getenv("DADET_WGRAD_BIG_SPLITS") && s < ceil_div(M, 4096)) s = ceil_div(M, 4096);
  const int max_s = ceil_div(M, 128);                 // at least four K-tiles per part
  if (s > max_s) s = max_s;
  const int rows = ceil_div(ceil_div(M, s), 32) * 32;
  *rps = rows;
  *splits = ceil_div(M, rows);
  return true;
}

constexpr size_t kWgradBigLds = 2 * 4 * 256 * 64 + 2 * 256 * sizeof(int4);      // two K-tile slots + group 1's row state

int launch_wgrad_big(WgradArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_big_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradBigLds);
    if (e != hipSuccess) {
      set_error("conv_wgrad(big): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_wgrad_big_kernel, dim3(a.tiles_co * a.tiles_kc * a.splits), dim3(512), kWgradBigLds, st, a);
  return check_launch("conv_wgrad(big)");
}

// can this weight gradient be a member of a grouped launch (the tile kernel's own conditions; no lower bound on the tiles)
bool wgrad_group_member(const dadet_conv_desc* d) {
  static const bool enabled = !(getenv("DADET_WGRAD_BIG") && getenv("DADET_WGRAD_BIG")[0] == '0');
  if (!enabled || g_big_mode == 0 || gemm_mode() != 4) return false;
  const int M = d->N * d->Ho * d->Wo, K = d->KH * d->KW * d->Cin;
  if (d->Cin % 4 != 0 || K % 4 != 0 || d->Cout % 4 != 0 || M >= (1 << 24) || M < 256) return false;
  if ((uint64_t)d->N * d->H * d->W * d->Cin * 4 >= 0x7FFFFF00ull || (uint64_t)M * d->Cout * 4 >= 0x7FFFFF00ull) return false;
  return g_big_mode == 2 || (d->Cout >= 256 && K >= 256 && M >= 2048);
}

// Rows per part R (a multiple of 32, the same for every problem of the group — every workgroup then runs the same number
// of K-tiles on one tile, whatever its problem): the smallest R whose parts fit the chip's workgroup slots (256 for the
// 256 x 256 tile, 2 x 256 for the 128 x 128 kernel), at least 128 rows (four K-tiles), at most 4096 (+ an eighth where that saves a round of workgroups).
// splits[i] = ceil(M_i / R).
// DADET_WGRAD_GROUP_ROWS forces R (tests).
void wgrad_group_plan(const int n, const dadet_conv_desc* d, const int tile, int* tiles_co, int* tiles_kc, int* splits,
                      int* rows) {
  const int slots = tile == 256 ? kNumCU : 2 * kNumCU;
  int max_m = 0;
  for (int i = 0; i < n; ++i) {
    tiles_co[i] = ceil_div(d[i].Cout, tile);
    tiles_kc[i] = ceil_div(d[i].KH * d[i].KW * d[i].Cin, tile);
    const int M = d[i].N * d[i].Ho * d[i].Wo;
    max_m = M > max_m ? M : max_m;
  }
  int R = 128;
  if (const char* e = getenv("DADET_WGRAD_GROUP_ROWS")) {
    const int v = atoi(e);
    R = v >= 32 ? v / 32 * 32 : R;
  } else {
    const int top = ceil_div(max_m, 32) * 32;
    for (; R < top; R += 32) {
      int wgs = 0;
      for (int i = 0; i < n; ++i) wgs += tiles_co[i] * tiles_kc[i] * ceil_div(d[i].N * d[i].Ho * d[i].Wo, R);
      if (wgs <= slots) break;
    }
    // one part is ONE fp32 accumulator chain over its rows: never more than 4096 of them (wgrad_big_plan's bound, for the
    // same reason) — a group whose tiles alone nearly fill the slots (the res5 head on 512 ROIs: ~100 tiles x 25088 rows)
    // then takes a second / third round of workgroups, cut into equal parts rather than 4096 + a remainder
    // (an eighth of slack: the res5 head on 256 ROIs — 68 tiles, 12544 rows — fits the slots in three parts of 4192 rows;
    // cutting it into four of 3136 is 272 workgroups, a second round for 16 of them: the family went 1.8 -> 2.2 ms per step)
    const int cap = ceil_div(ceil_div(max_m, ceil_div(max_m, 4096)), 32) * 32;
    if (R > 4096 + 512) R = cap;
  }
  *rows = R;
  for (int i = 0; i < n; ++i) splits[i] = ceil_div(d[i].N * d[i].Ho * d[i].Wo, R);
}

int launch_wgrad_big_group(const WgradArgs* a, const int n, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_big_group_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgradBigLds);
    if (e != hipSuccess) {
      set_error("conv_wgrad_group: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  WgradGroup g;
  g.n = n;
  int at = 0;
  for (int i = 0; i < kWgradGroupMax; ++i) {
    g.a[i] = a[i < n ? i : n - 1];
    g.first[i] = at;
    if (i < n) at += a[i].tiles_co * a[i].tiles_kc * a[i].splits;
  }
  g.first[kWgradGroupMax] = at;
  g.by_rows = 1;
  hipLaunchKernelGGL(conv_wgrad_big_group_kernel, dim3(at), dim3(512), kWgradBigLds, st, g);
  return check_launch("conv_wgrad_group");
}

}  // namespace dadet

extern "C" int dadet_set_big_gemm(int mode) {
  if (mode < 0 || mode > 2) {
    dadet::set_error("set_big_gemm: mode must be 0 (off), 1 (planned) or 2 (wherever applicable)");
    return DADET_EINVAL;
  }
  dadet::g_big_mode = mode;
  return DADET_OK;
}
extern "C" int dadet_get_big_gemm(void) { return dadet::g_big_mode; }
