// Greedy NMS entirely on the device (gfx950).
//
// Replaces _C.nms of the reference (maskrcnn_benchmark/csrc/nms.h:10-28; CPU kernel
// csrc/cpu/nms_cpu.cpp:6-75; CUDA kernel + host sweep csrc/cuda/nms.cu:23-131).
//
// Differences in HOW (results are the reference's):
//   * ranking: one workgroup bitonic-sorts (score desc, index asc) keys in LDS (n <= 16384 fits the
//     160 KB LDS; larger n falls back to a global-memory bitonic network), instead of ATen sort;
//   * IoU bitmask: only the upper triangle of 64x64 tiles is computed (the reference computes all
//     tiles and ignores half, nms.cu:27);
//   * greedy sweep: done by one workgroup on the device, 64 boxes per step — a wavefront resolves the
//     64x64 diagonal tile with cross-lane reads, then every lane ORs the kept rows into its own 64-bit
//     word of the removed-set.  The reference copies the whole mask (18 MB at n=12000) to the host
//     with a blocking cudaMemcpy and sweeps there (nms.cu:99-123);
//   * output compaction to ascending ORIGINAL indices (nms_cpu.cpp:64 / nms.cu:127-130) by a
//     workgroup scan, written as int64.
// Built with -ffp-contract=off so the IoU arithmetic is the reference's (nms_cpu.cpp:50-60).
#include <cstdlib>

#include "common.h"

namespace dadet {

constexpr int kSortLdsMax = 16384;  // keys that fit one workgroup's LDS (8 B each = 128 KB)

struct Key {
  float score;
  int idx;
};

// true when a must come before b: score descending, index ascending (stable-sort order)
__device__ inline bool key_before(const Key& a, const Key& b) {
  return (a.score > b.score) || (a.score == b.score && a.idx < b.idx);
}

// ---- ranking ------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void nms_sort_lds_kernel(const float* __restrict__ scores, int n,
                                                            int npow2, int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Key* keys = reinterpret_cast<Key*>(smem);
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    Key k;
    if (i < n) {
      k.score = scores[i];
      k.idx = i;
    } else {
      k.score = -INFINITY;
      k.idx = 0x7fffffff;  // padding sorts last
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (npow2 >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool ascending_block = ((lo & size) == 0);  // "before"-ordered block
        Key a = keys[lo], b = keys[hi];
        const bool swap = ascending_block ? key_before(b, a) : key_before(a, b);
        if (swap) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = keys[i].idx;
}

// global-memory fallback (n > 16384): one launch per (size, stride) step
__global__ void nms_sort_init_kernel(const float* __restrict__ scores, int n, int npow2,
                                     Key* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npow2) return;
  Key k;
  if (i < n) {
    k.score = scores[i];
    k.idx = i;
  } else {
    k.score = -INFINITY;
    k.idx = 0x7fffffff;
  }
  keys[i] = k;
}
__global__ void nms_sort_step_kernel(Key* __restrict__ keys, int npow2, int size, int stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (npow2 >> 1)) return;
  const int lo = 2 * t - (t & (stride - 1));
  const int hi = lo + stride;
  const bool ascending_block = ((lo & size) == 0);
  Key a = keys[lo], b = keys[hi];
  const bool swap = ascending_block ? key_before(b, a) : key_before(a, b);
  if (swap) {
    keys[lo] = b;
    keys[hi] = a;
  }
}
__global__ void nms_sort_finish_kernel(const Key* __restrict__ keys, int n, int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = keys[i].idx;
}

__global__ void nms_gather_boxes_kernel(const float4* __restrict__ boxes, const int* __restrict__ order,
                                        int n, float4* __restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted[i] = boxes[order[i]];
}

// ---- IoU bitmask, upper triangle of 64x64 tiles ----------------------------------------------
// IoU exactly as nms_cpu.cpp:23,50-59: area = (x2-x1+1)*(y2-y1+1); inter = max(0,xx2-xx1+1)*max(0,yy2-yy1+1);
// ovr = inter / (area_i + area_j - inter).
// diag_t / adj_t (optional): the diagonal tile and the tile right of it once more, TRANSPOSED — word i of diag_t holds, for
// box i, the bits of the boxes j < i of its own 64-block that suppress it; word i of adj_t the bits of the boxes of the
// PREVIOUS block that suppress it.  With its column in a lane, "is box i suppressed by the kept set K" is one AND, and the
// new kept set one ballot (nms_sweep_pipelined_kernel).  Same IoU expression with the same operands as the row form.
// (MEASURED, round 3: four tiles per 256-thread workgroup over a triangular grid — 4442 workgroups instead of 35 344 one-
// wavefront ones, half of which return at once — changes nothing, 0.192 vs 0.186 ms per NMS: the kernel is bound by its
// VALU work, 64 IEEE divisions per lane and tile, not by the dispatch rate.)
// "IoU against the threshold" without the division wherever the answer is not within rounding of the threshold.  The
// reference decides fl(inter / u) >= thresh (CPU rule; > for the CUDA rule) with u = fl(fl(area_a + area_b) - inter).  With
// p = fl(thresh * u): inter >= p (1 + 2^-20) puts the exact quotient above thresh (1 + 2^-21), whose rounding cannot come
// down to thresh; inter <= p (1 - 2^-20) puts it below thresh (1 - 2^-21), whose rounding cannot come up to it — in both
// cases the IEEE quotient's comparison is known.  Anything in between (a band of 2^-19 around the threshold) takes the
// division, so every decision is the reference's bit for bit (tests/test_ops_gpu.py: near-tie pairs against the oracle).
// The division was ~half of the mask kernel's VALU time (64 per lane and tile).
template <int TIE_RULE>
__device__ __forceinline__ bool iou_suppresses(const float inter, const float u, const float thresh) {
  if (u > 0.f && thresh > 0.f) {        // (degenerate boxes — u <= 0 — and NaNs take the division: whatever it gives is the rule)
    const float p = thresh * u;
    if (inter >= p * (1.0f + 0x1p-20f)) return true;
    if (inter <= p * (1.0f - 0x1p-20f)) return false;
  }
  const float ovr = inter / u;
  return (TIE_RULE == 0) ? (ovr >= thresh) : (ovr > thresh);
}

template <int TIE_RULE>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ sorted, int n,
                                                      float thresh, int col_blocks,
                                                      unsigned long long* __restrict__ mask,
                                                      unsigned long long* __restrict__ diag_t,
                                                      unsigned long long* __restrict__ adj_t,
                                                      unsigned long long* __restrict__ blk_t) {
  // linear block id -> (row_block <= col_block) pair
  const int row_start = blockIdx.y, col_start = blockIdx.x;
  if (col_start < row_start) return;
  // blk_t (nms_sweep_block_kernel): per box eight words — which boxes of the PREVIOUS 256-box block (words 0 - 3) and of
  // its OWN block before it (words 4 - 7) suppress it; row chunks 4 (B - 1) .. c of column chunk c (B = c / 4) fill them,
  // the diagonal tile zeroes what no tile writes (chunks behind c in its block; the previous block of block 0)
  const int blk_first = 4 * (col_start / 4 - 1);
  const bool want_b = blk_t && row_start >= blk_first;
  if (blk_t && row_start == col_start && (int)threadIdx.x < min(n - col_start * 64, 64)) {
    unsigned long long* mine = blk_t + (size_t)(col_start * 64 + threadIdx.x) * 8;
    for (int q = col_start % 4 + 1; q < 4; ++q) mine[4 + q] = 0ULL;
    if (blk_first < 0)
      for (int q = 0; q < 4; ++q) mine[q] = 0ULL;
  }
  // row n of the matrix: all zeros, what the sweep loads for "no kept box in this slot"
  if (row_start == 0 && col_start == 0)
    for (int i = threadIdx.x; i < col_blocks; i += 64) mask[(size_t)n * col_blocks + i] = 0ULL;
  const int row_size = min(n - row_start * 64, 64);
  const int col_size = min(n - col_start * 64, 64);
  __shared__ float4 cb[64];
  __shared__ float carea[64];
  __shared__ float4 rb[64];
  __shared__ float rarea[64];
  const bool want_t = (diag_t && (col_start == row_start || col_start == row_start + 1)) || want_b;
  if ((int)threadIdx.x < col_size) {
    const float4 b = sorted[col_start * 64 + threadIdx.x];
    cb[threadIdx.x] = b;
    carea[threadIdx.x] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  }
  if (want_t && (int)threadIdx.x < row_size) {
    const float4 b = sorted[row_start * 64 + threadIdx.x];
    rb[threadIdx.x] = b;
    rarea[threadIdx.x] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  }
  __syncthreads();
  if (want_t && (int)threadIdx.x < col_size) {
    // this thread = COLUMN box b; a = the row box, exactly as in the row form below (same operand order)
    const float4 b = cb[threadIdx.x];
    const float barea = carea[threadIdx.x];
    const int stop = (row_start == col_start) ? (int)threadIdx.x : row_size;    // diagonal: rows j < i only
    unsigned long long t = 0;
    for (int i = 0; i < stop; ++i) {
      const float4 a = rb[i];
      const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
      const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
      const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
      const float inter = w * h;
      if (iou_suppresses<TIE_RULE>(inter, rarea[i] + barea - inter, thresh)) t |= 1ULL << i;
    }
    if (diag_t && (col_start == row_start || col_start == row_start + 1))
      (row_start == col_start ? diag_t : adj_t)[col_start * 64 + threadIdx.x] = t;
    if (want_b) blk_t[(size_t)(col_start * 64 + threadIdx.x) * 8 + (row_start - blk_first)] = t;
  }
  if ((int)threadIdx.x < row_size) {
    const int cur = row_start * 64 + threadIdx.x;
    const float4 a = sorted[cur];
    const float iarea = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    unsigned long long t = 0;
    const int start = (row_start == col_start) ? (int)threadIdx.x + 1 : 0;
    for (int i = start; i < col_size; ++i) {
      const float4 b = cb[i];
      const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
      const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
      const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
      const float inter = w * h;
      if (iou_suppresses<TIE_RULE>(inter, iarea + carea[i] - inter, thresh)) t |= 1ULL << i;
    }
    mask[(size_t)cur * col_blocks + col_start] = t;
  }
}

// ---- greedy sweep, one workgroup ------------------------------------------------------------
// thread w owns word w of the removed-set.  Per 64-box chunk c: wave 0 resolves the diagonal tile
// serially in registers (cross-lane reads), publishes the chunk's keep bits through LDS, then every
// thread w > c ORs the kept rows' word w.
__global__ __launch_bounds__(1024) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                         int n, int col_blocks, int max_keep,
                                                         unsigned long long* __restrict__ keep_bits) {
  __shared__ unsigned long long s_removed_c;  // removed word of the current chunk
  __shared__ unsigned long long s_keep;       // keep bits of the current chunk
  __shared__ int s_kept_total;
  const int w = threadIdx.x;
  unsigned long long removed = 0;  // word w of the removed set (valid for w < col_blocks)
  if (threadIdx.x == 0) s_kept_total = 0;
  // The chunk loop is a dependent chain, so the latency of one iteration is what counts.  The diagonal words do not
  // depend on anything computed here: those of chunk c + 1 are fetched while chunk c is resolved (measured: no change —
  // the kept rows' loads below are what an iteration waits for).
  unsigned long long diag_next = 0;   // wave 0: this lane's diagonal word of the next chunk
  if (threadIdx.x < 64 && (int)threadIdx.x < n) diag_next = mask[(size_t)threadIdx.x * col_blocks];
  __syncthreads();
  for (int c = 0; c < col_blocks; ++c) {
    if (w == c) s_removed_c = removed;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const unsigned long long diag = diag_next;
      const int nbox = (c + 1) * 64 + lane;
      diag_next = (c + 1 < col_blocks && nbox < n) ? mask[(size_t)nbox * col_blocks + c + 1] : 0ULL;
      const unsigned lo = (unsigned)diag, hi = (unsigned)(diag >> 32);
      // the whole resolution is wavefront-uniform: keep it on the scalar unit (SGPR state, v_readlane of the
      // diagonal rows) and visit only the boxes that are still alive instead of all 64 positions
      const unsigned long long rem0 = s_removed_c;
      // (the builtin returns a signed int: widen through unsigned, or bit 31 smears over the upper word)
      const unsigned rem_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rem0);
      const unsigned rem_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rem0 >> 32));
      unsigned long long rem = ((unsigned long long)rem_hi << 32) | (unsigned long long)rem_lo;
      const int limit = min(64, n - c * 64);
      const unsigned long long valid = limit >= 64 ? ~0ULL : ((1ULL << limit) - 1ULL);
      unsigned long long keep = 0;
      int kept_total = __builtin_amdgcn_readfirstlane(s_kept_total);
      unsigned long long alive = ~rem & valid;
      while (alive) {
        if (max_keep > 0 && kept_total >= max_keep) break;
        const int j = __ffsll((long long)alive) - 1;
        keep |= 1ULL << j;
        ++kept_total;
        const unsigned dlo = (unsigned)__builtin_amdgcn_readlane((int)lo, j);
        const unsigned dhi = (unsigned)__builtin_amdgcn_readlane((int)hi, j);
        rem |= ((unsigned long long)dhi << 32) | dlo;
        // positions <= j are decided; the mask only holds pairs (i, k > i), so rows never clear earlier bits
        alive = ~rem & valid & ~((2ULL << j) - 1ULL);
      }
      if (lane == 0) {
        s_keep = keep;
        s_kept_total = kept_total;
        keep_bits[c] = keep;
      }
    }
    __syncthreads();
    const unsigned long long keep = s_keep;
    if (w > c && w < col_blocks) {
      // the keep word is wavefront-uniform: peel kPeel kept rows per step so that many independent loads are in
      // flight instead of one dependent round trip per kept box (a chunk of a fresh region keeps most of its 64 boxes:
      // with 8 per step that was 8 round trips in series per chunk; 32 make it 2)
      constexpr int kPeel = 32;
      unsigned long long k = keep;
      const unsigned long long* mrow = mask + (size_t)c * 64 * col_blocks + w;
      while (k) {
        int j[kPeel];
#pragma unroll
        for (int u = 0; u < kPeel; ++u) {
          j[u] = k ? (__ffsll((long long)k) - 1) : -1;
          k &= (k - 1);
        }
        unsigned long long v[kPeel];
#pragma unroll
        for (int u = 0; u < kPeel; ++u) v[u] = (j[u] >= 0) ? mrow[(size_t)j[u] * col_blocks] : 0ULL;
        unsigned long long acc = 0;
#pragma unroll
        for (int u = 0; u < kPeel; ++u) acc |= v[u];
        removed |= acc;
      }
    }
    // early out once the quota is filled: later chunks keep nothing
    if (max_keep > 0 && s_kept_total >= max_keep) {
      for (int cc = c + 1 + (int)threadIdx.x; cc < col_blocks; cc += blockDim.x) keep_bits[cc] = 0;
      break;
    }
    __syncthreads();
  }
}

// ---- greedy sweep, pipelined (round 3) --------------------------------------------------------------------------
// nms_sweep_kernel above is a dependent chain of col_blocks steps, each "barrier, resolve the diagonal tile box by box on
// the scalar unit (one v_readlane pair per KEPT box: ~3 us for a chunk that keeps most of its 64 boxes), barrier, every
// thread loads the kept rows' words (1 - 2 round trips to L2), barrier": 5 - 6 us per 64-box chunk on fresh regions,
// 0.44 ms inside the training step — the longest single item between the RPN head and the box head (rocprofv3 timeline,
// profiles/r03_step_timeline_*.txt).  Two changes:
//   * the chunk is resolved in PARALLEL by fixed-point iteration on transposed tiles.  Lane i holds the column of box i
//     (diag_t: which boxes j < i of its chunk suppress it; adj_t: which boxes of the previous chunk do).  "Suppressed by
//     the kept set K" is then one AND per lane and the next K one ballot: K <- alive & ~suppressed_by(K), starting from
//     K = alive.  The tile is strictly upper triangular, so round t fixes (at least) box t for good; the iteration ends
//     when K stops changing — after as many rounds as the longest suppression chain in the chunk, typically 2 - 5, never
//     more than 64 — with exactly the greedy result.  The previous chunk's influence comes through adj_t the same way, so
//     no memory access sits between two chunks' resolutions;
//   * the other threads' loads for chunk c (word w of every kept row, w >= c + 2) are ISSUED after chunk c's keep bits are
//     known and CONSUMED one iteration later, i.e. they are in flight while chunk c + 1 is resolved.
// Same keep bits as nms_sweep_kernel (tests/test_ops_gpu.py compares both against the CPU oracle; DADET_NMS_SWEEP=0).
__global__ __launch_bounds__(256) void nms_sweep_pipelined_kernel(const unsigned long long* __restrict__ mask,
                                                                  const unsigned long long* __restrict__ diag_t,
                                                                  const unsigned long long* __restrict__ adj_t, int n,
                                                                  int col_blocks, int max_keep,
                                                                  unsigned long long* __restrict__ keep_bits) {
  constexpr int kSlots = 24;                  // kept rows of a chunk whose loads stay in flight for two resolutions
  __shared__ unsigned long long s_removed_c;  // removed word of the current chunk from chunks <= c - 2
  __shared__ unsigned long long s_keep;       // keep bits of the current chunk
  __shared__ int s_kept_total;
  // mask rows of the current chunk's kept boxes in rank order; every other entry names the all-zero row behind the
  // matrix (row n, written by the mask kernel), so that EVERY load below is unconditional: with a load under an exec
  // branch the compiler can no longer count what is in flight and waits with vmcnt(0) — the pushed words then had one
  // resolution of flight instead of two and an iteration cost ~1700 instructions (ISA of round 3)
  // (stored as BYTE offsets of the rows, 32 bits: the matrix of 16 384 boxes is 32 MB)
  __shared__ unsigned s_rows[64 + kSlots];
  // the keep words, written out once at the end.  A global store inside the loop shares the vmcnt counter with the loads in
  // flight on gfx9-family targets and may retire out of order with them: the compiler then waits with vmcnt(0) at every
  // use of a loaded word (seen in the ISA) — again one resolution of flight instead of two
  __shared__ unsigned long long s_keep_all[258];
  const int w = threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool wave0 = threadIdx.x < 64;
  const unsigned pitch = (unsigned)col_blocks * 8u;
  const unsigned zero_row = (unsigned)n * pitch;
  // Thread w owns word w of the removed set.  No guard on w anywhere: a thread behind the current chunk (or beyond the
  // last word, clamped) ORs words nobody reads again — word w is published once, at iteration w, and only loads consumed
  // before that matter.
  const char* mcol = reinterpret_cast<const char*>(mask + min(w, col_blocks - 1));
  auto word_at = [&](const unsigned row_bytes) { return *reinterpret_cast<const unsigned long long*>(mcol + row_bytes); };
  unsigned long long removed = 0;
  // word w of the kept rows of the last EVEN chunk (fly_a) and of the last ODD chunk (fly_b): issued at the end of iteration
  // c, consumed at the start of iteration c + 2; what chunk c needs from chunk c - 1 comes through adj_t instead
  unsigned long long fly_a[kSlots], fly_b[kSlots];
  if (threadIdx.x == 0) s_kept_total = 0;
  for (int i = threadIdx.x; i < 64 + kSlots; i += blockDim.x) s_rows[i] = zero_row;
  for (int i = threadIdx.x; i < 258; i += blockDim.x) s_keep_all[i] = 0ULL;
  // wave 0, lane i: the transposed diagonal / previous-block words of box (c * 64 + i), fetched TWO chunks ahead into the
  // register pair of the same parity (no register copy between iterations: a copy is a wait for the load just issued).
  //
  // The compiler derives every s_waitcnt vmcnt(N) from the ORDER in which loads are issued, and at the loop header it
  // must assume the shorter of "came from the prologue" and "came from the previous iteration".  The prologue therefore
  // issues exactly the loop's sequence — (diag, adj) of the even chunk, kSlots words, (diag, adj) of the odd chunk, kSlots
  // words, all unconditional (clamped rows; the pushed words read the zero row) — and nothing inside the loop leaves it
  // early (the quota is tested once per iteration, at the very end): a branch around the loads that flows back to the
  // header makes the newest loads look like the oldest, and the waits collapse to vmcnt(0).
  const unsigned long long* zrow = mask + (size_t)n * (size_t)col_blocks;   // (distinct words: no merged loads)
  // (diag, adj) are loaded by EVERY thread, in the same block as the pushed words and in front of them: only wave 0 uses
  // the values, but a load under `if (wave0)` is one the compiler cannot order against the others
  const int dl = threadIdx.x & 63;
  unsigned long long d_even = diag_t[min(dl, n - 1)];
  unsigned long long a_even = adj_t[min(64 + dl, n - 1)];   // (chunk 0 has no previous block: unused, prev_keep = 0)
#pragma unroll
  for (int u = 0; u < kSlots; ++u) fly_a[u] = zrow[min(u, col_blocks - 1)];
  unsigned long long d_odd = diag_t[min(64 + dl, n - 1)];
  unsigned long long a_odd = adj_t[min(64 + dl, n - 1)];
#pragma unroll
  for (int u = 0; u < kSlots; ++u) fly_b[u] = zrow[min(u, col_blocks - 1)];
  __syncthreads();
  auto step = [&](const int c, unsigned long long (&fly)[kSlots], unsigned long long& dcol, unsigned long long& acol) {
    {
      unsigned long long acc = 0;
#pragma unroll
      for (int u = 0; u < kSlots; ++u) acc |= fly[u];
      removed |= acc;
    }
    if (w == c) s_removed_c = removed;        // chunks <= c - 2 (chunk c - 1's part comes through adj_t)
    __syncthreads();
    if (wave0) {
      const unsigned long long rem0 = s_removed_c;
      const unsigned rem_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rem0);
      const unsigned rem_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rem0 >> 32));
      const unsigned long long prev_keep = c ? s_keep : 0ULL;      // chunk c - 1's kept set (still in LDS)
      const int limit = min(64, n - c * 64);
      // alive: inside the range, not removed by chunks <= c - 2 (the pushed words), not removed by chunk c - 1's kept
      // boxes (one AND with the transposed previous-block column)
      const bool in_range = lane < limit;
      const bool pushed = ((((unsigned long long)rem_hi << 32) | rem_lo) >> lane) & 1ULL;
      const bool alive_i = in_range && !pushed && (acol & prev_keep) == 0ULL;
      const unsigned long long alive = __ballot(alive_i);
      // greedy keep set by fixed-point iteration: K = alive & ~suppressed_by(K).  The tile is strictly upper triangular, so
      // after t rounds the first t boxes are final: at most 64 rounds, in practice the depth of the longest suppression
      // chain in the chunk (a handful).  One AND + compare + ballot per round.
      unsigned long long K = alive;
      for (int it = 0; it < 64; ++it) {
        const unsigned long long Kn = __ballot(alive_i && (dcol & K) == 0ULL);
        if (Kn == K) break;
        K = Kn;
      }
      int kept_total = __builtin_amdgcn_readfirstlane(s_kept_total);
      if (max_keep > 0 && kept_total + __popcll(K) > max_keep) {
        // quota: keep the first (max_keep - kept_total) of them — later boxes never influence earlier ones
        int room = max(max_keep - kept_total, 0);
        unsigned long long first = 0, k2 = K;
        while (room-- > 0 && k2) {
          first |= k2 & (~k2 + 1ULL);
          k2 &= k2 - 1ULL;
        }
        K = first;
      }
      kept_total += __popcll(K);
      // kept rows in rank order (one wave: its LDS writes land in program order)
      s_rows[lane] = zero_row;
      if ((K >> lane) & 1ULL) s_rows[__popcll(K & ((1ULL << lane) - 1ULL))] = (unsigned)(c * 64 + lane) * pitch;
      if (lane == 0) {
        s_keep = K;
        s_kept_total = kept_total;
        s_keep_all[c] = K;
      }
    }
    __syncthreads();
    const unsigned long long kv = s_keep;
    const int cnt = __popc((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kv)) +
                    __popc((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kv >> 32)));
    // a chunk that keeps more than kSlots boxes: the rest at once, waited for here (rare past the first chunks — the
    // kept boxes thin out as the sweep goes on)
    for (int base = kSlots; base < cnt; base += kSlots) {
      unsigned long long t[kSlots];
#pragma unroll
      for (int u = 0; u < kSlots; ++u) t[u] = word_at(s_rows[base + u]);
      unsigned long long acc = 0;
#pragma unroll
      for (int u = 0; u < kSlots; ++u) acc |= t[u];
      removed |= acc;
    }
    // (diag, adj) of chunk c + 2, then the first kSlots kept rows: consumed two iterations on (word c + 1 goes through
    // adj_t)
    {
      const int nbox = min((c + 2) * 64 + dl, n - 1);
      dcol = diag_t[nbox];
      acol = adj_t[nbox];
    }
#pragma unroll
    for (int u = 0; u < kSlots; ++u) fly[u] = word_at(s_rows[u]);
  };
  int c_end = col_blocks;
  for (int c = 0; c < c_end; c += 2) {
    step(c, fly_a, d_even, a_even);
    step(c + 1, fly_b, d_odd, a_odd);      // (c + 1 == col_blocks: an empty chunk — nothing in range, keeps nothing)
    // quota filled: later chunks keep nothing (a chunk resolved after the quota was reached has room for 0 boxes).  One
    // exit, the loop condition: see above
    c_end = (max_keep > 0 && s_kept_total >= max_keep) ? 0 : col_blocks;
  }
  __syncthreads();
  if (w < col_blocks) keep_bits[w] = s_keep_all[w];
}

// ---- greedy sweep in blocks of 256 boxes (round 6) ---------------------------------------------------------------
// nms_sweep_pipelined_kernel resolves 64 boxes per iteration on ONE wave (two barriers, the wave's fixed-point rounds, LDS
// hand-overs: ~1.2 us per chunk, 188 chunks for 12 000 boxes: 0.22 ms, the longest kernel between the RPN head and the box
// head).  Here all four waves resolve a BLOCK of 256 boxes together: lane = box, blk_t holds its eight transposed words
// (suppressors in the previous block / in its own block before it), the greedy keep set is the same fixed point
// K <- alive & ~suppressed_by(K) over the 256 x 256 strictly upper triangular tile — the four waves exchange their K words
// through LDS, one barrier (with an OR reduction of "changed") per round, as many rounds as the longest suppression chain in
// the block.  Everything else is the pipelined kernel's scheme at block granularity: thread w owns word w of the removed set;
// the words of block B's kept rows are issued when B is resolved and consumed two blocks later (block B + 1 sees B through
// the transposed previous-block words), the first kSlots of them in flight across the next block's resolution.
// Same keep bits as the other two sweeps (tests/test_ops_gpu.py: all three against the CPU oracle; DADET_NMS_SWEEP=1 / 0).
// MEASURED (round 6): alone on the GPU the two pipelined sweeps take the same time (tools/nms_time.py: 0.20 ms per NMS of
// 12 000 boxes with a quota of 2000, mask kernel included) — what bounds them is not the resolution but the chain "resolve a
// block -> fetch its kept rows (1.5 KB each, ~3 MB per NMS, HBM latency, one workgroup: ~50 GB/s) -> two blocks later";
// inside the training step, beside the GEMM waves, fewer barriers are worth 0.03 (img_only) to 0.07 ms (da).
__global__ __launch_bounds__(256) void nms_sweep_block_kernel(const unsigned long long* __restrict__ mask,
                                                              const unsigned long long* __restrict__ blk_t, int n,
                                                              int col_blocks, int max_keep,
                                                              unsigned long long* __restrict__ keep_bits) {
  // (a block of 256 boxes keeps ~40 where 64-box chunks keep ~10: with 24 slots the rest of every block was fetched
  // synchronously, one exposed HBM round trip per block)
  constexpr int kSlots = 48;
  __shared__ unsigned long long s_removed[4];     // removed words of the current block from blocks <= B - 2
  __shared__ unsigned long long s_K[2][4];        // the fixed point's keep words, double buffered
  __shared__ unsigned long long s_prevK[4];       // block B - 1's kept set
  __shared__ int s_kept_total;
  __shared__ unsigned s_rows[256 + kSlots];       // byte offsets of the current block's kept rows, rank order; rest: zero row
  __shared__ unsigned long long s_keep_all[264];
  const int w = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned pitch = (unsigned)col_blocks * 8u;
  const unsigned zero_row = (unsigned)n * pitch;
  const char* mcol = reinterpret_cast<const char*>(mask + min(w, col_blocks - 1));
  auto word_at = [&](const unsigned row_bytes) { return *reinterpret_cast<const unsigned long long*>(mcol + row_bytes); };
  const unsigned long long* zrow = mask + (size_t)n * (size_t)col_blocks;
  const int nblk = (col_blocks + 3) / 4;
  unsigned long long removed = 0;
  unsigned long long fly_a[kSlots], fly_b[kSlots];
  unsigned long long t_even[8], t_odd[8];          // this lane's transposed words of the next even / odd block
  if (threadIdx.x == 0) s_kept_total = 0;
  if (threadIdx.x < 4) s_prevK[threadIdx.x] = 0ULL;
  for (int i = threadIdx.x; i < 256 + kSlots; i += blockDim.x) s_rows[i] = zero_row;
  for (int i = threadIdx.x; i < 264; i += blockDim.x) s_keep_all[i] = 0ULL;
  // the prologue issues exactly the loop's load sequence (see nms_sweep_pipelined_kernel on vmcnt): transposed words of
  // the even block, kSlots row words, transposed words of the odd block, kSlots row words — all unconditional
  auto load_t = [&](unsigned long long (&t)[8], const int blk) {
    const unsigned long long* src = blk_t + (size_t)min(blk * 256 + w, n - 1) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = src[q];
  };
  load_t(t_even, 0);
#pragma unroll
  for (int u = 0; u < kSlots; ++u) fly_a[u] = zrow[min(u, col_blocks - 1)];
  load_t(t_odd, 1);
#pragma unroll
  for (int u = 0; u < kSlots; ++u) fly_b[u] = zrow[min(u, col_blocks - 1)];
  __syncthreads();
  auto step = [&](const int B, unsigned long long (&fly)[kSlots], unsigned long long (&t)[8]) {
    {
      unsigned long long acc = 0;
#pragma unroll
      for (int u = 0; u < kSlots; ++u) acc |= fly[u];
      removed |= acc;
    }
    if ((w >> 2) == B) s_removed[w & 3] = removed;        // blocks <= B - 2 (block B - 1's part comes through t[0..3])
    __syncthreads();
    const int box = B * 256 + w;
    const bool in_range = box < n;
    const bool pushed = (s_removed[wave] >> lane) & 1ULL;
    const bool alive_i = in_range && !pushed &&
                         ((t[0] & s_prevK[0]) | (t[1] & s_prevK[1]) | (t[2] & s_prevK[2]) | (t[3] & s_prevK[3])) == 0ULL;
    unsigned long long K = __ballot(alive_i);
    int buf = 0;
    if (lane == 0) s_K[0][wave] = K;
    s_rows[w] = zero_row;                 // (the previous block's row loads were issued before the barrier above)
    __syncthreads();
    // fixed point over the block: a box is kept iff alive and no kept box before it (own block) suppresses it
    for (int it = 0; it < 256; ++it) {
      const unsigned long long sup = (t[4] & s_K[buf][0]) | (t[5] & s_K[buf][1]) | (t[6] & s_K[buf][2]) | (t[7] & s_K[buf][3]);
      const unsigned long long Kn = __ballot(alive_i && sup == 0ULL);
      if (lane == 0) s_K[buf ^ 1][wave] = Kn;
      const int changed = __syncthreads_or(Kn != K);
      K = Kn;
      buf ^= 1;
      if (!changed) break;
    }
    unsigned long long k0 = s_K[buf][0], k1 = s_K[buf][1], k2 = s_K[buf][2], k3 = s_K[buf][3];
    auto before_me = [&]() {      // kept boxes of the block in front of this lane's box
      int r = __popcll(K & ((1ULL << lane) - 1ULL));
      if (wave > 0) r += __popcll(k0);
      if (wave > 1) r += __popcll(k1);
      if (wave > 2) r += __popcll(k2);
      return r;
    };
    int kept_total = s_kept_total;
    int cnt = __popcll(k0) + __popcll(k1) + __popcll(k2) + __popcll(k3);
    if (max_keep > 0 && kept_total + cnt > max_keep) {
      // quota: the first (max_keep - kept_total) of them — later boxes never influence earlier ones
      const int room = max(max_keep - kept_total, 0);
      const bool mine = ((K >> lane) & 1ULL) && before_me() < room;
      __syncthreads();                                     // everybody has read s_K[buf]
      K = __ballot(mine);
      if (lane == 0) s_K[buf][wave] = K;
      __syncthreads();
      k0 = s_K[buf][0]; k1 = s_K[buf][1]; k2 = s_K[buf][2]; k3 = s_K[buf][3];
      cnt = __popcll(k0) + __popcll(k1) + __popcll(k2) + __popcll(k3);
    }
    if ((K >> lane) & 1ULL) s_rows[before_me()] = (unsigned)box * pitch;
    __syncthreads();                                       // s_kept_total / s_prevK read by everybody above
    if (lane == 0) {
      s_prevK[wave] = K;
      s_keep_all[B * 4 + wave] = K;
    }
    if (threadIdx.x == 0) s_kept_total = kept_total + cnt;
    // a block that keeps more than kSlots boxes: the rest at once, waited for here
    for (int base = kSlots; base < cnt; base += kSlots) {
      unsigned long long tt[kSlots];
#pragma unroll
      for (int u = 0; u < kSlots; ++u) tt[u] = word_at(s_rows[min(base + u, 256 + kSlots - 1)]);
      unsigned long long acc = 0;
#pragma unroll
      for (int u = 0; u < kSlots; ++u) acc |= tt[u];
      removed |= acc;
    }
    // transposed words of block B + 2, then the first kSlots kept rows: consumed two iterations on
    load_t(t, B + 2);
#pragma unroll
    for (int u = 0; u < kSlots; ++u) fly[u] = word_at(s_rows[u]);
  };
  int b_end = nblk;
  for (int B = 0; B < b_end; B += 2) {
    step(B, fly_a, t_even);
    step(B + 1, fly_b, t_odd);          // (B + 1 == nblk: an empty block — nothing in range, keeps nothing)
    __syncthreads();
    b_end = (max_keep > 0 && s_kept_total >= max_keep) ? 0 : nblk;
  }
  __syncthreads();
  if (w < col_blocks) keep_bits[w] = s_keep_all[w];
}

// ---- compaction to ascending original indices ----------------------------------------------
__global__ void nms_flag_kernel(const unsigned long long* __restrict__ keep_bits,
                                const int* __restrict__ order, int n, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // sorted position
  if (i >= n) return;
  const bool kept = (keep_bits[i >> 6] >> (i & 63)) & 1ULL;
  flag[order ? order[i] : i] = kept ? 1 : 0;
}

// One WAVE: the kernel runs on a side stream beside GEMM workgroups that fill the SIMDs' register files (two waves of
// 204 - 237 VGPRs each), and a 1024-thread workgroup (16 waves on one CU) was only placed when a GEMM workgroup retired —
// 0.2 - 0.45 ms for ~10 us of work, on the path between the NMS sweep and the box head.  A single wave with a few
// registers is placed at once; 12 000 flags are 188 ballots.
__global__ __launch_bounds__(64) void nms_compact_kernel(const unsigned char* __restrict__ flag, int n,
                                                         int64_t* __restrict__ keep_out,
                                                         int* __restrict__ num_out) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int start = 0; start < n; start += 4 * 64) {
    // four ballots per round: the byte loads of a round are independent and issue back to back
    int f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = start + u * 64 + lane;
      f[u] = (i < n) ? (int)flag[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long ballot = __ballot(f[u]);
      if (f[u]) keep_out[base + __popcll(ballot & ((1ULL << lane) - 1ULL))] = (int64_t)(start + u * 64 + lane);
      base += __popcll(ballot);
    }
  }
  if (lane == 0) *num_out = base;
}

// Pre-ranked input (the RPN hands its boxes over in score order and wants positions in THAT order back): the kept
// positions ascend with the keep words, so the flag pass + the one-wave compaction (5 + 22 us on the path between the sweep
// and the box head) are one small kernel — thread w counts word w, an LDS scan gives its first slot, it writes its bits.
__global__ __launch_bounds__(256) void nms_compact_bits_kernel(const unsigned long long* __restrict__ keep_bits,
                                                               int col_blocks, int64_t* __restrict__ keep_out,
                                                               int* __restrict__ num_out) {
  __shared__ int s_scan[256];
  const int w = threadIdx.x;
  unsigned long long bits = w < col_blocks ? keep_bits[w] : 0ULL;
  const int mine = __popcll(bits);
  s_scan[w] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {       // inclusive Hillis-Steele scan
    const int add = w >= off ? s_scan[w - off] : 0;
    __syncthreads();
    s_scan[w] += add;
    __syncthreads();
  }
  int at = s_scan[w] - mine;
  while (bits) {
    const int b = __ffsll((long long)bits) - 1;
    keep_out[at++] = (int64_t)(w * 64 + b);
    bits &= bits - 1ULL;
  }
  if (w == 255) *num_out = s_scan[255];
}

static int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

struct NmsWorkspace {
  size_t order_off, sorted_off, mask_off, keepbits_off, flag_off, diagt_off, adjt_off, blkt_off, keys_off, total;
};

static NmsWorkspace nms_layout(int n) {
  NmsWorkspace ws;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const int col_blocks = ceil_div(n, 64);
  size_t off = 0;
  ws.order_off = off;    off = align(off + sizeof(int) * (size_t)n);
  ws.sorted_off = off;   off = align(off + sizeof(float4) * (size_t)n);
  ws.mask_off = off;     off = align(off + sizeof(unsigned long long) * ((size_t)n + 1) * col_blocks);  // + the zero row
  ws.keepbits_off = off; off = align(off + sizeof(unsigned long long) * (size_t)col_blocks);
  ws.flag_off = off;     off = align(off + (size_t)n);
  ws.diagt_off = off;    off = align(off + sizeof(unsigned long long) * (size_t)col_blocks * 64);
  ws.adjt_off = off;     off = align(off + sizeof(unsigned long long) * (size_t)col_blocks * 64);
  ws.blkt_off = off;     off = align(off + sizeof(unsigned long long) * (size_t)col_blocks * 64 * 8);
  ws.keys_off = off;
  const int p2 = next_pow2(n);
  if (p2 > kSortLdsMax) off = align(off + sizeof(Key) * (size_t)p2);
  ws.total = off;
  return ws;
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_nms_workspace_bytes(int n, size_t* bytes_out) {
  DADET_REQUIRE(n >= 0 && bytes_out, "nms_workspace_bytes: bad args");
  *bytes_out = n == 0 ? 0 : nms_layout(n).total;
  return DADET_OK;
}

extern "C" int dadet_nms(const float* boxes_xyxy, const float* scores, int n, float thresh, int tie_rule,
                         int max_keep, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                         int* num_keep_out, void* stream) {
  DADET_REQUIRE(n >= 0, "nms: n < 0");
  DADET_REQUIRE(num_keep_out, "nms: num_keep_out is null");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    (void)hipMemsetAsync(num_keep_out, 0, sizeof(int), st);
    return check_launch("nms(empty)");
  }
  DADET_REQUIRE(boxes_xyxy && keep_out && workspace, "nms: null pointer");
  const bool presorted = scores == nullptr;   // boxes already ranked by the caller (e.g. the RPN's top-k sort)
  DADET_REQUIRE(tie_rule == 0 || tie_rule == 1, "nms: tie_rule must be 0 (>=) or 1 (>)");
  DADET_REQUIRE((reinterpret_cast<uintptr_t>(boxes_xyxy) & 15) == 0, "nms: boxes must be 16-byte aligned");
  const int col_blocks = ceil_div(n, 64);
  DADET_REQUIRE(col_blocks <= 1024, "nms: n=%d exceeds the single-workgroup sweep limit (65536)", n);
  const NmsWorkspace ws = nms_layout(n);
  if (workspace_bytes < ws.total) {
    set_error("nms: workspace %zu < required %zu", workspace_bytes, ws.total);
    return DADET_EWORKSPACE;
  }
  char* base = static_cast<char*>(workspace);
  int* order = reinterpret_cast<int*>(base + ws.order_off);
  float4* sorted = reinterpret_cast<float4*>(base + ws.sorted_off);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(base + ws.mask_off);
  unsigned long long* keep_bits = reinterpret_cast<unsigned long long*>(base + ws.keepbits_off);
  unsigned char* flag = reinterpret_cast<unsigned char*>(base + ws.flag_off);
  unsigned long long* diag_t = reinterpret_cast<unsigned long long*>(base + ws.diagt_off);
  unsigned long long* adj_t = reinterpret_cast<unsigned long long*>(base + ws.adjt_off);
  unsigned long long* blk_t = reinterpret_cast<unsigned long long*>(base + ws.blkt_off);

  const int p2 = next_pow2(n);
  if (presorted) {
    order = nullptr;
    sorted = const_cast<float4*>(reinterpret_cast<const float4*>(boxes_xyxy));
  } else if (p2 <= kSortLdsMax) {
    const int threads = p2 / 2 >= 1024 ? 1024 : (p2 / 2 < 64 ? 64 : p2 / 2);
    const size_t lds = sizeof(Key) * (size_t)p2;
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("nms: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        return DADET_ELAUNCH;
      }
    }
    hipLaunchKernelGGL(nms_sort_lds_kernel, dim3(1), dim3(threads), lds, st, scores, n, p2, order);
  } else {
    Key* keys = reinterpret_cast<Key*>(base + ws.keys_off);
    hipLaunchKernelGGL(nms_sort_init_kernel, dim3(ceil_div(p2, 256)), dim3(256), 0, st, scores, n, p2,
                       keys);
    for (int size = 2; size <= p2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1)
        hipLaunchKernelGGL(nms_sort_step_kernel, dim3(ceil_div(p2 / 2, 256)), dim3(256), 0, st, keys, p2,
                           size, stride);
    hipLaunchKernelGGL(nms_sort_finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, keys, n, order);
  }
  if (!presorted)
    hipLaunchKernelGGL(nms_gather_boxes_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(boxes_xyxy), order, n, sorted);
  const dim3 mgrid(col_blocks, col_blocks);
  const int sweep_threads = col_blocks <= 64 ? 64 : ((col_blocks + 63) / 64) * 64;
  // DADET_NMS_SWEEP: 0 = the plain sweep, 1 = the chunk-pipelined sweep of round 3, otherwise (default) the 256-box block
  // sweep of round 6; the latter two hold 48 loads per thread in a 256-thread workgroup: up to 16 384 boxes
  const char* sweep_env = getenv("DADET_NMS_SWEEP");       // read per call: tests run all three sweeps in one process
  const int sweep_kind = sweep_env ? atoi(sweep_env) : 2;
  const bool pipelined = sweep_kind == 1 && sweep_threads <= 256;
  const bool blocked = sweep_kind >= 2 && sweep_threads <= 256;
  if (tie_rule == 0)
    hipLaunchKernelGGL(nms_mask_kernel<0>, mgrid, dim3(64), 0, st, sorted, n, thresh, col_blocks, mask,
                       pipelined ? diag_t : nullptr, adj_t, blocked ? blk_t : nullptr);
  else
    hipLaunchKernelGGL(nms_mask_kernel<1>, mgrid, dim3(64), 0, st, sorted, n, thresh, col_blocks, mask,
                       pipelined ? diag_t : nullptr, adj_t, blocked ? blk_t : nullptr);
  if (blocked)
    hipLaunchKernelGGL(nms_sweep_block_kernel, dim3(1), dim3(256), 0, st, mask, blk_t, n, col_blocks, max_keep, keep_bits);
  else if (!pipelined)
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(sweep_threads), 0, st, mask, n, col_blocks, max_keep,
                       keep_bits);
  else
    hipLaunchKernelGGL(nms_sweep_pipelined_kernel, dim3(1), dim3(sweep_threads), 0, st, mask, diag_t, adj_t, n,
                       col_blocks, max_keep, keep_bits);
  if (presorted && col_blocks <= 256) {
    hipLaunchKernelGGL(nms_compact_bits_kernel, dim3(1), dim3(256), 0, st, keep_bits, col_blocks, keep_out, num_keep_out);
  } else {
    hipLaunchKernelGGL(nms_flag_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, keep_bits, order, n, flag);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(64), 0, st, flag, n, keep_out, num_keep_out);
  }
  return check_launch("nms");
}
