// Sorted top-k of the RPN objectness scores, one launch for a whole batch (one workgroup per image).
//
// Replaces `objectness.topk(pre_nms_top_n, dim=1, sorted=True)` of RPNPostProcessor.forward_for_single_feature_map
// (maskrcnn_benchmark/modeling/rpn/inference.py:93-95).  The library path behind it (torch.sort -> rocprim segmented merge
// sort of all 122 880 scores per image) is 36 launches and 0.48 ms of kernel time per step; only 12 000 of the scores
// are wanted.
//
// Order: score descending, EQUAL scores by ascending index — the order of `torch.sort(descending=True, stable=True)`,
// which is what the rest of this package defines as the ranking rule (the reference's topk leaves ties unspecified).
//
// Per image: (1) radix select of the k-th largest key (3 passes over the scores, 11 + 11 + 10 bits, LDS histograms);
// (2) every element above the threshold is appended to an LDS buffer (order irrelevant), the elements EQUAL to the
// threshold go to a tie list of which the lowest indices fill the remaining slots; (3) the <= 16384 (key, index) pairs are
// bitonic-sorted in LDS as 64-bit words key << 32 | ~index; (4) scores and indices are written in order.
// The scores are read 4 times, coalesced; 480 KB per image, L2 resident after the first pass.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace dadet {

constexpr int kTopkThreads = 1024;
constexpr int kTopkCap = 16384;      // largest k (pairs sorted in LDS: 128 KB)
constexpr int kTopkTieCap = 2048;

// order-preserving map float -> uint32 (ascending); -0.0 sorts below +0.0 and NaNs above +inf (torch treats -0 == +0:
// the inputs here are sigmoid outputs, neither occurs)
__device__ inline uint32_t float_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_float(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __builtin_bit_cast(float, u);
}

// histogram update that survives clustered keys: when many lanes of a wave hit the same bin (sigmoid scores share their
// exponent bits: the first radix pass would put ~10^5 same-address LDS atomics in a row), one lane adds the count per
// distinct bin; spread-out bins take the plain per-lane atomic
__device__ inline void hist_add(int* hist, int bin, bool active, int lane) {
  unsigned long long todo = __ballot(active);
  if (!todo) return;
  const int first = __ffsll((long long)todo) - 1;
  const unsigned long long same0 = __ballot(active && bin == __shfl(bin, first, 64));
  if (__popcll(same0) < 8) {
    if (active) atomicAdd(&hist[bin], 1);
    return;
  }
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lb = __shfl(bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == lb);
    if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    todo &= ~same;
  }
}

// slot for every lane with `take` set: one LDS atomic per wave instead of one per element
__device__ inline int wave_append(int* counter, bool take, int lane) {
  const unsigned long long m = __ballot(take);
  int base = 0;
  if (m) {
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
  }
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void topk_sorted_body(const float* __restrict__ row, int n, int k, int sort_n,
                                                 float* __restrict__ o_s, int64_t* __restrict__ o_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);                 // [sort_n]
  int* hist = reinterpret_cast<int*>(smem + sizeof(unsigned long long) * (size_t)sort_n);  // [2048]
  int* ties = hist + 2048;                                                               // [kTopkTieCap]
  __shared__ uint32_t s_prefix, s_mask;
  __shared__ int s_need, s_count, s_nties;
  __shared__ int s_scan[kTopkThreads];
  const int t = threadIdx.x, lane = threadIdx.x & 63;
  constexpr int U = 4;      // independent loads in flight per thread (the passes are latency bound: one CU reads 480 KB)
  const int n_round = (n + kTopkThreads * U - 1) / (kTopkThreads * U) * (kTopkThreads * U);
  auto pack = [](uint32_t key, int i) { return ((unsigned long long)key << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i); };

  if (t == 0) { s_prefix = 0u; s_mask = 0u; s_need = k; s_count = 0; s_nties = 0; }
  __syncthreads();
  if (n > k) {
    // ---- (1) threshold key: the k-th largest.  After a pass, s_prefix / s_mask fix the high bits of the threshold and
    // s_need is the number of elements still to take among those matching the prefix
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = shifts[pass], bins = 1 << widths[pass];
      for (int b = t; b < bins; b += kTopkThreads) hist[b] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix, mask = s_mask;
      for (int i0 = t; i0 < n_round; i0 += kTopkThreads * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i0 + u * kTopkThreads < n) ? row[i0 + u * kTopkThreads] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t key = float_key(v[u]);
          hist_add(hist, (int)((key >> shift) & (bins - 1)), (i0 + u * kTopkThreads < n) && (key & mask) == prefix, lane);
        }
      }
      __syncthreads();
      if (t == 0) {
        int need = s_need, b = bins - 1;
        while (b > 0 && hist[b] < need) {      // every key in a higher bin is taken
          need -= hist[b];
          --b;
        }
        s_need = need;                         // 1 <= need <= hist[b]
        s_prefix = prefix | ((uint32_t)b << shift);
        s_mask = mask | ((uint32_t)(bins - 1) << shift);
      }
      __syncthreads();
    }
    const uint32_t thr = s_prefix;
    const int need = s_need;                   // how many of the keys == thr are taken (lowest indices first)
    // ---- (2) collect
    for (int i0 = t; i0 < n_round; i0 += kTopkThreads * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i0 + u * kTopkThreads < n) ? row[i0 + u * kTopkThreads] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * kTopkThreads;
        const uint32_t key = float_key(v[u]);
        const bool above_thr = i < n && key > thr, tie = i < n && key == thr;
        const int slot = wave_append(&s_count, above_thr, lane);
        if (above_thr) buf[slot] = pack(key, i);
        const int tslot = wave_append(&s_nties, tie, lane);
        if (tie && tslot < kTopkTieCap) ties[tslot] = i;
      }
    }
    __syncthreads();
    const int above = s_count, nties = s_nties;
    if (nties <= kTopkTieCap) {
      if (nties > need) {                      // only some of the ties fit: the lowest indices
        int p2 = 1;
        while (p2 < nties) p2 <<= 1;
        for (int i = nties + t; i < p2; i += kTopkThreads) ties[i] = 0x7FFFFFFF;
        __syncthreads();
        for (int kk = 2; kk <= p2; kk <<= 1)
          for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = t; i < p2; i += kTopkThreads) {
              const int p = i ^ j;
              if (p > i) {
                const int a = ties[i], b = ties[p];
                if ((a > b) == ((i & kk) == 0)) { ties[i] = b; ties[p] = a; }
              }
            }
            __syncthreads();
          }
      }
      for (int i = t; i < need; i += kTopkThreads) buf[above + i] = pack(thr, ties[i]);
    } else {
      // more equal scores than the tie list holds (saturated scores): ordered compaction over the whole row, one chunk
      // of 1024 consecutive elements at a time; the first `need` of them in index order are taken
      int taken = 0;
      for (int base = 0; base < n && taken < need; base += kTopkThreads) {
        const int i = base + t;
        const int flag = (i < n && float_key(row[i]) == thr) ? 1 : 0;
        s_scan[t] = flag;
        __syncthreads();
        for (int off = 1; off < kTopkThreads; off <<= 1) {
          const int v = t >= off ? s_scan[t - off] : 0;
          __syncthreads();
          s_scan[t] += v;
          __syncthreads();
        }
        const int pos = taken + s_scan[t] - flag;
        if (flag && pos < need) buf[above + pos] = pack(thr, i);
        taken += s_scan[kTopkThreads - 1];
        __syncthreads();
      }
    }
  } else {
    for (int i = t; i < n; i += kTopkThreads) buf[i] = pack(float_key(row[i]), i);
  }
  const int have = n > k ? k : n;
  for (int i = have + t; i < sort_n; i += kTopkThreads) buf[i] = 0ull;     // padding sorts last
  __syncthreads();
  // ---- (3) bitonic sort, descending
  for (int kk = 2; kk <= sort_n; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int h = t; h < (sort_n >> 1); h += kTopkThreads) {
        const int i = ((h & ~(j - 1)) << 1) | (h & (j - 1));
        const int p = i | j;
        const unsigned long long a = buf[i], b = buf[p];
        if ((a < b) == ((i & kk) == 0)) { buf[i] = b; buf[p] = a; }
      }
      __syncthreads();
    }
  // ---- (4) output
  for (int i = t; i < have; i += kTopkThreads) {
    const unsigned long long v = buf[i];
    o_s[i] = key_float((uint32_t)(v >> 32));
    o_i[i] = (int64_t)(0xFFFFFFFFu - (uint32_t)v);
  }
}

__global__ __launch_bounds__(kTopkThreads) void topk_sorted_kernel(const float* __restrict__ scores, int n, int64_t row_stride,
                                                                  int k, int sort_n, float* __restrict__ out_scores,
                                                                  int64_t* __restrict__ out_idx) {
  topk_sorted_body(scores + (size_t)blockIdx.x * row_stride, n, k, sort_n, out_scores + (size_t)blockIdx.x * k,
                   out_idx + (size_t)blockIdx.x * k);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same ranking for SEVERAL rows of different lengths in five launches, the passes spread over the chip.
// Use: the multi-level proposal selection of a feature pyramid (rpn/inference.py:124-152 of the reference: per level
// `objectness.topk(pre_nms_top_n)` over N x 3 x H_l x W_l scores — five levels x two images = ten rows of 1 536 .. 393 216
// scores at 1024 x 2048, k = 2000).  The library route is one segmented sort per level (17 launches each plus the slicing
// around it: ~100 launches of a host-bound stretch in which no GEMM runs); the one-workgroup kernel above would walk the
// 1.5 MB row of the finest level four times from one CU.  Here every pass is ONE launch over all rows with one workgroup
// per 8192 scores:
//   topk_rows_hist_kernel<P>, P = 0, 1, 2: histogram of radix digit P (11 + 11 + 10 bits) of the keys that match the
//     threshold prefix found so far, per-workgroup in LDS, then added to the row's global histogram.  The prefix is not
//     handed over by a kernel of its own: every workgroup of pass P re-derives it from the global histograms of the
//     passes before (2048 bins, one scan) — same data, same answer in every workgroup.
//   topk_rows_collect_kernel: keys above the threshold go to the row's candidate list (wave-aggregated global appends),
//     keys equal to it to a tie list.
//   topk_rows_finish_kernel (one workgroup per row): the lowest-index ties fill the remaining slots, the k pairs are
//     sorted in LDS (score descending, index ascending) and written.  More ties than the tie list holds (saturated
//     scores): that row is redone by the one-workgroup algorithm above, so the result is always the same.
constexpr int kRowsMax = 16;
constexpr int kRowsChunk = 8192;
struct TopkRow {
  const float* scores;
  float* out_scores;
  int64_t* out_idx;
  int n, k, chunk_begin, pad;
};
struct TopkRows {
  TopkRow r[kRowsMax];
  int rows, total_chunks;
};
struct TopkRowState {            // per row, zeroed before the first pass
  int hist[3][2048];
  int above, nties, pad[2];
};
constexpr int kRowsTieCap = 2048;

__device__ inline void rows_locate(const TopkRows& tb, int block, int* row, int* chunk) {
  int r = 0;
  while (r + 1 < tb.rows && block >= tb.r[r + 1].chunk_begin) ++r;
  *row = r;
  *chunk = block - tb.r[r].chunk_begin;
}

// threshold prefix / mask / still-to-take count after `passes` radix passes, from the row's global histograms
__device__ inline void rows_threshold(const TopkRowState* st, int k, int passes, uint32_t* prefix, uint32_t* mask, int* need,
                                      int* scratch /* LDS, 257 ints */) {
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  const int t = threadIdx.x, nthreads = blockDim.x;
  uint32_t pf = 0u, mk = 0u;
  int nd = k;
  for (int p = 0; p < passes; ++p) {
    const int bins = 1 << widths[p];
    const int per = bins / 256;                      // 8 or 4 bins per scanning thread (the first 256 threads scan)
    // suffix sums from the top bin down: thread j owns bins [bins - (j + 1) per, bins - j per)
    int mine = 0;
    if (t < 256)
      for (int i = 0; i < per; ++i) mine += st->hist[p][bins - 1 - (t * per + i)];
    if (t < 256) scratch[t] = mine;
    __syncthreads();
    if (t == 0) {
      int acc = 0, j = 0;
      while (j < 255 && acc + scratch[j] < nd) {     // every key in these bins is taken
        acc += scratch[j];
        ++j;
      }
      int b = bins - 1 - j * per, left = nd - acc;
      while (b > bins - (j + 1) * per && st->hist[p][b] < left) {
        left -= st->hist[p][b];
        --b;
      }
      scratch[256] = b;
      scratch[0] = left;
    }
    __syncthreads();
    const int b = scratch[256];
    nd = scratch[0];
    pf |= (uint32_t)b << shifts[p];
    mk |= (uint32_t)(bins - 1) << shifts[p];
    __syncthreads();
    (void)nthreads;
  }
  *prefix = pf;
  *mask = mk;
  *need = nd;
}

template <int PASS>
__global__ __launch_bounds__(256) void topk_rows_hist_kernel(const TopkRows tb, TopkRowState* __restrict__ states) {
  __shared__ int hist[2048];
  __shared__ int scratch[257];
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  int row, chunk;
  rows_locate(tb, (int)blockIdx.x, &row, &chunk);
  const TopkRow& r = tb.r[row];
  if (r.n <= r.k) return;                            // the whole row is taken: nothing to select
  TopkRowState* st = states + row;
  const int t = threadIdx.x, lane = t & 63;
  uint32_t prefix, mask;
  int need;
  rows_threshold(st, r.k, PASS, &prefix, &mask, &need, scratch);
  const int bins = 1 << widths[PASS], shift = shifts[PASS];
  for (int b = t; b < bins; b += 256) hist[b] = 0;
  __syncthreads();
  const int lo = chunk * kRowsChunk;
  for (int i0 = lo + t; i0 < lo + kRowsChunk; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i0 + u * 256 < r.n) ? r.scores[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t key = float_key(v[u]);
      hist_add(hist, (int)((key >> shift) & (bins - 1)), (i0 + u * 256 < r.n) && (key & mask) == prefix, lane);
    }
  }
  __syncthreads();
  for (int b = t; b < bins; b += 256)
    if (hist[b]) atomicAdd(&st->hist[PASS][b], hist[b]);
}

__global__ __launch_bounds__(256) void topk_rows_collect_kernel(const TopkRows tb, TopkRowState* __restrict__ states,
                                                                unsigned long long* __restrict__ cand, int cand_stride,
                                                                int* __restrict__ ties) {
  __shared__ int scratch[257];
  int row, chunk;
  rows_locate(tb, (int)blockIdx.x, &row, &chunk);
  const TopkRow& r = tb.r[row];
  if (r.n <= r.k) return;
  TopkRowState* st = states + row;
  const int t = threadIdx.x, lane = t & 63;
  uint32_t thr, mask;
  int need;
  rows_threshold(st, r.k, 3, &thr, &mask, &need, scratch);
  unsigned long long* my_cand = cand + (size_t)row * cand_stride;
  int* my_ties = ties + (size_t)row * kRowsTieCap;
  const int lo = chunk * kRowsChunk;
  for (int i0 = lo + t; i0 < lo + kRowsChunk; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i0 + u * 256 < r.n) ? r.scores[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      const uint32_t key = float_key(v[u]);
      const bool above = i < r.n && key > thr, tie = i < r.n && key == thr;
      const int slot = wave_append(&st->above, above, lane);
      if (above) my_cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
      const int tslot = wave_append(&st->nties, tie, lane);
      if (tie && tslot < kRowsTieCap) my_ties[tslot] = i;
    }
  }
}

__global__ __launch_bounds__(kTopkThreads) void topk_rows_finish_kernel(const TopkRows tb, TopkRowState* __restrict__ states,
                                                                       const unsigned long long* __restrict__ cand,
                                                                       int cand_stride, const int* __restrict__ ties,
                                                                       int sort_n_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);                      // [sort_n]
  int* tie = reinterpret_cast<int*>(smem + sizeof(unsigned long long) * (size_t)sort_n_max) + 2048;   // the body's tie area
  __shared__ int scratch[257];
  const int row = (int)blockIdx.x;
  const TopkRow& r = tb.r[row];
  TopkRowState* st = states + row;
  const int t = threadIdx.x;
  int sort_n = 2;
  while (sort_n < r.k) sort_n <<= 1;
  const int have = r.n > r.k ? r.k : r.n;
  if (r.n > r.k) {
    uint32_t thr, mask;
    int need;
    rows_threshold(st, r.k, 3, &thr, &mask, &need, scratch);
    const int above = st->above, nties = st->nties;
    __syncthreads();
    if (nties > kRowsTieCap) {                       // saturated scores: the one-workgroup algorithm redoes this row
      topk_sorted_body(r.scores, r.n, r.k, sort_n, r.out_scores, r.out_idx);
      return;
    }
    for (int i = t; i < above; i += kTopkThreads) buf[i] = cand[(size_t)row * cand_stride + i];
    int p2 = 1;
    while (p2 < nties) p2 <<= 1;
    for (int i = t; i < p2; i += kTopkThreads) tie[i] = i < nties ? ties[(size_t)row * kRowsTieCap + i] : 0x7FFFFFFF;
    __syncthreads();
    for (int kk = 2; kk <= p2; kk <<= 1)
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int i = t; i < p2; i += kTopkThreads) {
          const int p = i ^ j;
          if (p > i) {
            const int a = tie[i], b = tie[p];
            if ((a > b) == ((i & kk) == 0)) { tie[i] = b; tie[p] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = t; i < need; i += kTopkThreads)
      buf[above + i] = ((unsigned long long)thr << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)tie[i]);
  } else {
    for (int i = t; i < r.n; i += kTopkThreads)
      buf[i] = ((unsigned long long)float_key(r.scores[i]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
  }
  for (int i = have + t; i < sort_n; i += kTopkThreads) buf[i] = 0ull;
  __syncthreads();
  for (int kk = 2; kk <= sort_n; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int h = t; h < (sort_n >> 1); h += kTopkThreads) {
        const int i = ((h & ~(j - 1)) << 1) | (h & (j - 1));
        const int p = i | j;
        const unsigned long long a = buf[i], b = buf[p];
        if ((a < b) == ((i & kk) == 0)) { buf[i] = b; buf[p] = a; }
      }
      __syncthreads();
    }
  for (int i = t; i < have; i += kTopkThreads) {
    const unsigned long long v = buf[i];
    r.out_scores[i] = key_float((uint32_t)(v >> 32));
    r.out_idx[i] = (int64_t)(0xFFFFFFFFu - (uint32_t)v);
  }
}

}  // namespace dadet

using namespace dadet;

extern "C" int dadet_topk_sorted_rows_workspace_bytes(int rows, int k_max, size_t* bytes_out) {
  DADET_REQUIRE(rows >= 0 && rows <= kRowsMax && k_max > 0 && k_max <= kTopkCap && bytes_out,
                "topk_sorted_rows_workspace_bytes: rows=%d (<= %d) k_max=%d (<= %d)", rows, kRowsMax, k_max, kTopkCap);
  *bytes_out = sizeof(TopkRowState) * (size_t)rows + sizeof(unsigned long long) * (size_t)rows * k_max +
               sizeof(int) * (size_t)rows * kRowsTieCap + 256;
  return DADET_OK;
}

extern "C" int dadet_topk_sorted_rows(const dadet_topk_row* rows_in, int rows, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  DADET_REQUIRE(rows >= 0 && rows <= kRowsMax, "topk_sorted_rows: %d rows (at most %d per call)", rows, kRowsMax);
  if (rows == 0) return DADET_OK;
  DADET_REQUIRE(rows_in && workspace, "topk_sorted_rows: null pointer");
  TopkRows tb;
  int chunks = 0, k_max = 0;
  for (int i = 0; i < rows; ++i) {
    const dadet_topk_row& s = rows_in[i];
    DADET_REQUIRE(s.scores && s.out_scores && s.out_idx && s.n > 0 && s.k > 0 && s.k <= kTopkCap && s.k <= s.n,
                  "topk_sorted_rows: row %d is malformed (n=%d k=%d, 0 < k <= min(n, %d))", i, s.n, s.k, kTopkCap);
    tb.r[i].scores = s.scores; tb.r[i].out_scores = s.out_scores; tb.r[i].out_idx = s.out_idx;
    tb.r[i].n = s.n; tb.r[i].k = s.k; tb.r[i].chunk_begin = chunks; tb.r[i].pad = 0;
    chunks += ceil_div(s.n, kRowsChunk);
    k_max = s.k > k_max ? s.k : k_max;
  }
  tb.rows = rows;
  tb.total_chunks = chunks;
  size_t need = 0;
  (void)dadet_topk_sorted_rows_workspace_bytes(rows, k_max, &need);
  DADET_REQUIRE(workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                "topk_sorted_rows: workspace of %zu bytes (need %zu, 8-byte aligned)", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  TopkRowState* states = static_cast<TopkRowState*>(workspace);
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(states + rows);
  int* ties = reinterpret_cast<int*>(cand + (size_t)rows * k_max);
  (void)hipMemsetAsync(states, 0, sizeof(TopkRowState) * (size_t)rows, st);
  hipLaunchKernelGGL(topk_rows_hist_kernel<0>, dim3(chunks), dim3(256), 0, st, tb, states);
  hipLaunchKernelGGL(topk_rows_hist_kernel<1>, dim3(chunks), dim3(256), 0, st, tb, states);
  hipLaunchKernelGGL(topk_rows_hist_kernel<2>, dim3(chunks), dim3(256), 0, st, tb, states);
  hipLaunchKernelGGL(topk_rows_collect_kernel, dim3(chunks), dim3(256), 0, st, tb, states, cand, k_max, ties);
  int sort_n = 2;
  while (sort_n < k_max) sort_n <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)sort_n + sizeof(int) * (2048 + kTopkTieCap);
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_finish_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(unsigned long long) * kTopkCap + sizeof(int) * (2048 + kTopkTieCap)));
    if (e != hipSuccess) {
      set_error("topk_sorted_rows: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(topk_rows_finish_kernel, dim3(rows), dim3(kTopkThreads), lds, st, tb, states, cand, k_max, ties, sort_n);
  return check_launch("topk_sorted_rows");
}

extern "C" int dadet_topk_sorted(const float* scores, int rows, int n, int64_t row_stride, int k, float* out_scores,
                                 int64_t* out_idx, void* stream) {
  DADET_REQUIRE(rows >= 0 && n >= 0 && k > 0 && k <= kTopkCap, "topk_sorted: rows=%d n=%d k=%d (k <= %d)", rows, n, k,
                kTopkCap);
  DADET_REQUIRE(k <= n, "topk_sorted: k=%d exceeds the row length %d", k, n);
  DADET_REQUIRE(row_stride >= n, "topk_sorted: row stride %lld < n", (long long)row_stride);
  if (rows == 0) return DADET_OK;
  DADET_REQUIRE(scores && out_scores && out_idx, "topk_sorted: null pointer");
  int sort_n = 2;
  while (sort_n < k) sort_n <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)sort_n + sizeof(int) * (2048 + kTopkTieCap);
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_sorted_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(unsigned long long) * kTopkCap + sizeof(int) * (2048 + kTopkTieCap)));
    if (e != hipSuccess) {
      set_error("topk_sorted: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(topk_sorted_kernel, dim3(rows), dim3(kTopkThreads), lds, as_stream(stream), scores, n, row_stride, k,
                     sort_n, out_scores, out_idx);
  return check_launch("topk_sorted");
}
