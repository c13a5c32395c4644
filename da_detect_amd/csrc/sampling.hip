// Box-head proposal sampling in ONE launch per image.
//
// Replaces, for one image, the chain BalancedPositiveNegativeSampler.__call__
// (maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:25-68: two nonzero, two randperm, two index_put)
// + FastRCNNLossComputation.subsample's nonzero over (pos | neg) and the BoxList gather of every field
// (modeling/roi_heads/box_head/loss.py:95-130) — ~60 ATen launches and a host round trip per image, all of them
// latency bound and in front of the box head (tools/gap_analysis.py: 1.6 ms with no GEMM running).
//
// Same distribution as the reference: num_pos = min(#positives, max_pos) positives and num_neg =
// min(#negatives, cap - num_pos) negatives, each a uniformly random subset; the survivors come out in ascending
// proposal order (the reference's nonzero over the union mask).  The random subset is "the k smallest of iid random
// keys" (splitmix64 of (seed, index)) instead of "the first k of a random permutation"; labels < 0 are never taken.
// One workgroup: (class, key, index) packed into 64 bits, bitonic sort in LDS, flag the survivors, block scan.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace dadet {

constexpr int kSampleThreads = 1024;
constexpr int kSampleMaxN = 4096;   // 4096 x (8 + 4) bytes of LDS

__device__ inline uint32_t sample_key(uint64_t seed, uint32_t idx) {
  uint64_t z = seed + (uint64_t)(idx + 1u) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

__global__ __launch_bounds__(kSampleThreads) void sample_rois_kernel(
    const float4* __restrict__ boxes, const int64_t* __restrict__ labels, const float4* __restrict__ reg, int n, int N,
    int cap, int max_pos, uint64_t seed, int is_source, int64_t* __restrict__ idx_out, float4* __restrict__ boxes_out,
    int64_t* __restrict__ labels_out, float4* __restrict__ reg_out, int64_t* __restrict__ loss_labels_out,
    unsigned char* __restrict__ domain_out, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);          // [N]
  int* flag = reinterpret_cast<int*>(smem + sizeof(uint64_t) * N);  // [N]
  __shared__ int s_cnt[2];
  __shared__ int s_scan[kSampleThreads];
  const int t = threadIdx.x;
  if (t < 2) s_cnt[t] = 0;
  __syncthreads();
  int my_pos = 0, my_neg = 0;
  for (int i = t; i < N; i += kSampleThreads) {
    uint64_t cls = 3;   // padding sorts last
    if (i < n) {
      const int64_t lab = labels ? labels[i] : 0;
      cls = lab >= 1 ? 0 : (lab == 0 ? 1 : 2);
      my_pos += cls == 0;
      my_neg += cls == 1;
    }
    keys[i] = (cls << 45) | ((uint64_t)sample_key(seed, (uint32_t)i) << 13) | (uint64_t)i;
    flag[i] = 0;
  }
  if (my_pos) atomicAdd(&s_cnt[0], my_pos);
  if (my_neg) atomicAdd(&s_cnt[1], my_neg);
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < N; i += kSampleThreads) {
        const int p = i ^ j;
        if (p > i) {
          const uint64_t a = keys[i], b = keys[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[p] = a;
          }
        }
      }
      __syncthreads();
    }
  const int n_pos = s_cnt[0], n_neg = s_cnt[1];
  const int num_pos = n_pos < max_pos ? n_pos : max_pos;
  const int num_neg = n_neg < cap - num_pos ? n_neg : cap - num_pos;
  // sorted order: positives (by key), negatives (by key), ignored, padding
  for (int p = t; p < N; p += kSampleThreads) {
    const bool take = p < num_pos || (p >= n_pos && p < n_pos + num_neg);
    if (take) flag[(int)(keys[p] & 0x1FFFu)] = 1;
  }
  __syncthreads();
  // ascending-index compaction: thread t owns indices [t*C, (t+1)*C)
  const int C = (N + kSampleThreads - 1) / kSampleThreads;
  int mine = 0;
  for (int c = 0; c < C; ++c) {
    const int i = t * C + c;
    if (i < N) mine += flag[i];
  }
  s_scan[t] = mine;
  __syncthreads();
  for (int off = 1; off < kSampleThreads; off <<= 1) {
    const int v = t >= off ? s_scan[t - off] : 0;
    __syncthreads();
    s_scan[t] += v;
    __syncthreads();
  }
  int out = s_scan[t] - mine;
  for (int c = 0; c < C; ++c) {
    const int i = t * C + c;
    if (i < N && flag[i]) {
      const int64_t lab = labels ? labels[i] : 0;
      idx_out[out] = i;
      boxes_out[out] = boxes[i];
      labels_out[out] = lab;
      reg_out[out] = reg ? reg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      loss_labels_out[out] = is_source ? lab : -1;
      domain_out[out] = is_source ? 1 : 0;
      ++out;
    }
  }
  const int total = num_pos + num_neg;
  for (int r = total + t; r < cap; r += kSampleThreads) {   // rows past the sample: defined, never read
    idx_out[r] = -1;
    boxes_out[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    labels_out[r] = 0;
    reg_out[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    loss_labels_out[r] = -1;
    domain_out[r] = is_source ? 1 : 0;
  }
  if (t == 0) {
    counts[0] = total;
    counts[1] = num_pos;
  }
}

}  // namespace dadet

using namespace dadet;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int dadet_sample_rois(const float* boxes, const int64_t* labels, const float* regression_targets, int n,
                                 int cap, int max_pos, uint64_t seed, int is_source, int64_t* idx_out,
                                 float* boxes_out, int64_t* labels_out, float* regression_targets_out,
                                 int64_t* loss_labels_out, unsigned char* domain_out, int* counts_out, void* stream) {
  DADET_REQUIRE(n >= 0 && n <= kSampleMaxN, "sample_rois: n=%d outside 0..%d", n, kSampleMaxN);
  DADET_REQUIRE(cap > 0 && max_pos >= 0 && max_pos <= cap, "sample_rois: bad cap=%d / max_pos=%d", cap, max_pos);
  DADET_REQUIRE((n == 0 || boxes) && idx_out && boxes_out && labels_out && regression_targets_out &&
                    loss_labels_out && domain_out && counts_out,
                "sample_rois: null pointer");
  DADET_REQUIRE(al16(boxes) && al16(regression_targets) && al16(boxes_out) && al16(regression_targets_out),
                "sample_rois: box arrays must be 16-byte aligned");
  int N = 2;
  while (N < n) N <<= 1;
  const size_t lds = (sizeof(uint64_t) + sizeof(int)) * (size_t)N;
  hipLaunchKernelGGL(sample_rois_kernel, dim3(1), dim3(kSampleThreads), lds, as_stream(stream),
                     reinterpret_cast<const float4*>(boxes), labels, reinterpret_cast<const float4*>(regression_targets),
                     n, N, cap, max_pos, (uint64_t)seed, is_source, idx_out, reinterpret_cast<float4*>(boxes_out),
                     labels_out, reinterpret_cast<float4*>(regression_targets_out), loss_labels_out, domain_out,
                     counts_out);
  return check_launch("sample_rois");
}
