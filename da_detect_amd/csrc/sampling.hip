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
#include <mutex>
#include <unordered_map>
#include <stdlib.h>
#include "box_match.h"

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

// ------------------------------------------------------------------------------------------------------------------
// Proposal hand-over on the device (round 3): NMS -> box-head sample of ONE image without the kept-count round trip.
//
// Between the RPN's NMS and the box head's sampler the reference (and this package until round 2) goes through the host:
// the kept count is read back (rpn/inference.py:102 `boxlist = boxlist[keep]`, a device->host synchronisation), the kept
// boxes and scores are gathered, the ground-truth boxes of source images are concatenated (rpn/inference.py:51-74), then
// FastRCNNLossComputation.prepare_targets / subsample run on a proposal list whose length the host knows
// (roi_heads/box_head/loss.py:55-130).  Here ONE single-workgroup launch reads the NMS result where it lies —
// keep[0 .. count) in device memory — and does all of it: proposal i is sorted_boxes[keep[i]] for i < min(count, post_n),
// then the G ground-truth boxes (source images); IoU / Matcher / label rules / BoxCoder.encode per proposal exactly as
// dadet_box_match_encode (same arithmetic order, match_encode_one in box_match.h); the balanced random sample exactly
// as sample_rois_kernel above (same keys: splitmix64(seed, position in the proposal list)).  It also leaves the proposal
// list itself (boxes, objectness, length) in device buffers, for callers that want it on the host later.
__global__ __launch_bounds__(kSampleThreads) void proposals_sample_kernel(
    const float4* __restrict__ sorted_boxes, const float* __restrict__ sorted_scores, const int64_t* __restrict__ keep,
    const int* __restrict__ count_dev, int post_n, const float4* __restrict__ app_gts, int G_app,
    const float4* __restrict__ gts, const int64_t* __restrict__ gt_labels, int G, float high, float low, float wx,
    float wy, float ww, float wh, int N, int cap, int max_pos, uint64_t seed,
    int is_source, float4* __restrict__ prop_boxes, float* __restrict__ prop_scores, int* __restrict__ n_props,
    int64_t* __restrict__ idx_out, float4* __restrict__ boxes_out, int64_t* __restrict__ labels_out,
    float4* __restrict__ reg_out, int64_t* __restrict__ loss_labels_out, unsigned char* __restrict__ domain_out,
    float* __restrict__ obj_out, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                     // [N]
  int* flag = reinterpret_cast<int*>(smem + sizeof(uint64_t) * N);        // [N]
  float4* g_box = reinterpret_cast<float4*>(smem + (sizeof(uint64_t) + sizeof(int)) * N);   // [G]
  float* g_area = reinterpret_cast<float*>(g_box + G);                     // [G]
  __shared__ int s_cnt[2];
  __shared__ int s_scan[kSampleThreads];
  const int t = threadIdx.x;
  if (t < 2) s_cnt[t] = 0;
  for (int g = t; g < G; g += kSampleThreads) {
    const float4 b = gts[g];
    g_box[g] = b;
    g_area[g] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  }
  int kept = *count_dev;
  if (kept > post_n) kept = post_n;
  // appended boxes: add_gt_proposals gives SOURCE images their own ground truth (rpn/inference.py:51-74).  The boxes
  // matched against (gts) are the target list the box head was called with — the same boxes, except in the aligned
  // triplet passes, which pool another image's proposals (generalized_rcnn.py:110-112)
  const int n = kept + G_app;
  __syncthreads();
  auto proposal = [&](int i, float* score) -> float4 {
    if (i < kept) {
      const int64_t src = keep[i];
      *score = sorted_scores[src];
      return sorted_boxes[src];
    }
    *score = 1.f;
    return app_gts[i - kept];
  };
  int my_pos = 0, my_neg = 0;
  for (int i = t; i < N; i += kSampleThreads) {
    uint64_t cls = 3;   // padding sorts last
    if (i < n) {
      float sc;
      const float4 p = proposal(i, &sc);
      prop_boxes[i] = p;
      prop_scores[i] = sc;
      int64_t lab = 0;
      if (is_source) {
        float4 reg;
        lab = match_encode_one(p, g_box, g_area, gt_labels, G, high, low, wx, wy, ww, wh, &reg);
      }
      cls = lab >= 1 ? 0 : (lab == 0 ? 1 : 2);
      my_pos += cls == 0;
      my_neg += cls == 1;
    }
    keys[i] = (cls << 45) | ((uint64_t)sample_key(seed, (uint32_t)i) << 13) | (uint64_t)i;
    flag[i] = 0;
  }
  if (t == 0) *n_props = n;
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
  for (int p = t; p < N; p += kSampleThreads) {
    const bool take = p < num_pos || (p >= n_pos && p < n_pos + num_neg);
    if (take) flag[(int)(keys[p] & 0x1FFFu)] = 1;
  }
  __syncthreads();
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
      float sc;
      const float4 p = proposal(i, &sc);
      int64_t lab = 0;
      float4 reg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (is_source) lab = match_encode_one(p, g_box, g_area, gt_labels, G, high, low, wx, wy, ww, wh, &reg);
      idx_out[out] = i;
      boxes_out[out] = p;
      labels_out[out] = lab;
      reg_out[out] = reg;
      loss_labels_out[out] = is_source ? lab : -1;
      domain_out[out] = is_source ? 1 : 0;
      obj_out[out] = sc;
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
    obj_out[r] = 0.f;
  }
  if (t == 0) {
    counts[0] = total;
    counts[1] = num_pos;
  }
}

// ---- multi-level proposal lists laid end to end (training-mode selection over a feature pyramid) -------------------------
// One (level, image) pair per blockIdx.y: its NMS result — positions `keep[0 .. count)` into the score-ordered candidates —
// is copied into the image's fixed-capacity buffers at the pair's offset; the slots behind the kept count get score -1
// (and the first candidate's box, never read).  Replaces clamp / arange / compare / two gathers / where per pair
// (rpn/inference.py `_select_over_all_levels_device`): one launch instead of ~100 on the host-bound stretch between the RPN
// head and the box head.
struct MergeEntry {
  const float4* boxes;
  const float* scores;
  const int64_t* keep;
  const int* count;
  float4* boxes_out;
  float* scores_out;
  int n, cap;
};
constexpr int kMergeMax = 24;
struct MergeTable {
  MergeEntry e[kMergeMax];
};

__global__ __launch_bounds__(256) void fpn_merge_levels_kernel(const MergeTable t) {
  const MergeEntry& e = t.e[blockIdx.y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= e.cap) return;
  const int cnt = min(*e.count, e.cap);
  const bool valid = j < cnt;
  int64_t idx = valid ? e.keep[j] : 0;
  if (idx < 0 || idx >= e.n) idx = 0;
  e.boxes_out[j] = e.boxes[idx];
  e.scores_out[j] = valid ? e.scores[idx] : -1.f;
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

extern "C" int dadet_proposals_sample(const float* sorted_boxes, const float* sorted_scores, const int64_t* keep,
                                      const int* count_dev, int post_n, const float* appended_boxes, int num_appended,
                                      const float* gt_boxes, const int64_t* gt_labels, int G, float high_threshold,
                                      float low_threshold, float wx, float wy, float ww,
                                      float wh, int cap, int max_pos, uint64_t seed, int is_source, float* prop_boxes,
                                      float* prop_scores, int* n_props, int64_t* idx_out, float* boxes_out,
                                      int64_t* labels_out, float* regression_targets_out, int64_t* loss_labels_out,
                                      unsigned char* domain_out, float* objectness_out, int* counts_out, void* stream) {
  using namespace dadet;
  DADET_REQUIRE(post_n >= 0 && G >= 0 && num_appended >= 0 && post_n + num_appended <= kSampleMaxN,
                "proposals_sample: post_n + appended = %d outside 0..%d", post_n + num_appended, kSampleMaxN);
  DADET_REQUIRE(num_appended == 0 || appended_boxes, "proposals_sample: null appended boxes");
  DADET_REQUIRE(G <= 1024, "proposals_sample: G=%d ground-truth boxes exceed the LDS table", G);
  DADET_REQUIRE(cap > 0 && max_pos >= 0 && max_pos <= cap, "proposals_sample: bad cap=%d / max_pos=%d", cap, max_pos);
  DADET_REQUIRE(sorted_boxes && sorted_scores && keep && count_dev && prop_boxes && prop_scores && n_props && idx_out &&
                    boxes_out && labels_out && regression_targets_out && loss_labels_out && domain_out &&
                    objectness_out && counts_out && (G == 0 || (gt_boxes && (!is_source || gt_labels))),
                "proposals_sample: null pointer");
  DADET_REQUIRE(!is_source || G > 0, "proposals_sample: a source image needs ground-truth boxes");
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  DADET_REQUIRE(a16(sorted_boxes) && a16(gt_boxes) && a16(appended_boxes) && a16(prop_boxes) && a16(boxes_out) &&
                    a16(regression_targets_out),
                "proposals_sample: box arrays must be 16-byte aligned");
  int N = 2;
  while (N < post_n + num_appended) N <<= 1;
  const size_t lds = (sizeof(uint64_t) + sizeof(int)) * (size_t)N + (size_t)G * 20;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proposals_sample_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((sizeof(uint64_t) + sizeof(int)) * kSampleMaxN + 1024 * 20));
    if (e != hipSuccess) {
      set_error("proposals_sample: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DADET_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(proposals_sample_kernel, dim3(1), dim3(kSampleThreads), lds, as_stream(stream),
                     reinterpret_cast<const float4*>(sorted_boxes), sorted_scores, keep, count_dev, post_n,
                     reinterpret_cast<const float4*>(appended_boxes), num_appended,
                     reinterpret_cast<const float4*>(gt_boxes), gt_labels, G, high_threshold, low_threshold, wx, wy, ww,
                     wh, N, cap, max_pos, (uint64_t)seed, is_source, reinterpret_cast<float4*>(prop_boxes), prop_scores,
                     n_props, idx_out, reinterpret_cast<float4*>(boxes_out), labels_out,
                     reinterpret_cast<float4*>(regression_targets_out), loss_labels_out, domain_out, objectness_out,
                     counts_out);
  return check_launch("proposals_sample");
}

// ------------------------------------------------------------------------------------------------------------------
// RPN anchor sampling of ONE image in one launch: replaces BalancedPositiveNegativeSampler.__call__ on the anchor
// labels (modeling/balanced_positive_negative_sampler.py:25-68: two nonzero over ~10^5 anchors, two randperm, two
// index_put per image) + the nonzero / gathers of RPNLossComputation.__call__ (modeling/rpn/loss.py:101-123).  On a
// busy GPU the library's nonzero kernels alone took 0.2 - 0.5 ms each, and on slower hosts the chain ended after
// the RPN head, i.e. on the critical path (tools/gemm_table.py --holes: 0.7 ms).
//
// Same rule as sample_rois_kernel — the k smallest random keys of each class — but the anchors do not fit a sort in
// LDS: a 4-pass radix select (8 bits per pass, both classes in the same pass) finds each class's threshold key, one
// more pass appends the selected indices to LDS lists, and the (<= 256-entry) lists are sorted to restore the
// ascending anchor order of the reference's nonzero.  One workgroup; all passes read labels[t + 1024 j] (coalesced,
// L2 resident after the first).
namespace dadet {

constexpr int kAnchorCapMax = 1024;   // largest batch_size_per_image supported
constexpr int kTieCap = 1024;

__device__ inline void lds_bitonic_sort(int* v, int n_pow2, int t, int nthreads) {
  for (int k = 2; k <= n_pow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < n_pow2; i += nthreads) {
        const int p = i ^ j;
        if (p > i) {
          const int a = v[i], b = v[p];
          if ((a > b) == ((i & k) == 0)) {
            v[i] = b;
            v[p] = a;
          }
        }
      }
      __syncthreads();
    }
}

__device__ __forceinline__ void sample_anchors_body(
    const float* __restrict__ labels, const float4* __restrict__ reg, int A, int cap, int max_pos, uint64_t seed,
    int64_t index_offset, int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out, float4* __restrict__ reg_pos_out,
    int* __restrict__ counts) {
  __shared__ int s_hist[2][256];
  __shared__ int s_n[2];          // candidates per class
  __shared__ int s_k[2];          // still to take inside the current prefix
  __shared__ unsigned s_prefix[2];
  __shared__ int s_all[2];        // class takes every candidate (no threshold)
  __shared__ int s_list[2][kAnchorCapMax];
  __shared__ int s_len[2];
  __shared__ int s_tie[2][kTieCap];
  __shared__ int s_tie_len[2];
  const int t = threadIdx.x;
  auto cls_of = [&](int i) -> int {
    const float l = labels[i];
    return l >= 1.f ? 0 : (l == 0.f ? 1 : 2);
  };
  if (t < 2) { s_n[t] = 0; s_len[t] = 0; s_tie_len[t] = 0; s_prefix[t] = 0u; }
  __syncthreads();
  int mine[2] = {0, 0};
  for (int i = t; i < A; i += kSampleThreads) {
    const int c = cls_of(i);
    if (c < 2) ++mine[c];
  }
  if (mine[0]) atomicAdd(&s_n[0], mine[0]);
  if (mine[1]) atomicAdd(&s_n[1], mine[1]);
  __syncthreads();
  const int n_pos = s_n[0], n_neg = s_n[1];
  const int num_pos = n_pos < max_pos ? n_pos : max_pos;
  const int num_neg = n_neg < cap - num_pos ? n_neg : cap - num_pos;
  if (t == 0) {
    s_k[0] = num_pos; s_k[1] = num_neg;
    s_all[0] = num_pos == n_pos; s_all[1] = num_neg == n_neg;
  }
  __syncthreads();
  // radix select, most significant byte first: after pass p the prefix holds the top 8(p+1) bits of the threshold key
  // and s_k the number of keys still to take among those sharing the prefix
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int b = t; b < 512; b += kSampleThreads) (&s_hist[0][0])[b] = 0;
    __syncthreads();
    const bool need0 = !s_all[0], need1 = !s_all[1];
    if (need0 || need1) {
      for (int i = t; i < A; i += kSampleThreads) {
        const int c = cls_of(i);
        if (c == 2 || (c == 0 ? !need0 : !need1)) continue;
        const unsigned key = sample_key(seed, (unsigned)i);
        const unsigned hi = pass == 0 ? 0u : key >> (shift + 8);
        if (hi == (pass == 0 ? 0u : s_prefix[c] >> (shift + 8))) atomicAdd(&s_hist[c][(key >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    if (t < 2 && !s_all[t]) {
      int k = s_k[t], b = 0;
      while (b < 255 && s_hist[t][b] < k) {   // keys in lower buckets are all taken
        k -= s_hist[t][b];
        ++b;
      }
      s_k[t] = k;                             // 1 <= k <= hist[b] keys of bucket b remain to be taken
      s_prefix[t] |= (unsigned)b << shift;
    }
    __syncthreads();
  }
  // selection: key < threshold -> taken; key == threshold -> tie list (s_k of them are taken, lowest indices first)
  for (int i = t; i < A; i += kSampleThreads) {
    const int c = cls_of(i);
    if (c == 2) continue;
    bool take = s_all[c] != 0, tie = false;
    if (!take) {
      const unsigned key = sample_key(seed, (unsigned)i);
      take = key < s_prefix[c];
      tie = key == s_prefix[c];
    }
    if (take) {
      const int slot = atomicAdd(&s_len[c], 1);
      if (slot < kAnchorCapMax) s_list[c][slot] = i;
    } else if (tie) {
      const int slot = atomicAdd(&s_tie_len[c], 1);
      if (slot < kTieCap) s_tie[c][slot] = i;
    }
  }
  __syncthreads();
  for (int c = 0; c < 2; ++c) {
    if (s_all[c]) continue;
    // ties (practically always exactly one key): lowest indices first
    const int nt = s_tie_len[c] < kTieCap ? s_tie_len[c] : kTieCap;
    int p2 = 1;
    while (p2 < nt) p2 <<= 1;
    for (int i = nt + t; i < p2; i += kSampleThreads) s_tie[c][i] = 0x7FFFFFFF;
    __syncthreads();
    if (p2 > 1) lds_bitonic_sort(s_tie[c], p2, t, kSampleThreads);
    const int want = c == 0 ? num_pos : num_neg;
    const int have = s_len[c];
    for (int i = t; i < want - have && i < nt; i += kSampleThreads) s_list[c][have + i] = s_tie[c][i];
    __syncthreads();
  }
  // ascending anchor order
  for (int c = 0; c < 2; ++c) {
    const int n = c == 0 ? num_pos : num_neg;
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (int i = n + t; i < p2; i += kSampleThreads) s_list[c][i] = 0x7FFFFFFF;
    __syncthreads();
    if (p2 > 1) lds_bitonic_sort(s_list[c], p2, t, kSampleThreads);
  }
  __syncthreads();
  for (int i = t; i < cap; i += kSampleThreads) {
    const bool vp = i < num_pos, vn = i < num_neg;
    pos_out[i] = vp ? index_offset + s_list[0][i] : -1;
    neg_out[i] = vn ? index_offset + s_list[1][i] : -1;
    reg_pos_out[i] = vp ? reg[s_list[0][i]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (t == 0) {
    counts[0] = num_pos;
    counts[1] = num_neg;
  }
}

__global__ __launch_bounds__(kSampleThreads) void sample_anchors_kernel(
    const float* __restrict__ labels, const float4* __restrict__ reg, int A, int cap, int max_pos, uint64_t seed,
    int64_t index_offset, int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out, float4* __restrict__ reg_pos_out,
    int* __restrict__ counts) {
  sample_anchors_body(labels, reg, A, cap, max_pos, seed, index_offset, pos_out, neg_out, reg_pos_out, counts);
}

// ---- round 4: the same sample from a chip-wide scan.  The one-workgroup form above walks all A labels six times from one
// CU: 0.51 ms for the 122 880 anchors of a C4 map, 1.88 ms for the 523 776 of a five-level pyramid (rocprofv3, round 3 —
// the 7th-largest kernel of the headline step).  Here `sample_anchors_scan_kernel` (one workgroup per 4096 anchors) reads
// the labels ONCE: it counts the two classes and appends every positive and every negative whose random key is at most
// `thr` to two candidate lists in device memory; thr = 8 cap / A of the key range, i.e. ~8 cap x (negatives / A) expected
// candidates for at most cap wanted ones.  `sample_anchors_finish_kernel` (one workgroup) then sorts the candidates by
// (key, index) in LDS — exactly the order "smaller key first, equal keys by ascending index" of the selection rule — takes
// the first num_pos / num_neg and restores ascending anchor order.  Whenever the lists do not provably contain the answer
// (more than kCandCap positives, fewer listed negatives than wanted, a list overflow) it runs the one-workgroup algorithm
// instead, so the result is ALWAYS the one of sample_anchors_kernel, bit for bit (tests/test_ops_gpu.py).
constexpr int kCandCap = 4096;
struct AnchorScratch {
  int n[2];            // class totals
  int len[2];          // appended candidates (may exceed kCandCap: overflow)
  int pad[4];
  unsigned long long cand[2][kCandCap];   // (key << 32) | anchor index
};

__global__ __launch_bounds__(256) void sample_anchors_scan_kernel(const float* __restrict__ labels, int A, uint64_t seed,
                                                                  unsigned thr, AnchorScratch* __restrict__ sc) {
  __shared__ int s_n[2];
  const int t = threadIdx.x;
  if (t < 2) s_n[t] = 0;
  __syncthreads();
  int mine[2] = {0, 0};
  for (int i = (int)blockIdx.x * 256 + t; i < A; i += (int)gridDim.x * 256) {
    const float l = labels[i];
    const int c = l >= 1.f ? 0 : (l == 0.f ? 1 : 2);
    if (c == 2) continue;
    ++mine[c];
    const unsigned key = sample_key(seed, (unsigned)i);
    if (c == 0 || key <= thr) {
      const int slot = atomicAdd(&sc->len[c], 1);
      if (slot < kCandCap) sc->cand[c][slot] = ((unsigned long long)key << 32) | (unsigned)i;
    }
  }
  if (mine[0]) atomicAdd(&s_n[0], mine[0]);
  if (mine[1]) atomicAdd(&s_n[1], mine[1]);
  __syncthreads();
  if (t < 2 && s_n[t]) atomicAdd(&sc->n[t], s_n[t]);
}

__device__ inline void lds_bitonic_sort64(unsigned long long* v, int n_pow2, int t, int nthreads) {
  for (int k = 2; k <= n_pow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < n_pow2; i += nthreads) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = v[i], b = v[p];
          if ((a > b) == ((i & k) == 0)) {
            v[i] = b;
            v[p] = a;
          }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(kSampleThreads) void sample_anchors_finish_kernel(
    const float* __restrict__ labels, const float4* __restrict__ reg, int A, int cap, int max_pos, uint64_t seed,
    int64_t index_offset, int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out, float4* __restrict__ reg_pos_out,
    int* __restrict__ counts, AnchorScratch* __restrict__ sc) {
  __shared__ unsigned long long s_keys[kCandCap];
  __shared__ int s_sel[2][kAnchorCapMax];
  __shared__ int s_hdr[4];
  const int t = threadIdx.x;
  if (t < 4) s_hdr[t] = t < 2 ? sc->n[t] : sc->len[t - 2];
  __syncthreads();
  const int n_pos = s_hdr[0], n_neg = s_hdr[1], len_pos = s_hdr[2], len_neg = s_hdr[3];
  __syncthreads();
  if (t < 4) (t < 2 ? sc->n[t] : sc->len[t - 2]) = 0;       // ready for the next launch on this stream
  const int num_pos = n_pos < max_pos ? n_pos : max_pos;
  const int num_neg = n_neg < cap - num_pos ? n_neg : cap - num_pos;
  // the lists hold the answer iff every positive is listed and at least num_neg negatives are (all unlisted negatives have
  // larger keys than every listed one)
  const bool ok = len_pos == n_pos && n_pos <= kCandCap && len_neg <= kCandCap && len_neg >= num_neg;
  if (!ok) {
    sample_anchors_body(labels, reg, A, cap, max_pos, seed, index_offset, pos_out, neg_out, reg_pos_out, counts);
    return;
  }
  for (int c = 0; c < 2; ++c) {
    const int len = c == 0 ? len_pos : len_neg, want = c == 0 ? num_pos : num_neg;
    int p2 = 1;
    while (p2 < len) p2 <<= 1;
    for (int i = t; i < p2; i += kSampleThreads) s_keys[i] = i < len ? sc->cand[c][i] : ~0ull;
    __syncthreads();
    if (p2 > 1) lds_bitonic_sort64(s_keys, p2, t, kSampleThreads);
    // the `want` smallest (key, index) pairs, then back into ascending anchor order
    int q2 = 1;
    while (q2 < want) q2 <<= 1;
    for (int i = t; i < q2; i += kSampleThreads) s_sel[c][i] = i < want ? (int)(unsigned)(s_keys[i] & 0xFFFFFFFFull) : 0x7FFFFFFF;
    __syncthreads();
    if (q2 > 1) lds_bitonic_sort(s_sel[c], q2, t, kSampleThreads);
    __syncthreads();
  }
  for (int i = t; i < cap; i += kSampleThreads) {
    const bool vp = i < num_pos, vn = i < num_neg;
    pos_out[i] = vp ? index_offset + s_sel[0][i] : -1;
    neg_out[i] = vn ? index_offset + s_sel[1][i] : -1;
    reg_pos_out[i] = vp ? reg[s_sel[0][i]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (t == 0) {
    counts[0] = num_pos;
    counts[1] = num_neg;
  }
}

}  // namespace dadet

// zero-initialised scratch of the chip-wide anchor sampler, one per stream (launches on a stream are ordered; the finishing
// kernel leaves the counters at zero)
static dadet::AnchorScratch* anchor_scratch(hipStream_t st) {
  static std::mutex m;
  static std::unordered_map<hipStream_t, dadet::AnchorScratch*> table;
  std::lock_guard<std::mutex> lock(m);
  dadet::AnchorScratch*& p = table[st];
  if (!p) {
    if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(dadet::AnchorScratch)) != hipSuccess) {
      p = nullptr;
      return nullptr;
    }
    if (hipMemset(p, 0, sizeof(dadet::AnchorScratch)) != hipSuccess) return nullptr;
  }
  return p;
}

extern "C" int dadet_sample_anchors(const float* labels, const float* regression_targets, int A, int cap, int max_pos,
                                    uint64_t seed, int64_t index_offset, int64_t* pos_inds_out, int64_t* neg_inds_out,
                                    float* regression_targets_pos_out, int* counts_out, void* stream) {
  DADET_REQUIRE(A >= 0 && cap > 0 && cap <= kAnchorCapMax && max_pos >= 0 && max_pos <= cap,
                "sample_anchors: A=%d cap=%d (<= %d) max_pos=%d", A, cap, kAnchorCapMax, max_pos);
  DADET_REQUIRE((A == 0 || (labels && regression_targets)) && pos_inds_out && neg_inds_out &&
                    regression_targets_pos_out && counts_out,
                "sample_anchors: null pointer");
  DADET_REQUIRE(al16(regression_targets) && al16(regression_targets_pos_out),
                "sample_anchors: regression targets must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const char* env = getenv("DADET_ANCHOR_SCAN");          // 0: the one-workgroup kernel (A/B runs, the equality test)
  if (A >= 32768 && !(env && env[0] == '0')) {
    dadet::AnchorScratch* sc = anchor_scratch(st);
    if (!sc) {
      dadet::set_error("sample_anchors: could not allocate the candidate lists");
      return DADET_ELAUNCH;
    }
    const double frac = 8.0 * cap / (double)A;
    const unsigned thr = frac >= 1.0 ? 0xFFFFFFFFu : (unsigned)(frac * 4294967296.0);
    hipLaunchKernelGGL(sample_anchors_scan_kernel, dim3(ceil_div(A, 4096)), dim3(256), 0, st, labels, A, (uint64_t)seed,
                       thr, sc);
    hipLaunchKernelGGL(sample_anchors_finish_kernel, dim3(1), dim3(kSampleThreads), 0, st, labels,
                       reinterpret_cast<const float4*>(regression_targets), A, cap, max_pos, (uint64_t)seed, index_offset,
                       pos_inds_out, neg_inds_out, reinterpret_cast<float4*>(regression_targets_pos_out), counts_out, sc);
    return check_launch("sample_anchors(scan)");
  }
  hipLaunchKernelGGL(sample_anchors_kernel, dim3(1), dim3(kSampleThreads), 0, st, labels,
                     reinterpret_cast<const float4*>(regression_targets), A, cap, max_pos, (uint64_t)seed,
                     index_offset, pos_inds_out, neg_inds_out, reinterpret_cast<float4*>(regression_targets_pos_out),
                     counts_out);
  return check_launch("sample_anchors");
}

extern "C" int dadet_fpn_merge_levels(const dadet_merge_entry* entries, int n, void* stream) {
  static_assert(sizeof(dadet_merge_entry) == sizeof(dadet::MergeEntry), "merge entry layout");
  DADET_REQUIRE(n >= 0 && n <= kMergeMax, "fpn_merge_levels: %d (level, image) pairs (at most %d per call)", n, kMergeMax);
  if (n == 0) return DADET_OK;
  DADET_REQUIRE(entries, "fpn_merge_levels: null table");
  MergeTable t;
  int max_cap = 0;
  for (int i = 0; i < n; ++i) {
    const dadet_merge_entry& s = entries[i];
    DADET_REQUIRE(s.boxes && s.scores && s.keep && s.count && s.boxes_out && s.scores_out && s.n > 0 && s.cap >= 0 &&
                      (reinterpret_cast<uintptr_t>(s.boxes) & 15) == 0 && (reinterpret_cast<uintptr_t>(s.boxes_out) & 15) == 0,
                  "fpn_merge_levels: entry %d is malformed", i);
    MergeEntry& e = t.e[i];
    e.boxes = reinterpret_cast<const float4*>(s.boxes);
    e.scores = s.scores;
    e.keep = s.keep;
    e.count = s.count;
    e.boxes_out = reinterpret_cast<float4*>(s.boxes_out);
    e.scores_out = s.scores_out;
    e.n = s.n;
    e.cap = s.cap;
    max_cap = s.cap > max_cap ? s.cap : max_cap;
  }
  if (max_cap == 0) return DADET_OK;
  hipLaunchKernelGGL(fpn_merge_levels_kernel, dim3(ceil_div(max_cap, 256), n), dim3(256), 0, as_stream(stream), t);
  return check_launch("fpn_merge_levels");
}
