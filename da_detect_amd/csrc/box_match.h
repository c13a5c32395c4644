// IoU -> Matcher -> label rules -> BoxCoder.encode for ONE proposal against G ground-truth boxes held in LDS; shared by
// box_match_encode_kernel (elementwise.hip) and proposals_sample_kernel (sampling.hip) so that both give the same bits.
// reference chain: boxlist_iou (structures/boxlist_ops.py:56-91), Matcher without low-quality matches
// (modeling/matcher.py:42-92), label rules of FastRCNNLossComputation.prepare_targets (roi_heads/box_head/loss.py:69-93:
// below-low -> 0, between thresholds -> -1) and BoxCoder.encode (modeling/box_coder.py:22-50).  Operation order is the
// reference's (contraction is off for these files), first maximum wins on equal IoU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dadet {

// returns the label (>= 1 matched class, 0 background, -1 between the thresholds); *matched_out (optional) the matcher's
// index (-1 below low, -2 between); *reg the encoded regression target against the matched (clamped) box
__device__ inline int64_t match_encode_one(const float4 p, const float4* g_box, const float* g_area,
                                           const int64_t* __restrict__ gt_labels, int G, float high, float low, float wx,
                                           float wy, float ww, float wh, float4* reg, int64_t* matched_out = nullptr) {
  const float area = (p.z - p.x + 1.f) * (p.w - p.y + 1.f);
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float4 b = g_box[g];
    const float w = fmaxf(fminf(b.z, p.z) - fmaxf(b.x, p.x) + 1.f, 0.f);
    const float h = fmaxf(fminf(b.w, p.w) - fmaxf(b.y, p.y) + 1.f, 0.f);
    const float inter = w * h;
    const float iou = inter / (g_area[g] + area - inter);
    if (iou > best) {
      best = iou;
      arg = g;
    }
  }
  int64_t m = arg;
  if (best < low) m = -1;
  else if (best < high) m = -2;
  if (matched_out) *matched_out = m;
  const int src = m < 0 ? 0 : (int)m;   // matched_idxs.clamp(min=0)
  const float4 r = g_box[src];
  const float ew = p.z - p.x + 1.f, eh = p.w - p.y + 1.f;
  const float ecx = p.x + 0.5f * ew, ecy = p.y + 0.5f * eh;
  const float gw = r.z - r.x + 1.f, gh = r.w - r.y + 1.f;
  const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
  *reg = make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
  return m == -1 ? 0 : (m == -2 ? -1 : gt_labels[src]);
}

}  // namespace dadet
