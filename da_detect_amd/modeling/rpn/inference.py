"""RPN proposal selection (reference: maskrcnn_benchmark/modeling/rpn/inference.py:13-181).

Same chain as the reference — sigmoid, per-image top-k, decode, clip, small-box filter, NMS, first
post_nms_top_n, (training) append ground truth of source images — with the gather + BoxCoder.decode +
clip_to_image done by one HIP kernel on the top-k anchors and NMS (rank, IoU bitmask, greedy sweep,
compaction) fully on the device.

Ranking rule: the reference uses `topk(sorted=True)`, whose order among EQUAL scores is unspecified (and
differs between its CPU and CUDA builds).  Here equal scores are ordered by ascending anchor index (a
stable descending sort), which makes proposal indices reproducible; with distinct scores the result is the
reference's.
"""
import contextlib

import torch

from ... import _C
from ...utils.streams import other_stream, record
from ...structures.bounding_box import BoxList, PendingProposals, is_source_image
from ...structures.boxlist_ops import cat_boxlist
from ..box_coder import BoxCoder
from .utils import permute_and_flatten


# the one-launch dadet_topk_sorted instead of torch.sort.  False: measured on the RPN's shape (2 x 122 880 scores,
# k = 12 000, tools/topk_bench.py) the single-workgroup-per-image kernel takes 0.45 ms against 0.20 ms for the library's
# segmented sort of ALL scores (36 launches, but they spread over the whole chip; one workgroup reads its 480 KB row four
# times from one CU and then sorts 16 384 pairs in LDS).  Indices are identical; tests/test_topk_gpu.py runs the selection
# chain both ways.  No environment switch: a caller that wants it sets the module attribute.
_TOPK_KERNEL = False
# single level, several images: one library sort per image (inside the per-image loop, on that image's stream) instead of
# one segmented sort of the batch; False: the batch sort (tools/probes/variant_ab.py)
_SORT_PER_IMAGE = True


# one batched ranking call for all pyramid levels (dadet_topk_sorted_rows); False: one library sort per level
_ROWS_TOPK = True
# (the same kernels for the single-level C4 ranking, k = 12 000 of 122 880, measured and removed: img_only 18.57 / 18.62 ->
# 19.01 / 18.97 ms per step, da unchanged — the finishing workgroup's 16 384-pair LDS sort costs more than the library's
# segmented sort; tools/probes/variant_ab.py)
# False: multi-level training selection with the reference's host round trips (the path taken on CPU tensors and in eval
# mode; tests/test_model_gpu.py compares the two)
_DEVICE_SELECT = True
# (measured in round 3 and removed: the device-side selection as one captured HIP graph — 64.5 - 65.5 ms per step against
# 59.4 - 60.8 launch by launch; replaying the ~250-node graph cost more than issuing its launches from Python)


class RPNPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder=None,
                 fpn_post_nms_top_n=None):
        super(RPNPostProcessor, self).__init__()
        self.pre_nms_top_n = pre_nms_top_n
        self.post_nms_top_n = post_nms_top_n
        self.nms_thresh = nms_thresh
        self.min_size = min_size
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.fpn_post_nms_top_n = post_nms_top_n if fpn_post_nms_top_n is None else fpn_post_nms_top_n

    # set by RPNModule for ONE call: the caller is the training path whose box head samples straight from the NMS result
    # on the device (FastRCNNLossComputation._subsample_fused): single-level proposals are then handed over as
    # PendingProposals — no kept-count round trip, no gathers, no concatenation here
    defer = False

    def add_gt_proposals(self, proposals, targets):
        """append the ground-truth boxes of SOURCE images with objectness 1 (inference.py:51-74)"""
        out = []
        for proposal, target in zip(proposals, targets):
            if getattr(type(proposal), "is_pending_proposals", False) and proposal.pending is not None:
                if is_source_image(target):
                    proposal.attach_ground_truth(target.copy_with_fields(["labels"], skip_missing=True))
                out.append(proposal)
                continue
            device = proposal.bbox.device
            if is_source_image(target):
                gt = target.copy_with_fields([])
                gt.add_field("objectness", torch.ones(len(gt), device=device))
                out.append(cat_boxlist((proposal, gt)))
            else:
                out.append(proposal)
        return out

    def _level_inputs(self, objectness, box_regression):
        """sigmoid scores [N, HWA], deltas [N, HWA, 4] (both in anchor order) and this level's pre-NMS count"""
        N, A, H, W = objectness.shape
        scores_all = permute_and_flatten(objectness, N, A, 1, H, W).reshape(N, -1).sigmoid()
        deltas_all = permute_and_flatten(box_regression, N, A, 4, H, W).contiguous()  # [N, HWA, 4]
        return scores_all, deltas_all, min(self.pre_nms_top_n, A * H * W)

    def forward_for_single_feature_map(self, anchors, objectness, box_regression, raw=False, prepared=None):
        """anchors: list[BoxList] (one per image); objectness [N,A,H,W]; box_regression [N,4A,H,W].
        raw: return the per-image (score-ordered boxes, scores, NMS keep buffer, kept count on the device, size) tuples
        without reading the counts back (the multi-level device-side selection, _select_over_all_levels_device).
        prepared: (deltas [N,HWA,4], [(sorted scores, indices) per image]) when the ranking was done for all levels at once
        (_device_selection)"""
        N = objectness.shape[0]
        if prepared is not None:
            deltas_all, ranked = prepared
            sorted_scores, topk_idx = [r[0] for r in ranked], [r[1] for r in ranked]
        else:
            scores_all, deltas_all, pre_nms_top_n = self._level_inputs(objectness, box_regression)
            if scores_all.is_cuda and pre_nms_top_n <= _C.TOPK_SORTED_MAX and _TOPK_KERNEL:
                # radix select + in-LDS sort of the selected scores, one launch for the batch (csrc/topk.hip)
                sorted_scores, topk_idx = _C.topk_sorted(scores_all, pre_nms_top_n)
            elif scores_all.is_cuda and N > 1 and _SORT_PER_IMAGE:
                # ranked image by image inside the loop below (on the stream that image's NMS runs on): the library's sort
                # of ONE row of 122 880 scores is 10 launches / ~70 us, its segmented sort of two rows 45 launches / ~290 us
                # (profiles/r06_step_timeline_da.txt) — and the rows are independent.  Same indices.
                sorted_scores = topk_idx = None
            else:
                sorted_scores, order = torch.sort(scores_all, dim=1, descending=True, stable=True)
                sorted_scores = sorted_scores[:, :pre_nms_top_n].contiguous()
                topk_idx = order[:, :pre_nms_top_n].contiguous()

        # The greedy sweep of one image is a single workgroup: images are independent, so every second image runs
        # on the side stream and the sweeps overlap; the kept counts come back in ONE host round trip.
        dev = objectness.device
        use_side = dev.type == "cuda" and N > 1 and self.nms_thresh > 0 and self.min_size <= 0
        if use_side:
            main, side = torch.cuda.current_stream(dev), other_stream(dev)
            side.wait_stream(main)      # sorted scores / deltas exist
        pending = []
        for i in range(N):
            im_w, im_h = anchors[i].size
            ctx = torch.cuda.stream(side) if (use_side and i % 2 == 1) else contextlib.nullcontext()
            with ctx:
                if topk_idx is None:
                    row_scores, row_order = torch.sort(scores_all[i], descending=True, stable=True)
                    scores, idx_i = row_scores[:pre_nms_top_n].contiguous(), row_order[:pre_nms_top_n].contiguous()
                else:
                    scores, idx_i = sorted_scores[i], topk_idx[i]
                boxes = _C.rpn_decode_clip(deltas_all[i], anchors[i].bbox.contiguous(), idx_i,
                                           self.box_coder.weights, self.box_coder.bbox_xform_clip, im_w, im_h)
                if self.min_size > 0:  # with min_size == 0 every clipped box passes (w, h >= 1)
                    keep = ((boxes[:, 2] - boxes[:, 0] + 1 >= self.min_size) &
                            (boxes[:, 3] - boxes[:, 1] + 1 >= self.min_size)).nonzero().squeeze(1)
                    boxes, scores = boxes[keep].contiguous(), scores[keep].contiguous()
                keep = count = None
                if self.nms_thresh > 0:
                    # boxes arrive in the order of the stable descending score sort above: NMS needs no ranking pass
                    # (a min_size filter keeps the relative order too)
                    keep, count = _C.nms_with_count(boxes, None, self.nms_thresh, max_keep=self.post_nms_top_n)
            pending.append((boxes, scores, keep, count, (im_w, im_h)))
        if use_side:
            if not torch.cuda.is_current_stream_capturing():     # (a captured graph owns its memory: nothing to register)
                record([p[:4] for p in pending[1::2]], main)
            main.wait_stream(side)
        if raw:
            return pending
        if self.defer and self.training and self.nms_thresh > 0 and self.min_size <= 0 and dev.type == "cuda":
            return [PendingProposals(boxes, scores, keep, count, self.post_nms_top_n, size)
                    for boxes, scores, keep, count, size in pending]
        counts = None
        if self.nms_thresh > 0:
            counts = torch.cat([p[3] for p in pending]).tolist()
        result = []
        for i, (boxes, scores, keep, _, size) in enumerate(pending):
            if keep is not None:
                keep = keep[: counts[i]]
                boxes, scores = boxes[keep], scores[keep]
            boxlist = BoxList(boxes, size, mode="xyxy")
            boxlist.add_field("objectness", scores)
            result.append(boxlist)
        return result

    def forward(self, anchors, objectness, box_regression, targets=None):
        sampled = []
        num_levels = len(objectness)
        if (self.defer and _DEVICE_SELECT and num_levels > 1 and targets is not None and self.training
                and self.nms_thresh > 0 and self.min_size <= 0 and objectness[0].is_cuda):
            self.defer = False
            parts = self._device_selection(anchors, objectness, box_regression)
            out = [PendingProposals(b, sc, keep, cnt, post_n, anchors[i][0].size)
                   for i, (b, sc, keep, cnt, post_n) in enumerate(parts)]
            return self.add_gt_proposals(out, targets)
        defer, self.defer = self.defer and num_levels == 1 and targets is not None, False
        for a, o, b in zip(list(zip(*anchors)), objectness, box_regression):
            self.defer = defer
            sampled.append(self.forward_for_single_feature_map(a, o, b))
            self.defer = False
        if defer and all(getattr(type(p), "is_pending_proposals", False) for p in sampled[0]):
            boxlists = list(sampled[0])
        else:
            boxlists = [cat_boxlist(per_image) for per_image in zip(*sampled)]
        if num_levels > 1:
            boxlists = self.select_over_all_levels(boxlists)
        if self.training and targets is not None:
            boxlists = self.add_gt_proposals(boxlists, targets)
        return boxlists

    def _device_selection(self, anchors, objectness, box_regression):
        """the whole multi-level selection as a function of tensors at fixed shapes with no host round trip: -> per image
        (boxes, scores, keep, count, upper bound)."""
        levels = list(zip(list(zip(*anchors)), objectness, box_regression))
        n_img = objectness[0].shape[0]
        prepared = [None] * len(levels)
        if _ROWS_TOPK and len(levels) * n_img <= _C.TOPK_ROWS_MAX and self.pre_nms_top_n <= _C.TOPK_SORTED_MAX:
            # the ranking of every (level, image) row in ONE call (five launches, csrc/topk.hip `dadet_topk_sorted_rows`)
            # instead of one segmented library sort per level (17 launches each + slicing): same indices
            inputs = [self._level_inputs(o, b) for _, o, b in levels]
            rows = [sc[i] for sc, _, _ in inputs for i in range(n_img)]
            ks = [k for _, _, k in inputs for _ in range(n_img)]
            ranked = _C.topk_sorted_rows(rows, ks)
            prepared = [(inputs[l][1], ranked[l * n_img:(l + 1) * n_img]) for l in range(len(levels))]
        per_level = [self.forward_for_single_feature_map(a, o, b, raw=True, prepared=p)
                     for (a, o, b), p in zip(levels, prepared)]
        return self._select_over_all_levels_device(per_level)

    def _select_over_all_levels_device(self, per_level):
        """Training-mode multi-level selection (inference.py:102-121 per level, :154-167 over the batch) with every count
        left on the device: per image the levels' NMS results are laid end to end in fixed-capacity buffers (entries behind
        a level's kept count carry score -1), the batch-wide top fpn_post_nms_top_n is a mask over those buffers, and each
        image leaves as PendingProposals (positions of its selected entries + their number in device memory) for the box
        head's sampler.  The host path did the same with one count round trip per level, boolean-mask indexing per image
        (two more) and ~340 launches; nothing overlaps that chain since the RPN head's backward runs over rows.
        per_level[l][i] = (boxes [n_l, 4] in score order, scores [n_l], keep int64 [n_l], count int32 [1], size)."""
        n_img = len(per_level[0])
        merged = _C.fpn_merge_levels([[lvl[i][:4] for lvl in per_level] for i in range(n_img)], self.post_nms_top_n)
        img_boxes, img_scores = [m[0] for m in merged], [m[1] for m in merged]
        sizes = [int(s.numel()) for s in img_scores]
        all_scores = torch.cat(img_scores, dim=0)
        k = min(self.fpn_post_nms_top_n, int(all_scores.numel()))
        _, inds = torch.topk(all_scores, k, dim=0, sorted=True)
        mask = torch.zeros_like(all_scores, dtype=torch.bool)
        mask.scatter_(0, inds, True)     # (not mask[inds] = True: that copies a host scalar, which a graph capture forbids)
        mask &= all_scores >= 0          # fewer valid entries than k: the rest of the top-k are padding
        out = []
        for i, m in enumerate(mask.split(sizes)):
            keep_i = torch.nonzero_static(m, size=sizes[i], fill_value=0).reshape(-1)
            count_i = m.sum().to(torch.int32).reshape(1)
            out.append((img_boxes[i], img_scores[i], keep_i, count_i, min(k, sizes[i])))
        return out

    def select_over_all_levels(self, boxlists):
        """FPN: keep fpn_post_nms_top_n over the whole batch (train) / per image (test) (inference.py:154-181)"""
        if self.training:
            objectness = torch.cat([b.get_field("objectness") for b in boxlists], dim=0)
            sizes = [len(b) for b in boxlists]
            k = min(self.fpn_post_nms_top_n, len(objectness))
            _, inds = torch.topk(objectness, k, dim=0, sorted=True)
            mask = torch.zeros_like(objectness, dtype=torch.bool)
            mask[inds] = 1
            for i, m in enumerate(mask.split(sizes)):
                boxlists[i] = boxlists[i][m]
        else:
            for i in range(len(boxlists)):
                objectness = boxlists[i].get_field("objectness")
                k = min(self.fpn_post_nms_top_n, len(objectness))
                _, inds = torch.topk(objectness, k, dim=0, sorted=True)
                boxlists[i] = boxlists[i][inds]
        return boxlists


def make_rpn_postprocessor(config, rpn_box_coder, is_train):
    rpn = config.MODEL.RPN
    return RPNPostProcessor(
        pre_nms_top_n=rpn.PRE_NMS_TOP_N_TRAIN if is_train else rpn.PRE_NMS_TOP_N_TEST,
        post_nms_top_n=rpn.POST_NMS_TOP_N_TRAIN if is_train else rpn.POST_NMS_TOP_N_TEST,
        nms_thresh=rpn.NMS_THRESH, min_size=rpn.MIN_SIZE, box_coder=rpn_box_coder,
        fpn_post_nms_top_n=rpn.FPN_POST_NMS_TOP_N_TRAIN if is_train else rpn.FPN_POST_NMS_TOP_N_TEST)
