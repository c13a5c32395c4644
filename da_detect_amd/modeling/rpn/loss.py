"""RPN losses (reference: maskrcnn_benchmark/modeling/rpn/loss.py:22-178).

Target-domain images carry no labels: they are skipped when targets are built, and — exactly as in the
reference — the sampled indices computed from the source images index the flattened predictions of the whole
batch, which is only meaningful because source images come first (asserted here)."""
import torch
from torch.nn import functional as F

from ... import _C
from ...layers import smooth_l1_loss
from ...layers.misc import rpn_loss_fused
from ...structures.bounding_box import is_source_image
from ...structures.boxlist_ops import boxlist_iou, cat_boxlist
from ...utils import rng
from ..balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from ..matcher import Matcher
from .utils import concat_box_prediction_layers



def _nz(mask, size):
    """nonzero with a host-known result size (no device->host round trip)"""
    if _STATIC:
        return torch.nonzero_static(mask, size=size)
    return torch.nonzero(mask)


_STATIC = True
# one-launch-per-image anchor sampling (dadet_sample_anchors); False keeps the ATen chain (the path of the reference's random stream, utils.rng)
_FUSED = True

class RPNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, generate_labels_func):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.box_coder = box_coder
        self.copied_fields = []
        self.generate_labels_func = generate_labels_func
        self.discard_cases = ["not_visibility", "between_thresholds"]

    def match_targets_to_anchors(self, anchor, target, copied_fields=[]):
        matched_idxs = self.proposal_matcher(boxlist_iou(target, anchor))
        target = target.copy_with_fields(copied_fields)
        matched = target[matched_idxs.clamp(min=0)]
        matched.add_field("matched_idxs", matched_idxs)
        return matched

    def prepare_targets(self, anchors, targets):
        labels, regression_targets, masks = [], [], []
        seen_target_domain = False
        for anchors_per_image, targets_per_image in zip(anchors, targets):
            is_source = is_source_image(targets_per_image)
            masks.append(is_source)
            if not is_source:
                seen_target_domain = True
                continue
            assert not seen_target_domain, "source-domain images must precede target-domain images in a batch"
            if (anchors_per_image.bbox.is_cuda and self.generate_labels_func is generate_rpn_labels
                    and self.proposal_matcher.allow_low_quality_matches and not self.copied_fields
                    and set(self.discard_cases) == {"not_visibility", "between_thresholds"}):
                # IoU + matcher (with low-quality matches) + label rules + encode: two launches instead of ~100
                if len(targets_per_image) == 0:
                    raise ValueError("No ground-truth boxes available for one of the images during training")
                lab, reg = _C.rpn_anchor_targets(anchors_per_image.bbox, anchors_per_image.get_field("visibility"),
                                                 targets_per_image.bbox, self.proposal_matcher.high_threshold,
                                                 self.proposal_matcher.low_threshold)
                labels.append(lab)
                regression_targets.append(reg)
                continue
            matched = self.match_targets_to_anchors(anchors_per_image, targets_per_image, self.copied_fields)
            matched_idxs = matched.get_field("matched_idxs")
            lab = self.generate_labels_func(matched).to(dtype=torch.float32)
            lab[matched_idxs == Matcher.BELOW_LOW_THRESHOLD] = 0
            if "not_visibility" in self.discard_cases:
                lab[~anchors_per_image.get_field("visibility")] = -1
            if "between_thresholds" in self.discard_cases:
                lab[matched_idxs == Matcher.BETWEEN_THRESHOLDS] = -1
            labels.append(lab)
            regression_targets.append(self.box_coder.encode(matched.bbox, anchors_per_image.bbox))
        return labels, regression_targets, masks

    def prepare(self, anchors, targets):
        """everything of the loss that depends on anchors and ground truth only (labels, regression targets, the
        sampled anchor indices — loss.py:101-123): none of it needs the network's output, so RPNModule issues it
        on a side stream while the backbone is still running; its host synchronisations (nonzero) then wait for
        that stream alone."""
        anchors = [cat_boxlist(a) for a in anchors]
        labels, regression_targets, _ = self.prepare_targets(anchors, targets)
        cap = self.fg_bg_sampler.batch_size_per_image
        if (_FUSED and not rng.cpu_stream_enabled() and labels and cap <= _C.SAMPLE_ANCHORS_MAX_CAP
                and all(l.is_cuda and l.dtype == torch.float32 for l in labels)):
            return self._prepare_fused(labels, regression_targets)
        pos_masks, neg_masks = self.fg_bg_sampler(labels)
        n_pos = sum(c[0] for c in self.fg_bg_sampler.last_counts)     # host-side counts: no round trips
        n_neg = sum(c[1] for c in self.fg_bg_sampler.last_counts)
        pos_inds = _nz(torch.cat(pos_masks, dim=0), size=n_pos).squeeze(1)
        neg_inds = _nz(torch.cat(neg_masks, dim=0), size=n_neg).squeeze(1)
        sampled_inds = torch.cat([pos_inds, neg_inds], dim=0)
        labels = torch.cat(labels, dim=0)
        regression_targets = torch.cat(regression_targets, dim=0)
        return dict(pos_inds=pos_inds, sampled_inds=sampled_inds, labels_sampled=labels[sampled_inds],
                    regression_targets_pos=regression_targets[pos_inds])

    def _prepare_fused(self, labels, regression_targets):
        """one dadet_sample_anchors launch per labelled image (device-side random keys, same counts and distribution
        as the sampler), then ONE host round trip for the counts.  Not used when the draws must come from the
        reference's random stream (utils.rng.use_cpu_stream, the loss-parity tests)."""
        sampler = self.fg_bg_sampler
        cap = sampler.batch_size_per_image
        max_pos = int(cap * sampler.positive_fraction)
        dev, n_img = labels[0].device, len(labels)
        counts = torch.empty((n_img, 2), dtype=torch.int32, device=dev)
        pos = torch.empty(n_img * cap, dtype=torch.int64, device=dev)
        neg = torch.empty(n_img * cap, dtype=torch.int64, device=dev)
        reg = torch.empty((n_img * cap, 4), dtype=torch.float32, device=dev)
        offset = 0
        for i, (lab, tgt) in enumerate(zip(labels, regression_targets)):
            rows = slice(i * cap, (i + 1) * cap)
            _C.sample_anchors(lab, tgt, cap, max_pos, rng.next_seed(dev), offset, counts[i],
                              out=dict(pos=pos[rows], neg=neg[rows], regression_targets_pos=reg[rows]))
            offset += lab.numel()
        host = counts.tolist()
        sampler.last_counts = [(h[0], h[1]) for h in host]
        if n_img == 1:
            (n_pos, n_neg), = host
            pos_inds, neg_inds, reg_pos = pos[:n_pos], neg[:n_neg], reg[:n_pos]
        else:
            pos_inds = torch.cat([pos[i * cap:i * cap + h[0]] for i, h in enumerate(host)])
            neg_inds = torch.cat([neg[i * cap:i * cap + h[1]] for i, h in enumerate(host)])
            reg_pos = torch.cat([reg[i * cap:i * cap + h[0]] for i, h in enumerate(host)])
            n_pos, n_neg = pos_inds.numel(), neg_inds.numel()
        sampled_inds = torch.cat([pos_inds, neg_inds], dim=0)
        labels_sampled = (torch.arange(n_pos + n_neg, device=dev) < n_pos).to(torch.float32)
        return dict(pos_inds=pos_inds, sampled_inds=sampled_inds, labels_sampled=labels_sampled,
                    regression_targets_pos=reg_pos)

    def finish(self, objectness, box_regression, prep):
        """the part that needs the predictions (loss.py:125-143); no host synchronisation"""
        if len(objectness) == 1 and objectness[0].is_cuda:
            # one level: the flattening of concat_box_prediction_layers IS the NHWC map's memory order, so the fused
            # kernel indexes the head's outputs directly (no permute / gather / elementwise chain, forward or backward)
            return rpn_loss_fused(objectness[0], box_regression[0], prep["sampled_inds"], prep["labels_sampled"],
                                  prep["pos_inds"], prep["regression_targets_pos"], 1.0 / 9)
        objectness, box_regression = concat_box_prediction_layers(objectness, box_regression)
        objectness = objectness.squeeze()
        pos_inds, sampled_inds = prep["pos_inds"], prep["sampled_inds"]
        box_loss = smooth_l1_loss(box_regression[pos_inds], prep["regression_targets_pos"], beta=1.0 / 9,
                                  size_average=False) / (sampled_inds.numel())
        objectness_loss = F.binary_cross_entropy_with_logits(objectness[sampled_inds], prep["labels_sampled"])
        return objectness_loss, box_loss

    def __call__(self, anchors, objectness, box_regression, targets):
        return self.finish(objectness, box_regression, self.prepare(anchors, targets))


def generate_rpn_labels(matched_targets):
    return matched_targets.get_field("matched_idxs") >= 0


def make_rpn_loss_evaluator(cfg, box_coder):
    matcher = Matcher(cfg.MODEL.RPN.FG_IOU_THRESHOLD, cfg.MODEL.RPN.BG_IOU_THRESHOLD,
                      allow_low_quality_matches=True)
    sampler = BalancedPositiveNegativeSampler(cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE, cfg.MODEL.RPN.POSITIVE_FRACTION)
    return RPNLossComputation(matcher, sampler, box_coder, generate_rpn_labels)
