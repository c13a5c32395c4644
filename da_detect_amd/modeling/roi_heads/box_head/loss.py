"""Fast R-CNN losses with the domain-adaptation sampling rules
(reference: maskrcnn_benchmark/modeling/roi_heads/box_head/loss.py:16-251)."""
import os

import torch
from torch.nn import functional as F

from .... import _C
from ....layers import smooth_l1_loss
from ....layers.misc import fast_rcnn_loss_fused, fast_rcnn_loss_rows_fused
from ....structures.bounding_box import BoxList, is_source_image
from ....structures.boxlist_ops import boxlist_iou
from ...balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from ...box_coder import BoxCoder
from ...matcher import Matcher
from ...utils import cat
from ....utils import rng



def _nz(mask, size):
    """nonzero with a host-known result size (no device->host round trip)"""
    if _STATIC:
        return torch.nonzero_static(mask, size=size)
    return torch.nonzero(mask)


_STATIC = True
# one-launch-per-image sampling (dadet_sample_rois); False keeps the ATen chain (the path of the reference's random stream, utils.rng)
_FUSED = True
# NMS -> sampler hand-over on the device (dadet_proposals_sample); False: the kept count goes through the host
_PENDING = True


def _is_pending(p):
    return getattr(type(p), "is_pending_proposals", False) and p.pending is not None

class FastRCNNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cls_agnostic_bbox_reg=False):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.box_coder = box_coder
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg

    def match_targets_to_proposals(self, proposal, target, is_source=True):
        matched_idxs = self.proposal_matcher(boxlist_iou(target, proposal))
        target = target.copy_with_fields("labels")
        # source: negatives (-1/-2) are clamped to gt 0; target domain: raw (negative = from the end) indexing,
        # its labels are overwritten with 0 below anyway (loss.py:55-66)
        matched = target[matched_idxs.clamp(min=0)] if is_source else target[matched_idxs]
        matched.add_field("matched_idxs", matched_idxs)
        return matched

    def prepare_targets(self, proposals, targets, sample_for_da=False):
        labels, regression_targets, domain_labels = [], [], []
        self._all_negative = []
        for proposals_per_image, targets_per_image in zip(proposals, targets):
            is_source = is_source_image(targets_per_image)
            self._all_negative.append(bool(not is_source or sample_for_da))
            if not is_source or sample_for_da:
                # every label is overwritten with 0 below (loss.py:85-88) and the regression targets of these rows
                # never reach a loss (target-domain rows are masked out, loss.py:193-198; subsample_for_da drops
                # them, loss.py:143): the IoU / matcher / encode work of the reference is dead here and skipped.
                # The sampler still sees an all-zero label vector, i.e. draws the same permutations.
                n, dev = len(proposals_per_image), proposals_per_image.bbox.device
                labels.append(torch.zeros(n, dtype=torch.int64, device=dev))
                regression_targets.append(torch.zeros((n, 4), dtype=torch.float32, device=dev))
                domain_labels.append(torch.full((n,), bool(is_source), dtype=torch.bool, device=dev))
                continue
            if proposals_per_image.bbox.is_cuda and not self.proposal_matcher.allow_low_quality_matches:
                # IoU + matcher + label rules + encode in ONE launch (the ATen chain below is ~60 launches)
                if len(targets_per_image) == 0:
                    raise ValueError("No ground-truth boxes available for one of the images during training")
                _, lab, reg = _C.box_match_encode(
                    proposals_per_image.bbox, targets_per_image.bbox, targets_per_image.get_field("labels"),
                    self.proposal_matcher.high_threshold, self.proposal_matcher.low_threshold, self.box_coder.weights)
                regression_targets.append(reg)
                domain_labels.append(torch.ones_like(lab, dtype=torch.bool))
                labels.append(lab)
                continue
            matched = self.match_targets_to_proposals(proposals_per_image, targets_per_image, is_source)
            matched_idxs = matched.get_field("matched_idxs")
            lab = matched.get_field("labels").to(dtype=torch.int64)
            lab[matched_idxs == Matcher.BELOW_LOW_THRESHOLD] = 0
            lab[matched_idxs == Matcher.BETWEEN_THRESHOLDS] = -1  # ignored by the sampler
            regression_targets.append(self.box_coder.encode(matched.bbox, proposals_per_image.bbox))
            dom = torch.ones_like(lab, dtype=torch.bool) if is_source else torch.zeros_like(lab, dtype=torch.bool)
            domain_labels.append(dom)
            if not is_source or sample_for_da:
                lab[:] = 0  # everything is a "negative": uniform random sampling (loss.py:85-88)
            labels.append(lab)
        return labels, regression_targets, domain_labels

    def _take_sampled(self, proposals, pos_masks, neg_masks):
        limit = self.fg_bg_sampler.batch_size_per_image
        for i, (pm, nm) in enumerate(zip(pos_masks, neg_masks)):
            if self._all_negative[i] and len(proposals[i]) <= limit:
                continue      # every row is taken, in ascending order: the gather would be the identity
            k = sum(self.fg_bg_sampler.last_counts[i])        # known on the host: no round trip for the index list
            proposals[i] = proposals[i][_nz(pm | nm, size=k).squeeze(1)]
        return proposals

    def accepts_pending(self, targets=None):
        """the sampler can take the RPN's NMS result where it lies on the device (PendingProposals): true whenever the
        one-launch device sampler is the one in use.  dadet_proposals_sample keeps an image's ground truth in an LDS
        table of _C.PROPOSALS_SAMPLE_MAX_GT boxes: a batch with a more crowded image gets ordinary BoxLists from the
        RPN instead (and then the two-launch box_match_encode + sample_rois path, which has no such bound)"""
        if targets is not None and any(len(t) > _C.PROPOSALS_SAMPLE_MAX_GT for t in targets):
            return False
        return (_FUSED and _PENDING and not rng.cpu_stream_enabled()
                and not self.proposal_matcher.allow_low_quality_matches)

    def _fused_ok(self, proposals):
        """the one-launch sampler draws its own random keys on the device: it is the default on the GPU, and it is
        switched off when the draws must come from the reference's random stream (utils.rng.use_cpu_stream, the
        parity tests) — the distribution of the sample is the same, the stream is not"""
        if not (_FUSED and not rng.cpu_stream_enabled() and not self.proposal_matcher.allow_low_quality_matches):
            return False
        for p in proposals:
            if _is_pending(p):        # device-resident list: CUDA fp32 by construction (do not touch .bbox: it would sync)
                if p.upper_bound() > _C.SAMPLE_ROIS_MAX:
                    return False
            elif not (p.bbox.is_cuda and p.bbox.dtype == torch.float32 and len(p) <= _C.SAMPLE_ROIS_MAX):
                return False
        return True

    def _subsample_fused(self, proposals, targets):
        """per image: box_match_encode (source domain only) + sample_rois, then ONE host round trip for the counts"""
        return self.subsample_finish(self.subsample_launch(proposals, targets))

    def subsample_launch(self, proposals, targets):
        """first half of subsample() on the one-launch device sampler: every kernel is queued on the current stream, the
        counts (rows taken, positives, per image) start their way to a pinned host buffer, and NOTHING waits for them.
        -> the state subsample_finish() completes, or None when this sampler does not serve the call (subsample() then
        does everything).  state["speculative"]: the sample as it looks when every image fills its BATCH_SIZE_PER_IMAGE
        rows — the usual case; rows behind an image's count are defined (zero boxes, csrc/sampling.hip) — for work that
        may be queued before the counts are known (ROIBoxHead.forward issues the pooler and the res5 head on it)."""
        if not self._fused_ok(proposals):
            return None
        sampler = self.fg_bg_sampler
        cap = sampler.batch_size_per_image
        max_pos = int(cap * sampler.positive_fraction)
        dev = targets[0].bbox.device
        n_img = len(proposals)
        counts = torch.empty((n_img, 2), dtype=torch.int32, device=dev)
        buf = _C.sample_rois_buffers(n_img * cap, dev)
        self._all_negative = []
        deferred = [False] * n_img
        for i, (prop, tgt) in enumerate(zip(proposals, targets)):
            is_source = is_source_image(tgt)
            self._all_negative.append(not is_source)
            lab = reg = None
            if is_source and len(tgt) == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            out_i = {k: v[i * cap:(i + 1) * cap] for k, v in buf.items()}
            if _is_pending(prop):
                # the RPN's NMS result is read where it lies: kept boxes + appended ground truth, target assignment and
                # the sample in ONE launch, no kept-count round trip in between (csrc/sampling.hip)
                deferred[i] = True
                _C.proposals_sample(prop.pending, tgt.bbox if is_source else None,
                                    tgt.get_field("labels") if is_source else None, self.proposal_matcher.high_threshold,
                                    self.proposal_matcher.low_threshold, self.box_coder.weights, cap, max_pos,
                                    rng.next_seed(dev), is_source, counts[i], out=out_i)
                continue
            if is_source:
                _, lab, reg = _C.box_match_encode(
                    prop.bbox, tgt.bbox, tgt.get_field("labels"), self.proposal_matcher.high_threshold,
                    self.proposal_matcher.low_threshold, self.box_coder.weights)
            _C.sample_rois(prop.bbox, lab, reg, cap, max_pos, rng.next_seed(dev), is_source, counts[i], out=out_i)
        state = dict(proposals=proposals, counts=counts, buf=buf, deferred=deferred, cap=cap, n_img=n_img)
        if dev.type == "cuda":
            # the counts' copy is issued HERE, behind the sampler on its stream, into pinned memory: the host later waits for
            # this copy alone (an event), not for whatever the caller has queued on the compute stream in the meantime
            pin = self._counts_pin.get(n_img) if hasattr(self, "_counts_pin") else None
            if pin is None:
                if not hasattr(self, "_counts_pin"):
                    self._counts_pin = {}
                pin = self._counts_pin[n_img] = torch.empty((n_img, 2), dtype=torch.int32).pin_memory()
            pin.copy_(counts, non_blocking=True)
            state["pin"], state["copied"] = pin, torch.cuda.current_stream(dev).record_event()
        state["speculative"] = [BoxList(buf["boxes"][i * cap:(i + 1) * cap], prop.size, prop.mode)
                                for i, prop in enumerate(proposals)] if all(deferred) else None
        return state

    def subsample_finish(self, state):
        """second half: the host reads the counts (its only wait) and builds the sampled lists.  state["exact"] tells the
        caller whether state["speculative"] WAS the sample (every image filled its rows)."""
        proposals, buf, deferred, cap, n_img = (state[k] for k in ("proposals", "buf", "deferred", "cap", "n_img"))
        sampler = self.fg_bg_sampler
        if "pin" in state:
            state["copied"].synchronize()
            host = state["pin"].tolist()
        else:
            host = state["counts"].tolist()
        state["exact"] = all(h[0] == cap for h in host)
        sampled = []
        for i, prop in enumerate(proposals):
            k = host[i][0]
            rows = slice(i * cap, i * cap + k)
            b = BoxList(buf["boxes"][rows], prop.size, prop.mode)
            if deferred[i]:
                b.add_field("objectness", buf["objectness"][rows])
            for name in ([] if deferred[i] else prop.fields()):
                if name not in ("labels", "regression_targets", "domain_labels"):
                    b.add_field(name, prop.get_field(name)[buf["idx"][rows]])
            b.add_field("labels", buf["labels"][rows])
            b.add_field("regression_targets", buf["regression_targets"][rows])
            b.add_field("domain_labels", buf["domain"][rows])
            sampled.append(b)
        self._proposals = sampled
        self._sampled_pos = [h[1] for h in host]
        self._is_source = [not neg for neg in self._all_negative]
        sampler.last_counts = [(h[1], h[0] - h[1]) for h in host]
        if state["exact"]:                     # the usual case: every image filled its slice, no copies
            rows_of = lambda name: buf[name]   # noqa: E731
        else:
            rows_of = lambda name: cat([buf[name][i * cap:i * cap + host[i][0]] for i in range(n_img)], dim=0)  # noqa: E731
        self._loss_prep = dict(rows=True, loss_labels=rows_of("loss_labels"),
                               regression_targets=rows_of("regression_targets"), domain_masks=rows_of("domain"))
        return self._proposals

    def subsample(self, proposals, targets):
        """sample BATCH_SIZE_PER_IMAGE proposals per image for the detection loss; keeps them in
        self._proposals for the following __call__ (loss.py:95-130)"""
        if self._fused_ok(proposals):
            return self._subsample_fused(proposals, targets)
        labels, regression_targets, domain_labels = self.prepare_targets(proposals, targets)
        pos_masks, neg_masks = self.fg_bg_sampler(labels, self._all_negative)
        proposals = list(proposals)
        for lab, reg, prop, dom in zip(labels, regression_targets, proposals, domain_labels):
            prop.add_field("labels", lab)
            prop.add_field("regression_targets", reg)
            prop.add_field("domain_labels", dom)
        self._proposals = self._take_sampled(proposals, pos_masks, neg_masks)
        self._sampled_pos = [c[0] for c in self.fg_bg_sampler.last_counts]
        self._is_source = [not neg for neg in self._all_negative]
        self._prepare_loss_indices()
        return self._proposals

    def _prepare_loss_indices(self):
        """index tensors of __call__ (source-domain rows, their positives, the per-class regression columns) depend
        on the sampled proposals only: they are built here, where the host synchronises anyway, so that the loss
        itself — issued behind the res5 head — needs no device->host round trip (loss.py:186-213)"""
        proposals = self._proposals
        labels = cat([p.get_field("labels") for p in proposals], dim=0)
        regression_targets = cat([p.get_field("regression_targets") for p in proposals], dim=0)
        domain_masks = cat([p.get_field("domain_labels") for p in proposals], dim=0)
        # both counts are known on the host (rows of source-domain images; positives the sampler kept among them)
        n_src = sum(len(p) for p, s in zip(proposals, self._is_source) if s)
        n_pos = sum(k for k, s in zip(self._sampled_pos, self._is_source) if s)
        src = _nz(domain_masks, size=n_src).squeeze(1)
        labels_src = labels[src]
        pos = _nz(labels_src > 0, size=n_pos).squeeze(1)
        labels_pos = labels_src[pos]
        if self.cls_agnostic_bbox_reg:
            map_inds = torch.arange(4, 8, device=labels.device)
        else:
            map_inds = 4 * labels_pos[:, None] + torch.arange(4, device=labels.device)
        self._loss_prep = dict(domain_masks=domain_masks, src=src, labels_src=labels_src,
                               rows_pos=src[pos][:, None], map_inds=map_inds,
                               regression_targets_pos=regression_targets[src][pos])

    def subsample_for_da(self, proposals, targets):
        """uniformly sampled proposals (all labels forced to 0) for the instance-level domain classifier
        (loss.py:132-163); does NOT replace self._proposals"""
        if self._fused_ok(proposals):
            return self._subsample_for_da_fused(proposals, targets)
        labels, _, domain_labels = self.prepare_targets(proposals, targets, sample_for_da=True)
        pos_masks, neg_masks = self.fg_bg_sampler(labels, self._all_negative)
        proposals = list(proposals)
        for prop, dom in zip(proposals, domain_labels):
            prop.add_field("domain_labels", dom)
        return self._take_sampled(proposals, pos_masks, neg_masks)

    def _subsample_for_da_fused(self, proposals, targets):
        sampler = self.fg_bg_sampler
        cap = sampler.batch_size_per_image
        out = []
        counts = None
        for prop, tgt in zip(proposals, targets):
            is_source = is_source_image(tgt)
            n, dev = len(prop), prop.bbox.device
            if n <= cap:
                # every row is taken, in ascending order (the reference's sampler would return them all)
                if not prop.has_field("domain_labels"):
                    prop.add_field("domain_labels", torch.full((n,), bool(is_source), dtype=torch.bool, device=dev))
                out.append(prop)
                continue
            if counts is None:
                counts = torch.empty((len(proposals), 2), dtype=torch.int32, device=dev)
            s = _C.sample_rois(prop.bbox, None, None, cap, int(cap * sampler.positive_fraction), rng.next_seed(dev),
                               is_source, counts[len(out)])
            b = BoxList(s["boxes"], prop.size, prop.mode)      # n > cap negatives: exactly cap rows are taken
            for name in prop.fields():
                if name != "domain_labels":
                    b.add_field(name, prop.get_field(name)[s["idx"]])
            b.add_field("domain_labels", s["domain"])
            out.append(b)
        return out

    def __call__(self, class_logits, box_regression):
        """-> (classification_loss, box_loss, domain_masks); only source-domain rows enter the detection
        losses (loss.py:165-221)"""
        class_logits = cat(class_logits, dim=0)
        box_regression = cat(box_regression, dim=0)
        if not hasattr(self, "_proposals"):
            raise RuntimeError("subsample needs to be called before")
        prep = self._loss_prep
        if prep.get("rows"):
            cls_loss, box_loss = fast_rcnn_loss_rows_fused(class_logits, box_regression, prep["loss_labels"],
                                                           prep["regression_targets"])
            return cls_loss, box_loss, prep["domain_masks"]
        labels = prep["labels_src"]
        if class_logits.is_cuda and not self.cls_agnostic_bbox_reg:
            cls_loss, box_loss = fast_rcnn_loss_fused(class_logits, box_regression, prep["src"], labels,
                                                      prep["rows_pos"].reshape(-1), prep["map_inds"],
                                                      prep["regression_targets_pos"])
            return cls_loss, box_loss, prep["domain_masks"]
        classification_loss = F.cross_entropy(class_logits.index_select(0, prep["src"]), labels)
        box_loss = smooth_l1_loss(box_regression[prep["rows_pos"], prep["map_inds"]], prep["regression_targets_pos"],
                                  size_average=False, beta=1)
        box_loss = box_loss / labels.numel()
        return classification_loss, box_loss, prep["domain_masks"]


def make_roi_box_loss_evaluator(cfg):
    matcher = Matcher(cfg.MODEL.ROI_HEADS.FG_IOU_THRESHOLD, cfg.MODEL.ROI_HEADS.BG_IOU_THRESHOLD,
                      allow_low_quality_matches=False)
    box_coder = BoxCoder(weights=cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS)
    sampler = BalancedPositiveNegativeSampler(cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
                                              cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION)
    return FastRCNNLossComputation(matcher, sampler, box_coder, cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)
