"""DA-head loss helpers that stay in host tensor code (small [R]-sized tensors)
(reference: maskrcnn_benchmark/modeling/da_heads/loss.py:28-228)."""
import torch
from torch import nn
from torch.nn import functional as F

from ...layers import consistency_loss
from ...structures.bounding_box import is_source_image
from .fused import triplet_margin_loss_w


_LABEL_CACHE = {}


def image_domain_labels(targets):
    """float [N]: 1 for source-domain images, 0 for target-domain images (loss.py:46-53 `prepare_masks`)"""
    # built from the host-side flags with ONE cached device tensor per flag pattern: torch.tensor(list, device=...)
    # is a pageable host->device copy that blocks until the compute stream has drained
    key = (tuple(bool(is_source_image(t)) for t in targets), str(targets[0].bbox.device))
    lab = _LABEL_CACHE.get(key)
    if lab is None:
        lab = torch.tensor([1.0 if f else 0.0 for f in key[0]], dtype=torch.float32, device=targets[0].bbox.device)
        _LABEL_CACHE[key] = lab
    return lab


def da_ins_loss(da_ins, da_ins_labels):
    """BCE-with-logits of the instance logits against the ROI domain labels (loss.py:95-97,169-173)"""
    return F.binary_cross_entropy_with_logits(torch.squeeze(da_ins), da_ins_labels.to(torch.float32))


def da_consist_loss(img_mean_sig, da_ins_consist, da_ins_labels):
    """img_mean_sig: per-image mean sigmoid of one level, or a list with one entry per level"""
    levels = img_mean_sig if isinstance(img_mean_sig, (list, tuple)) else [img_mean_sig]
    return consistency_loss(levels, da_ins_consist, da_ins_labels, size_average=True)


class TripletMargins(object):
    """margin state of the adaptive triplet losses (loss.py:128-222): the margin grows by `lr` whenever the
    previous loss was exactly 0, until int(margin) == int(max_margin)."""

    def __init__(self):
        self.margin_ins = 0.0
        self.margin_img = 0.0

    def triplet_img_loss(self, anchor, positive, negative, prev_loss, adaptive=True, lr=0.001, max_margin=3.0,
                         margin=1.0):
        if self.margin_img == 0.0:
            self.margin_img = margin
        if adaptive:
            if prev_loss == 0.0 and int(self.margin_img) != int(max_margin):
                self.margin_img = self.margin_img + lr
        else:
            self.margin_img = margin
        return triplet_margin_loss_w(anchor, positive, negative, self.margin_img)

    def triplet_ins_loss(self, anchor, positive, negative, prev_loss, adaptive=True, lr=0.001, max_margin=3.0,
                         margin=1.0):
        if self.margin_ins == 0.0:
            self.margin_ins = margin
        if adaptive:
            if prev_loss == 0.0 and int(self.margin_ins) != int(max_margin):
                self.margin_ins = self.margin_ins + lr
        else:
            self.margin_ins = margin
        return nn.TripletMarginLoss(margin=self.margin_ins, p=2)(anchor, positive, negative)
