"""ROI heads container (reference: maskrcnn_benchmark/modeling/roi_heads/roi_heads.py:20-98).  Only the box
head is on the DA Faster R-CNN path; mask / keypoint heads are out of scope."""
import torch

from .box_head.box_head import build_roi_box_head


class CombinedROIHeads(torch.nn.ModuleDict):
    def __init__(self, cfg, heads):
        super(CombinedROIHeads, self).__init__(heads)
        self.cfg = cfg.clone()

    def forward(self, features, proposals, targets=None):
        losses = {}
        x, detections, loss_box, da_ins_feas, da_ins_labels = self.box(features, proposals, targets)
        losses.update(loss_box)
        return x, detections, losses, da_ins_feas, da_ins_labels


def build_roi_heads(cfg):
    if cfg.MODEL.RETINANET_ON:
        return []
    if cfg.MODEL.MASK_ON or cfg.MODEL.KEYPOINT_ON:
        raise NotImplementedError("mask / keypoint heads are outside the DA Faster R-CNN path")
    heads = []
    if not cfg.MODEL.RPN_ONLY:
        heads.append(("box", build_roi_box_head(cfg)))
    return CombinedROIHeads(cfg, heads) if heads else heads
