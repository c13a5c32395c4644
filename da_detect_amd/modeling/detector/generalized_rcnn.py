"""Generalized R-CNN with domain-adaptation heads
(reference: maskrcnn_benchmark/modeling/detector/generalized_rcnn.py:37-156).

forward(images, targets) -> dict of scalar losses (training) | list[BoxList] detections (eval).
Batch layout contracts inherited from the reference trainer (engine/trainer.py:215-224): source images first;
plain DA = [source, target]; triplet DA = [source, target(positive), auxiliary(negative)].
"""
import os

import torch
from torch import nn

from ...structures.image_list import to_image_list
from ...utils.streams import record, side_stream
from ..backbone import build_backbone
from ..da_heads.da_heads import build_da_heads, build_da_heads_triplet
from ..elision import elision_enabled, leading_source_images
from ..roi_heads.roi_heads import build_roi_heads
from ..rpn.rpn import build_rpn


_DA_AFTER_RPN = True


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super(GeneralizedRCNN, self).__init__()
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg)
        self.roi_heads = build_roi_heads(cfg)
        self.da_heads = build_da_heads(cfg)
        self.triplet_use = cfg.MODEL.DA_HEADS.TRIPLET_USE
        self.da_heads_triplet = build_da_heads_triplet(cfg) if self.triplet_use else False
        self.Aligned = cfg.MODEL.DA_HEADS.ALIGNMENT

    def _images_with_read_proposals(self, targets):
        """number of leading images whose RPN proposals some loss reads: the source images always (detection losses);
        a target-domain image only through the instance-level features (or as the ROI set of the aligned triplet
        passes); the auxiliary image of a triplet batch never (generalized_rcnn.py:100 passes proposals[0:2] on)"""
        n_src = leading_source_images(targets)
        if n_src == 0:
            return None
        if self.da_heads_triplet:
            if len(targets) != 3 or n_src != 1:
                return None
            return 2 if (self.da_heads_triplet.needs_instance_features or self.Aligned) else 1
        return len(targets) if self.da_heads.needs_instance_features else n_src

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        if self.training and images.tensors.is_cuda:
            # lets the RPN prepare its loss targets on a side stream without waiting for the backbone
            self.rpn.inputs_ready = torch.cuda.current_stream(images.tensors.device).record_event()
        features = self.backbone(images.tensors)
        holder = {}
        if self.training and self.da_heads and not self.da_heads_triplet and features[0].is_cuda:
            # image-level DA loss + its backward (DomainAdaptationModule.early_image_level) on their own stream, beside
            # the RPN branch and the box head: they need nothing but the backbone features.  _DA_AFTER_RPN: queued behind
            # the RPN branch's backward instead of beside the RPN head's forward — the compute stream then has GEMM work
            # while the host waits for the proposals and samples ROIs (tools/gemm_table.py --holes: 0.25 - 0.45 ms)
            dev = features[0].device

            def run_early_da():
                main = torch.cuda.current_stream(dev)
                stream = side_stream(dev, 3)
                stream.wait_stream(main)
                with torch.cuda.stream(stream):
                    holder["grads"] = self.da_heads.early_image_level(features, targets)
                if holder["grads"] is not None:
                    holder["ready"] = stream.record_event()
                    record(features, stream)
                    record(holder["grads"], main)      # consumed on the compute stream, in the backward pass

            if _DA_AFTER_RPN and self.rpn.early_backward:
                self.rpn.after_early_backward = run_early_da
            else:
                run_early_da()
        if self.training and self.roi_heads and (self.da_heads or self.da_heads_triplet) and elision_enabled():
            self.rpn.live_images = self._images_with_read_proposals(targets)
        if self.training and self.roi_heads and features[0].is_cuda:
            # the box head's sampler reads the NMS result on the device: the RPN need not bring the kept count to the host
            self.rpn.box_selector_train.defer = self.roi_heads.box.loss_evaluator.accepts_pending(targets)
        proposals, proposal_losses = self.rpn(images, features, targets)
        if self.training:
            pending, self.rpn.after_early_backward = self.rpn.after_early_backward, None
            if pending is not None:
                pending()                    # the RPN took a path without an early backward
            features = self.rpn.bridge_features(features, holder.get("grads"), holder.get("ready"))
            if self.roi_heads:
                self.roi_heads.box.proposals_ready, self.rpn.proposals_ready = self.rpn.proposals_ready, None
        da_losses, detector_losses = {}, {}
        if self.roi_heads:
            if self.training and self.da_heads_triplet:
                f = features[0]
                assert f.shape[0] == 3, "triplet training expects [source, target, auxiliary] batches"
                da_img_fea_set = [[f[0:1]], [f[1:2]], [f[2:3]]]
                ori_features, ori_targets = [f[0:2]], targets[0:2]
                self.roi_heads.box.ins_features_unused = not self.da_heads_triplet.needs_instance_features
                x, result, detector_losses, da_ins_feas, da_ins_labels = self.roi_heads(
                    ori_features, proposals[0:2], ori_targets)
                if self.Aligned:
                    # all three domains are pooled with the TARGET image's proposals (generalized_rcnn.py:110-112)
                    ins_set = []
                    for fea, tgt in zip(da_img_fea_set, (targets[0], targets[1], targets[2])):
                        _, _, _, feas, _ = self.roi_heads(fea, [proposals[1]], [tgt])
                        ins_set.append(feas)
                else:
                    ins_set = [0, 0, 0]
                da_losses = self.da_heads_triplet(ori_features, da_ins_feas, da_ins_labels, ins_set,
                                                  da_img_fea_set, ori_targets)
            elif self.training and self.da_heads:
                self.roi_heads.box.ins_features_unused = not self.da_heads.needs_instance_features
                x, result, detector_losses, da_ins_feas, da_ins_labels = self.roi_heads(features, proposals, targets)
                da_losses = self.da_heads(features, da_ins_feas, da_ins_labels, targets)
            else:
                # evaluation, and plain (non-DA) training — the latter raises UnboundLocalError in the
                # reference (generalized_rcnn.py:150); upstream behaviour is used instead
                x, result, detector_losses, _, _ = self.roi_heads(features, proposals, targets)
        else:
            result = proposals
        if self.training:
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            losses.update(da_losses)
            return losses
        return result
