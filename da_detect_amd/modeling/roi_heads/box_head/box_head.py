"""Box head with the extra domain-adaptation ROI pass
(reference: maskrcnn_benchmark/modeling/roi_heads/box_head/box_head.py:22-118)."""
import torch

from .inference import make_roi_box_post_processor
from .loss import make_roi_box_loss_evaluator
from .roi_box_feature_extractors import make_roi_box_feature_extractor
from .roi_box_predictors import make_roi_box_predictor


class ROIBoxHead(torch.nn.Module):
    def __init__(self, cfg):
        super(ROIBoxHead, self).__init__()
        self.feature_extractor = make_roi_box_feature_extractor(cfg)
        self.predictor = make_roi_box_predictor(cfg)
        self.post_processor = make_roi_box_post_processor(cfg)
        self.loss_evaluator = make_roi_box_loss_evaluator(cfg)

    def forward(self, features, proposals, targets=None):
        """-> (x, proposals | detections, losses, da_ins_feas, da_ins_labels).  Training runs two passes of
        pooler + res5 + predictor: the sampled detection ROIs, then BATCH_SIZE_PER_IMAGE uniformly sampled
        ROIs per image whose features / domain labels feed the instance-level domain classifier."""
        if self.training:
            with torch.no_grad():
                proposals = self.loss_evaluator.subsample(proposals, targets)
        x = self.feature_extractor(features, proposals)
        class_logits, box_regression = self.predictor(x)
        if not self.training:
            return x, self.post_processor((class_logits, box_regression), proposals), {}, x, None
        loss_classifier, loss_box_reg, _ = self.loss_evaluator([class_logits], [box_regression])
        with torch.no_grad():
            da_proposals = self.loss_evaluator.subsample_for_da(proposals, targets)
        da_ins_feas = self.feature_extractor(features, da_proposals)
        # the reference runs the predictor on the DA features too and only keeps the domain mask, which
        # depends on self._proposals alone (box_head.py:107-110); the dead predictor call is skipped
        da_ins_labels = torch.cat([p.get_field("domain_labels") for p in self.loss_evaluator._proposals], dim=0)
        return (x, proposals, dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg), da_ins_feas,
                da_ins_labels)


def build_roi_box_head(cfg):
    return ROIBoxHead(cfg)
