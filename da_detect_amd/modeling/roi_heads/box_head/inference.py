"""Detection post-processing for evaluation (reference:
maskrcnn_benchmark/modeling/roi_heads/box_head/inference.py:12-150): softmax, per-class decode, clip,
score threshold, per-class NMS (HIP), top detections_per_img."""
import torch
import torch.nn.functional as F
from torch import nn

from ....structures.bounding_box import BoxList
from ....structures.boxlist_ops import boxlist_nms, cat_boxlist
from ...box_coder import BoxCoder


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None,
                 cls_agnostic_bbox_reg=False):
        super(PostProcessor, self).__init__()
        self.score_thresh = score_thresh
        self.nms = nms
        self.detections_per_img = detections_per_img
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg

    def forward(self, x, boxes):
        class_logits, box_regression = x
        class_prob = F.softmax(class_logits, -1)
        counts = [len(b) for b in boxes]
        concat_boxes = torch.cat([b.bbox for b in boxes], dim=0)
        if self.cls_agnostic_bbox_reg:
            box_regression = box_regression[:, -4:]
        proposals = self.box_coder.decode(box_regression.reshape(sum(counts), -1), concat_boxes)
        if self.cls_agnostic_bbox_reg:
            proposals = proposals.repeat(1, class_prob.shape[1])
        num_classes = class_prob.shape[1]
        results = []
        for prob, boxes_per_img, box in zip(class_prob.split(counts, dim=0), proposals.split(counts, dim=0), boxes):
            boxlist = BoxList(boxes_per_img.reshape(-1, 4), box.size, mode="xyxy")
            boxlist.add_field("scores", prob.reshape(-1))
            boxlist = boxlist.clip_to_image(remove_empty=False)
            results.append(self.filter_results(boxlist, num_classes))
        return results

    def filter_results(self, boxlist, num_classes):
        boxes = boxlist.bbox.reshape(-1, num_classes * 4)
        scores = boxlist.get_field("scores").reshape(-1, num_classes)
        device = scores.device
        above = scores > self.score_thresh
        per_class = []
        for j in range(1, num_classes):  # class 0 is background
            inds = above[:, j].nonzero().squeeze(1)
            bl = BoxList(boxes[inds, j * 4:(j + 1) * 4], boxlist.size, mode="xyxy")
            bl.add_field("scores", scores[inds, j])
            bl = boxlist_nms(bl, self.nms)
            bl.add_field("labels", torch.full((len(bl),), j, dtype=torch.int64, device=device))
            per_class.append(bl)
        result = cat_boxlist(per_class)
        n = len(result)
        if n > self.detections_per_img > 0:
            cls_scores = result.get_field("scores")
            thresh, _ = torch.kthvalue(cls_scores.cpu(), n - self.detections_per_img + 1)
            result = result[torch.nonzero(cls_scores >= thresh.item()).squeeze(1)]
        return result


def make_roi_box_post_processor(cfg):
    box_coder = BoxCoder(weights=cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS)
    return PostProcessor(cfg.MODEL.ROI_HEADS.SCORE_THRESH, cfg.MODEL.ROI_HEADS.NMS,
                         cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG, box_coder, cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)
