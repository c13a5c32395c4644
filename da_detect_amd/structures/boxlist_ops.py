"""BoxList operations (reference: maskrcnn_benchmark/structures/boxlist_ops.py:11-131)."""
import torch

from .bounding_box import BoxList


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """NMS over a BoxList; keeps the first `max_proposals` survivors (boxlist_ops.py:11-34)."""
    from ..layers import nms as _box_nms

    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)


def remove_small_boxes(boxlist, min_size):
    """keep boxes whose both sides are >= min_size (boxlist_ops.py:37-51)"""
    wh = boxlist.convert("xywh").bbox
    keep = ((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).nonzero().squeeze(1)
    return boxlist[keep]


def boxlist_iou(boxlist1, boxlist2):
    """[N,M] IoU with the +1 convention (boxlist_ops.py:56-91)"""
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    area1, area2 = boxlist1.area(), boxlist2.area()
    b1, b2 = boxlist1.bbox, boxlist2.bbox
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def _cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def cat_boxlist(bboxes):
    """concatenate BoxLists of one image (same size / mode / field set) (boxlist_ops.py:105-131)"""
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size, mode, fields = bboxes[0].size, bboxes[0].mode, set(bboxes[0].fields())
    assert all(b.size == size and b.mode == mode and set(b.fields()) == fields for b in bboxes)
    out = BoxList(_cat([b.bbox for b in bboxes], dim=0), size, mode)
    for f in fields:
        out.add_field(f, _cat([b.get_field(f) for b in bboxes], dim=0))
    return out
