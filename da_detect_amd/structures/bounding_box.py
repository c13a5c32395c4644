"""BoxList — a set of boxes of one image plus per-box fields.

API and semantics of the reference container (reference: maskrcnn_benchmark/structures/bounding_box.py:9-255):
xyxy / xywh modes with the "+1" pixel convention, `size` = (image_width, image_height), per-box
`extra_fields`, indexing applies to every field.
"""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1
_MODES = ("xyxy", "xywh")


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(bbox.ndimension()))
        if bbox.size(-1) != 4:
            raise ValueError("last dimenion of bbox should have a size of 4, got {}".format(bbox.size(-1)))
        if mode not in _MODES:
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox = bbox
        self.size = image_size  # (width, height)
        self.mode = mode
        self.extra_fields = {}

    # ---- fields -----------------------------------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        self.extra_fields.update(other.extra_fields)

    def _carry_fields(self, dst, op):
        """copy fields to dst; non-tensor fields (masks, keypoints) are transformed with `op`"""
        for k, v in self.extra_fields.items():
            dst.add_field(k, v if isinstance(v, torch.Tensor) else op(v))
        return dst

    # ---- representation ---------------------------------------------------------------------
    def _split_into_xyxy(self):
        if self.mode == "xyxy":
            return self.bbox.split(1, dim=-1)
        x, y, w, h = self.bbox.split(1, dim=-1)
        return x, y, x + (w - 1).clamp(min=0), y + (h - 1).clamp(min=0)

    def convert(self, mode):
        if mode not in _MODES:
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._split_into_xyxy()
        if mode == "xyxy":
            data = torch.cat((x1, y1, x2, y2), dim=-1)
        else:
            data = torch.cat((x1, y1, x2 - x1 + 1, y2 - y1 + 1), dim=-1)
        out = BoxList(data, self.size, mode=mode)
        out._copy_extra_fields(self)
        return out

    # ---- geometry ---------------------------------------------------------------------------
    def resize(self, size, *args, **kwargs):
        """boxes scaled to an image of `size` = (width, height) (bounding_box.py:91-126)"""
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        op = lambda v: v.resize(size, *args, **kwargs)  # noqa: E731
        if rw == rh:
            return self._carry_fields(BoxList(self.bbox * rw, size, mode=self.mode), op)
        x1, y1, x2, y2 = self._split_into_xyxy()
        data = torch.cat((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=-1)
        return self._carry_fields(BoxList(data, size, mode="xyxy"), op).convert(self.mode)

    def transpose(self, method):
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        W, H = self.size
        x1, y1, x2, y2 = self._split_into_xyxy()
        if method == FLIP_LEFT_RIGHT:
            data = torch.cat((W - x2 - 1, y1, W - x1 - 1, y2), dim=-1)
        else:  # the reference omits the -1 for vertical flips (bounding_box.py:150-154)
            data = torch.cat((x1, H - y2, x2, H - y1), dim=-1)
        out = BoxList(data, self.size, mode="xyxy")
        return self._carry_fields(out, lambda v: v.transpose(method)).convert(self.mode)

    def crop(self, box):
        x1, y1, x2, y2 = self._split_into_xyxy()
        w, h = box[2] - box[0], box[3] - box[1]
        data = torch.cat(((x1 - box[0]).clamp(min=0, max=w), (y1 - box[1]).clamp(min=0, max=h),
                          (x2 - box[0]).clamp(min=0, max=w), (y2 - box[1]).clamp(min=0, max=h)), dim=-1)
        out = BoxList(data, (w, h), mode="xyxy")
        return self._carry_fields(out, lambda v: v.crop(box)).convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        """in-place clamp to [0, W-1] x [0, H-1] (bounding_box.py:214-224)"""
        W, H = self.size
        self.bbox[:, 0].clamp_(min=0, max=W - 1)
        self.bbox[:, 1].clamp_(min=0, max=H - 1)
        self.bbox[:, 2].clamp_(min=0, max=W - 1)
        self.bbox[:, 3].clamp_(min=0, max=H - 1)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    # ---- container protocol -------------------------------------------------------------------
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        if not isinstance(fields, (list, tuple)):
            fields = [fields]
        for f in fields:
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return out

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)


class PendingProposals(BoxList):
    """RPN proposals of one image whose NUMBER is still on the device.

    The RPN's NMS leaves `keep` (positions into the score-sorted candidate list) and a kept count in device memory; the
    reference reads the count back to slice the list (rpn/inference.py:102) and then concatenates the ground-truth boxes
    of source images (rpn/inference.py:51-74).  The training path does not need the list on the host at all — the box
    head's sampler (dadet_proposals_sample) reads keep / count where they lie — so this object just carries the pieces.
    Anything else that looks at it (`.bbox`, `len()`, a field, indexing — tests, evaluation code, a custom head) gets
    an ordinary BoxList: the first such access materialises it, with the device->host synchronisation the reference
    pays at that point."""

    is_pending_proposals = True

    def __init__(self, sorted_boxes, sorted_scores, keep, count_dev, post_n, image_size):
        # deliberately not BoxList.__init__: bbox / extra_fields are properties here
        self._pending = dict(sorted_boxes=sorted_boxes, sorted_scores=sorted_scores, keep=keep, count_dev=count_dev,
                             post_n=int(post_n), gt=None)
        self._bbox = None
        self._fields = None
        self.size = image_size
        self.mode = "xyxy"

    # ---- the deferred form -------------------------------------------------------------------------------------------
    @property
    def pending(self):
        """dict(sorted_boxes, sorted_scores, keep, count_dev, post_n, gt) or None once materialised"""
        return self._pending

    def attach_ground_truth(self, gt_boxlist):
        """add_gt_proposals (rpn/inference.py:51-74) for a source image: appended when the list is formed"""
        if self._pending is None:
            raise RuntimeError("proposals already materialised")
        self._pending["gt"] = gt_boxlist

    def upper_bound(self):
        p = self._pending
        return len(self) if p is None else p["post_n"] + (len(p["gt"]) if p["gt"] is not None else 0)

    def pending_tensors(self):
        p = self._pending
        if p is None:
            return [self._bbox] + [v for v in self._fields.values() if isinstance(v, torch.Tensor)]
        out = [p["sorted_boxes"], p["sorted_scores"], p["keep"], p["count_dev"]]
        if p["gt"] is not None:
            out.append(p["gt"].bbox)
        return out

    def _materialize(self):
        p, self._pending = self._pending, None
        if p is None:
            return
        n = min(int(p["count_dev"].item()), p["post_n"])       # the reference's round trip, paid only when asked for
        keep = p["keep"][:n]
        boxes, scores = p["sorted_boxes"][keep], p["sorted_scores"][keep]
        if p["gt"] is not None:
            gt = p["gt"]
            boxes = torch.cat([boxes, gt.bbox.to(boxes.dtype)], dim=0)
            scores = torch.cat([scores, torch.ones(len(gt), dtype=scores.dtype, device=scores.device)], dim=0)
        self._bbox = boxes
        self._fields = {"objectness": scores}

    @property
    def bbox(self):
        self._materialize()
        return self._bbox

    @bbox.setter
    def bbox(self, value):
        self._materialize()
        self._bbox = value

    @property
    def extra_fields(self):
        self._materialize()
        return self._fields

    @extra_fields.setter
    def extra_fields(self, value):
        self._materialize()
        self._fields = value


def is_source_image(target):
    """True when the image's ground truth comes from the source domain (`is_source` field, any element set).
    The answer is read back from the device ONCE per BoxList and kept on the object: the training path asks for
    it in several places per step (rpn/loss.py:66, rpn/inference.py:64, box_head/loss.py:80, da_heads/loss.py:85
    of the reference each synchronise on it)."""
    flag = getattr(target, "_is_source_host", None)
    if flag is None:
        flag = bool(target.get_field("is_source").any())
        target._is_source_host = flag
    return flag
