"""Box <-> regression-target codec (reference: maskrcnn_benchmark/modeling/box_coder.py:6-95)."""
import math

import torch


class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    @staticmethod
    def _whc(b):
        w = b[:, 2] - b[:, 0] + 1
        h = b[:, 3] - b[:, 1] + 1
        return w, h, b[:, 0] + 0.5 * w, b[:, 1] + 0.5 * h

    def encode(self, reference_boxes, proposals):
        """targets of `reference_boxes` relative to `proposals` (box_coder.py:22-50)"""
        ew, eh, ecx, ecy = self._whc(proposals)
        gw, gh, gcx, gcy = self._whc(reference_boxes)
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * torch.log(gw / ew),
                            wh * torch.log(gh / eh)), dim=1)

    def decode(self, rel_codes, boxes):
        """apply [N, 4k] deltas to [N, 4] boxes (box_coder.py:52-95)"""
        boxes = boxes.to(rel_codes.dtype)
        w, h, cx, cy = self._whc(boxes)
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx = dx * w[:, None] + cx[:, None]
        pcy = dy * h[:, None] + cy[:, None]
        pw = torch.exp(dw) * w[:, None]
        ph = torch.exp(dh) * h[:, None]
        out = torch.zeros_like(rel_codes)
        out[:, 0::4] = pcx - 0.5 * pw
        out[:, 1::4] = pcy - 0.5 * ph
        out[:, 2::4] = pcx + 0.5 * pw - 1
        out[:, 3::4] = pcy + 0.5 * ph - 1
        return out
