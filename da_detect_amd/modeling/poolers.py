"""ROI feature pooler (reference: maskrcnn_benchmark/modeling/poolers.py:11-121)."""
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from .. import _C
from ..layers import ROIAlign
from .utils import cat


class _PyramidROIAlign(torch.autograd.Function):
    """ROIAlign over a feature pyramid without per-level index lists (reference: poolers.py:108-121 — per level one
    `nonzero`, i.e. one device->host round trip, a gather of the ROIs, `_C.roi_align_forward` and an index_put into the
    result).  `levels` stays on the device: one launch per level over ALL ROIs, each pooling the ROIs of its level into
    their own rows (dadet_roi_align_forward_level); backward is one deterministic gather per level
    (dadet_roi_align_backward_level)."""

    @staticmethod
    def forward(ctx, rois, levels, conf, *feats):
        out_size, scales, sampling_ratio = conf
        out = torch.empty((rois.shape[0], feats[0].shape[1], out_size, out_size), dtype=feats[0].dtype,
                          device=feats[0].device).contiguous(memory_format=torch.channels_last)
        for lvl, (f, s) in enumerate(zip(feats, scales)):
            _C.roi_align_forward_level(f, rois, levels, lvl, out, s, out_size, out_size, sampling_ratio)
        ctx.conf = conf
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.save_for_backward(rois, levels)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        rois, levels = ctx.saved_tensors
        out_size, scales, sampling_ratio = ctx.conf
        grads = []
        for lvl, (shape, s) in enumerate(zip(ctx.shapes, scales)):
            if not ctx.needs_input_grad[3 + lvl]:
                grads.append(None)
                continue
            n, c, h, w = shape
            grads.append(_C.roi_align_backward_level(grad, rois, levels, lvl, s, out_size, out_size, n, c, h, w,
                                                     sampling_ratio))
        return (None, None, None) + tuple(grads)


# DADET_PYRAMID_ROIALIGN=0: the reference's per-level split (nonzero + gather + index_put per level)
_PYRAMID_KERNELS = __import__("os").environ.get("DADET_PYRAMID_ROIALIGN", "1") == "1"


class LevelMapper(object):
    """FPN level assignment k = floor(k0 + log2(sqrt(area) / 224)) (poolers.py:11-42)"""

    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max = k_min, k_max
        self.s0, self.lvl0, self.eps = canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(cat([b.area() for b in boxlists]))
        lvls = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        # clamped again AFTER the cast: a NaN / inf area (degenerate box) survives the float clamp as NaN and casts to an
        # arbitrary integer — _PyramidROIAlign would then leave that ROI's rows of its torch.empty output unwritten
        lvls = torch.clamp(lvls, min=self.k_min, max=self.k_max).to(torch.int64) - self.k_min
        return lvls.clamp_(0, self.k_max - self.k_min)


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super(Pooler, self).__init__()
        self.poolers = nn.ModuleList(
            [ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio) for s in scales])
        self.output_size = output_size
        lvl_min = -torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item()
        lvl_max = -torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item()
        self.map_levels = LevelMapper(lvl_min, lvl_max)

    def convert_to_roi_format(self, boxes):
        """list[BoxList] -> [R,5] (batch index, x1, y1, x2, y2) (poolers.py:78-89)"""
        concat = cat([b.bbox for b in boxes], dim=0)
        ids = cat([torch.full((len(b), 1), i, dtype=concat.dtype, device=concat.device)
                   for i, b in enumerate(boxes)], dim=0)
        return torch.cat([ids, concat], dim=1)

    def forward(self, x, boxes, bin_stride=1):
        rois = self.convert_to_roi_format(boxes)
        if len(self.poolers) == 1:
            # the batch index of a ROI is its BoxList's position: images behind the last BoxList have no ROI
            return self.poolers[0](x[0], rois, bin_stride, live_images=len(boxes))
        if bin_stride != 1:
            raise NotImplementedError("bin_stride over several pyramid levels")
        levels = self.map_levels(boxes)
        out_size = self.output_size[0]
        if (_PYRAMID_KERNELS and rois.is_cuda and len(rois) > 0 and x[0].shape[1] % 4 == 0 and out_size <= 14
                and self.output_size[0] == self.output_size[1]):
            conf = (out_size, tuple(p.spatial_scale for p in self.poolers), self.poolers[0].sampling_ratio)
            # (the mapper's result is a float tensor, as the reference's: int64 minus the Python float k_min)
            return _PyramidROIAlign.apply(rois, levels.to(torch.int64).contiguous(), conf, *x[:len(self.poolers)])
        result = torch.zeros((len(rois), x[0].shape[1], out_size, out_size), dtype=x[0].dtype,
                             device=x[0].device).contiguous(memory_format=torch.channels_last)
        for level, (feat, pooler) in enumerate(zip(x, self.poolers)):
            idx = torch.nonzero(levels == level).squeeze(1)
            result[idx] = pooler(feat, rois[idx])
        return result
