"""ROIAlign layer (reference: maskrcnn_benchmark/layers/roi_align.py:11-68) on the HIP kernels."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ..utils import streams


def _hw(output_size):
    return (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)


class _ROIAlign(Function):
    """bilinear ROI pooling; forward bit-exact with the reference's CPU operator, backward a deterministic gather"""

    @staticmethod
    def forward(ctx, features, rois, output_size, spatial_scale, sampling_ratio, bin_stride=1, live_images=None):
        ph, pw = _hw(output_size)
        ctx.geometry = (spatial_scale, ph, pw, sampling_ratio) + tuple(features.shape)
        ctx.bin_stride = int(bin_stride)
        ctx.live_images = live_images      # ROIs reference the leading `live_images` images only (None: unknown)
        ctx.save_for_backward(rois)
        return _C.roi_align_forward(features, rois, spatial_scale, ph, pw, sampling_ratio, bin_stride=ctx.bin_stride)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_pooled):
        (rois,) = ctx.saved_tensors
        scale, ph, pw, ratio, n, c, h, w = ctx.geometry
        grad_features = _C.roi_align_backward(grad_pooled, rois, scale, ph, pw, n, c, h, w, ratio,
                                              bin_stride=ctx.bin_stride, live_images=ctx.live_images)
        return grad_features, None, None, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlign, self).__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois, bin_stride=1, live_images=None):
        """bin_stride s > 1 (not in the reference's layer): every s-th bin of the output_size grid in both directions, as
        a compact tensor — for a consumer that reads nothing else (a stride-s 1x1 convolution).  live_images: the ROIs'
        batch indices are all < live_images (lets the backward skip the other images' pixels)"""
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, bin_stride,
                         live_images)

    def extra_repr(self):
        return "output_size=%s, spatial_scale=%s, sampling_ratio=%s" % (self.output_size, self.spatial_scale,
                                                                        self.sampling_ratio)
