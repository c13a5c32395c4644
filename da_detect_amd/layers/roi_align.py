"""ROIAlign layer (reference: maskrcnn_benchmark/layers/roi_align.py:11-68) on the HIP kernels."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C
from ..utils import streams


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        return _C.roi_align_forward(input, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                    sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        # weight gradients queued by the preceding node (ROI head's first block) run on the lane beside this gather
        after = None
        if grad_output.is_cuda and streams.deferred_pending():
            after = torch.cuda.current_stream(grad_output.device).record_event()
        grad_input = _C.roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0],
                                           ctx.output_size[1], bs, ch, h, w, ctx.sampling_ratio)
        streams.flush_deferred_wgrads(grad_output.device, after)
        return grad_input, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlign, self).__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "{}(output_size={}, spatial_scale={}, sampling_ratio={})".format(
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
