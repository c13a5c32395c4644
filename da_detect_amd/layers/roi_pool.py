"""ROIPool (max over each bin) on the HIP kernels — operator surface of maskrcnn_benchmark/layers/roi_pool.py:11-63:
`roi_pool(input, rois, output_size, spatial_scale)` and the `ROIPool` module."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


def _hw(output_size):
    return (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)


class _ROIPool(Function):
    """forward keeps the arg-max cell of every bin; backward routes each bin's gradient to that cell"""

    @staticmethod
    def forward(ctx, features, rois, output_size, spatial_scale):
        ph, pw = _hw(output_size)
        pooled, winners = _C.roi_pool_forward(features, rois, spatial_scale, ph, pw)
        ctx.geometry = (spatial_scale, ph, pw) + tuple(features.shape)
        ctx.save_for_backward(features, rois, winners)
        return pooled

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_pooled):
        features, rois, winners = ctx.saved_tensors
        scale, ph, pw, n, c, h, w = ctx.geometry
        return _C.roi_pool_backward(grad_pooled, features, rois, winners, scale, ph, pw, n, c, h, w), None, None, None


roi_pool = _ROIPool.apply


class ROIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super(ROIPool, self).__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale

    def forward(self, input, rois):
        return roi_pool(input, rois, self.output_size, self.spatial_scale)

    def extra_repr(self):
        return "output_size=%s, spatial_scale=%s" % (self.output_size, self.spatial_scale)
