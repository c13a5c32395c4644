"""Deformable position-sensitive ROI pooling — operator API of the reference's vendored tree
(tools/cityscapes/maskrcnn_benchmark/layers/dcn/deform_pool_func.py:10-95, deform_pool_module.py:6-150),
served by csrc/deform.hip (NHWC)."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import _C
from ..conv import linear


class DeformRoIPoolingFunction(Function):
    """argument order / checks of deform_pool_func.py:12-50"""

    @staticmethod
    def forward(ctx, data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1,
                part_size=None, sample_per_part=4, trans_std=.0):
        ctx.spatial_scale = spatial_scale
        ctx.out_size = out_size
        ctx.out_channels = out_channels
        ctx.no_trans = bool(no_trans)
        ctx.group_size = group_size
        ctx.part_size = out_size if part_size is None else part_size
        ctx.sample_per_part = sample_per_part
        ctx.trans_std = trans_std
        assert 0.0 <= ctx.trans_std <= 1.0
        if not data.is_cuda:
            raise NotImplementedError("deform_roi_pooling has no CPU path")
        output, count = _C.deform_psroi_pool_forward(data, rois, offset, ctx.no_trans, ctx.spatial_scale,
                                                        ctx.out_channels, ctx.group_size, ctx.out_size,
                                                        ctx.part_size, ctx.sample_per_part, ctx.trans_std)
        if data.requires_grad or rois.requires_grad or offset.requires_grad:
            ctx.save_for_backward(data, rois, offset)
        ctx.output_count = count
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        data, rois, offset = ctx.saved_tensors
        grad_input, grad_offset = _C.deform_psroi_pool_backward(
            grad_output, data, rois, offset, ctx.output_count, ctx.no_trans, ctx.spatial_scale, ctx.out_channels,
            ctx.group_size, ctx.out_size, ctx.part_size, ctx.sample_per_part, ctx.trans_std)
        return (grad_input, None, grad_offset, None, None, None, None, None, None, None, None)


deform_roi_pooling = DeformRoIPoolingFunction.apply


class DeformRoIPooling(nn.Module):
    """deform_pool_module.py:6-33"""

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0):
        super().__init__()
        self.spatial_scale = spatial_scale
        self.out_size = out_size
        self.out_channels = out_channels
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def _pool(self, data, rois, offset, no_trans):
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size, self.out_channels, no_trans,
                                  self.group_size, self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, data, rois, offset):
        if self.no_trans:
            offset = data.new_empty(0)
        return self._pool(data, rois, offset, self.no_trans)


class _FC(nn.Linear):
    """nn.Linear whose contraction runs on the library GEMM"""

    def __init__(self, cin, cout, relu=False):
        super().__init__(cin, cout)
        self.relu = relu

    def forward(self, x):
        return linear(x, self.weight, self.bias, relu=self.relu)


def _offset_fc(out_size, out_channels, hidden):
    fc = nn.Sequential(_FC(out_size * out_size * out_channels, hidden, relu=True), _FC(hidden, hidden, relu=True),
                       _FC(hidden, out_size * out_size * 2))
    fc[-1].weight.data.zero_()
    fc[-1].bias.data.zero_()
    return fc


class DeformRoIPoolingPack(DeformRoIPooling):
    """deform_pool_module.py:36-86: offsets predicted from a first, undeformed pooling pass"""

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, deform_fc_channels=1024):
        super().__init__(spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part,
                         trans_std)
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            self.offset_fc = _offset_fc(self.out_size, self.out_channels, deform_fc_channels)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        empty = data.new_empty(0)
        if self.no_trans:
            return self._pool(data, rois, empty, True)
        n = rois.shape[0]
        x = self._pool(data, rois, empty, True)
        offset = self.offset_fc(x.reshape(n, -1)).view(n, 2, self.out_size, self.out_size)
        return self._pool(data, rois, offset, False)


class ModulatedDeformRoIPoolingPack(DeformRoIPooling):
    """deform_pool_module.py:89-150: v2 adds a sigmoid modulation of every pooled bin"""

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, deform_fc_channels=1024):
        super().__init__(spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part,
                         trans_std)
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            self.offset_fc = _offset_fc(self.out_size, self.out_channels, deform_fc_channels)
            self.mask_fc = nn.Sequential(_FC(self.out_size * self.out_size * self.out_channels, deform_fc_channels,
                                             relu=True),
                                         nn.Identity(),
                                         _FC(deform_fc_channels, self.out_size * self.out_size), nn.Sigmoid())
            self.mask_fc[2].weight.data.zero_()
            self.mask_fc[2].bias.data.zero_()

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        empty = data.new_empty(0)
        if self.no_trans:
            return self._pool(data, rois, empty, True)
        n = rois.shape[0]
        x = self._pool(data, rois, empty, True).reshape(n, -1)
        offset = self.offset_fc(x).view(n, 2, self.out_size, self.out_size)
        mask = self.mask_fc(x).view(n, 1, self.out_size, self.out_size)
        return self._pool(data, rois, offset, False) * mask
