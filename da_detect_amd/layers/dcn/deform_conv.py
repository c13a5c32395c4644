"""Deformable convolution v1 / v2 on the HIP kernels.

Function / module names and argument order follow the reference's vendored tree
(tools/cityscapes/maskrcnn_benchmark/layers/dcn/deform_conv_func.py:9-259, deform_conv_module.py:10-176,
layers/misc.py:114-203).  One autograd function serves both versions (v1 = no mask):
  forward   cols = deform_sample(x, offset, mask)            (HIP, csrc/deform.hip)
            y    = cols (*) W  (+ bias)                      (implicit-GEMM kernel, 1x1 over K = kh*kw*Cin)
  backward  gcols = gy (*) W^T ; (gx, goffset, gmask) = deform_sample_backward(...) ; gW = cols^T (*) gy
`groups` and `im2col_step` exist for signature compatibility; only groups == 1 is on the HIP path (as used by
every DCN config of the reference, configs/dcn/*.yaml)."""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from ... import _C
from ..misc import Conv2d

CL = torch.channels_last


def _one(v, name):
    v = _pair(v)
    if v[0] != v[1]:
        raise NotImplementedError("deformable convolution: %s must be the same in both dimensions" % name)
    return int(v[0])


def _as_1x1(weight):
    """[Cout,Cin,kh,kw] (physically [Cout][kh][kw][Cin]) -> [Cout, kh*kw*Cin, 1, 1] with K = (tap, ci)"""
    cout, cin, kh, kw = weight.shape
    return weight.contiguous(memory_format=CL).permute(0, 2, 3, 1).reshape(cout, kh * kw * cin, 1, 1)


class _DeformConv(Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
        cout, cin, kh, kw = weight.shape
        cols = _C.deform_sample_forward(x, offset, mask, kh, kw, stride, padding, dilation, deformable_groups)
        w1x1 = _as_1x1(weight)
        y = _C.conv_forward(cols, w1x1, None, bias)
        ctx.save_for_backward(x, offset, mask, weight, cols)
        ctx.conf = (kh, kw, stride, padding, dilation, deformable_groups, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, offset, mask, weight, cols = ctx.saved_tensors
        kh, kw, stride, padding, dilation, dg, has_bias = ctx.conf
        cout, cin = weight.shape[0], weight.shape[1]
        gy = gy.contiguous(memory_format=CL)
        w1x1 = _as_1x1(weight)
        gcols = _C.conv_forward(gy, _C.conv_weight_transpose(w1x1))
        gx, goffset, gmask = _C.deform_sample_backward(x, offset, mask, gcols, kh, kw, stride, padding, dilation, dg,
                                                       need_x=ctx.needs_input_grad[0])
        gw = None
        if ctx.needs_input_grad[3]:
            gw1 = _C.conv_wgrad(cols, gy, (cout, kh * kw * cin, 1, 1))          # [Cout, K, 1, 1], K = (tap, ci)
            gw = gw1.reshape(cout, kh, kw, cin).permute(0, 3, 1, 2)             # channels_last [Cout,Cin,kh,kw]
        gb = _C.colsum(gy) if (has_bias and ctx.needs_input_grad[4]) else None
        return gx, goffset, gmask, gw, gb, None, None, None, None


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    """DCNv1 (deform_conv_func.py:9-147)"""
    if groups != 1:
        raise NotImplementedError("deform_conv: groups > 1")
    return _DeformConv.apply(input, offset, None, weight, None, _one(stride, "stride"), _one(padding, "padding"),
                             _one(dilation, "dilation"), deformable_groups)


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """DCNv2 (deform_conv_func.py:150-259)"""
    if groups != 1:
        raise NotImplementedError("modulated_deform_conv: groups > 1")
    return _DeformConv.apply(input, offset, mask, weight, bias, _one(stride, "stride"), _one(padding, "padding"),
                             _one(dilation, "dilation"), deformable_groups)


class _DeformBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(_DeformBase, self).__init__()
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size)
                                   .contiguous(memory_format=CL))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        n = in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)  # deform_conv_module.py:48-54


class DeformConv(_DeformBase):
    def __init__(self, *args, **kwargs):
        assert not kwargs.get("bias", False), "DeformConv has no bias (deform_conv_module.py:24)"
        super(DeformConv, self).__init__(*args, **kwargs)

    def forward(self, input, offset):
        return deform_conv(input, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class ModulatedDeformConv(_DeformBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                  groups, deformable_groups, bias)

    def forward(self, input, offset, mask):
        return modulated_deform_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """DCNv2 with its own offset/mask-generating conv (deform_conv_module.py:140-176)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConvPack, self).__init__(in_channels, out_channels, kernel_size, stride, padding,
                                                      dilation, groups, deformable_groups, bias)
        k = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = Conv2d(in_channels, deformable_groups * 3 * k, kernel_size=self.kernel_size,
                                       stride=_pair(stride), padding=_pair(padding), bias=True)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, input):
        out = self.conv_offset_mask(input)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return modulated_deform_conv(input, offset, torch.sigmoid(mask), self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups, self.deformable_groups)


class DFConv2d(nn.Module):
    """offset-predicting conv + (modulated) deformable conv (vendored layers/misc.py:114-203); the hard-coded
    [:18] / [-9:] channel slices of the reference (3x3, one deformable group) are generalised to k*k*dg"""

    def __init__(self, in_channels, out_channels, with_modulated_dcn=True, kernel_size=3, stride=1, groups=1,
                 dilation=1, deformable_groups=1, bias=False):
        super(DFConv2d, self).__init__()
        k = _one(kernel_size, "kernel_size")
        padding = dilation * (k - 1) // 2
        self.k2 = k * k * deformable_groups
        self.offset = Conv2d(in_channels, self.k2 * (3 if with_modulated_dcn else 2), kernel_size=k, stride=stride,
                             padding=padding)
        if dilation != 1:
            raise NotImplementedError("DFConv2d: dilation > 1 needs a dilated offset conv")
        nn.init.kaiming_uniform_(self.offset.weight, a=1)
        nn.init.constant_(self.offset.bias, 0.0)
        block = ModulatedDeformConv if with_modulated_dcn else DeformConv
        self.conv = block(in_channels, out_channels, kernel_size=k, stride=stride, padding=padding, dilation=dilation,
                          groups=groups, deformable_groups=deformable_groups, bias=bias)
        self.with_modulated_dcn = with_modulated_dcn

    def forward(self, x):
        if x.numel() == 0:
            raise NotImplementedError("DFConv2d on an empty batch")
        om = self.offset(x)
        if not self.with_modulated_dcn:
            return self.conv(x, om)
        offset = om[:, : 2 * self.k2, :, :]
        mask = om[:, -self.k2:, :, :].sigmoid()
        return self.conv(x, offset, mask)
