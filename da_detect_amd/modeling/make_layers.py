"""Layer factories (reference: maskrcnn_benchmark/modeling/make_layers.py:43-125).  GroupNorm variants are not on
the DA Faster R-CNN path (every shipped yaml has USE_GN False) and raise."""
import torch
from torch import nn

from ..layers import Conv2d, linear


class Linear(nn.Linear):
    """nn.Linear (same parameter names) whose contraction runs on the implicit-GEMM kernel; `relu` fuses the
    activation into the epilogue."""

    def forward(self, x, relu=False):
        return linear(x, self.weight, self.bias, relu=relu)


def _no_gn(use_gn):
    if use_gn:
        raise NotImplementedError("GroupNorm layers are outside the DA Faster R-CNN path")


def make_conv3x3(in_channels, out_channels, dilation=1, stride=1, use_gn=False, use_relu=False, kaiming_init=True):
    _no_gn(use_gn)
    conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=dilation, dilation=dilation)
    if kaiming_init:
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    else:
        torch.nn.init.normal_(conv.weight, std=0.01)
    nn.init.constant_(conv.bias, 0)
    return nn.Sequential(conv, nn.ReLU(inplace=True)) if use_relu else conv


def make_fc(dim_in, hidden_dim, use_gn=False):
    """XavierFill of Caffe2 == kaiming_uniform_(a=1) (make_layers.py:83-95)"""
    _no_gn(use_gn)
    fc = Linear(dim_in, hidden_dim)
    nn.init.kaiming_uniform_(fc.weight, a=1)
    nn.init.constant_(fc.bias, 0)
    return fc


def conv_with_kaiming_uniform(use_gn=False, use_relu=False):
    _no_gn(use_gn)

    def make_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1):
        conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                      padding=dilation * (kernel_size - 1) // 2, dilation=dilation)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        nn.init.constant_(conv.bias, 0)
        return nn.Sequential(conv, nn.ReLU(inplace=True)) if use_relu else conv

    return make_conv
