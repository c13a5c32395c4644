"""Feature pyramid (reference: maskrcnn_benchmark/modeling/backbone/fpn.py:7-99) on the implicit-GEMM kernels.

Same module tree / parameter names (fpn_inner{i}, fpn_layer{i}).  The lateral 1x1 conv, its bias and the top-down
addition are ONE kernel launch: the nearest-upsampled coarser map is the GEMM epilogue's addend."""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers import Conv2d


class FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels, conv_block, top_blocks=None):
        super(FPN, self).__init__()
        self.inner_blocks, self.layer_blocks = [], []
        for idx, in_channels in enumerate(in_channels_list, 1):
            if in_channels == 0:
                continue
            inner, layer = "fpn_inner{}".format(idx), "fpn_layer{}".format(idx)
            self.add_module(inner, conv_block(in_channels, out_channels, 1))
            self.add_module(layer, conv_block(out_channels, out_channels, 3, 1))
            self.inner_blocks.append(inner)
            self.layer_blocks.append(layer)
        self.top_blocks = top_blocks

    @staticmethod
    def _lateral(block, feature, top_down):
        if isinstance(block, Conv2d):
            return block(feature, residual=top_down)       # conv + bias + top_down in the epilogue
        return block(feature) + top_down                   # conv_block with an activation (FPN.USE_RELU)

    def forward(self, x):
        """x: list of maps, increasing depth -> tuple of pyramid maps, highest resolution first (fpn.py:43-74)"""
        last_inner = getattr(self, self.inner_blocks[-1])(x[-1])
        results = [getattr(self, self.layer_blocks[-1])(last_inner)]
        for feature, inner, layer in zip(x[:-1][::-1], self.inner_blocks[:-1][::-1], self.layer_blocks[:-1][::-1]):
            top_down = F.interpolate(last_inner, scale_factor=2, mode="nearest")
            last_inner = self._lateral(getattr(self, inner), feature, top_down)
            results.insert(0, getattr(self, layer)(last_inner))
        if isinstance(self.top_blocks, LastLevelP6P7):
            results.extend(self.top_blocks(x[-1], results[-1]))
        elif isinstance(self.top_blocks, LastLevelMaxPool):
            results.extend(self.top_blocks(results[-1]))
        return tuple(results)


class LastLevelMaxPool(nn.Module):
    """max_pool2d(kernel 1, stride 2) == stride-2 subsampling (fpn.py:77-79)"""

    def forward(self, x):
        return [x[:, :, ::2, ::2].contiguous(memory_format=torch.channels_last)]


class LastLevelP6P7(nn.Module):
    """RetinaNet's extra levels (fpn.py:82-99); kept for the module tree of *-FPN-RETINANET bodies"""

    def __init__(self, in_channels, out_channels):
        super(LastLevelP6P7, self).__init__()
        raise NotImplementedError("RetinaNet is outside the DA Faster R-CNN path (SURVEY.md section 2.1)")
