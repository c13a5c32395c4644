"""Deformable convolution / pooling layers (operator API of the reference's vendored tree:
tools/cityscapes/maskrcnn_benchmark/layers/dcn/{deform_conv_func,deform_conv_module,deform_pool_func,
deform_pool_module}.py, layers/misc.py:114-203)."""
from .deform_conv import (DeformConv, DFConv2d, ModulatedDeformConv, ModulatedDeformConvPack, deform_conv,
                          modulated_deform_conv)
from .deform_pool import (DeformRoIPooling, DeformRoIPoolingFunction, DeformRoIPoolingPack,
                          ModulatedDeformRoIPoolingPack, deform_roi_pooling)

__all__ = ["deform_conv", "modulated_deform_conv", "DeformConv", "ModulatedDeformConv", "ModulatedDeformConvPack",
           "DFConv2d", "deform_roi_pooling", "DeformRoIPoolingFunction", "DeformRoIPooling", "DeformRoIPoolingPack",
           "ModulatedDeformRoIPoolingPack"]
