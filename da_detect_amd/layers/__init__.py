"""Operator API of the reference (`maskrcnn_benchmark.layers`, reference layers/__init__.py:4-22) served by the
HIP library."""
from .. import _C
from .conv import conv2d_affine_act, linear, conv1x1_multi
from .misc import (Conv2d, FrozenBatchNorm2d, GradientScalarLayer, SigmoidFocalLoss, consistency_loss,
                   global_avg_pool, smooth_l1_loss)
from .roi_align import ROIAlign, roi_align
from .roi_pool import ROIPool, roi_pool

nms = _C.nms

__all__ = ["nms", "roi_align", "ROIAlign", "roi_pool", "ROIPool", "smooth_l1_loss", "Conv2d", "FrozenBatchNorm2d",
           "SigmoidFocalLoss", "GradientScalarLayer", "consistency_loss", "conv2d_affine_act", "linear",
           "conv1x1_multi", "global_avg_pool"]
