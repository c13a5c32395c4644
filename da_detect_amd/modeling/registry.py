"""Model component registries (reference: maskrcnn_benchmark/modeling/registry.py:5-8)."""
from ..utils.registry import Registry

BACKBONES = Registry()
RPN_HEADS = Registry()
ROI_BOX_FEATURE_EXTRACTORS = Registry()
ROI_BOX_PREDICTOR = Registry()
