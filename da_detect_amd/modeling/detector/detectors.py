from .generalized_rcnn import GeneralizedRCNN

_DETECTION_META_ARCHITECTURES = {"GeneralizedRCNN": GeneralizedRCNN}


def build_detection_model(cfg):
    """reference: maskrcnn_benchmark/modeling/detector/detectors.py:5-10"""
    return _DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg)
