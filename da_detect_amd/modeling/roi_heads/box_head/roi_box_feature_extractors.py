"""ROI feature extractors (reference: maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_feature_extractors.py)."""
from torch import nn

from ... import registry
from ...backbone import resnet
from ...poolers import Pooler


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("ResNet50Conv5ROIFeatureExtractor")
class ResNet50Conv5ROIFeatureExtractor(nn.Module):
    """ROIAlign (14x14) -> res5 (3 bottlenecks, first 1x1 has stride 2) -> [R, 2048, 7, 7]
    (roi_box_feature_extractors.py:13-45).  The FLOP majority of the training step."""

    def __init__(self, config):
        super(ResNet50Conv5ROIFeatureExtractor, self).__init__()
        resolution = config.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler(output_size=(resolution, resolution), scales=config.MODEL.ROI_BOX_HEAD.POOLER_SCALES,
                             sampling_ratio=config.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO)
        stage = resnet.StageSpec(index=4, block_count=3, return_features=False)
        self.head = resnet.ResNetHead(
            block_module=config.MODEL.RESNETS.TRANS_FUNC, stages=(stage,),
            num_groups=config.MODEL.RESNETS.NUM_GROUPS, width_per_group=config.MODEL.RESNETS.WIDTH_PER_GROUP,
            stride_in_1x1=config.MODEL.RESNETS.STRIDE_IN_1X1, stride_init=None,
            res2_out_channels=config.MODEL.RESNETS.RES2_OUT_CHANNELS, dilation=config.MODEL.RESNETS.RES5_DILATION)

    def forward(self, x, proposals):
        return self.head(self.pooler(x, proposals))


def make_roi_box_feature_extractor(cfg):
    return registry.ROI_BOX_FEATURE_EXTRACTORS[cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR](cfg)
