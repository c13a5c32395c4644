"""ROI feature extractors (reference: maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_feature_extractors.py)."""
import os

from torch import nn

from ....layers import conv2d_affine_act
from ... import registry
from ...backbone import resnet
from ...make_layers import make_fc
from ...poolers import Pooler


_SUBGRID = os.environ.get("DADET_ROI_SUBGRID", "1") == "1"
_BWD_LIST = os.environ.get("DADET_ROI_BWD_LIST", "1")[:1] != "0"      # csrc/roi_align.hip reads the same switch


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
        # With STRIDE_IN_1X1 the head's first block reads the pooled 14 x 14 grid through two stride-2 1x1 convolutions
        # (conv1 and the shortcut, resnet.py:236-262): bins (2i, 2j) only, and zeros flow back into the other three
        # quarters.  Those bins are pooled alone, into a 7 x 7 grid, and the block runs with stride 1 on it: the same
        # values (bit for bit) from a quarter of the ROIAlign work, forward and backward.  DADET_ROI_SUBGRID=0: the
        # reference's full grid.
        # (the sub-grid backward is the list kernel's: channels in 16-byte groups, pooled grid of at most 14 x 14,
        # DADET_ROI_BWD_LIST on — a configuration it does not cover pools the full grid instead of failing mid-backward)
        stride = self.head.input_bin_stride() if (_SUBGRID and x[0].is_cuda and len(self.pooler.poolers) == 1
                                                  and sum(len(p) for p in proposals) > 0
                                                  and x[0].shape[1] % 4 == 0 and max(self.pooler.output_size) <= 14
                                                  and _BWD_LIST) else 1
        if stride > 1:
            return self.head(self.pooler(x, proposals, bin_stride=stride), first_stride=1)
        return self.head(self.pooler(x, proposals))


@registry.ROI_BOX_FEATURE_EXTRACTORS.register("FPN2MLPFeatureExtractor")
class FPN2MLPFeatureExtractor(nn.Module):
    """per-level ROIAlign (7x7) -> fc6 -> ReLU -> fc7 -> ReLU (roi_box_feature_extractors.py:48-79).

    fc6 consumes the pooled tensor flattened in (c, h, w) order.  The pooled ROIs live as NHWC here, so instead of
    transposing every ROI, fc6 runs as a 7x7 VALID convolution with its weight viewed as [rep, C, 7, 7] — the same
    contraction, summed over (h, w, c) in the layout the data already has."""

    def __init__(self, cfg):
        super(FPN2MLPFeatureExtractor, self).__init__()
        resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler(output_size=(resolution, resolution), scales=cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES,
                             sampling_ratio=cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO)
        self.resolution, self.channels = resolution, cfg.MODEL.BACKBONE.OUT_CHANNELS
        input_size = self.channels * resolution ** 2
        representation_size = cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
        use_gn = cfg.MODEL.ROI_BOX_HEAD.USE_GN
        self.fc6 = make_fc(input_size, representation_size, use_gn)
        self.fc7 = make_fc(representation_size, representation_size, use_gn)

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        w6 = self.fc6.weight.view(-1, self.channels, self.resolution, self.resolution)
        x = conv2d_affine_act(x, w6, None, self.fc6.bias, relu=True)
        return self.fc7(x.reshape(x.shape[0], -1), relu=True)


def make_roi_box_feature_extractor(cfg):
    return registry.ROI_BOX_FEATURE_EXTRACTORS[cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR](cfg)
