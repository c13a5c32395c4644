"""Box predictors (reference: maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_predictors.py)."""
from torch import nn

from ....layers import conv1x1_multi, global_avg_pool
from ... import registry


@registry.ROI_BOX_PREDICTOR.register("FastRCNNPredictor")
class FastRCNNPredictor(nn.Module):
    """avgpool 7x7 -> Linear(2048, classes), Linear(2048, 4*classes) (roi_box_predictors.py:7-33); the two
    linears run as one GEMM."""

    def __init__(self, config, pretrained=None):
        super(FastRCNNPredictor, self).__init__()
        num_inputs = config.MODEL.RESNETS.RES2_OUT_CHANNELS * 8
        num_classes = config.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        self.avgpool = nn.AvgPool2d(kernel_size=7, stride=7)  # kept for the module tree; forward uses the HIP pool
        self.cls_score = nn.Linear(num_inputs, num_classes)
        num_bbox_reg_classes = 2 if config.MODEL.CLS_AGNOSTIC_BBOX_REG else num_classes
        self.bbox_pred = nn.Linear(num_inputs, num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, mean=0, std=0.01)
        nn.init.constant_(self.cls_score.bias, 0)
        nn.init.normal_(self.bbox_pred.weight, mean=0, std=0.001)
        nn.init.constant_(self.bbox_pred.bias, 0)

    def forward(self, x):
        assert x.shape[2] == 7 and x.shape[3] == 7, "FastRCNNPredictor expects 7x7 ROI features"
        v = global_avg_pool(x)
        cls, box = conv1x1_multi(v.view(v.shape[0], v.shape[1], 1, 1), [self.cls_score.weight, self.bbox_pred.weight],
                                 [self.cls_score.bias, self.bbox_pred.bias])
        return cls.reshape(cls.shape[0], -1), box.reshape(box.shape[0], -1)


@registry.ROI_BOX_PREDICTOR.register("FPNPredictor")
class FPNPredictor(nn.Module):
    """Linear(rep, classes), Linear(rep, 4*classes) on the MLP head's vector (roi_box_predictors.py:36-58), one GEMM"""

    def __init__(self, cfg):
        super(FPNPredictor, self).__init__()
        num_classes = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        representation_size = cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
        self.cls_score = nn.Linear(representation_size, num_classes)
        num_bbox_reg_classes = 2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else num_classes
        self.bbox_pred = nn.Linear(representation_size, num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)

    def forward(self, x):
        cls, box = conv1x1_multi(x.reshape(x.shape[0], x.shape[1], 1, 1),
                                 [self.cls_score.weight, self.bbox_pred.weight],
                                 [self.cls_score.bias, self.bbox_pred.bias])
        return cls.reshape(cls.shape[0], -1), box.reshape(box.shape[0], -1)


def make_roi_box_predictor(cfg):
    return registry.ROI_BOX_PREDICTOR[cfg.MODEL.ROI_BOX_HEAD.PREDICTOR](cfg)
