"""Box predictors: class scores and per-class box deltas from the ROI head's features
(reference behaviour: maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_predictors.py:7-58).

Both predictors are "two linear layers on one vector per ROI".  Parameter names (`cls_score`, `bbox_pred`) and
initialisers (normal 0.01 / 0.001, zero bias) are the reference's, so its checkpoints load by name; the two layers
are evaluated as ONE GEMM of width classes + 4 * box classes (layers.conv1x1_multi)."""
from torch import nn

from ....layers import conv1x1_multi, global_avg_pool
from ... import registry


class _TwoLinearPredictor(nn.Module):
    def _build(self, cfg, in_features):
        classes = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        box_classes = 2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else classes
        self.cls_score = nn.Linear(in_features, classes)
        self.bbox_pred = nn.Linear(in_features, 4 * box_classes)
        for layer, std in ((self.cls_score, 0.01), (self.bbox_pred, 0.001)):
            nn.init.normal_(layer.weight, mean=0, std=std)
            nn.init.constant_(layer.bias, 0)

    def _predict(self, vectors):
        """vectors [R, F] -> (scores [R, classes], deltas [R, 4 * box classes])"""
        rows = vectors.shape[0]
        scores, deltas = conv1x1_multi(vectors.reshape(rows, -1, 1, 1),
                                       [self.cls_score.weight, self.bbox_pred.weight],
                                       [self.cls_score.bias, self.bbox_pred.bias])
        return scores.reshape(rows, -1), deltas.reshape(rows, -1)


@registry.ROI_BOX_PREDICTOR.register("FastRCNNPredictor")
class FastRCNNPredictor(_TwoLinearPredictor):
    """C4 head: the res5 output [R, 2048, 7, 7] is average-pooled first (roi_box_predictors.py:7-33)"""

    def __init__(self, config, pretrained=None):
        super(FastRCNNPredictor, self).__init__()
        self.avgpool = nn.AvgPool2d(kernel_size=7, stride=7)  # kept for the module tree; forward uses the HIP pool
        self._build(config, config.MODEL.RESNETS.RES2_OUT_CHANNELS * 8)

    def forward(self, x):
        if tuple(x.shape[2:]) != (7, 7):
            raise ValueError("FastRCNNPredictor expects 7x7 ROI features, got %s" % (tuple(x.shape),))
        return self._predict(global_avg_pool(x))


@registry.ROI_BOX_PREDICTOR.register("FPNPredictor")
class FPNPredictor(_TwoLinearPredictor):
    """FPN head: the MLP feature extractor already yields one vector per ROI (roi_box_predictors.py:36-58)"""

    def __init__(self, cfg):
        super(FPNPredictor, self).__init__()
        self._build(cfg, cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM)

    def forward(self, x):
        return self._predict(x)


def make_roi_box_predictor(cfg):
    return registry.ROI_BOX_PREDICTOR[cfg.MODEL.ROI_BOX_HEAD.PREDICTOR](cfg)
