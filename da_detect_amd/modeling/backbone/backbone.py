"""Backbone builders keyed by MODEL.BACKBONE.CONV_BODY (reference: maskrcnn_benchmark/modeling/backbone/backbone.py)."""
from collections import OrderedDict

from torch import nn

from .. import registry
from . import resnet


@registry.BACKBONES.register("R-50-C4")
@registry.BACKBONES.register("R-50-C5")
@registry.BACKBONES.register("R-101-C4")
@registry.BACKBONES.register("R-101-C5")
def build_resnet_backbone(cfg):
    return nn.Sequential(OrderedDict([("body", resnet.ResNet(cfg))]))


def build_backbone(cfg):
    body = cfg.MODEL.BACKBONE.CONV_BODY
    assert body in registry.BACKBONES, \
        "cfg.MODEL.BACKBONE.CONV_BODY: {} are not registered in registry".format(body)
    return registry.BACKBONES[body](cfg)
