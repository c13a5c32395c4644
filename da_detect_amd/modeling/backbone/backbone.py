"""Backbone builders keyed by MODEL.BACKBONE.CONV_BODY (reference: maskrcnn_benchmark/modeling/backbone/backbone.py)."""
from collections import OrderedDict

from torch import nn

from .. import registry
from ..make_layers import conv_with_kaiming_uniform
from . import fpn as fpn_module
from . import resnet


@registry.BACKBONES.register("R-50-C4")
@registry.BACKBONES.register("R-50-C5")
@registry.BACKBONES.register("R-101-C4")
@registry.BACKBONES.register("R-101-C5")
def build_resnet_backbone(cfg):
    return nn.Sequential(OrderedDict([("body", resnet.ResNet(cfg))]))


@registry.BACKBONES.register("R-50-FPN")
@registry.BACKBONES.register("R-101-FPN")
@registry.BACKBONES.register("R-152-FPN")
def build_resnet_fpn_backbone(cfg):
    """backbone.py:21-42: C2..C5 -> P2..P5 (+ P6 by stride-2 subsampling)"""
    body = resnet.ResNet(cfg)
    c2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    fpn = fpn_module.FPN(in_channels_list=[c2, c2 * 2, c2 * 4, c2 * 8], out_channels=cfg.MODEL.BACKBONE.OUT_CHANNELS,
                         conv_block=conv_with_kaiming_uniform(cfg.MODEL.FPN.USE_GN, cfg.MODEL.FPN.USE_RELU),
                         top_blocks=fpn_module.LastLevelMaxPool())
    return nn.Sequential(OrderedDict([("body", body), ("fpn", fpn)]))


def build_backbone(cfg):
    body = cfg.MODEL.BACKBONE.CONV_BODY
    assert body in registry.BACKBONES, \
        "cfg.MODEL.BACKBONE.CONV_BODY: {} are not registered in registry".format(body)
    return registry.BACKBONES[body](cfg)
