"""Caffe2 / Detectron `.pkl` weights -> this model's parameter names (reference:
maskrcnn_benchmark/utils/c2_model_loading.py:11-176).  The MSRA ImageNet ResNets that every DA yaml starts from
(`MODEL.WEIGHT: catalog://ImageNetPretrained/MSRA/R-50`) come in that format.  The renaming is an ORDERED list of
substring rewrites (order matters: e.g. "_" -> "." first, then ".w" -> ".weight"); it is kept here as data."""
import logging
import pickle
from collections import OrderedDict

import torch

_BASIC_RULES = [
    # Caffe2 blob-name punctuation -> module paths, parameter suffixes
    ("_", "."), (".w", ".weight"), (".bn", "_bn"), (".b", ".bias"), ("_bn.s", "_bn.scale"),
    (".biasranch", ".branch"), ("bbox.pred", "bbox_pred"), ("cls.score", "cls_score"), ("res.conv1_", "conv1_"),
    # RPN / Faster R-CNN heads
    (".biasbox", ".bbox"), ("conv.rpn", "rpn.conv"), ("rpn.bbox.pred", "rpn.bbox_pred"),
    ("rpn.cls.logits", "rpn.cls_logits"),
    # AffineChannel -> (frozen) BatchNorm
    ("_bn.scale", "_bn.weight"),
    # torchvision-style stage / layer names
    ("conv1_bn.", "bn1."), ("res2.", "layer1."), ("res3.", "layer2."), ("res4.", "layer3."), ("res5.", "layer4."),
    (".branch2a.", ".conv1."), (".branch2a_bn.", ".bn1."), (".branch2b.", ".conv2."), (".branch2b_bn.", ".bn2."),
    (".branch2c.", ".conv3."), (".branch2c_bn.", ".bn3."), (".branch1.", ".downsample.0."),
    (".branch1_bn.", ".downsample.1."),
    # GroupNorm variants
    ("conv1.gn.s", "bn1.weight"), ("conv1.gn.bias", "bn1.bias"), ("conv2.gn.s", "bn2.weight"),
    ("conv2.gn.bias", "bn2.bias"), ("conv3.gn.s", "bn3.weight"), ("conv3.gn.bias", "bn3.bias"),
    ("downsample.0.gn.s", "downsample.1.weight"), ("downsample.0.gn.bias", "downsample.1.bias"),
]
_HEAD_RULES = [
    ("mask.fcn.logits", "mask_fcn_logits"), (".[mask].fcn", "mask_fcn"), ("conv5.mask", "conv5_mask"),
    ("kps.score.lowres", "kps_score_lowres"), ("kps.score", "kps_score"), ("conv.fcn", "conv_fcn"),
    ("rpn.", "rpn.head."),       # this package's RPN keeps its convs under `head`
]
# last block of each stage in the Caffe2 graph ("res<stage>_<block>"), per architecture
_C2_STAGE_NAMES = {"R-50": ["1.2", "2.3", "3.5", "4.2"], "R-101": ["1.2", "2.3", "3.22", "4.2"],
                   "R-152": ["1.2", "2.7", "3.35", "4.2"]}


def _fpn_rules(stage_names):
    rules = []
    for idx, stage in enumerate(stage_names, 1):
        lateral = ".lateral" if idx < 4 else ""
        rules.append(("fpn.inner.layer{}.sum{}".format(stage, lateral), "fpn_inner{}".format(idx)))
        rules.append(("fpn.layer{}.sum".format(stage), "fpn_layer{}".format(idx)))
    rules += [("rpn.conv.fpn2", "rpn.conv"), ("rpn.bbox_pred.fpn2", "rpn.bbox_pred"),
              ("rpn.cls_logits.fpn2", "rpn.cls_logits")]
    return rules


def rename_c2_keys(keys, stage_names):
    """ordered list of Caffe2 blob names -> list of parameter names (same order)"""
    out = ["fc1000_b" if k == "pred_b" else "fc1000_w" if k == "pred_w" else k for k in keys]
    for old, new in _BASIC_RULES + _fpn_rules(stage_names) + _HEAD_RULES:
        out = [k.replace(old, new) for k in out]
    return out


def _rename_weights_for_resnet(weights, stage_names):
    original = sorted(weights.keys())
    mapped = dict(zip(original, rename_c2_keys(original, stage_names)))
    logger = logging.getLogger(__name__)
    out = OrderedDict()
    for k in original:
        if "_momentum" in k:
            continue
        logger.info("C2 name: %s mapped name: %s", k, mapped[k])
        out[mapped[k]] = torch.from_numpy(weights[k])
    return out


def _load_c2_pickled_weights(file_path):
    with open(file_path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    return data["blobs"] if "blobs" in data else data


def rename_for_deformable_convs(state_dict, stage_with_dcn):
    """stages built with deformable 3x3 convs keep that conv under `conv2.conv` (DFConv2d) — vendored reference
    tools/cityscapes/maskrcnn_benchmark/utils/c2_model_loading.py:146-170.  Without this the pretrained 3x3 weights of
    those stages match no parameter and silently keep their random init."""
    out = OrderedDict()
    for key, value in state_dict.items():
        for ix, with_dcn in enumerate(stage_with_dcn, 1):
            if with_dcn and ("layer%d." % ix) in key and ".conv2." in key and ".conv2.conv." not in key \
                    and key.rsplit(".", 1)[-1] in ("weight", "bias"):
                key = key.replace(".conv2.", ".conv2.conv.")
                break
        out[key] = value
    return out


def load_c2_format(cfg, f):
    body = cfg.MODEL.BACKBONE.CONV_BODY
    arch = body.replace("-C4", "").replace("-C5", "").replace("-FPN", "").replace("-RETINANET", "")
    if arch not in _C2_STAGE_NAMES:
        raise KeyError("no Caffe2 weight mapping for CONV_BODY {}".format(body))
    state = _rename_weights_for_resnet(_load_c2_pickled_weights(f), _C2_STAGE_NAMES[arch])
    return dict(model=rename_for_deformable_convs(state, cfg.MODEL.RESNETS.STAGE_WITH_DCN))
