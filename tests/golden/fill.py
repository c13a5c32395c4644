"""Deterministic, construction-order-independent parameter fill shared by the golden-vector generator (which
applies it to the imported reference model) and the tests (which apply it to this repo's model / the oracle):
every state_dict entry is drawn from its own generator seeded by a hash of the key name."""
import hashlib

import torch


def _gen(key, seed):
    h = int(hashlib.sha256(("%s|%d" % (key, seed)).encode()).hexdigest()[:12], 16)
    return torch.Generator().manual_seed(h)


def fill_state_dict(sd, seed=0):
    """returns a new dict key -> CPU fp32 tensor with the same shapes as `sd`"""
    out = {}
    for k, v in sd.items():
        g = _gen(k, seed)
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.05
        elif ".bn" in k or "downsample.1" in k:
            if k.endswith("weight"):
                t = torch.rand(shape, generator=g) * 0.3 + (0.15 if (".bn3." in k) else 0.85)
            else:
                t = torch.randn(shape, generator=g) * 0.05
        elif "cell_anchors" in k:
            t = v.detach().clone().float()
        elif k.endswith("bias"):
            t = torch.randn(shape, generator=g) * 0.02
            if "cls_logits" in k:
                t = t - 3.0  # negative objectness logits: finer fp32 spacing of sigmoid -> no tied scores
        else:  # conv / linear weights
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            std = (2.0 / fan_in) ** 0.5
            if "stem.conv1" in k:
                std = std / 60.0
            if "fpn_inner" in k:
                std = std * 0.1  # keeps the pyramid O(1): saturated RPN sigmoids would tie at exactly 1.0
            if "predictor" in k or "rpn.head.cls_logits" in k or "rpn.head.bbox_pred" in k or "_da" in k:
                std = std * 0.5
            t = torch.randn(shape, generator=g) * std
        out[k] = t.to(torch.float32)
    return out
