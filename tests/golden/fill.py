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


def structured_da_heads(sd, prefix, dir_img, off_img, dir_ins, off_ins, gain_img, gain_ins, blend=0.25):
    """Domain classifiers that actually separate the two domains (AdvGRL fixtures, make_golden_advgrl.py): each head
    projects its input on a given direction `d` (x -> z = d.x - off, positive for source-like inputs) with half of its
    hidden units computing relu(+z) and the other half relu(-z), and reads them out as gain * z; `blend` of the random
    fill stays on top so that the weights are not exactly rank one.  gain 0 leaves a head as filled.
    Image head `prefix.imghead.conv{1,2}_da` (1x1 convs C->512->1), instance head `prefix.inshead.fc{1,2,3}_da`
    (2048->1024->1024->1, a dropout of 0.5 after each hidden layer)."""
    out = {k: v.clone() for k, v in sd.items()}
    dir_img = torch.as_tensor(dir_img, dtype=torch.float32)
    dir_ins = torch.as_tensor(dir_ins, dtype=torch.float32)
    if gain_img:
        p = prefix + ".imghead."
        w1 = out[p + "conv1_da.weight"] * blend
        half = w1.shape[0] // 2
        w1[:half, :, 0, 0] += dir_img
        w1[half:, :, 0, 0] -= dir_img
        b1 = out[p + "conv1_da.bias"] * blend
        b1[:half] -= float(off_img)
        b1[half:] += float(off_img)
        w2 = out[p + "conv2_da.weight"] * blend
        w2[0, :half, 0, 0] += float(gain_img) / half
        w2[0, half:, 0, 0] -= float(gain_img) / half
        out[p + "conv1_da.weight"], out[p + "conv1_da.bias"], out[p + "conv2_da.weight"] = w1, b1, w2
    if gain_ins:
        p = prefix + ".inshead."
        w1 = out[p + "fc1_da.weight"] * blend
        half = w1.shape[0] // 2
        w1[:half] += dir_ins
        w1[half:] -= dir_ins
        b1 = out[p + "fc1_da.bias"] * blend
        b1[:half] -= float(off_ins)
        b1[half:] += float(off_ins)
        w2 = out[p + "fc2_da.weight"] * blend + torch.eye(w1.shape[0])
        w3 = out[p + "fc3_da.weight"] * blend
        w3[0, :half] += float(gain_ins) / half
        w3[0, half:] -= float(gain_ins) / half
        out[p + "fc1_da.weight"], out[p + "fc1_da.bias"] = w1, b1
        out[p + "fc2_da.weight"], out[p + "fc3_da.weight"] = w2, w3
    return out
