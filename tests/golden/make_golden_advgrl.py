"""Golden vectors for the ACTIVE AdvGRL branch — runs ONLY in the authoring container (needs /root/reference).

Adv_GRL (reference da_heads.py:173-195) replaces the fixed reversal weight -GRL_WEIGHT by
-advGRL_WEIGHT * min(DA_ADV_GRL_THRESHOLD, 1 / loss) once the current domain-classifier loss is <= BCE([.7,.3],[1,0])
= 0.6288.  A randomly initialised classifier sits at ~0.69, so the branch is dormant in every other fixture.  Here the
domain classifiers are made GOOD: the target / auxiliary images are darkened (so their C4 / ROI features differ
systematically from the source's) and the two heads get structured weights (tests/golden/fill.py
`structured_da_heads`) that project on the source-minus-target mean-feature direction measured in a first pass.

The gradient-reversal weight only acts in backward, so the fixture stores GRADIENTS: d loss_da_image / d C4 features
and d loss_da_instance / d ROI features (the reference cannot back-propagate further on the CPU: ROIAlign backward is
not implemented there, csrc/ROIAlign.h:44), next to the losses and the effective weights.

Two documented adaptations of the reference call (SURVEY.md fact 10 — as written the branch raises): the loss handed to
Adv_GRL is detached (the instance branch's requires grad and `.numpy()` refuses it), and the threshold is a 0-d tensor
(`min(30, tensor)` returns the int, which has no `.numpy()`).  Nothing of the reference is stored.

    python tests/golden/make_golden_advgrl.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shims  # noqa: E402

ref_shims.install()
from fill import fill_state_dict, structured_da_heads  # noqa: E402
from maskrcnn_benchmark.config import cfg as ref_cfg  # noqa: E402
from maskrcnn_benchmark.modeling.detector import build_detection_model as ref_build  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList as RefBoxList  # noqa: E402
from maskrcnn_benchmark.structures.image_list import to_image_list as ref_to_image_list  # noqa: E402

from da_detect_amd.config import cfg as my_cfg  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from oracle import model_ref  # noqa: E402

YAML = "/root/reference/configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
H, W, NIMG, SEED = 160, 288, 3, 0
from cases import ADVGRL_CASES as CASES, ADVGRL_IMG_SCALE as IMG_SCALE  # noqa: E402


def scaled_batch(mc):
    images, targets = make_batch(mc, NIMG, H, W, seed=SEED, device=torch.device("cpu"))
    for i, s in enumerate(IMG_SCALE):
        images.tensors[i] *= s
    return images, targets


def run_reference(overrides, weights_fn):
    c = ref_cfg.clone()
    c.merge_from_file(YAML)
    c.merge_from_list(["MODEL.DEVICE", "cpu"] + list(overrides))
    model = ref_build(c)
    weights = weights_fn(fill_state_dict(model.state_dict(), SEED))
    model.load_state_dict(weights)
    model.train()
    mc = my_cfg.clone()
    mc.merge_from_file(YAML)
    mc.merge_from_list(list(overrides))
    images, targets = scaled_batch(mc)
    ref_targets = []
    for t in targets:
        b = RefBoxList(t.bbox.clone(), t.size, mode="xyxy")
        b.add_field("labels", t.get_field("labels").clone())
        b.add_field("is_source", t.get_field("is_source").clone())
        ref_targets.append(b)
    inter = {"weights_used": [], "trainable": [n for n, p in model.named_parameters() if p.requires_grad]}
    model.backbone.register_forward_hook(lambda m, i, o: inter.__setitem__("feat", o[0]))
    model.rpn.head.register_forward_hook(
        lambda m, i, o: inter.update(objectness=o[0][0].detach().clone(), deltas=o[1][0].detach().clone()))
    da = model.da_heads_triplet
    da.register_forward_pre_hook(lambda m, args: inter.update(ins_feat=args[1], ins_labels=args[2].clone()))
    da.advGRL_threshold = torch.tensor(float(da.advGRL_threshold))          # adaptation 2 (see the module docstring)
    orig = da.Adv_GRL

    def adv_grl(loss_iter, feats, list_option=True):
        out = orig(loss_iter.detach(), feats, list_option)                   # adaptation 1
        active = bool(loss_iter.detach() <= da.bce)
        w = float(da.advGRL_optimized.weight) if active else float((da.grl_img if list_option else da.grl_ins).weight)
        inter["weights_used"].append(("img" if list_option else "ins", float(loss_iter.detach()), active, w))
        return out

    da.Adv_GRL = adv_grl
    torch.manual_seed(SEED)
    losses = model(ref_to_image_list(images.tensors), ref_targets)
    return mc, weights, images, targets, losses, inter


def main():
    # pass 1: plain fill -> mean-feature directions of the two heads
    _, _, _, _, _, inter = run_reference([], lambda sd: sd)
    feat = inter["feat"].detach()
    mu_s, mu_t = feat[0].mean(dim=(1, 2)), feat[1].mean(dim=(1, 2))
    pooled = inter["ins_feat"].detach().mean(dim=(2, 3))
    lab = inter["ins_labels"].bool()
    nu_s, nu_t = pooled[lab].mean(0), pooled[~lab].mean(0)

    def direction(a, b):
        d = a - b
        d = d / float(d.dot(d))
        return d, float(d.dot(0.5 * (a + b)))

    dir_img, off_img = direction(mu_s, mu_t)
    dir_ins, off_ins = direction(nu_s, nu_t)
    out = {"seed": np.int64(SEED), "H": np.int64(H), "W": np.int64(W), "nimg": np.int64(NIMG),
           "img_scale": np.asarray(IMG_SCALE, np.float32), "dir_img": dir_img.numpy(), "off_img": np.float32(off_img),
           "dir_ins": dir_ins.numpy(), "off_ins": np.float32(off_ins)}
    for name, (overrides, gain_img, gain_ins) in CASES.items():
        shape = lambda sd: structured_da_heads(sd, "da_heads_triplet", dir_img, off_img, dir_ins, off_ins,  # noqa: E731
                                               gain_img, gain_ins)
        mc, weights, images, targets, losses, inter = run_reference(overrides, shape)
        g_feat, = torch.autograd.grad(losses["loss_da_image"], inter["feat"], retain_graph=True)
        g_ins, = torch.autograd.grad(losses["loss_da_instance"], inter["ins_feat"], retain_graph=True)
        print(name, {k: round(float(v), 5) for k, v in losses.items()})
        print("   Adv_GRL calls (branch, current loss, active, weight):", inter["weights_used"])
        # pin the oracle on the same case: losses and both gradients
        sd = {k: v.clone() for k, v in weights.items()}
        for n in inter["trainable"]:
            sd[n].requires_grad_(True)
        o_inter = {}
        torch.manual_seed(SEED)
        o_losses = model_ref.training_losses(sd, mc, images.tensors, model_ref.targets_to_dicts(targets), state={},
                                             intermediates=o_inter, grad_probe=True)
        for k in losses:
            r, o = float(losses[k]), float(o_losses[k])
            assert abs(r - o) <= 1e-5 * max(abs(r), 1.0), (name, k, r, o)
        og_feat, = torch.autograd.grad(o_losses["loss_da_image"], o_inter["feat_graph"], retain_graph=True)
        og_ins, = torch.autograd.grad(o_losses["loss_da_instance"], o_inter["ins_feat_graph"], retain_graph=True)
        for tag, a, b in (("feat", g_feat, og_feat), ("ins", g_ins, og_ins)):
            err = float((a - b).abs().max()) / float(a.abs().max())
            print("   oracle gradient (%s) max rel err %.2e" % (tag, err))
            assert err < 1e-4, (name, tag, err)
        for k, v in losses.items():
            out["%s/loss/%s" % (name, k)] = v.detach().numpy()
        for branch, cur, active, w in inter["weights_used"]:
            out["%s/weight_%s" % (name, branch)] = np.float32(w)
            out["%s/current_%s" % (name, branch)] = np.float32(cur)
            out["%s/active_%s" % (name, branch)] = np.bool_(active)
        assert float(g_feat[2].abs().max()) == 0.0                          # the auxiliary image is not in the DA loss
        out["%s/g_feat" % name] = g_feat[:2, ::16].numpy()                 # every 16th channel of [2,1024,Hf,Wf]
        out["%s/g_feat_absmax" % name] = g_feat.abs().max().numpy()
        assert torch.equal(g_ins[:, :, :1, :1].expand_as(g_ins), g_ins)
        out["%s/g_ins" % name] = g_ins[:, ::64, 0, 0].numpy()              # avg-pool: uniform over the 7x7 window
        out["%s/g_ins_absmax" % name] = g_ins.abs().max().numpy()
        if name == "active":
            out["objectness"], out["deltas"] = inter["objectness"].numpy(), inter["deltas"].numpy()
    path = os.path.join(HERE, "advgrl.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
