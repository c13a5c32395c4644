"""Golden vectors for the EVALUATION path — runs ONLY in the authoring container (needs /root/reference).

Imports the reference model on CPU (ref_shims), fills it with tests/golden/fill.py, runs `model.eval()` forward on a
seeded synthetic batch and stores the detections (boxes / scores / labels per image) together with the RPN maps,
the test-mode proposals and the box-head outputs.  The same inputs go through oracle/model_ref.inference and must
agree (that pins the oracle's eval path).  Also records the reference model's state_dict key -> shape table
(checkpoint-compatibility fixture).  Fixtures hold seeds, shapes and OUTPUT tensors only.

    python tests/golden/make_golden_eval.py
"""
import json
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
from fill import fill_state_dict  # noqa: E402
from maskrcnn_benchmark.config import cfg as ref_cfg  # noqa: E402
from maskrcnn_benchmark.modeling.detector import build_detection_model as ref_build  # noqa: E402
from maskrcnn_benchmark.structures.image_list import to_image_list as ref_to_image_list  # noqa: E402

from da_detect_amd.config import cfg as my_cfg  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from oracle import model_ref  # noqa: E402

REF_CFG_DIR = "/root/reference/configs/da_faster_rcnn"
YAML = "e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
STATE_DICT_CASES = {
    "da_plain": (YAML, []),
    "da_triplet": ("e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml", []),
    "fpn": ("../e2e_faster_rcnn_R_50_FPN_1x.yaml", []),
}


def run_case(name, yaml_path, oracle_fn, H=192, W=320, nimg=2, seed=0):
    c = ref_cfg.clone()
    c.merge_from_file(yaml_path)
    c.merge_from_list(["MODEL.DEVICE", "cpu"])
    model = ref_build(c)
    weights = fill_state_dict(model.state_dict(), seed)
    model.load_state_dict(weights)
    model.eval()
    mc = my_cfg.clone()
    mc.merge_from_file(yaml_path)
    images, _ = make_batch(mc, nimg, H, W, seed=seed, device=torch.device("cpu"))
    inter = {}
    model.rpn.head.register_forward_hook(
        lambda m, i, o: inter.update(objectness=[t.detach().clone() for t in o[0]],
                                     deltas=[t.detach().clone() for t in o[1]]))
    model.rpn.register_forward_hook(
        lambda m, i, o: inter.update(proposals=[(p.bbox.clone(), p.get_field("objectness").clone()) for p in o[0]]))
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: inter.update(class_logits=o[0].detach().clone(), box_regression=o[1].detach().clone()))
    with torch.no_grad():
        dets = model(ref_to_image_list(images.tensors))
    o_inter = {}
    o_dets = oracle_fn({k: v.clone() for k, v in weights.items()}, mc, images.tensors, o_inter)
    out = {"seed": np.int64(seed), "H": np.int64(H), "W": np.int64(W), "nimg": np.int64(nimg),
           "class_logits": inter["class_logits"].numpy()}
    rows = 8 if inter["box_regression"].numel() > 100000 else 1      # keep the fixture small: every 8th ROI
    out["box_regression_rows"] = np.int64(rows)
    out["box_regression"] = inter["box_regression"][::rows].numpy()
    if len(inter["objectness"]) == 1:
        out["objectness"], out["deltas"] = inter["objectness"][0].numpy(), inter["deltas"][0].numpy()
    else:
        for l, (o, d) in enumerate(zip(inter["objectness"], inter["deltas"])):
            out["objectness/%d" % l], out["deltas/%d" % l] = o.numpy(), d.numpy()
    assert torch.allclose(inter["class_logits"], o_inter["class_logits"], rtol=1e-4, atol=1e-5), "class logits"
    for i, (d, o) in enumerate(zip(dets, o_dets)):
        b, s, l = d.bbox, d.get_field("scores"), d.get_field("labels")
        rb, rs = inter["proposals"][i]
        ob, os_ = o_inter["proposals"][i]
        print("%s image %d: %d proposals, %d detections (oracle %d / %d)" % (name, i, len(rb), len(b), len(ob),
                                                                           len(o["boxes"])))
        assert rb.shape == ob.shape and torch.allclose(rb, ob, atol=1e-3), "proposals"
        assert len(b) == len(o["boxes"]) and torch.equal(l, o["labels"]), "detections"
        assert torch.allclose(b, o["boxes"], atol=2e-3) and torch.allclose(s, o["scores"], atol=1e-6), "detections"
        out["proposals/%d/boxes" % i], out["proposals/%d/objectness" % i] = rb.numpy(), rs.numpy()
        out["det/%d/boxes" % i], out["det/%d/scores" % i], out["det/%d/labels" % i] = b.numpy(), s.numpy(), l.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def run_untied(name, yaml_path, oracle_fn):
    """exactly tied objectness scores leave the reference's topk order unspecified -> take the first seed whose
    proposals come out in the oracle's (stable, ascending-index) order"""
    for seed in range(8):
        try:
            return run_case(name, yaml_path, oracle_fn, seed=seed)
        except AssertionError as e:
            if str(e) not in ("proposals", "class logits"):
                raise
            print("%s: seed %d has tied proposal scores (%s) -> next seed" % (name, seed, e))
    raise SystemExit("no tie-free seed for " + name)


def main():
    run_untied("eval_da_plain", os.path.join(REF_CFG_DIR, YAML), model_ref.inference)
    run_untied("eval_fpn", "/root/reference/configs/e2e_faster_rcnn_R_50_FPN_1x.yaml", model_ref.inference_fpn)

    table = {}
    for name, (yaml, overrides) in STATE_DICT_CASES.items():
        c = ref_cfg.clone()
        c.merge_from_file(os.path.join(REF_CFG_DIR, yaml))
        c.merge_from_list(["MODEL.DEVICE", "cpu"] + list(overrides))
        table[name] = {k: list(v.shape) for k, v in ref_build(c).state_dict().items()}
    with open(os.path.join(HERE, "reference_state_dict_keys.json"), "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("wrote reference_state_dict_keys.json:", {k: len(v) for k, v in table.items()})


if __name__ == "__main__":
    main()
