"""Golden data for checkpoint loading — runs ONLY in the authoring container (needs /root/reference): the reference's
Caffe2 key renamer and its suffix matcher applied to synthetic key lists; outputs stored as data."""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from maskrcnn_benchmark.utils import c2_model_loading as ref_c2  # noqa: E402
from maskrcnn_benchmark.utils import model_serialization as ref_ms  # noqa: E402

C2_KEYS = ["conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "conv1_w_momentum", "res2_0_branch2a_w",
           "res2_0_branch2a_bn_s", "res2_0_branch2a_bn_b", "res2_0_branch2b_w", "res2_0_branch2b_bn_s",
           "res2_0_branch2c_w", "res2_0_branch2c_bn_b", "res2_0_branch1_w", "res2_0_branch1_bn_s",
           "res3_3_branch2a_w", "res4_5_branch2c_bn_s", "res4_22_branch2b_w", "res5_2_branch2c_bn_b",
           "res5_0_branch1_bn_b", "fc1000_w", "fc1000_b", "pred_w", "pred_b", "conv_rpn_w", "conv_rpn_b",
           "rpn_cls_logits_w", "rpn_cls_logits_b", "rpn_bbox_pred_w", "rpn_bbox_pred_b", "cls_score_w", "cls_score_b",
           "bbox_pred_w", "bbox_pred_b", "fc6_w", "fc7_b", "fpn_inner_res5_2_sum_w", "fpn_inner_res5_2_sum_b",
           "fpn_inner_res4_5_sum_lateral_w", "fpn_inner_res3_3_sum_lateral_b", "fpn_inner_res2_2_sum_lateral_w",
           "fpn_res5_2_sum_w", "fpn_res4_5_sum_b", "fpn_res3_3_sum_w", "fpn_res2_2_sum_w", "conv_rpn_fpn2_w",
           "rpn_cls_logits_fpn2_b", "rpn_bbox_pred_fpn2_w", "_[mask]_fcn1_w", "conv5_mask_w", "mask_fcn_logits_b",
           "kps_score_lowres_w", "kps_score_w", "conv_fcn1_w", "res2_0_branch2a_gn_s", "res2_0_branch2a_gn_b",
           "res2_0_branch1_gn_s", "res2_0_branch1_gn_b"]


def main():
    out = {"c2": {}, "suffix": []}
    for arch, stages in ref_c2._C2_STAGE_NAMES.items():
        mapping = {}
        for k in C2_KEYS:      # one key per call: several Caffe2 names legitimately collapse onto one parameter name
            renamed = ref_c2._rename_weights_for_resnet({k: np.zeros(1, dtype=np.float32), "zz_anchor_w":
                                                         np.zeros(1, dtype=np.float32)}, stages)
            names = [n for n in renamed.keys() if n != "zz.anchor.weight"]
            mapping[k] = names[0] if names else None      # None: dropped (momentum blobs)
        out["c2"][arch] = mapping
    cases = [
        (["backbone.body.layer1.0.conv1.weight", "backbone.body.stem.conv1.weight", "rpn.head.conv.weight",
          "roi_heads.box.feature_extractor.head.layer4.0.conv1.weight", "roi_heads.box.predictor.cls_score.weight",
          "da_heads.imghead.conv1_da.weight"],
         ["conv1.weight", "layer1.0.conv1.weight", "layer4.0.conv1.weight", "rpn.head.conv.weight", "cls_score.weight",
          "fc1000.weight"]),
        (["a.b.c", "x.b.c", "c"], ["b.c", "c", "a.b.c"]),
        (["module.a.w", "module.b.w"], ["a.w", "zz.b.w"]),
    ]
    for model_keys, loaded_keys in cases:
        msd = {k: torch.zeros(1) for k in model_keys}
        lsd = {k: torch.full((1,), float(i + 1)) for i, k in enumerate(sorted(loaded_keys))}
        ref_ms.align_and_update_state_dicts(msd, lsd)
        got = {}
        for k in model_keys:
            v = float(msd[k])
            got[k] = sorted(loaded_keys)[int(v) - 1] if v > 0 else None
        out["suffix"].append({"model_keys": model_keys, "loaded_keys": loaded_keys, "matches": got})
    with open(os.path.join(HERE, "reference_checkpoint_maps.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({a: len(m) for a, m in out["c2"].items()}, len(out["suffix"]))


if __name__ == "__main__":
    main()
