"""Configuration of each golden case (mirrors CASES in make_golden.py, built from this repo's own YAMLs)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_PLAIN = "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
_TRIPLET = "configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
_FPN = "configs/e2e_faster_rcnn_R_50_FPN_1x.yaml"
_C4_PLAIN = "configs/e2e_faster_rcnn_R_50_C4_1x.yaml"      # BASELINE.json configs[0]
_CASES = {
    "fpn": (_FPN, []),
    "c4_plain": (_C4_PLAIN, []),
    "da_plain": (_PLAIN, []),
    "da_img_only": (_PLAIN, ["MODEL.DA_HEADS.DA_INS_LOSS_WEIGHT", 0.0, "MODEL.DA_HEADS.DA_CST_LOSS_WEIGHT", 0.0]),
    "da_triplet": (_TRIPLET, []),
    "da_triplet_aligned": (_TRIPLET, ["MODEL.DA_HEADS.ALIGNMENT", True, "MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", 1.0,
                                      "MODEL.DA_HEADS.DA_CST_LOSS_WEIGHT", 0.1]),
}


def case_cfg(name):
    from da_detect_amd.config import cfg

    yaml, overrides = _CASES[name]
    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, yaml))
    c.merge_from_list(list(overrides))
    return c


def fpn_dcn_da_cfg():
    """BASELINE.json configs[4]: R-101-FPN + DCN under DA (this repo's documented extension, no reference yaml)"""
    from da_detect_amd.config import cfg

    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, "configs/da_faster_rcnn/"
                                         "e2e_da_faster_rcnn_R_101_FPN_DCN_cityscapes_to_foggy_cityscapes.yaml"))
    return c


# AdvGRL fixtures (make_golden_advgrl.py -> advgrl.npz): triplet recipe + consistency term, the domain classifiers made
# good by fill.structured_da_heads; name -> (overrides, gain of the image head, gain of the instance head)
ADVGRL_IMG_SCALE = (1.0, 0.35, 0.6)     # source, target (darkened), auxiliary
ADVGRL_CASES = {
    "active": (["MODEL.DA_HEADS.DA_CST_LOSS_WEIGHT", 0.1], 3.0, 3.0),          # threshold 30: weight = -0.1 / loss
    "clamped": (["MODEL.DA_HEADS.DA_ADV_GRL_THRESHOLD", 2, "MODEL.DA_HEADS.DA_CST_LOSS_WEIGHT", 0.1], 3.0, 3.0),
    "instance_dormant": ([], 3.0, 0.0),     # image branch active, instance branch above the gate (fixed -GRL weight)
}


def advgrl_setup(z, name, device):
    """-> (cfg, model state dict, ImageList, targets) of AdvGRL case `name` for fixture `z` (np.load of advgrl.npz)"""
    import torch

    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.fill import fill_state_dict, structured_da_heads

    overrides, gain_img, gain_ins = ADVGRL_CASES[name]
    c = case_cfg("da_triplet")
    c.merge_from_list(list(overrides))
    model = build_detection_model(c)
    sd = structured_da_heads(fill_state_dict(model.state_dict(), int(z["seed"])), "da_heads_triplet", z["dir_img"],
                             float(z["off_img"]), z["dir_ins"], float(z["off_ins"]), gain_img, gain_ins)
    images, targets = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]),
                                 device=torch.device("cpu"))
    for i, sc in enumerate(z["img_scale"]):
        images.tensors[i] *= float(sc)
    return c, model, sd, images.to(device), [t.to(device) for t in targets]
