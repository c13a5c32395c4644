"""Configuration of each golden case (mirrors CASES in make_golden.py, built from this repo's own YAMLs)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_PLAIN = "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
_TRIPLET = "configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"
_FPN = "configs/e2e_faster_rcnn_R_50_FPN_1x.yaml"
_CASES = {
    "fpn": (_FPN, []),
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
