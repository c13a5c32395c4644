"""CPU tests of the host-side logic and of the C-ABI surface (no kernels are launched)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    from da_detect_amd import _lib

    header = open(os.path.join(ROOT, "include", "dadet.h")).read()
    declared = set(re.findall(r"\b(dadet_[a-z0-9_]+)\s*\(", header))
    declared -= {"dadet_conv_desc", "dadet_sgd_entry"}
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libdadet_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.dadet_version() >= 100


def test_product_path_has_no_cpu_fallback():
    from da_detect_amd import _C, _lib

    with pytest.raises(_lib.DadetError):
        _C.roi_align_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 0.25, 7, 7, 2)
    with pytest.raises(_lib.DadetError):
        _C.conv_forward(torch.zeros(1, 4, 8, 8), torch.zeros(4, 4, 1, 1))
    with pytest.raises(_lib.DadetError):
        _C.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)


def test_cfgnode_yacs_semantics(tmp_path):
    from da_detect_amd.config import cfg

    c = cfg.clone()
    p = tmp_path / "a.yaml"
    p.write_text("MODEL:\n  ROI_BOX_HEAD:\n    NUM_CLASSES: 9\nINPUT:\n  MIN_SIZE_TRAIN: (600,)\nSOLVER:\n  BASE_LR: 1\n")
    c.merge_from_file(str(p))
    assert c.MODEL.ROI_BOX_HEAD.NUM_CLASSES == 9 and c.INPUT.MIN_SIZE_TRAIN == (600,) and c.SOLVER.BASE_LR == 1.0
    c.merge_from_list(["MODEL.DEVICE", "cpu", "SOLVER.STEPS", "(1, 2)"])
    assert c.MODEL.DEVICE == "cpu" and c.SOLVER.STEPS == (1, 2)
    with pytest.raises(KeyError):
        c.merge_from_list(["MODEL.NOPE", 1])
    with pytest.raises(ValueError):
        c.merge_from_list(["MODEL.RPN.NMS_THRESH", "high"])
    assert cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES == 81  # clone is deep
    c.freeze()
    with pytest.raises(AttributeError):
        c.MODEL.DEVICE = "cuda"


def test_every_shipped_yaml_merges():
    import glob

    from da_detect_amd.config import cfg

    files = glob.glob(os.path.join(ROOT, "configs", "**", "*.yaml"), recursive=True)
    assert len(files) >= 4
    for f in files:
        cfg.clone().merge_from_file(f)


def test_box_coder_reference_known_answer_and_roundtrip():
    from da_detect_amd.modeling.box_coder import BoxCoder

    ka = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))
    for case in ka["box_decode"]:
        got = BoxCoder(tuple(case["weights"])).decode(torch.tensor(case["deltas"]), torch.tensor(case["boxes"]))
        np.testing.assert_allclose(got.numpy(), np.array(case["expected"], np.float32), atol=1e-4)
    g = torch.Generator().manual_seed(0)
    a = torch.rand(50, 2, generator=g) * 300
    boxes = torch.cat([a, a + 10 + torch.rand(50, 2, generator=g) * 200], 1)
    b = torch.rand(50, 2, generator=g) * 300
    gts = torch.cat([b, b + 10 + torch.rand(50, 2, generator=g) * 200], 1)
    coder = BoxCoder((10.0, 10.0, 5.0, 5.0))
    torch.testing.assert_close(coder.decode(coder.encode(gts, boxes), boxes), gts, rtol=1e-4, atol=1e-3)


def test_boxlist_semantics():
    from da_detect_amd.structures import BoxList, boxlist_iou, cat_boxlist, remove_small_boxes

    b = BoxList(torch.tensor([[0.0, 0, 9, 9], [5, 5, 14, 24], [3, 3, 3, 3]]), (20, 30))
    b.add_field("labels", torch.tensor([1, 2, 3]))
    assert b.area().tolist() == [100.0, 200.0, 1.0]
    assert b.convert("xywh").bbox[1].tolist() == [5, 5, 10, 20]
    assert b.convert("xywh").convert("xyxy").bbox.tolist() == b.bbox.tolist()
    assert b.transpose(0).bbox[0].tolist() == [10, 0, 19, 9]
    assert b.resize((40, 60)).bbox[0].tolist() == [0, 0, 18, 18]
    assert b[torch.tensor([2, 0])].get_field("labels").tolist() == [3, 1]
    assert len(remove_small_boxes(b, 2)) == 2
    iou = boxlist_iou(b, b)
    assert torch.allclose(torch.diag(iou), torch.ones(3))
    assert abs(float(iou[0, 1]) - 25.0 / 275.0) < 1e-6
    c = cat_boxlist([b, b])
    assert len(c) == 6 and c.get_field("labels").tolist() == [1, 2, 3, 1, 2, 3]
    clipped = BoxList(torch.tensor([[-5.0, -5, 50, 50]]), (20, 30)).clip_to_image(remove_empty=False)
    assert clipped.bbox[0].tolist() == [0, 0, 19, 29]


def test_matcher_and_sampler():
    from da_detect_amd.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from da_detect_amd.modeling.matcher import Matcher

    q = torch.tensor([[0.9, 0.2, 0.45, 0.1], [0.1, 0.6, 0.45, 0.05]])
    assert Matcher(0.7, 0.3, False)(q.clone()).tolist() == [0, -2, -2, -1]
    # low-quality matches: every gt keeps its best prediction (ties included)
    assert Matcher(0.7, 0.3, True)(q.clone()).tolist() == [0, 1, -2, -1]
    with pytest.raises(ValueError):
        Matcher(0.5, 0.5)(torch.zeros(0, 3))
    torch.manual_seed(0)
    labels = torch.tensor([1] * 10 + [0] * 100 + [-1] * 5)
    pos, neg = BalancedPositiveNegativeSampler(16, 0.25)([labels])
    assert int(pos[0].sum()) == 4 and int(neg[0].sum()) == 12
    assert not bool((pos[0] & (labels != 1)).any()) and not bool((neg[0] & (labels != 0)).any())


def test_anchor_generator_matches_detectron_values():
    from da_detect_amd.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors
    from da_detect_amd.structures import ImageList

    cell = generate_anchors(16, (32, 64, 128, 256, 512), (0.5, 1.0, 2.0))
    assert cell.shape == (15, 4)
    # Detectron's stride-16 anchors: ratio 0.5 / scale 32 and ratio 1 / scale 512
    assert cell[0].tolist() == [-15.0, -4.0, 30.0, 19.0]
    assert cell[9].tolist() == [-248.0, -248.0, 263.0, 263.0]
    ag = AnchorGenerator((32, 64, 128, 256, 512), (0.5, 1.0, 2.0), (16,), 0)
    out = ag(ImageList(torch.zeros(1, 3, 64, 96), [(64, 96)]), [torch.zeros(1, 8, 4, 6)])
    a = out[0][0]
    assert a.bbox.shape == (4 * 6 * 15, 4)
    assert a.bbox[15].tolist() == [1.0, -4.0, 46.0, 19.0]  # next cell along x
    vis = a.get_field("visibility")
    assert bool(vis.any()) and not bool(vis.all())


def test_model_structure_matches_reference_state_dict_layout():
    from da_detect_amd.config import cfg
    from da_detect_amd.modeling.detector import build_detection_model

    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"))
    m = build_detection_model(c)
    sd = m.state_dict()
    assert len(sd) == 286
    total = sum(p.numel() for p in m.parameters())
    trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert (total, trainable) == (36736314, 36513914)  # SURVEY.md section 2.3 [probe]
    for k in ["backbone.body.stem.conv1.weight", "backbone.body.layer3.5.bn3.running_var",
              "rpn.anchor_generator.cell_anchors.0", "rpn.head.bbox_pred.bias",
              "roi_heads.box.feature_extractor.head.layer4.0.downsample.0.weight",
              "roi_heads.box.predictor.cls_score.weight", "da_heads.imghead.conv2_da.weight",
              "da_heads.inshead.fc3_da.bias"]:
        assert k in sd, k
    assert tuple(sd["backbone.body.stem.conv1.weight"].shape) == (64, 3, 7, 7)


def test_cosine_scheduler_restated_from_call_site():
    from da_detect_amd.solver import CosineLRScheduler

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([{"params": [p], "lr": 1e-3}], lr=1e-3)
    s = CosineLRScheduler(opt, t_initial=1000, lr_min=1e-6, warmup_lr_init=1e-4, warmup_t=100, t_in_epochs=False)
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12
    s.step_update(50)
    assert abs(opt.param_groups[0]["lr"] - (1e-4 + 50 * (1e-3 - 1e-4) / 100)) < 1e-12
    s.step_update(500)
    assert abs(opt.param_groups[0]["lr"] - (1e-6 + 0.5 * (1e-3 - 1e-6))) < 1e-9
    s.step_update(5000)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12


@pytest.mark.parametrize("case", ["da_plain", "da_triplet", "fpn"])
def test_state_dict_keys_match_reference(case):
    """released DA checkpoints must load: same state_dict key names and shapes as the reference model
    (tests/golden/reference_state_dict_keys.json, written by make_golden_eval.py from the imported reference)"""
    import json

    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg

    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_state_dict_keys.json")))[case]
    mine = {k: list(v.shape) for k, v in build_detection_model(case_cfg(case)).state_dict().items()}
    assert set(mine) == set(ref), (sorted(set(mine) - set(ref))[:5], sorted(set(ref) - set(mine))[:5])
    assert all(mine[k] == ref[k] for k in ref)


def test_prepare_for_coco_detection_records():
    """bbox.json record boundary (coco_eval.py:81-112): resize to the original size, xywh with the +1 convention"""
    from da_detect_amd.engine.inference import _accumulate_predictions_from_multiple_gpus, prepare_for_coco_detection
    from da_detect_amd.structures.bounding_box import BoxList

    class DS:
        id_to_img_map = {0: 17, 1: 42}
        contiguous_category_id_to_json_id = {1: 24, 2: 26}

        def get_img_info(self, i):
            return {"width": 200, "height": 100}

    a = BoxList(torch.tensor([[10.0, 20.0, 29.0, 39.0]]), (100, 50), mode="xyxy")   # half-size network input
    a.add_field("scores", torch.tensor([0.9]))
    a.add_field("labels", torch.tensor([2]))
    empty = BoxList(torch.zeros((0, 4)), (100, 50), mode="xyxy")
    empty.add_field("scores", torch.zeros(0))
    empty.add_field("labels", torch.zeros(0, dtype=torch.int64))
    preds = _accumulate_predictions_from_multiple_gpus({1: empty, 0: a})
    recs = prepare_for_coco_detection(preds, DS())
    assert len(recs) == 1
    r = recs[0]
    assert r["image_id"] == 17 and r["category_id"] == 26 and abs(r["score"] - 0.9) < 1e-6
    assert r["bbox"] == [20.0, 40.0, 39.0, 39.0]    # scaled x2 -> (20,40,58,78) -> w = 58-20+1, h = 78-40+1


def test_weight_gradient_split_plan_respects_the_workgroup_slots():
    """dadet_conv_wgrad_workspace_bytes is pure host code (the split plan of conv_igemm.hip::wgrad_plan): the number
    of workgroups (tiles x splits) must not land just above a multiple of the chip's 512 slots — the configuration the
    sweep in profiles/r01_wgrad_split_sweep.txt showed to cost a whole extra pass — and every split keeps >= 4 K-steps"""
    import ctypes

    from da_detect_amd import _C, _lib

    def splits(N, H, W, Cin, Cout, k, stride):
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        d = _C._desc(N, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo)
        nbytes = ctypes.c_size_t(0)
        _lib.call("dadet_conv_wgrad_workspace_bytes", ctypes.byref(d), ctypes.byref(nbytes))
        per = 4 * Cout * Cin * k * k
        assert nbytes.value % per == 0
        return max(1, nbytes.value // per), N * Ho * Wo

    shapes = [(512, 7, 7, 512, 512, 3, 1), (2, 64, 128, 1024, 1024, 3, 1), (512, 7, 7, 512, 2048, 1, 1),
              (512, 14, 14, 1024, 2048, 1, 2), (2, 64, 128, 256, 256, 3, 1), (2, 128, 256, 128, 128, 3, 1),
              (2, 64, 128, 256, 1024, 1, 1), (2, 128, 256, 128, 512, 1, 1), (1, 8, 8, 64, 64, 3, 1)]
    lib = _lib.load()
    plan = lib.dadet_get_big_gemm()
    try:
        lib.dadet_set_big_gemm(0)       # the 128 x 128 kernel's plan
        for N, H, W, Cin, Cout, k, stride in shapes:
            s, M = splits(N, H, W, Cin, Cout, k, stride)
            tiles = -(-Cout // 128) * -(-(Cin * k * k) // 128)
            wgs = tiles * s
            rows = -(-M // s)
            assert s == 1 or rows >= 128, (N, H, W, Cin, Cout, k, s, rows)
            over = wgs % 512
            assert wgs <= 512 or over == 0 or over > 128, ("%d workgroups: a nearly empty extra pass" % wgs, Cin, Cout, k)
        # the shapes the sweep pinned down
        assert splits(512, 7, 7, 512, 512, 3, 1)[0] == 7           # 144 tiles -> 1008 workgroups
        assert splits(512, 7, 7, 512, 2048, 1, 1)[0] == 8          # 64 tiles  -> 512 workgroups
        assert splits(2, 64, 128, 256, 1024, 1, 1)[0] == 32        # 16 tiles  -> 512 workgroups
        # the 256 x 256-tile kernel (round 5; contraction mode 4, weights of at least eight tiles): one workgroup per CU, the
        # parts fill the 256 slots once and keep at least four K-tiles of rows each
        lib.dadet_set_big_gemm(1)
        big = 0
        for N, H, W, Cin, Cout, k, stride in shapes:
            pad = k // 2
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
            d = _C._desc(N, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo)
            if lib.dadet_conv_wgrad_variant(ctypes.byref(d)) != 1:
                continue
            big += 1
            s, M = splits(N, H, W, Cin, Cout, k, stride)
            tiles = -(-Cout // 256) * -(-(Cin * k * k) // 256)
            # (a part is one fp32 accumulator chain: never over more than 4096 rows, even where that costs a second pass)
            assert tiles >= 8 and tiles * s <= max(256, tiles * -(-M // 4096)) and -(-M // s) <= 4096 + 31 \
                and (s == 1 or -(-M // s) >= 128), (Cin, Cout, k, tiles, s)
        assert big >= 4
        # several weight gradients in one launch: every part of every problem reduces the same number of rows, the parts fit
        # the 256 slots, and a four-tile weight (below the bound for a launch of its own) is a member
        group = [(2, 64, 128, 256, 1024, 1, 1), (2, 64, 128, 256, 256, 3, 1), (2, 64, 128, 1024, 256, 1, 1)]
        descs = (_lib.ConvDesc * 3)()
        for i, (N, H, W, Cin, Cout, k, stride) in enumerate(group):
            pad = k // 2
            descs[i] = _C._desc(N, H, W, Cin, Cout, k, k, stride, pad, (H + 2 * pad - k) // stride + 1,
                                (W + 2 * pad - k) // stride + 1)
        sp, nb = (ctypes.c_int * 3)(), (ctypes.c_size_t * 3)()
        assert lib.dadet_conv_wgrad_group_plan(descs, 3, sp, nb) == 256
        tiles = [-(-d.Cout // 256) * -(-(d.Cin * d.KH * d.KW) // 256) for d in descs]
        assert tiles == [4, 9, 4] and len(set(sp)) == 1 and sum(t * s for t, s in zip(tiles, sp)) <= 256
        assert sum(t * (s + 1) for t, s in zip(tiles, sp)) > 256 - 17 * 2      # ... and no fewer parts than fit
        assert all(b == 4 * s * d.Cout * d.Cin * d.KH * d.KW for b, s, d in zip(nb, sp, descs))
        # a res3 block (128 / 512 channels): the 128 x 128 kernel's grouped form, two workgroups per CU
        group = [(2, 128, 256, 128, 512, 1, 1), (2, 128, 256, 128, 128, 3, 1), (2, 128, 256, 512, 128, 1, 1)]
        for i, (N, H, W, Cin, Cout, k, stride) in enumerate(group):
            descs[i] = _C._desc(N, H, W, Cin, Cout, k, k, stride, k // 2, H, W)
        assert lib.dadet_conv_wgrad_group_plan(descs, 3, sp, nb) == 128
        tiles = [-(-d.Cout // 128) * -(-(d.Cin * d.KH * d.KW) // 128) for d in descs]
        assert tiles == [4, 9, 4] and len(set(sp)) == 1 and 512 - 34 < sum(t * s for t, s in zip(tiles, sp)) <= 512
        # the res5 head on 512 ROIs (M = 25088, ~100 tiles: the slots alone would allow two parts of 12544 rows): a part of
        # a GROUP is one accumulator chain as well — never more than 4096 (+31) rows, in equal parts
        group = [(512, 7, 7, 2048, 512, 1, 1), (512, 7, 7, 512, 512, 3, 1), (512, 7, 7, 512, 2048, 1, 1)]
        for i, (N, H, W, Cin, Cout, k, stride) in enumerate(group):
            descs[i] = _C._desc(N, H, W, Cin, Cout, k, k, stride, k // 2, H, W)
        assert lib.dadet_conv_wgrad_group_plan(descs, 3, sp, nb) == 256
        assert len(set(sp)) == 1 and sp[0] == 7 and -(-25088 // sp[0]) <= 4096 + 31
        # ... with an eighth of slack where the slots alone ask for slightly more: the same head on 256 ROIs (68 tiles,
        # 12544 rows) stays at three parts of 4192 rows (one round of 204 workgroups) instead of four (272: a second round)
        for i, (N, H, W, Cin, Cout, k, stride) in enumerate([(256,) + g[1:] for g in group]):
            descs[i] = _C._desc(N, H, W, Cin, Cout, k, k, stride, k // 2, H, W)
        assert lib.dadet_conv_wgrad_group_plan(descs, 3, sp, nb) == 256
        assert len(set(sp)) == 1 and sp[0] == 3 and -(-12544 // sp[0]) <= 4096 + 512
        odd = (_lib.ConvDesc * 1)(_C._desc(2, 128, 256, 128, 18, 3, 3, 1, 1, 128, 256))
        assert lib.dadet_conv_wgrad_group_plan(odd, 1, sp, nb) == 0          # 18 output channels: rows padded beyond Cout
    finally:
        lib.dadet_set_big_gemm(plan)


def test_unread_work_rules():
    """modeling/elision.py + GeneralizedRCNN._images_with_read_proposals: which images' RPN proposals some loss reads
    (source images always; a target image only through instance-level features or aligned passes; a triplet batch's
    auxiliary image never) — host logic, no kernels"""
    from da_detect_amd.config import cfg as base
    from da_detect_amd.modeling import elision
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.structures.bounding_box import BoxList

    def tgt(source):
        b = BoxList(torch.tensor([[1.0, 2.0, 30.0, 40.0]]), (64, 64), "xyxy")
        b.add_field("labels", torch.tensor([1]))
        b.add_field("is_source", torch.tensor([source]))
        return b

    assert elision.leading_source_images([tgt(True), tgt(False)]) == 1
    assert elision.leading_source_images([tgt(True), tgt(True), tgt(False)]) == 2
    assert elision.leading_source_images([tgt(False), tgt(True)]) == 0          # not a prefix: no rule applies
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def model_for(yaml, *overrides):
        c = base.clone()
        c.merge_from_file(os.path.join(root, "configs/da_faster_rcnn", yaml))
        c.merge_from_list(list(overrides))
        return build_detection_model(c)

    pair, triple = [tgt(True), tgt(False)], [tgt(True), tgt(False), tgt(False)]
    m = model_for("e2e_da_faster_rcnn_R_50_C4_img_only.yaml")
    assert not m.da_heads.needs_instance_features and m._images_with_read_proposals(pair) == 1
    m = model_for("e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")
    assert m.da_heads.needs_instance_features and m._images_with_read_proposals(pair) == 2
    tri = [f for f in os.listdir(os.path.join(root, "configs/da_faster_rcnn")) if "triplet" in f][0]
    m = model_for(tri, "MODEL.DA_HEADS.ALIGNMENT", False)
    need = m.da_heads_triplet.needs_instance_features
    assert m._images_with_read_proposals(triple) == (2 if need else 1)
    m = model_for(tri, "MODEL.DA_HEADS.ALIGNMENT", True)
    assert m._images_with_read_proposals(triple) == 2                          # aligned passes pool the target's proposals
    assert m._images_with_read_proposals(pair) is None                         # not a triplet batch: no rule


def test_lane_tuner_keeps_one_stream_unless_clearly_faster(monkeypatch):
    """engine.trainer.WgradLaneTuner's decision rule (the timing itself needs a GPU: tests/test_default_path_gpu.py)"""
    from da_detect_amd.engine.trainer import WgradLaneTuner
    from da_detect_amd.utils import streams

    monkeypatch.setattr(streams, "WGRAD_LANE_ROWS", 0)
    t = WgradLaneTuner(torch.device("cpu"))
    assert not t.active                                                         # CPU: nothing to tune
    for times, want in (({0: 20.0, 17000: 19.7}, 0), ({0: 20.0, 17000: 19.3}, 17000), ({0: 20.0, 17000: 21.0}, 0)):
        t = WgradLaneTuner(torch.device("cpu"))
        t.active, t.times = True, {}
        t._cand, t._count = len(t.CANDIDATES) - 1, t.settle + t.measure - 1
        t.times = {k: v for k, v in times.items() if k != t.CANDIDATES[-1]}
        monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
        monkeypatch.setattr(streams, "join_wgrad_lane", lambda *a, **k: None)
        import time as _time
        t._t0 = _time.perf_counter() - times[t.CANDIDATES[-1]] * t.measure
        t.step_end()
        assert not t.active and streams.WGRAD_LANE_ROWS == want, (times, streams.WGRAD_LANE_ROWS)
    monkeypatch.setattr(streams, "WGRAD_LANE_ROWS", 0)


def test_later_value_falls_back_to_a_host_tensor_on_cpu():
    """da_heads._later_value: on CPU the previous triplet loss is kept as the reference keeps it (a detached CPU tensor);
    the adaptive-margin rule compares it with 0.0 either way"""
    from da_detect_amd.modeling.da_heads.da_heads import _later_value
    from da_detect_amd.modeling.da_heads.loss import TripletMargins

    zero, one = _later_value(torch.tensor(0.0, requires_grad=True) * 1.0), _later_value(torch.tensor(1.0))
    assert not zero.requires_grad and (zero == 0.0) and not (one == 0.0)
    m = TripletMargins()
    a, p, n = torch.randn(4, 8), torch.randn(4, 8), torch.randn(4, 8)      # (the instance form runs on the CPU)
    m.triplet_ins_loss(a, p, n, one, adaptive=True, lr=0.5, max_margin=3.0, margin=1.0)
    assert m.margin_ins == 1.0
    m.triplet_ins_loss(a, p, n, zero, adaptive=True, lr=0.5, max_margin=3.0, margin=1.0)
    assert m.margin_ins == 1.5          # grows only after an exactly-zero loss


def test_offset_branch_padding_on_cpu_matches_zero_padding():
    """backbone.resnet._pad_out_channels: the persistent-buffer cache is a CUDA-only shortcut; on CPU (and for non-leaf
    weights) the output channels are zero-padded on the spot — same values either way"""
    from da_detect_amd.modeling.backbone.resnet import _pad_out_channels

    w = torch.nn.Parameter(torch.randn(27, 8, 3, 3))
    b = torch.nn.Parameter(torch.randn(27))
    wp, bp = _pad_out_channels(w, b)
    assert wp.shape == (28, 8, 3, 3) and bp.shape == (28,)
    assert torch.equal(wp[:27], w) and torch.equal(bp[:27], b) and float(wp[27].abs().sum()) == 0.0 and float(bp[27]) == 0.0
    w4 = torch.nn.Parameter(torch.randn(28, 8, 3, 3))
    assert _pad_out_channels(w4, None)[0] is w4


def test_deferred_weight_gradient_reductions_never_share_a_buffer_in_one_launch(monkeypatch):
    """utils.streams: the merged reduction launch adds every queued item into its dw with a plain read-modify-write, so two
    items with the same dw must not meet in one launch (a weight used twice in a graph).  The queue flushes the
    earlier one first; zero_grad() drops what a backward without step() left behind."""
    import types

    import torch
    from da_detect_amd import _C
    from da_detect_amd.utils import streams

    launches = []
    monkeypatch.setattr(_C, "conv_wgrad_reduce_batch", lambda items: launches.append([it[0].dw for it in items]))
    monkeypatch.setattr(streams, "DIRECT_WGRAD", True)
    ws = torch.zeros(1)

    def item(dw):
        return (types.SimpleNamespace(dw=dw), ws, None, None)

    streams.discard_wgrad_reductions()
    lane = streams.WgradLane(torch.device("cpu"))
    lane.reduce_batch([item(0x1000), item(0x2000)])
    lane.reduce_batch([item(0x3000)])
    assert launches == [] and len(streams._PENDING_REDUCES) == 3           # distinct buffers: merged, deferred
    lane.reduce_batch([item(0x2000), item(0x4000)])                         # 0x2000 again: the queued pass goes first
    assert launches == [[0x1000, 0x2000, 0x3000]]
    assert [it[0].dw for it in streams._PENDING_REDUCES] == [0x2000, 0x4000]
    lane.reduce_batch([item(0x5000), item(0x5000)])                         # twice inside one node's own batch
    assert launches[1] == [0x2000, 0x4000, 0x5000] and [it[0].dw for it in streams._PENDING_REDUCES] == [0x5000]
    streams.flush_wgrad_reductions(torch.device("cpu"))
    assert launches[2] == [0x5000] and not streams._PENDING_REDUCES and not streams._PENDING_DW
    # a backward that no step() followed: the next zero_grad() drops its queued passes instead of adding them later
    from da_detect_amd.parallel.reducer import BucketedGradReducer

    lane.reduce_batch([item(0x6000)])
    red = BucketedGradReducer([torch.nn.Parameter(torch.ones(4))])
    red.zero_grad()
    assert not streams._PENDING_REDUCES and not streams._PENDING_DW and len(launches) == 3


def test_crowded_images_take_the_unbounded_sampler_path():
    """dadet_proposals_sample holds an image's ground truth in an LDS table of 1024 boxes: a batch with a more crowded image
    must not be handed over as PendingProposals — the RPN then returns BoxLists and the box head's
    box_match_encode + sample_rois path, which has no such bound, runs"""
    import torch
    from da_detect_amd import _C
    from da_detect_amd.modeling.roi_heads.box_head import loss as L
    from da_detect_amd.structures.bounding_box import BoxList

    ev = L.FastRCNNLossComputation.__new__(L.FastRCNNLossComputation)
    ev.proposal_matcher = type("M", (), {"allow_low_quality_matches": False})()

    def target(n):
        return BoxList(torch.zeros((n, 4)), (100, 100), mode="xyxy")

    base = ev.accepts_pending()
    assert ev.accepts_pending([target(8), target(_C.PROPOSALS_SAMPLE_MAX_GT)]) == base
    assert ev.accepts_pending([target(8), target(_C.PROPOSALS_SAMPLE_MAX_GT + 1)]) is False


def test_large_tile_plan_of_the_forward_and_data_gradient_gemms():
    """csrc/conv_big.hip: big_variant through dadet_conv_forward_variant (no launch): which tile serves a layer under the
    default contraction — 3: weight-stationary (K <= 256 1x1), 4: 256 x 256 tile, 5: 256 x 128 tile, 0 - 2: the 128-wide
    kernels.  Round 6: 129 .. 256 output channels go to the 256 x 256 tile when one column of tiles fills the chip (>= 96 row
    tiles: the pyramid's P2 / P3 layers), and stay on the 256 x 128 tile below that (res4: 64 row tiles)."""
    import ctypes

    from da_detect_amd import _C, _lib

    lib = _lib.load()
    if lib.dadet_get_gemm_mode() != 4 or lib.dadet_get_big_gemm() != 1:
        import pytest
        pytest.skip("not the default contraction / plan")

    def variant(N, H, W, Cin, Cout, k):
        d = _C._desc(N, H, W, Cin, Cout, k, k, 1, k // 2, H, W)
        return lib.dadet_conv_forward_variant(ctypes.byref(d))

    assert variant(2, 256, 512, 256, 256, 3) == 4        # P2 3x3: 1024 row tiles x 1
    assert variant(2, 128, 256, 256, 256, 3) == 4        # P3 3x3: 256 row tiles
    assert variant(2, 128, 256, 512, 256, 1) == 4        # C3 lateral, K = 512
    assert variant(2, 64, 128, 256, 256, 3) == 5         # res4 3x3: 64 row tiles -> 256 x 128 tile, two K parts
    assert variant(2, 64, 128, 1024, 256, 1) == 5        # res4 conv1
    assert variant(2, 128, 256, 128, 128, 3) == 5        # res3 3x3: 128 channels never take the wide tile
    assert variant(256, 7, 7, 512, 512, 3) == 4          # res5 3x3 on 256 ROIs
    assert variant(2, 64, 128, 256, 1024, 1) == 3        # K = 256 1x1: weight-stationary
    assert variant(2, 256, 512, 64, 64, 3) in (0, 1, 2)  # res2: the 128-wide kernels
