"""Checkpoint I/O: Caffe2 key renaming and suffix matching against maps recorded from
the reference's own functions (tests/golden/reference_checkpoint_maps.json), save / load round trip, loading an
MSRA-style pickle into the R-50-C4 model."""
import json
import os
import pickle

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_checkpoint_maps.json")))


def test_c2_renaming_matches_the_reference():
    from da_detect_amd.utils.c2_model_loading import _C2_STAGE_NAMES, _rename_weights_for_resnet, rename_c2_keys

    for arch, mapping in GOLD["c2"].items():
        keys = sorted(mapping)
        got = dict(zip(keys, rename_c2_keys(keys, _C2_STAGE_NAMES[arch])))
        for k in keys:
            if mapping[k] is not None:
                assert got[k] == mapping[k], (arch, k, got[k], mapping[k])
        w = _rename_weights_for_resnet({k: np.zeros(1, np.float32) for k in ("conv1_w", "conv1_w_momentum")},
                                       _C2_STAGE_NAMES[arch])
        assert list(w) == ["conv1.weight"]            # momentum blobs are dropped


def test_suffix_matching_matches_the_reference():
    from da_detect_amd.utils.model_serialization import match_keys, strip_prefix_if_present

    for case in GOLD["suffix"]:
        got = match_keys(case["model_keys"], case["loaded_keys"])
        for k, want in case["matches"].items():
            assert got.get(k) == want, (k, got.get(k), want)
    sd = {"module.a": 1, "module.b.c": 2}
    assert dict(strip_prefix_if_present(sd, "module.")) == {"a": 1, "b.c": 2}
    assert strip_prefix_if_present({"module.a": 1, "b": 2}, "module.") == {"module.a": 1, "b": 2}


def _small_cfg():
    from golden.cases import case_cfg

    return case_cfg("da_plain")


def test_checkpointer_round_trip_and_c2_pickle(tmp_path):
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.utils.checkpoint import DetectronCheckpointer

    cfg = _small_cfg()
    model = build_detection_model(cfg)
    ck = DetectronCheckpointer(cfg, model, save_dir=str(tmp_path), save_to_disk=True)
    assert not ck.has_checkpoint() and ck.load(None) == {}
    with torch.no_grad():
        model.rpn.head.conv.weight.fill_(0.125)
    ck.save("model_0000010", iteration=10)
    assert ck.has_checkpoint() and ck.get_checkpoint_file().endswith("model_0000010.pth")
    with torch.no_grad():
        model.rpn.head.conv.weight.zero_()
    extra = ck.load(ck.get_checkpoint_file())
    assert extra["iteration"] == 10 and float(model.rpn.head.conv.weight.detach().mean()) == 0.125
    # DistributedDataParallel-style "module." prefix is stripped
    wrapped = {"module." + k: v for k, v in model.state_dict().items()}
    torch.save({"model": wrapped}, str(tmp_path / "ddp.pth"))
    with torch.no_grad():
        model.rpn.head.conv.weight.zero_()
    ck.load(str(tmp_path / "ddp.pth"))
    assert float(model.rpn.head.conv.weight.detach().mean()) == 0.125
    # an MSRA-style Caffe2 pickle initialises both the backbone and the res5 box head (suffix matching)
    # (every bottleneck's first conv / bn must be present, as in the real file: a bare "conv1.weight" would otherwise be
    # the longest suffix of every "...layerX.Y.conv1.weight" — the matcher's documented behaviour)
    blobs = {"conv1_w": np.full((64, 3, 7, 7), 2.0, np.float32), "res_conv1_bn_s": np.full(64, 3.0, np.float32)}
    cin = 64
    for stage, (nblocks, mid) in enumerate([(3, 64), (4, 128), (6, 256), (3, 512)], 2):
        for b in range(nblocks):
            val = 4.0 if (stage, b) == (2, 0) else 5.0 if (stage, b) == (5, 0) else 1.0
            blobs["res%d_%d_branch2a_w" % (stage, b)] = np.full((mid, cin if b == 0 else mid * 4, 1, 1), val, np.float32)
            blobs["res%d_%d_branch2a_bn_s" % (stage, b)] = np.full(mid, 1.5, np.float32)
        cin = mid * 4
    blobs["res5_0_branch2a_w_momentum"] = np.zeros((512, 1024, 1, 1), np.float32)
    with open(str(tmp_path / "R-50.pkl"), "wb") as f:
        pickle.dump({"blobs": blobs}, f)
    ck.load(str(tmp_path / "R-50.pkl"))
    sd = model.state_dict()
    assert float(sd["backbone.body.stem.conv1.weight"].mean()) == 2.0
    assert float(sd["backbone.body.stem.bn1.weight"].mean()) == 3.0
    assert float(sd["backbone.body.layer1.0.conv1.weight"].mean()) == 4.0
    assert float(sd["roi_heads.box.feature_extractor.head.layer4.0.conv1.weight"].mean()) == 5.0
    assert float(model.rpn.head.conv.weight.detach().mean()) == 0.125          # untouched by the pickle
    # catalog://NAME resolves through ModelCatalog of cfg.PATHS_CATALOG to a LOCAL file (no network): missing -> a
    # clear error, present -> loaded like any .pkl (checkpoint.py:118-125 of the reference downloads instead)
    import importlib

    import pytest

    with pytest.raises(FileNotFoundError):
        ck.load("catalog://ImageNetPretrained/MSRA/R-50")
    with pytest.raises(ValueError):
        ck.load("https://example.invalid/R-50.pkl")
    os.makedirs(str(tmp_path / "ImageNetPretrained" / "MSRA"))
    os.replace(str(tmp_path / "R-50.pkl"), str(tmp_path / "ImageNetPretrained" / "MSRA" / "R-50.pkl"))
    catalog = importlib.import_module("da_detect_amd.config.paths_catalog")
    old = catalog.ModelCatalog.MODEL_DIR
    os.environ["DADET_MODEL_DIR"] = str(tmp_path)      # read by ModelCatalog.get at call time
    try:
        with torch.no_grad():
            model.backbone.body.stem.conv1.weight.zero_()
        ck.load("catalog://ImageNetPretrained/MSRA/R-50")
        assert float(model.state_dict()["backbone.body.stem.conv1.weight"].mean()) == 2.0
    finally:
        os.environ.pop("DADET_MODEL_DIR")
        catalog.ModelCatalog.MODEL_DIR = old


def test_catalog_module_is_loaded_once_so_run_time_registrations_are_seen():
    """DatasetCatalog.register(...) / an edited ModelCatalog table must reach make_data_loader and catalog://
    loading — the PATHS_CATALOG file is executed once, not once per call"""
    from da_detect_amd.config import cfg
    from da_detect_amd.data import build as B
    from da_detect_amd.utils.imports import load_paths_catalog

    cat = B._catalog(cfg)
    cat.register("unit_test_cocostyle", "/abs/images", "/abs/ann.json")
    try:
        again = B._catalog(cfg)
        assert again is cat
        assert again.get("unit_test_cocostyle")["args"] == {"root": "/abs/images", "ann_file": "/abs/ann.json"}
        assert load_paths_catalog(cfg.PATHS_CATALOG).DatasetCatalog is cat
    finally:
        cat.DATASETS.pop("unit_test_cocostyle", None)


def test_saved_state_is_the_live_state_not_the_loaded_one(tmp_path):
    """the trainers merge checkpointer.load()'s leftovers (which include the LOADED optimizer / scheduler
    state, as in the reference fork) into `arguments` and later pass them to save(): the live objects must win"""
    from da_detect_amd.utils.checkpoint import Checkpointer

    model = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(model.parameters(), lr=0.5, momentum=0.9)
    ck = Checkpointer(model, opt, None, save_dir=str(tmp_path / "out" / "nested"), save_to_disk=True)
    ck.save("a", iteration=3)                                   # also creates the directory
    rest = ck.load(ck.get_checkpoint_file())
    assert rest["iteration"] == 3 and "optimizer" in rest       # reference behaviour: optimizer state is handed back
    opt.param_groups[0]["lr"] = 0.125
    ck.save("b", **rest)
    saved = torch.load(str(tmp_path / "out" / "nested" / "b.pth"), weights_only=False)
    assert saved["optimizer"]["param_groups"][0]["lr"] == 0.125 and saved["iteration"] == 3
    rest2 = ck.load(str(tmp_path / "out" / "nested" / "a.pth"), load_optimizer=True)
    assert "optimizer" not in rest2 and opt.param_groups[0]["lr"] == 0.5


def test_c2_weights_reach_deformable_convs():
    """with MODEL.RESNETS.STAGE_WITH_DCN the 3x3 conv of those stages lives under `conv2.conv` (DFConv2d);
    the vendored reference renames the pickle's keys accordingly (tools/cityscapes/.../c2_model_loading.py:146-170)"""
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.utils.c2_model_loading import rename_for_deformable_convs
    from da_detect_amd.utils.model_serialization import match_keys
    from golden.cases import fpn_dcn_da_cfg

    cfg = fpn_dcn_da_cfg()
    stages = cfg.MODEL.RESNETS.STAGE_WITH_DCN
    assert any(stages)
    loaded = {"layer%d.0.conv2.weight" % s: 0 for s in (1, 2, 3, 4)}
    loaded.update({"layer%d.0.conv1.weight" % s: 0 for s in (1, 2, 3, 4)})
    renamed = rename_for_deformable_convs(loaded, stages)
    for s, with_dcn in enumerate(stages, 1):
        assert ("layer%d.0.conv2.conv.weight" % s in renamed) == bool(with_dcn)
        assert ("layer%d.0.conv2.weight" % s in renamed) != bool(with_dcn)
        assert "layer%d.0.conv1.weight" % s in renamed
    model_keys = sorted(build_detection_model(cfg).state_dict().keys())
    dcn_keys = [k for k in model_keys if k.endswith(".conv2.conv.weight")]
    assert dcn_keys
    full = {k.split("backbone.body.")[-1].replace(".conv2.conv.", ".conv2."): 0 for k in dcn_keys}
    matches = match_keys(model_keys, sorted(rename_for_deformable_convs(full, stages)))
    for k in dcn_keys:
        assert matches.get(k) is not None, "%s keeps its random init" % k
