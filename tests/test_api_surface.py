"""Drop-in Python boundary: every `maskrcnn_benchmark.*` name the reference's training entry point
imports resolves after `da_detect_amd.compat.install()`, has the reference's parameters in the reference's order, and
every call the script makes binds.  Fixture: tests/golden/reference_api_surface.json, recorded from the reference's
SOURCE by tests/golden/make_golden_api.py (tools/train_net_triplet.py:16-36 imports, :60-325 calls)."""
import importlib
import inspect
import json
import logging
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
API = json.load(open(os.path.join(HERE, "golden", "reference_api_surface.json")))


@pytest.fixture(scope="module")
def aliased():
    from da_detect_amd import compat

    compat.install()
    yield
    for k in [k for k in sys.modules if k == "maskrcnn_benchmark" or k.startswith("maskrcnn_benchmark.")]:
        del sys.modules[k]


def _resolve(rec):
    return getattr(importlib.import_module(rec["module"]), rec["name"])


@pytest.mark.parametrize("rec", API["imports"], ids=lambda r: "%s.%s" % (r["module"], r["name"]))
def test_every_imported_name_resolves_with_the_reference_parameters(aliased, rec):
    obj = _resolve(rec)
    if rec["kind"] == "third_party":
        assert callable(obj)
        return
    if rec["kind"] == "object":
        assert rec["name"] == "cfg" and hasattr(obj, "merge_from_file") and hasattr(obj, "merge_from_list")
        return
    assert callable(obj), rec
    want = rec["signature"]
    if want is None:
        return
    target = obj.__init__ if rec["kind"] == "class" else obj
    got = [p for p in inspect.signature(target).parameters.values()]
    got_names = [p.name for p in got if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    want_names = [p["name"] for p in want["params"]]
    # the reference's parameters, in the reference's order, form a prefix (this package may append optional ones)
    assert got_names[: len(want_names)] == want_names, (rec["name"], got_names, want_names)
    for p in got[len(want_names):]:
        assert p.default is not inspect.Parameter.empty or p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL), (rec["name"], p)
    # a parameter that is optional in the reference is optional here
    by_name = {p.name: p for p in got}
    for p in want["params"]:
        if p["default"] is not None:
            assert by_name[p["name"]].default is not inspect.Parameter.empty, (rec["name"], p["name"])
    if rec["kind"] == "class":
        for m in rec.get("methods", []):
            assert callable(getattr(obj, m, None)), "%s.%s missing" % (rec["name"], m)


@pytest.mark.parametrize("call", API["calls"], ids=lambda c: "%s@%d" % (c["name"], c["line"]))
def test_every_call_of_the_reference_script_binds(aliased, call):
    rec = [r for r in API["imports"] if r["name"] == call["name"]][0]
    obj = _resolve(rec)
    sig = inspect.signature(obj)
    sig.bind(*[None] * call["positional"], **{k: None for k in call["keywords"]})


def test_attribute_calls_of_the_script_exist(aliased):
    from maskrcnn_benchmark.config import cfg
    from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer

    for owner, attr in API["attribute_calls"]:
        if owner == "cfg":
            assert callable(getattr(cfg, attr)), attr
        elif owner == "checkpointer":
            assert callable(getattr(DetectronCheckpointer, attr)), attr
        elif owner == "logger":
            assert callable(getattr(logging.getLogger("x"), attr)), attr


def test_the_script_prologue_runs_on_the_aliased_package(tmp_path):
    """the part of the reference script's flow that needs no GPU, with its own call forms
    (tools/train_net_triplet.py:54-116, 300-330): config merge + OUTPUT_DIR join, mkdir, logger, env report, model,
    per-tensor optimizer, timm-style cosine schedule, checkpointer (no weight), meters"""
    code = r'''
import os, sys
sys.path.insert(0, %r)
from da_detect_amd import compat
compat.install()
from maskrcnn_benchmark.utils.env import setup_environment  # noqa
import torch
from maskrcnn_benchmark.config import cfg
from maskrcnn_benchmark.data import make_data_loader, make_data_loader_da
from maskrcnn_benchmark.engine.inference import inference
from maskrcnn_benchmark.engine.trainer import do_train, do_da_train
from maskrcnn_benchmark.modeling.detector import build_detection_model
from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer
from maskrcnn_benchmark.utils.collect_env import collect_env_info
from maskrcnn_benchmark.utils.comm import synchronize, get_rank
from maskrcnn_benchmark.utils.imports import import_file
from maskrcnn_benchmark.utils.logger import setup_logger
from maskrcnn_benchmark.utils.miscellaneous import mkdir
from maskrcnn_benchmark.solver import make_lr_scheduler
from maskrcnn_benchmark.solver import make_optimizer
from maskrcnn_benchmark.utils.metric_logger import (MetricLogger, TensorboardLogger)
cfg.merge_from_file(%r)
cfg.MODEL.OUTPUT_DIR = os.path.join(%r, cfg.MODEL.OUTPUT_SAVE_NAME)
cfg.merge_from_list(["MODEL.WEIGHT", "", "SOLVER.MAX_ITER", 10])
mkdir(cfg.MODEL.OUTPUT_DIR)
mkdir(cfg.MODEL.OUTPUT_DIR)
logger = setup_logger("maskrcnn_benchmark", cfg.MODEL.OUTPUT_DIR, get_rank())
logger.info("Using {} GPUs".format(1))
assert "PyTorch version" in collect_env_info()
model = build_detection_model(cfg)
optimizer = make_optimizer(cfg, model)
from timm.scheduler.cosine_lr import CosineLRScheduler
scheduler = CosineLRScheduler(optimizer, t_initial=cfg.SOLVER.MAX_ITER, lr_min=cfg.SOLVER.LR_MIN,
                              warmup_lr_init=cfg.SOLVER.WARMUP_LR, warmup_t=cfg.SOLVER.WARMUP_ITERS, cycle_limit=1,
                              t_in_epochs=False)
arguments = {"iteration": 0}
checkpointer = DetectronCheckpointer(cfg, model, optimizer, scheduler, cfg.MODEL.OUTPUT_DIR, get_rank() == 0)
arguments.update(checkpointer.load(cfg.MODEL.WEIGHT))
meters = MetricLogger(delimiter="  ")
meters.update(loss=torch.tensor(2.0), time=0.5)
meters.update(loss=4.0, time=0.5)
assert abs(meters.loss.global_avg - 3.0) < 1e-6 and "loss: " in str(meters)
scheduler.step_update(3); scheduler.step(1)
checkpointer.save("model_{:07d}".format(3), **arguments)
assert os.path.exists(os.path.join(cfg.MODEL.OUTPUT_DIR, "model_0000003.pth"))
assert os.path.exists(os.path.join(cfg.MODEL.OUTPUT_DIR, "log.txt"))
synchronize()
print("PROLOGUE-OK")
''' % (ROOT, os.path.join(ROOT, "configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"),
       str(tmp_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "PROLOGUE-OK" in res.stdout, res.stderr[-3000:]


def test_reference_signature_loaders_from_the_path_catalog(tmp_path):
    """make_data_loader / make_data_loader_da with the reference's signatures (data/build.py:232,332): dataset NAMES from
    cfg.DATASETS resolved by DatasetCatalog of the file cfg.PATHS_CATALOG"""
    import numpy as np

    from da_detect_amd.config import cfg
    from da_detect_amd.data import make_data_loader, make_data_loader_da
    from test_data_pipeline import _write_coco

    rng = np.random.default_rng(0)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng, sizes=[(96, 192)] * 4) for k in ("src", "tgt", "aux")}
    catalog = tmp_path / "my_catalog.py"
    catalog.write_text(
        "class DatasetCatalog(object):\n"
        "    DATASETS = %r\n"
        "    @staticmethod\n"
        "    def get(name):\n"
        "        ann, root = DatasetCatalog.DATASETS[name]\n"
        "        return dict(factory='COCODataset', args=dict(root=root, ann_file=ann))\n"
        % {k + "_cocostyle": v for k, v in specs.items()})
    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, "configs/da_faster_rcnn/"
                                         "e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"))
    c.merge_from_list(["PATHS_CATALOG", str(catalog), "DATASETS.SOURCE_TRAIN", ("src_cocostyle",),
                       "DATASETS.TARGET_TRAIN", ("tgt_cocostyle",), "DATASETS.TARGET_TRAIN_negative", ("aux_cocostyle",),
                       "DATASETS.TEST", ("tgt_cocostyle",), "SOLVER.MAX_ITER", 3, "DATALOADER.NUM_WORKERS", 0,
                       "INPUT.MIN_SIZE_TRAIN", (96,), "INPUT.MAX_SIZE_TRAIN", 192, "INPUT.MIN_SIZE_TEST", 96,
                       "INPUT.MAX_SIZE_TEST", 192, "TEST.IMS_PER_BATCH", 1])
    src = make_data_loader(c, is_train=True, is_source=True, is_negative=False, is_distributed=False, start_iter=0)
    neg = make_data_loader(c, is_train=True, is_source=False, is_negative=True, is_distributed=False, start_iter=1)
    assert len(src) == 3 and len(list(neg)) == 2          # start_iter shortens the stream, not its nominal length
    images, targets, ids = next(iter(src))
    assert images.tensors.shape[0] == 1 and bool(targets[0].get_field("is_source").all())
    _, t_neg, _ = next(iter(neg))
    assert not bool(t_neg[0].get_field("is_source").any())
    val = make_data_loader(c, is_train=False, is_distributed=False, is_for_period=False)
    assert isinstance(val, list) and len(val) == 1 and len(val[0].dataset) == 4
    trip = make_data_loader_da(c, is_train=True, is_source=[True, False, False], is_negative=False, is_distributed=False,
                               start_iter=0)
    batch = next(iter(trip))
    assert len(batch) == 9
    s_img, s_tgt, p_img, p_tgt, n_img, n_tgt = batch[:6]
    assert bool(s_tgt[0].get_field("is_source").all()) and not bool(p_tgt[0].get_field("is_source").any())
    assert torch.equal(s_tgt[0].bbox, p_tgt[0].bbox) and torch.equal(s_tgt[0].bbox, n_tgt[0].bbox)   # build.py:34-46
    assert (s_img + p_img + n_img).tensors.shape[0] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("aligned", [False, True])
def test_reference_training_flow_on_the_aliased_package(tmp_path, aligned):
    """the reference script's train() body (tools/train_net_triplet.py:54-197) with its own call forms — catalog
    datasets, make_data_loader / make_data_loader_da, the 14-positional + 2-keyword do_da_train call — run through
    `maskrcnn_benchmark.*` aliases on the HIP device"""
    import numpy as np

    from test_data_pipeline import _write_coco

    rng = np.random.default_rng(5)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng, sizes=[(96, 192)] * 4) for k in ("src", "tgt", "aux")}
    catalog = tmp_path / "my_catalog.py"
    catalog.write_text(
        "class DatasetCatalog(object):\n"
        "    DATASETS = %r\n"
        "    @staticmethod\n"
        "    def get(name):\n"
        "        ann, root = DatasetCatalog.DATASETS[name]\n"
        "        return dict(factory='COCODataset', args=dict(root=root, ann_file=ann))\n"
        % {k + "_cocostyle": v for k, v in specs.items()})
    code = r'''
import os, sys
sys.path.insert(0, %(root)r)
from da_detect_amd import compat
compat.install()
from maskrcnn_benchmark.utils.env import setup_environment  # noqa
import torch
from maskrcnn_benchmark.config import cfg
from maskrcnn_benchmark.data import make_data_loader, make_data_loader_da
from maskrcnn_benchmark.engine.trainer import do_train, do_da_train
from maskrcnn_benchmark.modeling.detector import build_detection_model
from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer
from maskrcnn_benchmark.utils.comm import synchronize, get_rank
from maskrcnn_benchmark.utils.logger import setup_logger
from maskrcnn_benchmark.utils.miscellaneous import mkdir
from maskrcnn_benchmark.solver import make_optimizer
from maskrcnn_benchmark.utils.metric_logger import MetricLogger
cfg.merge_from_file(%(yaml)r)
cfg.MODEL.OUTPUT_DIR = os.path.join(%(out)r, cfg.MODEL.OUTPUT_SAVE_NAME)
cfg.merge_from_list(["MODEL.WEIGHT", "", "SOLVER.MAX_ITER", 4, "SOLVER.CHECKPOINT_PERIOD", 2, "PATHS_CATALOG", %(cat)r,
                     "DATASETS.SOURCE_TRAIN", ("src_cocostyle",), "DATASETS.TARGET_TRAIN", ("tgt_cocostyle",),
                     "DATASETS.TARGET_TRAIN_negative", ("aux_cocostyle",), "DATASETS.TEST", ("tgt_cocostyle",),
                     "DATALOADER.NUM_WORKERS", 0, "INPUT.MIN_SIZE_TRAIN", (96,), "INPUT.MAX_SIZE_TRAIN", 192,
                     "INPUT.MIN_SIZE_TEST", 96, "INPUT.MAX_SIZE_TEST", 192, "TEST.IMS_PER_BATCH", 1,
                     "MODEL.DA_HEADS.ALIGNMENT", %(aligned)r, "MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", %(insw)r])
mkdir(cfg.MODEL.OUTPUT_DIR)
logger = setup_logger("maskrcnn_benchmark", cfg.MODEL.OUTPUT_DIR, get_rank())
distributed = False
torch.manual_seed(100)
model = build_detection_model(cfg)
device = torch.device(cfg.MODEL.DEVICE)
model.to(device)
optimizer = make_optimizer(cfg, model)
from timm.scheduler.cosine_lr import CosineLRScheduler
scheduler = CosineLRScheduler(optimizer, t_initial=cfg.SOLVER.MAX_ITER, lr_min=cfg.SOLVER.LR_MIN,
                              warmup_lr_init=cfg.SOLVER.WARMUP_LR, warmup_t=2, cycle_limit=1, t_in_epochs=False)
arguments = {}
arguments["iteration"] = 0
output_dir = cfg.MODEL.OUTPUT_DIR
checkpointer = DetectronCheckpointer(cfg, model, optimizer, scheduler, output_dir, get_rank() == 0)
arguments.update(checkpointer.load(cfg.MODEL.WEIGHT))
checkpoint_period = cfg.SOLVER.CHECKPOINT_PERIOD
meters = MetricLogger(delimiter="  ")
data_loader_val = make_data_loader(cfg, is_train=False, is_distributed=distributed, is_for_period=False)
triplet_data_loading = cfg.MODEL.DA_HEADS.TRIPLET_USE
triplet_data_aligned = cfg.MODEL.DA_HEADS.ALIGNMENT
if triplet_data_aligned:
    Positive_target_data_loader = make_data_loader_da(cfg, is_train=True, is_source=[True, False, False],
                                                      is_negative=False, is_distributed=distributed,
                                                      start_iter=arguments["iteration"])
    source_data_loader = []
    Negative_target_data_loader = []
else:
    source_data_loader = make_data_loader(cfg, is_train=True, is_source=True, is_negative=False,
                                          is_distributed=distributed, start_iter=arguments["iteration"])
    Negative_target_data_loader = make_data_loader(cfg, is_train=True, is_source=False, is_negative=True,
                                                   is_distributed=distributed, start_iter=arguments["iteration"])
    Positive_target_data_loader = make_data_loader(cfg, is_train=True, is_source=False, is_negative=False,
                                                   is_distributed=distributed, start_iter=arguments["iteration"])
do_da_train(model, source_data_loader, Positive_target_data_loader, Negative_target_data_loader, data_loader_val,
            optimizer, scheduler, checkpointer, device, checkpoint_period, arguments, cfg, distributed, meters,
            triplet_data_loading=triplet_data_loading, triplet_data_aligned=triplet_data_aligned)
assert arguments["iteration"] == 3
assert os.path.exists(os.path.join(output_dir, "model_0000002.pth")) and os.path.exists(os.path.join(output_dir, "model_final.pth"))
print("FLOW-OK", str(meters))
''' % dict(root=ROOT, out=str(tmp_path), cat=str(catalog), aligned=bool(aligned), insw=1.0 if aligned else 0.0,
           yaml=os.path.join(ROOT, "configs/da_faster_rcnn/"
                                   "e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "FLOW-OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
    assert "triplet_loss_image: " in res.stdout and "Start evaluation on [Validation]" in res.stdout
    if aligned:
        assert "triplet_loss_instance: " in res.stdout


# ---- the native module: maskrcnn_benchmark._C of BOTH reference trees ---------------------------------------------------------
NATIVE = json.load(open(os.path.join(HERE, "golden", "reference_native_api.json")))
_NATIVE_FNS = {}
for _tree, _rec in NATIVE.items():
    for _f in _rec["functions"]:
        _NATIVE_FNS.setdefault(_f["name"], (_tree, _f))


@pytest.mark.parametrize("name", sorted(_NATIVE_FNS))
def test_native_module_has_every_bound_function_with_the_reference_arguments(name):
    """every `m.def` of csrc/vision.cpp:7-15 and of the vendored tools/cityscapes/.../csrc/vision.cpp:9-23 (14 names) exists in
    da_detect_amd._C, takes the reference's arguments positionally in the reference's order (this package may append optional
    ones), and — for the functions written against the vendored tree this round — under the reference's C++ parameter names.
    Fixture: tests/golden/reference_native_api.json (tests/golden/make_golden_native_api.py: parsed from the headers' text)."""
    from da_detect_amd import _C

    tree, rec = _NATIVE_FNS[name]
    fn = getattr(_C, name, None)
    assert callable(fn), "_C.%s missing (%s)" % (name, rec["header"])
    params = [p for p in inspect.signature(fn).parameters.values()
              if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    n = len(rec["params"])
    assert len(params) >= n, (name, [p.name for p in params], [p["name"] for p in rec["params"]])
    for p in params[n:]:
        assert p.default is not inspect.Parameter.empty, "%s: extra parameter %s must be optional" % (name, p.name)
    inspect.signature(fn).bind(*[None] * n)
    if name.startswith(("deform_", "modulated_")):
        assert [p.name for p in params[:n]] == [p["name"] for p in rec["params"]], name


@pytest.mark.parametrize("call", [c for r in NATIVE.values() for c in r["calls"]],
                         ids=lambda c: "%s@%s:%d" % (c["name"], os.path.basename(c["file"]), c["line"]))
def test_every_native_call_of_the_reference_layers_binds(call):
    """each `_C.name(...)` call form in the reference's layers/*.py and layers/dcn/*.py (argument count, keywords) binds to
    this package's function of that name"""
    from da_detect_amd import _C

    inspect.signature(getattr(_C, call["name"])).bind(*[None] * call["positional"], **{k: None for k in call["keywords"]})


def test_native_module_aliases_exist():
    from da_detect_amd import _C

    for r in NATIVE.values():
        for al in r["aliases"]:      # e.g. layers/nms.py: `nms = _C.nms`
            assert callable(getattr(_C, al["name"]))
