"""Drop-in aliasing: after `compat.install()`, `import maskrcnn_benchmark.<x>` resolves to `da_detect_amd.<x>`.

The reference's entry points import e.g. `maskrcnn_benchmark.config.cfg`, `maskrcnn_benchmark.modeling.detector.
build_detection_model`, `maskrcnn_benchmark.solver.make_optimizer`, `maskrcnn_benchmark.structures.image_list.
to_image_list`, `maskrcnn_benchmark.layers.*`, `maskrcnn_benchmark._C` (reference: tools/train_net_triplet.py:16-36,
maskrcnn_benchmark/layers/nms.py:2).  This package mirrors those module paths one to one, so aliasing the package
prefix is enough.  Every `maskrcnn_benchmark` name the reference's tools/train_net_triplet.py imports, and every call it
makes on them, is covered by tests/test_api_surface.py against a fixture recorded from the reference's sources.

`timm.scheduler.cosine_lr.CosineLRScheduler` (imported inside the reference's train(), tools/train_net_triplet.py:67)
is a third-party class; when timm is not installed the alias below maps that module path to this package's restatement
of the schedule (solver/lr_scheduler.py)."""
import importlib
import sys
import types

_SUBMODULES = [
    "_C", "config", "config.defaults", "layers", "layers.roi_align", "layers.roi_pool", "layers.dcn", "layers.misc", "structures",
    "structures.bounding_box", "structures.boxlist_ops", "structures.image_list", "modeling", "modeling.registry",
    "modeling.box_coder", "modeling.matcher", "modeling.balanced_positive_negative_sampler", "modeling.poolers",
    "modeling.utils", "modeling.backbone", "modeling.backbone.resnet", "modeling.backbone.backbone", "modeling.rpn",
    "modeling.rpn.rpn", "modeling.rpn.anchor_generator", "modeling.rpn.inference", "modeling.rpn.loss",
    "modeling.rpn.utils", "modeling.roi_heads", "modeling.roi_heads.roi_heads", "modeling.roi_heads.box_head",
    "modeling.roi_heads.box_head.box_head", "modeling.roi_heads.box_head.loss",
    "modeling.roi_heads.box_head.inference", "modeling.roi_heads.box_head.roi_box_feature_extractors",
    "modeling.roi_heads.box_head.roi_box_predictors", "modeling.da_heads", "modeling.da_heads.da_heads",
    "modeling.da_heads.loss", "modeling.detector", "modeling.detector.detectors",
    "modeling.detector.generalized_rcnn", "solver", "solver.build", "solver.lr_scheduler", "engine",
    "engine.trainer", "engine.inference", "data", "data.build", "data.transforms", "data.samplers", "data.collate_batch", "data.datasets", "utils", "utils.comm", "utils.registry", "utils.checkpoint", "utils.model_serialization", "utils.c2_model_loading",
    "utils.imports", "utils.env", "utils.logger", "utils.miscellaneous", "utils.metric_logger", "utils.collect_env",
    "config.paths_catalog", "engine.trainer", "parallel", "parallel.reducer",
]


def _alias_timm():
    try:
        import timm.scheduler.cosine_lr  # noqa: F401
        return
    except Exception:
        pass
    from .solver.lr_scheduler import CosineLRScheduler

    pkg, sched, cos = (types.ModuleType(n) for n in ("timm", "timm.scheduler", "timm.scheduler.cosine_lr"))
    cos.CosineLRScheduler = sched.CosineLRScheduler = CosineLRScheduler
    pkg.scheduler, sched.cosine_lr = sched, cos
    pkg.__dadet_alias__ = True
    sys.modules.update({"timm": pkg, "timm.scheduler": sched, "timm.scheduler.cosine_lr": cos})


def install(prefix="maskrcnn_benchmark"):
    if prefix in sys.modules and not getattr(sys.modules[prefix], "__dadet_alias__", False):
        raise RuntimeError("%s is already imported from elsewhere; install the alias before importing it" % prefix)
    root = importlib.import_module("da_detect_amd")
    root.__dadet_alias__ = True
    sys.modules[prefix] = root
    for sub in _SUBMODULES:
        mod = importlib.import_module("da_detect_amd." + sub)
        sys.modules[prefix + "." + sub] = mod
    _alias_timm()
    return root
