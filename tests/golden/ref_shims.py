"""Import shims that let the REFERENCE package (/root/reference, read-only) run on this image's CPU, used only
by tests/golden/make_golden.py in the authoring container (the reference never travels to the GPU box).
Nothing here touches the reference tree: missing third-party modules are stubbed in sys.modules, the built
reference extension oracle/_ref/ref_C.so is registered as maskrcnn_benchmark._C."""
import os
import sys
import types

REF_ROOT = "/root/reference"


def install():
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    if root not in sys.path:
        sys.path.insert(0, root)
    import numpy as np
    import torch

    # yacs: the repo's own CfgNode has the subset the reference uses
    from da_detect_amd.config.cfgnode import CfgNode
    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs_config
    # torch._six (removed from torch), np.float (removed from numpy), torch.cuda.FloatTensor on a CPU box
    six = types.ModuleType("torch._six")
    six.PY3, six.PY37, six.string_classes, six.int_classes = True, True, (str,), (int,)
    sys.modules["torch._six"] = six
    torch._six = six
    if not hasattr(np, "float"):
        np.float = float
    torch.cuda.FloatTensor = torch.FloatTensor
    # modules imported at module load by parts of the reference that are never executed here
    for name in ["cv2", "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval"]:
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["cv2"].log = lambda *a, **k: None
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    # the reference's compiled CPU operators
    from oracle import build_ref
    build_ref.build()
    ref_c = build_ref.load()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import maskrcnn_benchmark  # noqa: F401  (package __init__ is empty)
    sys.modules["maskrcnn_benchmark._C"] = ref_c
    maskrcnn_benchmark._C = ref_c
    return ref_c
