"""aligned-triplet step: do gradients change when the host synchronises at chosen points?  (hunting a cross-stream hazard:
the fully serialised run matches the fp64 oracle, the normal one is 1e-3 off)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.modeling.roi_heads.box_head import box_head as bh  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from golden.cases import case_cfg  # noqa: E402
from golden.fill import fill_state_dict  # noqa: E402

device = torch.device("cuda", 0)
POINTS = set()


def maybe(name):
    if name in POINTS or "all" in POINTS:
        torch.cuda.synchronize()


orig_fwd = bh.ROIBoxHead.forward


def fwd(self, features, proposals, targets=None):
    maybe("before_box_head")
    out = orig_fwd(self, features, proposals, targets)
    maybe("after_box_head")
    return out


bh.ROIBoxHead.forward = fwd
orig_sub = None
from da_detect_amd import _lib  # noqa: E402

orig_call = _lib.call
SYNC_NAMES = set()


def call(name, *args):
    if "every_call" in POINTS or name in SYNC_NAMES:
        torch.cuda.synchronize()
    r = orig_call(name, *args)
    if "every_call" in POINTS or name in SYNC_NAMES:
        torch.cuda.synchronize()
    return r


_lib.call = call


def run(points, seed=11, case="da_triplet_aligned", overrides=("MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", 0.0)):
    POINTS.clear()
    POINTS.update(points)
    c = case_cfg(case)
    c.merge_from_list(list(overrides))
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed))
    model = model.to(device).train()
    images, targets = make_batch(c, 3, 192, 320, seed=seed, device=device)
    opt = make_optimizer(c, model)
    opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
    enable_overlapped_rpn_backward(model)
    ev = model.roi_heads.box.loss_evaluator
    o_sub = ev.subsample

    def sub(*a, **k):
        r = o_sub(*a, **k)
        maybe("after_subsample")
        return r

    ev.subsample = sub
    torch.manual_seed(seed)
    train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}


def diff(a, b):
    worst, name = 0.0, None
    for n, g in a.items():
        d = float((b[n].double() - g.double()).norm()) / (float(g.double().norm()) + 1e-30)
        if d > worst:
            worst, name = d, n
    return worst, name


base = run({"every_call"})
again = run({"every_call"})
print("every-call-sync vs itself:", diff(base, again))
g = run(set())
print("no sync vs every-call-sync: worst %.2e at %s" % diff(base, g))
names = ["dadet_sample_rois", "dadet_box_match_encode", "dadet_roi_align_forward_sub", "dadet_roi_align_forward_ws",
         "dadet_conv_forward", "dadet_fast_rcnn_loss_rows", "dadet_nms", "dadet_rpn_decode_clip", "dadet_sample_anchors",
         "dadet_rpn_anchor_targets", "dadet_conv_wgrad_partials", "dadet_conv_wgrad", "dadet_roi_align_backward_sub",
         "dadet_rpn_loss_rows", "dadet_scatter_pixel_taps_add", "dadet_gather_pixel_taps", "dadet_conv_wgrad_reduce_batch"]
for nme in names:
    SYNC_NAMES.clear()
    SYNC_NAMES.add(nme)
    g = run(set())
    print("%-34s synced: vs every-call-sync worst %.2e at %s" % ((nme,) + diff(base, g)))
