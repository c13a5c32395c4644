"""un-traced timing (HIP events only) of the stretch between the RPN head and the box head: when, relative to the start
of the step on the GPU, do (a) the RPN branch's backward end on the compute stream, (b) the proposal selection start / end
on the side stream, (c) the box head's ROIAlign start — and when did the HOST issue each of them"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.rpn import rpn as rpn_mod  # noqa: E402

device = torch.device("cuda", 0)
c, model, opt, reducer = bench.build(bench.YAML, device, seed=100)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, 2, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
marks = {}
host = {}
t_step = [0.0]


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream(device))
    marks[name] = e
    host[name] = (time.perf_counter() - t_step[0]) * 1e3


sel = model.rpn.box_selector_train
orig_sel = sel.forward


def sel_forward(*a, **k):
    mark("selection start (side)")
    r = orig_sel(*a, **k)
    mark("selection end (side)")
    return r


sel.forward = sel_forward
orig_bw = torch.autograd.backward
state = {"first": True}


def backward(*a, **k):
    r = orig_bw(*a, **k)
    if state["first"]:
        state["first"] = False
        mark("rpn backward end (main)")
    return r


torch.autograd.backward = backward
rpn_mod.torch.autograd.backward = backward
orig_roi = _C.roi_align_forward


def roi_fwd(*a, **k):
    mark("roi_align fwd start (main)")
    return orig_roi(*a, **k)


_C.roi_align_forward = roi_fwd
head = model.rpn.head
orig_head = head.forward


def head_fwd(*a, **k):
    r = orig_head(*a, **k)
    mark("rpn head fwd end (main)")
    return r


head.forward = head_fwd
for _ in range(8):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
for it in range(3):
    state["first"] = True
    marks.clear()
    t_step[0] = time.perf_counter()
    mark("step start")
    train_step(model, opt, images, targets)
    mark("step end")
    torch.cuda.synchronize()
    s = marks["step start"]
    print("step %d" % it)
    for k, e in sorted(marks.items(), key=lambda kv: s.elapsed_time(kv[1])):
        print("   GPU +%7.2f ms   host issued at +%7.2f ms   %s" % (s.elapsed_time(e), host[k], k))
