"""How far ahead of the GPU is the host at the phase boundaries of one training step?

At every boundary (entry / exit of the backbone, the RPN, the proposal selection, the box head's sampler, its feature
extractor, the DA heads, backward, the optimizer) the host notes its own clock and records an event on the stream it is
issuing to.  After the step: host time and GPU time of every boundary from a common origin (a synchronised point), and
    lead = GPU time - host time
A lead of a few microseconds means the GPU executed the boundary as soon as it was issued — it had been WAITING for the
host; a lead of milliseconds means the host is far ahead and the GPU never starves there.

usage (GPU box): python tools/probes/host_lead.py [--workload img_only] [--hw 1024x2048]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine import trainer  # noqa: E402

MARKS = []
ON = [False]


def mark(name):
    if not ON[0]:
        return
    ev = torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    ev.record(torch.cuda.current_stream())
    MARKS.append((name, t, ev, int(torch.cuda.current_stream().cuda_stream)))


def wrap(obj, attr, name):
    fn = getattr(obj, attr)

    def wrapped(*a, **k):
        mark(name + " >")
        try:
            return fn(*a, **k)
        finally:
            mark(name + " <")

    if isinstance(fn, torch.nn.Module):       # a child module: wrap its forward instead of replacing the attribute
        fn, obj, attr = fn.forward, fn, "forward"
    setattr(obj, attr, wrapped)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="img_only")
    ap.add_argument("--hw", default="1024x2048")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--lane", type=int, default=17000, help="weight-gradient lane for GEMMs of up to this many rows (0: one stream)")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[args.workload]
    c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
    h, w = [int(v) for v in args.hw.lower().split("x")]
    images, targets = make_batch(c, images_per_gpu, h, w, seed=100, device=device)
    net = model
    trainer.enable_overlapped_rpn_backward(model, True)      # the schedule bench.py and the training loops run
    from da_detect_amd.utils import streams
    streams.WGRAD_LANE_ROWS = args.lane
    wrap(net.backbone, "forward", "backbone")
    wrap(net.rpn, "forward", "rpn")
    wrap(net.rpn, "_prepare_loss_targets", "  rpn loss targets (side stream, host sync)")
    wrap(net.rpn, "box_selector_train", "  proposal selection")
    wrap(net.rpn, "_finish_overlapped", "  rpn losses + early backward + selection")
    wrap(net.rpn.head, "forward", "  rpn head")
    box = net.roi_heads.box
    wrap(net.roi_heads, "forward", "roi_heads")
    wrap(box.loss_evaluator, "subsample", "  box sampler (host sync)")
    wrap(box.feature_extractor, "forward", "  box feature extractor (ROIAlign + res5)")
    wrap(box.predictor, "forward", "  box predictor")
    wrap(box.loss_evaluator, "subsample_for_da", "  DA ROI draw")
    wrap(box.feature_extractor.pooler, "forward", "    pooler")
    wrap(box.feature_extractor.pooler, "convert_to_roi_format", "      roi format")
    if hasattr(box.feature_extractor, "head"):
        wrap(box.feature_extractor.head, "forward", "    res5 head")
    if net.da_heads:
        wrap(net.da_heads, "forward", "da_heads")
    wrap(opt, "step", "optimizer.step")
    wrap(opt, "zero_grad", "zero_grad")
    orig_backward = torch.Tensor.backward

    def backward(self, *a, **k):
        mark("backward >")
        try:
            return orig_backward(self, *a, **k)
        finally:
            mark("backward <")

    torch.Tensor.backward = backward
    for _ in range(args.steps):
        trainer.train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    rows_all = []
    for rep in range(3):
        torch.cuda.synchronize()
        del MARKS[:]
        e0 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(torch.cuda.current_stream())
        trainer.train_step(model, opt, images, targets)      # unmarked: the marked step starts with the lead a loop has
        trainer.train_step(model, opt, images, targets)
        ON[0] = True
        mark("step begins")
        trainer.train_step(model, opt, images, targets)
        mark("step issued")
        ON[0] = False
        t_issued = time.perf_counter()
        torch.cuda.synchronize()
        t_done = time.perf_counter()
        rows = [(n, (t - t0) * 1e3, e0.elapsed_time(ev), s) for n, t, ev, s in MARKS]
        rows_all.append((rows, (t_issued - t0) * 1e3, (t_done - t0) * 1e3))
    rows, issued, done = rows_all[-1]
    main_stream = rows[0][3]
    print("# tools/probes/host_lead.py --workload %s --hw %s: the third of three steps issued back to back from an idle GPU"
          " (times from the start of the first); host finished issuing at %.2f ms, GPU finished at %.2f ms" % (args.workload, args.hw, issued, done))
    print("%-58s %10s %10s %10s   %s" % ("boundary", "host ms", "GPU ms", "lead ms", "stream"))
    for n, th, tg, s in rows:
        print("%-58s %10.3f %10.3f %10.3f   %s" % (n, th, tg, tg - th, "compute" if s == main_stream else "side"))
    print("\nall three repetitions: host issue time / GPU end: " + ", ".join("%.2f / %.2f ms" % (i, d) for _, i, d in rows_all))


if __name__ == "__main__":
    main()
