"""Per-shape table of the GEMM launches of one training step (forward / data-gradient / weight-gradient):
launches, time, algorithmic TFLOP/s and GB/s, and the roofline bound max(flops / MFMA peak, bytes / HBM peak) next
to the measured time.  The weight-gradient lane is switched off so that no second GEMM shares the GPU with the
bracketed one.  Usage (GPU box): python tools/gemm_table.py [--workload img_only] [--steps 3] [--top 40]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def holes_report(prof, step_marks):
    """stretches of the LAST step in which no GEMM kernel runs (HIP events around every GEMM launch, no tracer: the
    host runs at its normal speed), with the GEMMs on either side"""
    t0 = prof.origin.elapsed_time(step_marks[-2])
    t1 = prof.origin.elapsed_time(step_marks[-1])
    spans = []
    for name, recs in prof.records.items():
        hosts = prof.host_times.get(name, [None] * len(recs))
        for (s, e, _), h in zip(recs, hosts):
            a, b = prof.origin.elapsed_time(s), prof.origin.elapsed_time(e)
            if a >= t0 and b <= t1 + 1.0:
                spans.append((a, b, name, h))
    spans.sort()
    holes, cur_e, cur_name = [], t0, "step start"
    for a, b, name, h in spans:
        if a > cur_e:
            # lead = how long before its start on the GPU the host had issued the launch that ends the stretch
            holes.append((a - cur_e, cur_e - t0, cur_name, "%s [issued %.2f ms earlier]" % (name, a - h)
                          if h is not None else name))
        if b > cur_e:
            cur_e, cur_name = b, name
    if t1 > cur_e:
        holes.append((t1 - cur_e, cur_e - t0, cur_name, "step end"))
    total = sum(h[0] for h in holes)
    print("last step %.2f ms: no GEMM running for %.2f ms in %d stretches" % (t1 - t0, total, len(holes)))
    for d, at, before, after in sorted(holes, reverse=True)[:25]:
        print("  %7.1f us at +%6.2f ms   after %-62s before %s" % (d * 1e3, at, before.replace("conv_", "")[:62],
                                                                  after.replace("conv_", "")[:90]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="img_only", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--overlap", action="store_true", help="keep the weight-gradient lane on (concurrent kernels)")
    ap.add_argument("--holes", action="store_true", help="list the stretches of the last step without a GEMM kernel")
    args = ap.parse_args()
    from da_detect_amd import _C
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step
    from da_detect_amd.utils import streams

    device = torch.device("cuda", 0)
    yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[args.workload]
    c, model, opt, _ = bench.build(yaml_path, device, seed=100, overrides=overrides)
    enable_overlapped_rpn_backward(model, True)
    images, targets = make_batch(c, images_per_gpu, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
    for _ in range(3):
        train_step(model, opt, images, targets)
    streams.WGRAD_OVERLAP = bool(args.overlap)
    prof = _C.KernelProfiler()
    prof.detail = True
    _C.PROFILER = prof
    step_marks = []
    for _ in range(args.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        step_marks.append(ev)
        train_step(model, opt, images, targets)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    step_marks.append(ev)
    torch.cuda.synchronize()
    _C.PROFILER = None
    if args.holes:
        holes_report(prof, step_marks)
        return
    table = prof.summary()
    mode = _C.get_gemm_mode()
    peak = bench.FP32_MFMA_PEAK_TFLOPS if mode == 0 else bench.BF16_MFMA_PEAK_TFLOPS / bench.GEMM_MODES[mode][1]
    hbm = 8.0e12
    rows = []
    for name, k in table.items():
        per_step = k["launches"] / args.steps
        bound_ms = max(k["work_per_launch"] / (peak * 1e12), k["bytes_per_launch"] / hbm) * 1e3
        rows.append((k["total_ms"] / args.steps, name, per_step, k["avg_ms"], k["work_per_launch"] / 1e9,
                     k["bytes_per_launch"] / 1e6, k["achieved"] / 1e12,
                     k["bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9, bound_ms))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    lost = sum(r[0] - r[8] * r[2] for r in rows)
    print("GEMM time per step %.2f ms, of which above the roofline bound %.2f ms (MFMA peak %.1f TF/s algorithmic, "
          "HBM 8 TB/s)" % (total, lost, peak))
    print("%-46s %-34s %5s %8s %8s %8s %8s %8s %8s %6s %8s" % ("kernel", "shape", "n", "ms/step", "avg ms", "GF",
                                                               "MB", "TF/s", "GB/s", "eff", "lost ms"))
    for r in rows[:args.top]:
        kern, _, shape = r[1].partition("|")
        print("%-46s %-34s %5.1f %8.3f %8.4f %8.2f %8.1f %8.1f %8.0f %6.2f %8.3f" % (
            kern, shape, r[2], r[0], r[3], r[4], r[5], r[6], r[7], r[8] / r[3], r[0] - r[8] * r[2]))


if __name__ == "__main__":
    main()
