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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="img_only", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--overlap", action="store_true", help="keep the weight-gradient lane on (concurrent kernels)")
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
    for _ in range(args.steps):
        train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    _C.PROFILER = None
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
