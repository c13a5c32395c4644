"""Sweep of the weight-gradient split count (workgroups just below multiples of the chip's slots) over the step's
weight-gradient shapes, next to the library's own plan.  Usage (GPU box): python tools/wgrad_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
# (name, N, H, W, Cin, Cout, k, stride, launches per step)
SHAPES = [
    ("res5 3x3", 512, 7, 7, 512, 512, 3, 1, 3),
    ("rpn 3x3", 2, 64, 128, 1024, 1024, 3, 1, 1),
    ("res5 1x1 512->2048", 512, 7, 7, 512, 2048, 1, 1, 3),
    ("res5 1x1 2048->512", 512, 7, 7, 2048, 512, 1, 1, 2),
    ("res5 ds 1024->2048 s2", 512, 14, 14, 1024, 2048, 1, 2, 1),
    ("res4 3x3", 2, 64, 128, 256, 256, 3, 1, 6),
    ("res3 3x3", 2, 128, 256, 128, 128, 3, 1, 4),
    ("res4 1x1 256->1024", 2, 64, 128, 256, 1024, 1, 1, 6),
    ("res4 1x1 1024->256", 2, 64, 128, 1024, 256, 1, 1, 5),
    ("res3 1x1 128->512", 2, 128, 256, 128, 512, 1, 1, 4),
    ("res3 1x1 512->128", 2, 128, 256, 512, 128, 1, 1, 3),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def planned_splits(x, gy, Cout, Cin, k, stride, pad):
    import ctypes
    from da_detect_amd import _lib
    d = _C._desc(x.shape[0], x.shape[2], x.shape[3], Cin, Cout, k, k, stride, pad, gy.shape[2], gy.shape[3])
    nbytes = ctypes.c_size_t(0)
    _lib.call("dadet_conv_wgrad_workspace_bytes", ctypes.byref(d), ctypes.byref(nbytes))
    return max(1, nbytes.value // (4 * Cout * Cin * k * k))


def main():
    dev = torch.device("cuda", 0)
    slots = [0, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096]
    print("%-26s %5s " % ("shape", "tiles") + "".join("%16s" % ("<=%d" % p if p else "plan") for p in slots))
    best_total = plan_total = 0.0
    for name, N, H, W, Cin, Cout, k, stride, per_step in SHAPES:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=CL)
        gy = torch.randn(N, Cout, Ho, Wo, device=dev).contiguous(memory_format=CL)
        tiles = -(-Cout // 128) * -(-(Cin * k * k) // 128)
        row = []
        for n in slots:
            s = max(1, n // tiles)
            if n:
                os.environ["DADET_WGRAD_SPLITS"] = str(s)
            else:
                os.environ.pop("DADET_WGRAD_SPLITS", None)   # the library's own plan
            ms = timeit(lambda: _C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride, pad))
            if not n:
                s = planned_splits(x, gy, Cout, Cin, k, stride, pad)
            row.append((s, ms))
        best_total += min(v for _, v in row) * per_step
        plan_total += row[0][1] * per_step
        print("%-26s %5d " % (name, tiles) + "".join("%16s" % ("s=%d %.4f" % v) for v in row))
    print("per step with the best plan of every shape: %.3f ms; with the library's plan: %.3f ms" % (best_total,
                                                                                                    plan_total))


if __name__ == "__main__":
    main()
