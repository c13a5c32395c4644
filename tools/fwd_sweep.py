"""Tile-variant sweep of the forward / data-gradient GEMM (0: 128x128, 1: 128x64, 2: 64x64 tiles) over the step's
shapes, next to the library's own choice.  Usage (GPU box): python tools/fwd_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402
from tools.wgrad_sweep import timeit  # noqa: E402

CL = torch.channels_last
# (name, N, H, W, Cin, Cout, k, stride, with addend, launches per step)
SHAPES = [
    ("res5 3x3", 512, 7, 7, 512, 512, 3, 1, False, 6),
    ("rpn 3x3", 2, 64, 128, 1024, 1024, 3, 1, False, 2),
    ("res5 1x1 512->2048 +add", 512, 7, 7, 512, 2048, 1, 1, True, 5),
    ("res5 1x1 2048->512", 512, 7, 7, 2048, 512, 1, 1, False, 5),
    ("res5 1x1 2048->1024", 512, 7, 7, 2048, 1024, 1, 1, False, 1),
    ("res5 ds 1024->2048 s2", 512, 14, 14, 1024, 2048, 1, 2, False, 1),
    ("res5 c1 1024->512 s2", 512, 14, 14, 1024, 512, 1, 2, False, 1),
    ("res4 3x3", 2, 64, 128, 256, 256, 3, 1, False, 12),
    ("res4 1x1 256->1024 +add", 2, 64, 128, 256, 1024, 1, 1, True, 11),
    ("res4 1x1 1024->256", 2, 64, 128, 1024, 256, 1, 1, False, 11),
    ("res4 1x1 512->1024", 2, 64, 128, 512, 1024, 1, 1, False, 2),
    ("da 1x1 1024->512", 2, 64, 128, 1024, 512, 1, 1, False, 2),
    ("res3 3x3", 2, 128, 256, 128, 128, 3, 1, False, 8),
    ("res3 1x1 128->512 +add", 2, 128, 256, 128, 512, 1, 1, True, 7),
    ("res3 1x1 512->128", 2, 128, 256, 512, 128, 1, 1, False, 7),
    ("res3 ds 256->512 s2", 2, 256, 512, 256, 512, 1, 2, False, 1),
    ("res2 3x3", 2, 256, 512, 64, 64, 3, 1, False, 3),
    ("res2 1x1 64->256 +add", 2, 256, 512, 64, 256, 1, 1, True, 4),
    ("res2 1x1 256->64", 2, 256, 512, 256, 64, 1, 1, False, 2),
    ("rpn 1x1 dgrad 76->1024", 2, 64, 128, 76, 1024, 1, 1, False, 1),
]


def main():
    dev = torch.device("cuda", 0)
    print("%-28s %8s %8s %8s %8s   (ms; * = best)" % ("shape", "library", "128x128", "128x64", "64x64"))
    lib_total = best_total = 0.0
    for name, N, H, W, Cin, Cout, k, stride, add, per_step in SHAPES:
        pad = k // 2
        x = torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=CL)
        y = _C.conv_forward(x, w, stride=stride, pad=pad)
        addend = torch.randn_like(y) if add else None
        row = []
        for v in (None, 0, 1, 2):
            if v is None:
                os.environ.pop("DADET_FWD_VARIANT", None)
            else:
                os.environ["DADET_FWD_VARIANT"] = str(v)
            row.append(timeit(lambda: _C.conv_forward(x, w, addend=addend, stride=stride, pad=pad, relu_mode=1)))
        os.environ.pop("DADET_FWD_VARIANT", None)
        best = min(row[1:])
        lib_total += row[0] * per_step
        best_total += best * per_step
        print("%-28s %8.4f " % (name, row[0]) + " ".join("%7.4f%s" % (t, "*" if t == best else " ") for t in row[1:]))
    print("per step: library's choice %.3f ms, best variant per shape %.3f ms" % (lib_total, best_total))


if __name__ == "__main__":
    main()
