"""Launch time of the large-tile forward kernels against the number of K parts per tile, on the step's own shapes.

The 256 x 256 grid of the box head (M = 256 ROIs x 7 x 7 = 12544 rows: 49 row tiles) leaves CUs without a tile at the
planned part counts (98 tiles x 2 parts = 196 workgroups on 256 CUs; 392 tiles = 1.53 rounds): does another part count,
or the 256 x 128 tile, buy the idle share back?

usage (GPU box): python tools/probes/big_splits_sweep.py [--reps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C, _lib  # noqa: E402

CL = torch.channels_last
SHAPES = [  # N, Cin, H, W, Cout, k, pad
    (256, 512, 7, 7, 512, 3, 1),      # res5 3x3
    (256, 2048, 7, 7, 512, 1, 0),     # res5 conv1
    (256, 512, 7, 7, 2048, 1, 0),     # res5 conv3
    (256, 1024, 7, 7, 2048, 1, 0),    # res5 downsample (stride folded into ROIAlign)
    (256, 2048, 7, 7, 1024, 1, 0),    # its data gradient
    (256, 1024, 7, 7, 512, 1, 0),
    (1, 1024, 64, 128, 1024, 3, 1),   # RPN 3x3
    (2, 256, 64, 128, 256, 3, 1),     # res4 3x3
    (2, 1024, 64, 128, 256, 1, 0),    # res4 conv1
    (2, 128, 128, 256, 128, 3, 1),    # res3 3x3
    (2, 256, 64, 128, 1024, 1, 0),    # res4 conv3 (+ residual): short reductions, served by the weight-stationary kernel
    (2, 128, 128, 256, 512, 1, 0),    # res3 conv3 (+ residual)
    (2, 512, 64, 128, 256, 1, 0),     # res4 conv1 of the first block
]
SHORT = 3                              # the last SHORT shapes run with a residual addend, like their layers


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for si, (N, Cin, H, W, Cout, k, pad) in enumerate(SHAPES):
        x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
        w = (torch.randn((Cout, Cin, k, k), device=dev) * (2.0 / (Cin * k * k)) ** 0.5).contiguous(memory_format=CL)
        M, K = N * H * W, Cin * k * k
        add = (torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL)
               if si >= len(SHAPES) - SHORT and si != len(SHAPES) - 1 else None)
        gf = 2.0 * M * Cout * K * 1e-9
        cells = []
        lib.dadet_set_big_gemm(1)
        os.environ.pop("DADET_BIG_SPLITS", None)
        us = timed(lambda: _C.conv_forward(x, w, pad=pad, addend=add), args.reps)
        cells.append("plan %.1f us (%.0f TF/s)" % (us, gf / us * 1e3))
        lib.dadet_set_big_gemm(2)
        for tile_n in (256, 128):
            os.environ["DADET_BIG_TILE_N"] = str(tile_n)
            tiles = -(-M // 256) * -(-Cout // tile_n)
            row = []
            for s in (1, 2, 3, 4, 5, 6, 8):
                if K // 32 // s < 2 and s > 1:
                    continue
                os.environ["DADET_BIG_SPLITS"] = str(s)
                us = timed(lambda: _C.conv_forward(x, w, pad=pad, addend=add), args.reps)
                row.append("S=%d(%d wg) %.1f" % (s, tiles * s, us))
            cells.append("tile %d: " % tile_n + "  ".join(row))
        os.environ.pop("DADET_BIG_SPLITS", None)
        os.environ.pop("DADET_BIG_TILE_N", None)
        lib.dadet_set_big_gemm(1)
        print("M=%d N=%d K=%d k%d | " % (M, Cout, K, k) + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
