"""what a grouped data-gradient + weight-gradient launch could return: the two GEMMs of a backward layer (both consume the
same output gradient, neither feeds the other) one after the other on one stream, against the same two on two streams
(their workgroups then share the chip: one fill and one drain instead of two) — an upper bound for a single launch whose
tile scheduler covers both problems"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from da_detect_amd import _C

dev = torch.device("cuda:0")
CL = torch.channels_last
side = torch.cuda.Stream(dev)


def timeit(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


LAYERS = [  # name, N, H, W, Cin (forward), Cout (forward), k
    ("res4 conv3 1x1 256->1024", 2, 64, 128, 256, 1024, 1),
    ("res4 conv2 3x3 256->256", 2, 64, 128, 256, 256, 3),
    ("res4 conv1 1x1 1024->256", 2, 64, 128, 1024, 256, 1),
    ("res3 conv2 3x3 128->128", 2, 128, 256, 128, 128, 3),
    ("res3 conv3 1x1 128->512", 2, 128, 256, 128, 512, 1),
    ("res5 conv2 3x3 512->512 (256 ROIs)", 256, 7, 7, 512, 512, 3),
    ("res5 conv3 1x1 512->2048 (256 ROIs)", 256, 7, 7, 512, 2048, 1),
    ("res5 conv1 1x1 2048->512 (256 ROIs)", 256, 7, 7, 2048, 512, 1),
]
for name, N, H, W, Cin, Cout, k in LAYERS:
    pad = k // 2
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    gy = torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
    wt = _C.conv_weight_transpose(w)
    mask = torch.randn((N, Cin, H, W), device=dev).clamp_min(0).contiguous(memory_format=CL)
    gx = torch.empty_like(x)
    dw = torch.zeros_like(w)

    def dgrad():
        _C.conv_forward(gy, wt, pad=pad, relu_mode=2, mask_ref=mask, out=gx)

    def wgrad():
        _C.conv_wgrad(x, gy, tuple(w.shape), 1, pad, dw=dw, accumulate=True)

    def serial():
        wgrad()
        dgrad()

    def two_streams():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        torch.cuda.current_stream().wait_stream(side)

    td, tw, ts, tp = timeit(dgrad), timeit(wgrad), timeit(serial), timeit(two_streams)
    print("%-40s dgrad %6.1f  wgrad %6.1f  one stream %6.1f  two streams %6.1f us  (%+.0f%%)" % (
        name, td, tw, ts, tp, 100 * (ts / tp - 1)), flush=True)
