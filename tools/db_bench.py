"""A/B of the double-buffered K-step-16 variant (DADET_DB=1) against the single-buffer 128x128 kernel (stream-K off for
both) + correctness of the variant against it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


SHAPES = [  # name, N, Cin, H, W, Cout, k, pad
    ("rpn 3x3 1024->1024 (1024 tiles)", 2, 1024, 64, 128, 1024, 3, 1),
    ("res5 3x3 512->512 (784 tiles)", 512, 512, 7, 7, 512, 3, 1),
    ("res5 1x1 2048->512 (784 tiles)", 512, 2048, 7, 7, 512, 1, 0),
    ("res5 1x1 512->2048 (3136 tiles)", 512, 512, 7, 7, 2048, 1, 0),
    ("res4 3x3 256->256 (256 tiles)", 2, 256, 64, 128, 256, 3, 1),
    ("res4 1x1 1024->256 (256 tiles)", 2, 1024, 64, 128, 256, 1, 0),
    ("da img 1x1 1024->512 (512 tiles)", 2, 1024, 64, 128, 512, 1, 0),
    ("res3 3x3 128->128 (512 tiles)", 2, 128, 128, 256, 128, 3, 1),
    ("ragged 3x3 48->200, 37x53", 3, 48, 37, 53, 200, 3, 1),
]
os.environ["DADET_STREAMK"] = "0"
for name, N, Cin, H, W, Cout, k, pad in SHAPES:
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.02).contiguous(memory_format=CL)
    bias = torch.randn(Cout, device=dev)
    res, out = [], []
    for flag in ("0", "1"):
        os.environ["DADET_DB"] = flag
        out.append(_C.conv_forward(x, w, pad=pad, bias=bias, relu_mode=1))
        y = out[-1].clone()
        res.append(timeit(lambda: _C.conv_forward(x, w, pad=pad, out=y)))
    err = float((out[0] - out[1]).abs().max()) / (float(out[0].abs().max()) + 1e-30)
    gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
    print("%-36s single %.4f ms (%.0f TF/s)   double-buffered %.4f ms (%.0f TF/s)   %+.1f%%   max rel diff %.1e" % (
        name, res[0], gf / res[0], res[1], gf / res[1], 100 * (res[0] / res[1] - 1), err))
