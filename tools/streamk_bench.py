"""A/B of the stream-K tail (DADET_STREAMK=0/1) on the forward / data-gradient GEMM shapes of the BASELINE step whose
128 x 128 tile grid leaves the last pass over the 512 workgroup slots partly empty."""
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
    ("res5/256 3x3 512->512 (392 tiles)", 256, 512, 7, 7, 512, 3, 1),
    ("res5/256 1x1 2048->512 (392 tiles)", 256, 2048, 7, 7, 512, 1, 0),
    ("res5/256 1x1 1024->512 (392 tiles)", 256, 1024, 7, 7, 512, 1, 0),
    ("res5/256 1x1 512->2048 (1568 tiles)", 256, 512, 7, 7, 2048, 1, 0),
    ("res5/256 1x1 1024->2048 (1568 tiles)", 256, 1024, 7, 7, 2048, 1, 0),
    ("rpn/1 3x3 1024->1024 (512 tiles)", 1, 1024, 64, 128, 1024, 3, 1),
    ("res5 3x3 512->512 (784 tiles)", 512, 512, 7, 7, 512, 3, 1),
    ("res5 1x1 2048->512 (784 tiles)", 512, 2048, 7, 7, 512, 1, 0),
    ("res5 1x1 512->2048 (3136 tiles)", 512, 512, 7, 7, 2048, 1, 0),
    ("res5 1x1 1024->2048 dgrad-like (3136)", 512, 1024, 7, 7, 2048, 1, 0),
    ("res4 3x3 256->256 (256 tiles)", 2, 256, 64, 128, 256, 3, 1),
    ("res4 1x1 1024->256 (256 tiles)", 2, 1024, 64, 128, 256, 1, 0),
    ("rpn 3x3 1024->1024 (1024 tiles)", 2, 1024, 64, 128, 1024, 3, 1),
    ("da img 1x1 1024->512 (512 tiles)", 2, 1024, 64, 128, 512, 1, 0),
    ("res3 3x3 128->128 (512 tiles)", 2, 128, 128, 256, 128, 3, 1),
]
for name, N, Cin, H, W, Cout, k, pad in SHAPES:
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.02).contiguous(memory_format=CL)
    y = _C.conv_forward(x, w, pad=pad)
    res = []
    for flag in ("0", "1"):
        os.environ["DADET_STREAMK"] = flag
        res.append(timeit(lambda: _C.conv_forward(x, w, pad=pad, out=y)))
    gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
    print("%-40s plain %.4f ms (%.0f TF/s)   stream-K %.4f ms (%.0f TF/s)   %+.1f%%" % (
        name, res[0], gf / res[0], res[1], gf / res[1], 100 * (res[0] / res[1] - 1)))
