"""A/B of the 16-byte (LDS-transposed) epilogue against the 4-byte one on the short-K layers of the backbone"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=30):
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


SHAPES = [  # name, N, Cin, H, W, Cout, k, pad, addend
    ("res2 1x1 64->256 +add", 2, 64, 256, 512, 256, 1, 0, True),
    ("res2 1x1 256->64", 2, 256, 256, 512, 64, 1, 0, False),
    ("res2 3x3 64->64", 2, 64, 256, 512, 64, 3, 1, False),
    ("res3 1x1 128->512 +add", 2, 128, 128, 256, 512, 1, 0, True),
    ("res3 1x1 512->128", 2, 512, 128, 256, 128, 1, 0, False),
    ("res4 1x1 256->1024 +add", 2, 256, 64, 128, 1024, 1, 0, True),
    ("res4 1x1 1024->256", 2, 1024, 64, 128, 256, 1, 0, False),
    ("res5 1x1 512->2048 +add", 512, 512, 7, 7, 2048, 1, 0, True),
    ("res5 3x3 512->512", 512, 512, 7, 7, 512, 3, 1, False),
    ("rpn 3x3 1024->1024", 2, 1024, 64, 128, 1024, 3, 1, False),
]
for name, N, Cin, H, W, Cout, k, pad, add in SHAPES:
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.02).contiguous(memory_format=CL)
    scale, bias = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    addend = torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL) if add else None
    y = _C.conv_forward(x, w, pad=pad, scale=scale, bias=bias, addend=addend, relu_mode=1)
    res = []
    for flag in ("0", "1"):
        os.environ["DADET_EPILOGUE_V4"] = flag
        res.append(timeit(lambda: _C.conv_forward(x, w, pad=pad, scale=scale, bias=bias, addend=addend, relu_mode=1, out=y)))
    mb = 4.0 * (x.numel() + w.numel() + y.numel() * (2 if add else 1)) / 1e6
    print("%-26s 4-byte %.4f ms (%.0f GB/s)   16-byte %.4f ms (%.0f GB/s)   %+.1f%%" % (
        name, res[0], mb / res[0], res[1], mb / res[1], 100 * (res[0] / res[1] - 1)))
