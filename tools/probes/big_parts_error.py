"""RMS / max error against float64 of the 256 x 256-tile kernel with its reduction cut into 1 .. 4 parts, next to the 128 x 128 kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C, _lib

CL = torch.channels_last
dev = torch.device("cuda:0")
lib = _lib.load()
for (N, Cin, H, W, Cout, k, stride, pad) in [(1, 256, 20, 20, 512, 3, 2, 1), (2, 128, 24, 40, 512, 3, 1, 1), (1, 1024, 16, 16, 256, 1, 1, 0)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=CL)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    row = []
    lib.dadet_set_big_gemm(0)
    y = _C.conv_forward(x, w, stride=stride, pad=pad)
    e = (y.double() - ref).abs()
    row.append("128-tile %.2e / %.2e" % (float(e.pow(2).mean().sqrt()), float(e.max())))
    lib.dadet_set_big_gemm(2)
    for s in (1, 2, 3, 4):
        os.environ["DADET_BIG_SPLITS"] = str(s)
        y = _C.conv_forward(x, w, stride=stride, pad=pad)
        e = (y.double() - ref).abs()
        row.append("S=%d %.2e / %.2e" % (s, float(e.pow(2).mean().sqrt()), float(e.max())))
    os.environ.pop("DADET_BIG_SPLITS")
    print((N, Cin, H, W, Cout, k, stride, pad), " | ".join(row))
