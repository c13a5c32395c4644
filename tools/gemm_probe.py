"""a handful of conv shapes, launched repeatedly — target for rocprofv3 --pmc passes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")
SHAPES = [(2, 1024, 64, 128, 1024, 3, 1, 1), (512, 512, 7, 7, 512, 3, 1, 1), (2, 256, 64, 128, 1024, 1, 1, 0),
          (512, 2048, 7, 7, 512, 1, 1, 0), (2, 128, 128, 256, 128, 3, 1, 1)]
which = [int(a) for a in sys.argv[1:]] or list(range(len(SHAPES)))
for i in which:
    N, Cin, H, W, Cout, k, stride, pad = SHAPES[i]
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
    y = _C.conv_forward(x, w, stride=stride, pad=pad)
    gy = torch.randn_like(y)
    for _ in range(3):
        _C.conv_forward(x, w, stride=stride, pad=pad, out=y)
        _C.conv_wgrad(x, gy, tuple(w.shape), stride, pad)
torch.cuda.synchronize()
