"""one conv shape, a few launches of fwd and wgrad (for rocprofv3 --pmc runs)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")
N, Cin, H, W, Cout, k, stride, pad = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else
                                                       (2, 1024, 64, 128, 1024, 3, 1, 1))]
x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.05).contiguous(memory_format=CL)
y = _C.conv_forward(x, w, stride=stride, pad=pad)
gy = torch.randn_like(y)
for _ in range(4):
    _C.conv_forward(x, w, stride=stride, pad=pad, out=y)
    _C.conv_wgrad(x, gy, tuple(w.shape), stride, pad)
torch.cuda.synchronize()
