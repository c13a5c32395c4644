"""Does the parked-part hand-over of conv_big_kernel ever read a stale workspace line?  Two different inputs alternate on
one stream (the workspace addresses are reused with other contents every launch); each result is compared bit for bit
with the first one of its input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C, _lib

CL = torch.channels_last
dev = torch.device("cuda:0")
lib = _lib.load()
lib.dadet_set_big_gemm(2)
for (N, Cin, H, W, Cout, k, pad) in [(1, 1024, 64, 128, 1024, 3, 1), (1, 512, 112, 112, 512, 3, 1), (1, 256, 128, 128, 256, 3, 1)]:
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn((N, Cin, H, W), generator=g).to(dev).contiguous(memory_format=CL) for _ in range(2)]
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=CL)
    first = [_C.conv_forward(x, w, pad=pad).clone() for x in xs]
    bad = 0
    worst = 0.0
    for it in range(40):
        for j in range(2):
            y = _C.conv_forward(xs[j], w, pad=pad)
            if not torch.equal(y, first[j]):
                bad += 1
                worst = max(worst, float((y - first[j]).abs().max()))
    print((N, Cin, H, W, Cout, k), "mismatching launches %d / 80, worst |diff| %.3e" % (bad, worst))
