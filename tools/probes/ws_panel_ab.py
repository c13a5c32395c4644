"""conv1x1_ws_kernel, K = 256: 64-column against 128-column panels on the step's own shapes (launch time, HIP events)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")
for (N, Cin, H, W, Cout, gate) in [(2, 256, 64, 128, 1024, False), (2, 256, 64, 128, 1024, True), (2, 256, 128, 256, 512, False)]:
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, 1, 1), device=dev) * 0.05).contiguous(memory_format=CL)
    add = torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL)
    kw = dict(addend=add, relu_mode=1)
    if gate:
        kw = dict(addend=add, relu_mode=2, mask_ref=add)
    row = []
    for bn in ("64", "128"):
        os.environ["DADET_WS_K256_BN"] = bn
        for _ in range(3):
            _C.conv_forward(x, w, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            _C.conv_forward(x, w, **kw)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 40 * 1e3
        mb = 4e-6 * (x.numel() + 2 * add.numel() * (1.5 if gate else 1))
        row.append("panel %s: %.1f us (%.2f TB/s)" % (bn, us, mb / us))
    print("M=%d N=%d K=%d %s | " % (N * H * W, Cout, Cin, "gate" if gate else "relu") + " | ".join(row))
