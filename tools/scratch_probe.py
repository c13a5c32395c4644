import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C
from tools.wgrad_sweep import timeit, planned_splits
CL = torch.channels_last
dev = torch.device("cuda", 0)
cases = [
 ("res5 1x1 512->2048 7x7x512", 512, 7, 7, 512, 2048, 1),
 ("same M as 2x98x128", 2, 98, 128, 512, 2048, 1),
 ("M=24576 2x96x128", 2, 96, 128, 512, 2048, 1),
 ("M=16384 2x64x128", 2, 64, 128, 512, 2048, 1),
 ("M=32768 2x128x128", 2, 128, 128, 512, 2048, 1),
 ("res5 3x3 7x7x512", 512, 7, 7, 512, 512, 3),
 ("3x3 M=25088 2x98x128", 2, 98, 128, 512, 512, 3),
 ("3x3 M=16384 2x64x128 512", 2, 64, 128, 512, 512, 3),
 ("rpn 3x3", 2, 64, 128, 1024, 1024, 3),
 ("rpn-like 1024 M=25088", 2, 98, 128, 1024, 1024, 3),
]
for name, N, H, W, Cin, Cout, k in cases:
    pad = k // 2
    x = torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=CL)
    gy = torch.randn(N, Cout, H, W, device=dev).contiguous(memory_format=CL)
    w = torch.randn(Cout, Cin, k, k, device=dev).contiguous(memory_format=CL)
    for _ in range(2):
        mw = timeit(lambda: _C.conv_wgrad(x, gy, (Cout, Cin, k, k), 1, pad))
        mf = timeit(lambda: _C.conv_forward(x, w, stride=1, pad=pad))
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    s = planned_splits(x, gy, Cout, Cin, k, 1, pad)
    print("%-30s M=%6d  wgrad s=%3d %.4f ms %6.1f TF/s | fwd %.4f ms %6.1f TF/s" % (name, N*H*W, s, mw, fl/mw/1e9, mf, fl/mf/1e9))
