"""where the time of the short-K 1x1 "+add" layers goes (res4 conv3: M = 16384, N = 1024, K = 256): epilogue variants, a K
sweep (intercept = streaming cost of the tile grid, slope = contraction), and a pure streaming kernel over the same bytes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from da_detect_amd import _C

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


N, H, W, Cout = 2, 64, 128, 1024
for Cin in (64, 128, 256, 512, 1024):
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, 1, 1), device=dev) * 0.02).contiguous(memory_format=CL)
    scale, bias = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    addend = torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL)
    y = torch.empty_like(addend)
    full = timeit(lambda: _C.conv_forward(x, w, scale=scale, bias=bias, addend=addend, relu_mode=1, out=y))
    noadd = timeit(lambda: _C.conv_forward(x, w, scale=scale, bias=bias, relu_mode=1, out=y))
    plain = timeit(lambda: _C.conv_forward(x, w, out=y))
    gf = 2.0 * N * H * W * Cout * Cin / 1e9
    print("K = %4d: affine + add + relu %6.1f us, affine + relu %6.1f us, plain store %6.1f us   (%.1f GF: %.0f / %.0f / %.0f TF/s)"
          % (Cin, full, noadd, plain, gf, gf / full * 1e-3 * 1e3, gf / noadd * 1e-3 * 1e3, gf / plain * 1e-3 * 1e3), flush=True)
a = torch.randn((N, Cout, H, W), device=dev).contiguous(memory_format=CL)
b = torch.randn_like(a)
o = torch.empty_like(a)
t = timeit(lambda: torch.add(a, b, out=o))
print("streaming add over the same output (67 MB + 67 MB -> 67 MB): %.1f us (%.0f GB/s)" % (t, 3 * a.numel() * 4 / t / 1e3))
t = timeit(lambda: o.copy_(a))
print("copy 67 MB -> 67 MB: %.1f us (%.0f GB/s)" % (t, 2 * a.numel() * 4 / t / 1e3))
