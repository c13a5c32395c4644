import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C
CL = torch.channels_last
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
N, Cin, H, W, Cout, k = 2, 1024, 64, 128, 1024, 3
gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
for name, fx, fw in (("zeros", torch.zeros, torch.zeros), ("ones", torch.ones, torch.ones),
                     ("randn", torch.randn, lambda *a, **k: torch.randn(*a, **k) * 0.02),
                     ("rand 0..1", torch.rand, torch.rand)):
    x = fx((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = fw((Cout, Cin, k, k), device=dev).contiguous(memory_format=CL)
    y = _C.conv_forward(x, w, pad=1)
    ms = timeit(lambda: _C.conv_forward(x, w, pad=1, out=y))
    print("%-10s %.4f ms  %.0f TF/s algorithmic  %.0f TF/s executed" % (name, ms, gf / ms, 6 * gf / ms))
