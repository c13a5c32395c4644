"""deformable sampling backward on the R-101-FPN-DCN shapes: new (4 pixels x 3 taps per wavefront step) vs old kernel,
with stage ablations (DADET_DEFORM_ABLATE, read once per process: run one configuration per process)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
for name, N, C, H, W in (("res3", 2, 128, 128, 256), ("res4", 2, 256, 64, 128), ("res5", 2, 512, 32, 64)):
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=CL)
    om = (torch.randn((N, 20, H, W), device=dev) * sigma).contiguous(memory_format=CL)
    gcols = torch.randn((N, 9 * C, H, W), device=dev).contiguous(memory_format=CL)
    t = timeit(lambda: _C.deform_sample_backward_om(x, om, gcols, 3, 3, 1, 1, 1, 1, False))
    t0 = timeit(lambda: (torch.empty_like(x).zero_(), torch.empty_like(om).zero_()))
    tf = timeit(lambda: _C.deform_sample_forward_om(x, om, 3, 3, 1, 1, 1, 1, False))
    print("%s C=%d %dx%d sigma %.1f: backward %.3f ms (of which the two zero fills %.3f), forward %.3f ms; gcols %.0f MB"
          % (name, C, H, W, sigma, t, t0, tf, gcols.numel() * 4 / 1e6), flush=True)
