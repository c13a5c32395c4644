"""does the operand DATA set the GEMM's speed?  The RPN 3x3 convolution (2 x 1024 x 64 x 128 -> 1024) in every contraction
mode on all-zero, all-one, normal and uniform operands: same instruction stream, different bit activity in the matrix
pipe.  (profiles/r02_gemm_ceiling.txt was the first run of this probe, mode 3 only.)
usage: data_dependence.py [modes=4,3,2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C  # noqa: E402

CL = torch.channels_last
dev = torch.device("cuda:0")
MFMAS = {4: 3, 3: 6, 2: 3, 0: 1}


def timeit(fn, n=20):
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


modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "4,3,2").split(",")]
N, Cin, H, W, Cout, k = 2, 1024, 64, 128, 1024, 3
gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
for mode in modes:
    _C.set_gemm_mode(mode)
    for name, fx, fw in (("zeros", torch.zeros, torch.zeros), ("ones", torch.ones, torch.ones),
                         ("randn", torch.randn, lambda *a, **k: torch.randn(*a, **k) * 0.02),
                         ("rand 0..1", torch.rand, torch.rand)):
        x = fx((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
        w = fw((Cout, Cin, k, k), device=dev).contiguous(memory_format=CL)
        y = _C.conv_forward(x, w, pad=1)
        ms = timeit(lambda: _C.conv_forward(x, w, pad=1, out=y))
        print("mode %d %-10s %.4f ms  %.0f TF/s algorithmic  %.0f TF/s executed" % (mode, name, ms, gf / ms,
                                                                                   MFMAS[mode] * gf / ms), flush=True)
