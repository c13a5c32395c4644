"""the weight-stationary 1x1 kernel (csrc/conv_ws.hip) against the tiled split kernel on the short-K layers of the
BASELINE step: time per launch, algorithmic TF/s and GB/s, both kernels on the same tensors (DADET_WS_1X1 = 0 / 1)"""
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


SHAPES = [  # name, N, K, H, W, Cout, stride, epilogue
    ("res4 conv3 fwd   256->1024 +add relu", 2, 256, 64, 128, 1024, 1, "add_relu"),
    ("res4 conv1 dgrad 256->1024 +add gate", 2, 256, 64, 128, 1024, 1, "add_gate"),
    ("res4 shortcut    512->1024 s2 (K=512: tiled only)", 2, 512, 128, 256, 1024, 2, "affine"),
    ("res3 conv3 fwd   128->512 +add relu", 2, 128, 128, 256, 512, 1, "add_relu"),
    ("res3 conv1 dgrad 128->512 +add gate", 2, 128, 128, 256, 512, 1, "add_gate"),
    ("res3 shortcut    256->512 s2", 2, 256, 256, 512, 512, 2, "affine"),
    ("res2 conv3 fwd   64->256 +add relu", 2, 64, 256, 512, 256, 1, "add_relu"),
    ("res2 shortcut    64->256", 2, 64, 256, 512, 256, 1, "affine"),
]
for name, N, K, H, W, Cout, stride, epi in SHAPES:
    x = torch.randn((N, K, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, K, 1, 1), device=dev) * 0.05).contiguous(memory_format=CL)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    kw = dict(scale=torch.rand(Cout, device=dev) + 0.5, bias=torch.randn(Cout, device=dev))
    nbytes = 4.0 * (N * Ho * Wo * K + Cout * K + N * Ho * Wo * Cout)
    if epi in ("add_relu", "add_gate"):
        kw["addend"] = torch.randn((N, Cout, Ho, Wo), device=dev).contiguous(memory_format=CL)
        nbytes += 4.0 * N * Ho * Wo * Cout
    if epi == "add_relu":
        kw["relu_mode"] = 1
    if epi == "add_gate":
        kw["relu_mode"] = 2
        kw["mask_ref"] = torch.randn((N, Cout, Ho, Wo), device=dev).clamp_min(0).contiguous(memory_format=CL)
        nbytes += 4.0 * N * Ho * Wo * Cout
    y = torch.empty((N, Cout, Ho, Wo), device=dev).contiguous(memory_format=CL)
    gf = 2.0 * N * Ho * Wo * Cout * K / 1e9
    res = []
    for flag in ("0", "1"):
        os.environ["DADET_WS_1X1"] = flag
        res.append(timeit(lambda: _C.conv_forward(x, w, stride=stride, out=y, **kw)))
    os.environ.pop("DADET_WS_1X1")
    print("%-52s tiled %6.1f us (%5.1f TF/s %5.0f GB/s)   weight-stationary %6.1f us (%5.1f TF/s %5.0f GB/s)   %+.0f%%" % (
        name, res[0], gf / res[0] * 1e3, nbytes / res[0] / 1e3, res[1], gf / res[1] * 1e3, nbytes / res[1] / 1e3,
        100 * (res[0] / res[1] - 1)), flush=True)
