"""A/B of the forward / data-gradient GEMM with the weights as pre-split bf16 planes streamed into LDS by DMA
(dadet_conv_forward_wp, conv_split.hip "BP") against the register-staged kernel, + bit-identity of the two."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


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


SHAPES = [  # name, N, Cin, H, W, Cout, k, pad
    ("rpn 3x3 1024->1024 (M 16384)", 2, 1024, 64, 128, 1024, 3, 1),
    ("rpn 3x3 1024->1024 (M 8192)", 1, 1024, 64, 128, 1024, 3, 1),
    ("res5 3x3 512->512 (256 rois)", 256, 512, 7, 7, 512, 3, 1),
    ("res5 3x3 512->512 (512 rois)", 512, 512, 7, 7, 512, 3, 1),
    ("res5 1x1 2048->512 (256 rois)", 256, 2048, 7, 7, 512, 1, 0),
    ("res5 1x1 512->2048 (256 rois)", 256, 512, 7, 7, 2048, 1, 0),
    ("res5 1x1 1024->2048 (256 rois)", 256, 1024, 7, 7, 2048, 1, 0),
    ("res4 3x3 256->256", 2, 256, 64, 128, 256, 3, 1),
    ("res4 1x1 1024->256", 2, 1024, 64, 128, 256, 1, 0),
    ("res4 1x1 256->1024", 2, 256, 64, 128, 1024, 1, 0),
    ("da img 1x1 1024->512", 2, 1024, 64, 128, 512, 1, 0),
    ("res3 3x3 128->128", 2, 128, 128, 256, 128, 3, 1),
    ("res3 1x1 512->128", 2, 512, 128, 256, 128, 1, 0),
    ("res3 1x1 128->512", 2, 128, 128, 256, 512, 1, 0),
    ("res2 1x1 64->256", 2, 64, 256, 512, 256, 1, 0),
    ("res2 3x3 64->64", 2, 64, 256, 512, 64, 3, 1),
    ("fc 2048->1024 (512 rows)", 512, 2048, 1, 1, 1024, 1, 0),
    ("ragged 3x3 48->200, 37x53", 3, 48, 37, 53, 200, 3, 1),
    ("ragged 1x1 72->100, 19x23", 2, 72, 19, 23, 100, 1, 0),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot = [0.0, 0.0]
for name, N, Cin, H, W, Cout, k, pad in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn((N, Cin, H, W), device=dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), device=dev) * 0.02).contiguous(memory_format=CL)
    bias = torch.randn(Cout, device=dev)
    wp = _C.weight_planes(w)
    t_split = timeit(lambda: _C.weight_planes(w))
    y0 = _C.conv_forward(x, w, pad=pad, bias=bias, relu_mode=1)
    y1 = _C.conv_forward(x, w, pad=pad, bias=bias, relu_mode=1, w_planes=wp)
    same = torch.equal(y0, y1)
    ya, yb = y0.clone(), y0.clone()
    res = []
    for rep in range(2):      # interleaved A/B
        res.append((timeit(lambda: _C.conv_forward(x, w, pad=pad, out=ya)),
                    timeit(lambda: _C.conv_forward(x, w, pad=pad, out=yb, w_planes=wp))))
    a = min(r[0] for r in res)
    b = min(r[1] for r in res)
    tot[0] += a
    tot[1] += b
    gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
    print("%-34s staged %.4f ms (%5.0f TF/s)   planes+DMA %.4f ms (%5.0f TF/s)   %+5.1f%%   identical %s   split pass %.4f ms"
          % (name, a, gf / a, b, gf / b, 100 * (a / b - 1), same, t_split), flush=True)
    if not same:
        d = (y0 - y1).abs()
        print("    MISMATCH: max abs diff %.3e at %d of %d elements" % (float(d.max()), int((d > 0).sum()), d.numel()))
print("sum staged %.3f ms, planes %.3f ms (%+.1f%%)" % (tot[0], tot[1], 100 * (tot[0] / tot[1] - 1)))
