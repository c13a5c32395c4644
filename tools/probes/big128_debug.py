import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from da_detect_amd import _C, _lib
CL = torch.channels_last
dev = torch.device("cuda:0")
lib = _lib.load()
lib.dadet_set_big_gemm(2)
os.environ["DADET_BIG_TILE_N"] = sys.argv[1] if len(sys.argv) > 1 else "128"
os.environ["DADET_BIG_SPLITS"] = sys.argv[2] if len(sys.argv) > 2 else "1"
for (N, Cin, H, W, Cout, k, pad) in [(1, 64, 1, 512, 128, 1, 0), (1, 64, 1, 256, 256, 1, 0), (1, 64, 1, 512, 256, 1, 0), (1, 64, 16, 16, 128, 3, 1),
                                     (1, 128, 16, 16, 128, 3, 1), (2, 128, 20, 28, 320, 3, 1), (1, 32, 16, 32, 128, 3, 1)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, Cin, H, W), generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / (Cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=CL)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=pad)
    ys = [_C.conv_forward(x, w, pad=pad) for _ in range(3)]
    M = N * H * W
    e = (ys[0].double() - ref).abs().permute(0, 2, 3, 1).reshape(M, Cout)
    print("%s: max err %.2e, repeat equal %s %s" % ((N, Cin, H, W, Cout, k), float(e.max()), torch.equal(ys[0], ys[1]), torch.equal(ys[0], ys[2])))
    if float(e.max()) > 1e-4:
        Mp, Np = (M + 31) // 32 * 32, (Cout + 31) // 32 * 32
        ee = torch.zeros(Mp, Np, device=dev, dtype=torch.float64); ee[:M, :Cout] = e
        blocks = ee.reshape(Mp // 32, 32, Np // 32, 32).amax(dim=(1, 3))
        bad = (blocks > 1e-4).int()
        print("bad 32x32 blocks: rows", bad.any(dim=1).nonzero().flatten().tolist(), "cols", bad.any(dim=0).nonzero().flatten().tolist())
