"""dadet_topk_sorted against torch.sort on the RPN's shape: 2 x 122880 sigmoid scores, k = 12000"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=30):
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


for name, x in (("sigmoid(N(-3, 2))", (torch.randn((2, 122880), device=dev) * 2 - 3).sigmoid()),
                ("uniform(0, 1)", torch.rand((2, 122880), device=dev)),
                ("N(0, 100)", torch.randn((2, 122880), device=dev) * 100)):
    for k in (12000, 6000, 2000):
        a = timeit(lambda: _C.topk_sorted(x, k))
        b = timeit(lambda: torch.sort(x, dim=1, descending=True, stable=True))
        print("%-20s k=%5d  topk_sorted %.3f ms   torch.sort (all 122880) %.3f ms" % (name, k, a, b))
