"""A/B of the ROIAlign entry points with / without the scratch buffer (spatial ROI order, per-tile ROI masks) at the
BASELINE shape: 512 ROIs (256 per image, proposal-like sizes) on a [2, 1024, 64, 128] map, 14 x 14 bins."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


g = torch.Generator().manual_seed(0)
feat = torch.randn((2, 1024, 64, 128), device=dev).contiguous(memory_format=CL)
go = torch.randn((512, 1024, 14, 14), device=dev).contiguous(memory_format=CL)
for name, wh in (("uniform 16..316 px", torch.rand((512, 2), generator=g) * 300 + 16),
                 ("skewed (many small, few large)", torch.rand((512, 2), generator=g) ** 3 * 900 + 16)):
    xy = torch.rand((512, 2), generator=g) * torch.tensor([1800.0, 900.0])
    rois = torch.cat([(torch.arange(512) >= 256).float().view(-1, 1), xy,
                      torch.minimum(xy + wh, torch.tensor([2047.0, 1023.0]))], 1).to(dev)
    alg = (512 * 1024 * 196 * 4 + feat.numel() * 4) / 1e6
    for flag in (False, True):
        _C.ROI_ALIGN_WORKSPACE = flag        # fwd kernel choice: DADET_ROI_FWD_SWEEP=0 restores four-taps-per-sample
        f = timeit(lambda: _C.roi_align_forward(feat, rois, 1 / 16.0, 14, 14, 0))
        b = timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16.0, 14, 14, 2, 1024, 64, 128, 0))
        print("%-32s workspace=%-5s fwd %.3f ms (%.0f GB/s algorithmic)  bwd %.3f ms (%.0f GB/s)" % (
            name, flag, f, alg / f, b, alg / b))
