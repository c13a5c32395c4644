"""launch time of the ROIAlign backward of the box head on a real step's ROIs (256 sampled ROIs of a 1024 x 2048 image, 1024 channels,
7 x 7 sub-grid of the 14 x 14 bins): the kernel sits between the box head's and the backbone's backward with nothing beside it"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine import trainer  # noqa: E402

dev = torch.device("cuda", 0)
yaml_path, overrides, ipg, _ = bench.WORKLOADS["img_only"]
c, model, opt, reducer = bench.build(yaml_path, dev, seed=100, overrides=overrides)
trainer.enable_overlapped_rpn_backward(model, True)
images, targets = make_batch(c, ipg, 1024, 2048, seed=100, device=dev)
calls = []
orig = _C.roi_align_backward


def spy(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)


for _ in range(4):
    trainer.train_step(model, opt, images, targets)
_C.roi_align_backward = spy
trainer.train_step(model, opt, images, targets)
_C.roi_align_backward = orig
torch.cuda.synchronize()
for a, k in calls:
    for _ in range(3):
        orig(*a, **k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        orig(*a, **k)
    e1.record()
    torch.cuda.synchronize()
    print("roi_align_backward grad %s -> %.1f us per launch  (%s)" % (tuple(a[0].shape), e0.elapsed_time(e1) / 20 * 1e3,
                                                                   {kk: vv for kk, vv in k.items() if not torch.is_tensor(vv)}))
