"""Operator-level golden vectors from the REFERENCE's compiled CPU operators (oracle/_ref/ref_C.so, built from
/root/reference/maskrcnn_benchmark/csrc by oracle/build_ref.py): seeded inputs + reference outputs of `nms` and
`roi_align_forward`, stored in tests/golden/ref_ops.npz.  Authoring container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import build_ref  # noqa: E402

build_ref.build()
ref = build_ref.load()
rng = np.random.default_rng(2024)
out = {}
# ROIAlign: adaptive and fixed sampling, degenerate / out-of-image / whole-image ROIs
x = rng.standard_normal((2, 16, 24, 40)).astype(np.float32)
R = 24
xy = np.stack([rng.uniform(-30, 600, R), rng.uniform(-30, 360, R)], 1)
wh = np.stack([rng.uniform(1, 500, R), rng.uniform(1, 300, R)], 1)
rois = np.concatenate([rng.integers(0, 2, (R, 1)), xy, xy + wh], 1).astype(np.float32)
rois[0, 1:] = [10, 10, 10, 10]
rois[1, 1:] = [0, 0, 639, 383]
rois[2, 1:] = [630, 375, 700, 420]
out["roi/input"], out["roi/rois"] = x, rois
for ph, sr in ((7, 0), (14, 0), (7, 2)):
    out["roi/out_%d_%d" % (ph, sr)] = ref.roi_align_forward(torch.from_numpy(x), torch.from_numpy(rois), 1 / 16.0,
                                                          ph, ph, sr).numpy()
# NMS: 1500 boxes with duplicated scores are avoided (the reference's sort leaves tie order unspecified)
n = 1500
xy = rng.uniform(0, 800, (n, 2))
boxes = np.concatenate([xy, xy + rng.uniform(4, 160, (n, 2))], 1).astype(np.float32)
scores = rng.permutation(n).astype(np.float32) / n
out["nms/boxes"], out["nms/scores"] = boxes, scores
for thr in (0.3, 0.5, 0.7):
    out["nms/keep_%.1f" % thr] = ref.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
path = os.path.join(HERE, "ref_ops.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB")
