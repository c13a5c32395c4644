#!/usr/bin/env python
"""time of one NMS (mask + sweep + compaction) on 12 000 pre-ranked boxes.

  uniform   boxes spread over the image: most of them survive (50 kept per 64-box chunk)
  clustered jittered copies of a few thousand anchors-like boxes, ranked at random — what the RPN hands over during
            training: the whole list is swept and ~10 boxes per chunk are kept (2000 of 12 000)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
W, H = 2048, 1024


def uniform(n, side):
    xy = np.stack([rng.uniform(0, W - 2, n), rng.uniform(0, H - 2, n)], 1)
    wh = np.stack([rng.uniform(8, side, n), rng.uniform(8, side, n)], 1)
    return np.concatenate([xy, np.minimum(xy + wh, [W - 1, H - 1])], 1).astype(np.float32)


def clustered(n, centres, jitter):
    base = uniform(centres, 400)
    pick = rng.integers(0, centres, n)
    b = base[pick] + rng.normal(0, jitter, (n, 4)).astype(np.float32)
    b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 4)
    return np.clip(b, 0, [W - 1, H - 1, W - 1, H - 1]).astype(np.float32)


cases = [("uniform side 300", uniform(12000, 300)), ("uniform side 80", uniform(12000, 80)),
         ("clustered 1800 x jitter 6", clustered(12000, 1800, 6.0)), ("clustered 1000 x jitter 10", clustered(12000, 1000, 10.0))]
for name, arr in cases:
    b = torch.from_numpy(arr).to(dev)
    for mk in (2000, -1):
        keep, cnt = _C.nms_with_count(b, None, 0.7, max_keep=mk)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            _C.nms_with_count(b, None, 0.7, max_keep=mk)
        e.record()
        torch.cuda.synchronize()
        print("%-28s max_keep %5d: kept %5d, %.3f ms per NMS" % (name, mk, int(cnt), s.elapsed_time(e) / 20))
