#!/usr/bin/env python
"""cProfile of the host side of the bench step (where does the CPU spend its time between kernel launches?)
usage: host_profile.py [steps=5] [workload=img_only]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

device = torch.device("cuda", 0)
workload = sys.argv[2] if len(sys.argv) > 2 else "img_only"
yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, images_per_gpu, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
for _ in range(3):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(steps):
    train_step(model, opt, images, targets)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, +drain %.2f ms/step" % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
