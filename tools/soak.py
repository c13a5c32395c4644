#!/usr/bin/env python
"""Soak run: many training steps of the bench workload; watches the allocator and the losses (stream hand-overs,
cached tables and event churn must not grow memory or destabilise the step time)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
device = torch.device("cuda", 0)
c, model, opt, reducer = bench.build(bench.YAML, device, seed=100)
enable_overlapped_rpn_backward(model)
batches = [make_batch(c, 2, bench.HEIGHT, bench.WIDTH, seed=100 + i, device=device) for i in range(4)]
t0 = time.perf_counter()
for it in range(steps):
    images, targets = batches[it % len(batches)]
    loss = train_step(model, opt, images, targets)
    if it % 50 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        tot = float(sum(v.detach() for v in loss.values()))
        from da_detect_amd import _C
        _C.check_nonfinite()      # the GEMMs' guard (also reports a two-part meeting that timed out, csrc/conv_big.hip)
        print("it %4d  loss %.4f  alloc %.2f GB  reserved %.2f GB  %.1f ms/it" % (
            it, tot, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            (time.perf_counter() - t0) / (it + 1) * 1e3), flush=True)
        assert tot == tot, "NaN loss"
