"""does Python's cyclic garbage collector cost step time?  alternating blocks of steps with the collector on / off in one
process (per-step wall times; a collection shows as a step a few ms longer than its neighbours).
usage: gc_probe.py [workload=img_only] [blocks=4] [steps=40]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
workload = sys.argv[1] if len(sys.argv) > 1 else "img_only"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sys.argv = [sys.argv[0]]
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

device = torch.device("cuda", 0)
yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, images_per_gpu, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
for _ in range(10):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
for b in range(blocks):
    on = b % 2 == 0
    if on:
        gc.enable()
    else:
        gc.collect()
        gc.disable()
    n0 = sum(s["collections"] for s in gc.get_stats())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print("%s gc %s: %.3f ms/step, %d collections in %d steps" % (workload, "on " if on else "off", dt,
                                                                   sum(s["collections"] for s in gc.get_stats()) - n0, steps), flush=True)
gc.enable()
