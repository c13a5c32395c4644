"""why is a recipe slower as the SECOND model built in a process (bench.py's comment at `isolate`: da 29.9 vs 27.3 ms)?
usage: second_model_probe.py first|second [workload=da]
`second` builds and steps img_only first (and drops it), then the workload; `first` runs the workload alone.  Prints the
step time, the host enqueue time per step, sizes of the module-level tables that outlive a model, and the host functions
with the largest own time — compare the two outputs."""
import cProfile
import gc
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.utils import streams  # noqa: E402

device = torch.device("cuda", 0)
order = sys.argv[1] if len(sys.argv) > 1 else "first"
workload = sys.argv[2] if len(sys.argv) > 2 else "da"


def run(name, steps, profile=False):
    yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[name]
    c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
    enable_overlapped_rpn_backward(model)
    images, targets = make_batch(c, images_per_gpu, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
    for _ in range(8):
        train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, opt, images, targets)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s (%s): %.3f ms/step, host enqueue %.3f ms/step" % (name, order, (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3))
    if profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            train_step(model, opt, images, targets)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    tc = _C._TRANSPOSES
    print("tables: transposed-weight cache %d entries; pending reductions %d; side streams %d; hooks on params %d" % (
        len(getattr(tc, "entries", {}) or {}), len(streams._PENDING_REDUCES), len(streams._SIDE),
        sum(len(getattr(p, "_post_accumulate_grad_hooks", None) or {}) for p in model.parameters())))
    print("allocator: reserved %.2f GB, allocated %.2f GB, segments %d" % (
        torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9,
        torch.cuda.memory_stats()["segment.all.current"]))


if order == "second":
    run("img_only", 20)
    gc.collect()
    torch.cuda.empty_cache()
run(workload, 30, profile=True)
