"""A/B of a module-level constant (no environment switch exists for it) inside one process per setting:
usage: variant_ab.py <module>.<NAME> <value> [workload=img_only] [steps=30]   e.g. da_detect_amd.modeling.rpn.inference._ROWS_TOPK_SINGLE True"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
target, value = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "img_only"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
sys.argv = [sys.argv[0]]
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

mod, name = target.rsplit(".", 1)
setattr(importlib.import_module(mod), name, {"True": True, "False": False}.get(value, value))
device = torch.device("cuda", 0)
yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, images_per_gpu, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
for _ in range(10):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
print("%s = %s, %s: %.3f ms/step" % (target, value, workload, (time.perf_counter() - t0) / steps * 1e3))
