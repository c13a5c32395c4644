"""contraction mode 4: which GEMM operands reach `_C.conv_forward` / `_C.conv_wgrad` without a largest magnitude attached
(each costs a `dadet_amax` pass over the tensor)?  Runs a few steps, then one step with `amax.measure` wrapped: prints the
number of measurements per step and the call sites (innermost frames outside _C.py / amax.py), with the bytes each reads.
usage: amax_sites.py [workload=img_only] [HxW]"""
import collections
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ["DADET_GEMM_MODE"] = "4"
workload = sys.argv[1] if len(sys.argv) > 1 else "img_only"
hw = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else None
sys.argv = [sys.argv[0]]
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd import amax  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

device = torch.device("cuda", 0)
yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
enable_overlapped_rpn_backward(model)
H, W = hw if hw else (bench.HEIGHT, bench.WIDTH)
images, targets = make_batch(c, images_per_gpu, H, W, seed=100, device=device)
for _ in range(6):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
n0 = amax.MEASURED
t0 = time.perf_counter()
for _ in range(10):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
print("%s: %.3f ms/step, %.1f measurements per step" % (workload, (time.perf_counter() - t0) * 100,
                                                        (amax.MEASURED - n0) / 10.0))
sites = collections.Counter()
nbytes = collections.Counter()
orig = amax.measure


def traced(t):
    frames = [f for f in traceback.extract_stack()[:-1]
              if not f.filename.endswith(("_C.py", "amax.py", "amax_sites.py")) and "/torch/" not in f.filename]
    key = " <- ".join("%s:%d %s" % (os.path.relpath(f.filename, ROOT).replace("da_detect_amd/", ""), f.lineno, f.name)
                      for f in frames[-1:-6:-1])
    sites[key] += 1
    nbytes[key] += t.numel() * 4
    return orig(t)


amax.measure = traced
train_step(model, opt, images, targets)
torch.cuda.synchronize()
for k, n in sorted(sites.items(), key=lambda kv: -nbytes[kv[0]]):
    print("%3d x %9.2f MB  %s" % (n, nbytes[k] / 1e6, k))
