"""which python call sites launch the small ATen kernels of one training step (torch.profiler, CPU-side op records with
stacks; usage: op_sites.py [workload=fpn_dcn_da])"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl = sys.argv[1] if len(sys.argv) > 1 else "fpn_dcn_da"
sys.argv = [sys.argv[0]]
import torch
import bench
from da_detect_amd.data.synthetic import make_batch
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step

dev = torch.device("cuda", 0)
yaml_path, overrides, ipg, _ = bench.WORKLOADS[wl]
c, model, opt, reducer = bench.build(yaml_path, dev, seed=100, overrides=overrides)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, ipg, bench.HEIGHT, bench.WIDTH, seed=100, device=dev)
for _ in range(3):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

sites = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.as_strided", "aten.slice", "aten.select", "aten.detach", "aten.alias",
        "aten.permute", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.t.", "aten.empty",
        "aten.reshape", "aten._reshape_alias", "aten.empty_like", "aten.empty_strided", "aten.new_empty", "aten.unbind",
        "aten.split", "aten.is_pinned", "aten.record_stream", "aten._local_scalar_dense", "aten.lift_fresh", "aten.narrow")


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack()[:-1] if "da_detect_amd" in f.filename or "bench.py" in f.filename]
            where = " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in fr[-3:][::-1]) \
                if fr else "(autograd engine)"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Count():
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
print("ATen calls that launch something, one step of %s: %d" % (wl, sum(sites.values())))
for (name, where), n in sites.most_common(90):
    print("%4d  %-28s %s" % (n, name, where))
