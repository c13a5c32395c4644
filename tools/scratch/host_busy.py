"""is the step host bound?  wall time per step against the CPU time the process burns per step (all threads: the main
thread issues the forward pass, autograd's device thread the backward pass), and against the time the host spends
blocked in its three device->host round trips"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402

device = torch.device("cuda", 0)
c, model, opt, reducer = bench.build(bench.YAML, device, seed=100)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, 2, bench.HEIGHT, bench.WIDTH, seed=100, device=device)
for _ in range(5):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()

blocked = [0.0]
orig_tolist = torch.Tensor.tolist


def tolist(self):
    t = time.perf_counter()
    r = orig_tolist(self)
    if self.is_cuda:
        blocked[0] += time.perf_counter() - t
    return r


torch.Tensor.tolist = tolist
steps = 20
w0, c0 = time.perf_counter(), time.process_time()
for _ in range(steps):
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.process_time()
print("wall %.2f ms/step, process CPU time %.2f ms/step, blocked in .tolist() of device tensors %.2f ms/step" % (
    (w1 - w0) / steps * 1e3, (c1 - c0) / steps * 1e3, blocked[0] / steps * 1e3))
