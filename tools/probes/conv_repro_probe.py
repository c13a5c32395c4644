"""Which convolution launch of a full-size forward pass is not reproducible bit for bit?  Every dadet_conv_forward* result
is check-summed (integer sum of the fp32 bit patterns, on the launch's own stream) in two runs from one seed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from da_detect_amd import _C
from da_detect_amd.data.synthetic import make_batch

workload = sys.argv[1] if len(sys.argv) > 1 else "img_only"
device = torch.device("cuda:0")
yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[workload]
c, model, opt, reducer = bench.build(yaml_path, device, seed=100, overrides=overrides)
images, targets = make_batch(c, images_per_gpu, 1024, 2048, seed=100, device=device)
LOG = []
orig = _C._conv_forward_call


def call(d, x, w, scale, bias, addend, mask_ref, out, ax, aw):
    orig(d, x, w, scale, bias, addend, mask_ref, out, ax, aw)
    LOG.append(((d.N * d.Ho * d.Wo, d.Cout, d.KH * d.KW * d.Cin, d.KH), out.view(torch.int32).sum(dtype=torch.int64),
                x.view(torch.int32).sum(dtype=torch.int64)))


_C._conv_forward_call = call
runs = []
for r in range(3):
    LOG.clear()
    torch.manual_seed(7)
    opt.zero_grad()
    losses = model(images, targets)
    sum(losses.values()).backward()
    reducer.finalize()
    torch.cuda.synchronize()
    runs.append([(k, int(o), int(i)) for k, o, i in LOG])
print("launches per run", [len(r) for r in runs])
for r in (1, 2):
    for idx, (a, b) in enumerate(zip(runs[0], runs[r])):
        if a != b:
            print("run %d: first difference at launch %d: shape (M, N, K, k) %s, input equal: %s, output equal: %s" % (
                r, idx, a[0], a[2] == b[2], a[1] == b[1]))
            break
    else:
        print("run %d: identical" % r)
