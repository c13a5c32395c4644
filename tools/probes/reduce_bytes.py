"""bytes moved by the deferred weight-gradient reduction passes of one img_only step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import torch
import bench
from da_detect_amd import _C
from da_detect_amd.config import cfg
from da_detect_amd.data.synthetic import make_batch
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step
from da_detect_amd.modeling.detector import build_detection_model
from da_detect_amd.solver import make_optimizer
from da_detect_amd.parallel.reducer import BucketedGradReducer

wl = sys.argv[1] if len(sys.argv) > 1 else "img_only"
yaml_path, overrides, ipg, _ = bench.WORKLOADS[wl]
c = cfg.clone(); c.merge_from_file(os.path.join(ROOT, yaml_path))
if overrides: c.merge_from_list(list(overrides))
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = build_detection_model(c); bench.benchmark_init(model, 1); model = model.to(dev).train()
opt = make_optimizer(c, model)
red = BucketedGradReducer([p for p in model.parameters() if p.requires_grad]); opt.attach_reducer(red)
enable_overlapped_rpn_backward(model)
images, targets = make_batch(c, ipg, 1024, 2048, seed=1, device=dev)
log = []
orig = _C.conv_wgrad_reduce_batch
def hook(batch):
    log.append([(it[0].count, it[0].splits, it[0].accumulate, it[0].K) for it in batch])
    return orig(batch)
_C.conv_wgrad_reduce_batch = hook
for i in range(3):
    del log[:]
    train_step(model, opt, images, targets)
torch.cuda.synchronize()
tot_r = tot_w = 0
for b in log:
    r = sum(cnt * 4 * (s + (1 if a else 0)) for cnt, s, a, k in b); w = sum(cnt * 4 for cnt, s, a, k in b)
    print("launch with %d items: read %.1f MB write %.1f MB" % (len(b), r / 1e6, w / 1e6))
    for cnt, s, a, k in sorted(b, key=lambda t: -t[0] * t[1])[:12]:
        print("    count %9d splits %3d acc %d K %d" % (cnt, s, a, k))
    tot_r += r; tot_w += w
print("total read %.1f MB write %.1f MB" % (tot_r / 1e6, tot_w / 1e6))
