"""one aligned-triplet step on the GPU with everything the CPU-side comparison needs dumped to gpurun_out/aligned_dump.pt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from da_detect_amd import _C  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from da_detect_amd.utils import rng  # noqa: E402
from golden.cases import case_cfg  # noqa: E402
from golden.fill import fill_state_dict  # noqa: E402

device = torch.device("cuda", 0)
seed = 11
c = case_cfg("da_triplet_aligned")
model = build_detection_model(c)
model.load_state_dict(fill_state_dict(model.state_dict(), seed))
model = model.to(device).train()
images, targets = make_batch(c, 3, 192, 320, seed=seed, device=device)
opt = make_optimizer(c, model)
opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
enable_overlapped_rpn_backward(model)
rec = dict(seeds=[], masks=[], rois=[], maps=[], ins_set=None, pooled=[])
o_seed, o_mask, o_sr = rng.next_seed, rng.dropout_mask, _C.sample_rois
rng.next_seed = lambda dev: rec["seeds"].append(o_seed(dev)) or rec["seeds"][-1]


def dmask(shape, p, dev):
    m = o_mask(shape, p, dev)
    rec["masks"].append(m)
    return m


rng.dropout_mask = dmask


def sr(boxes, labels, reg, cap, max_pos, seed_, is_source, counts, out=None):
    o = o_sr(boxes, labels, reg, cap, max_pos, seed_, is_source, counts, out=out)
    rec["rois"].append((o["idx"], counts, boxes.shape[0]))
    return o


_C.sample_rois = sr
model.rpn.head.register_forward_hook(lambda m, i, o: rec["maps"].append((o[0][0].detach(), o[1][0].detach())))
fe = model.roi_heads.box.feature_extractor
o_pool = fe.pooler.forward


def pool(x, boxes, **k):
    y = o_pool(x, boxes, **k)
    rec["pooled"].append((y.detach().mean(dim=(2, 3)).cpu(), [b.bbox.detach().cpu() for b in boxes]))
    return y


fe.pooler.forward = pool
tri = model.da_heads_triplet
o_tri = tri.forward


def tri_fwd(img_features, da_ins_feature, da_ins_labels, da_ins_feas_set, img_fea_set, targets=None):
    rec["ins_set"] = [f.detach().mean(dim=(2, 3)).cpu() for f in da_ins_feas_set]
    return o_tri(img_features, da_ins_feature, da_ins_labels, da_ins_feas_set, img_fea_set, targets)


tri.forward = tri_fwd
torch.manual_seed(seed)
losses = train_step(model, opt, images, targets)
torch.cuda.synchronize()
out = dict(seeds=rec["seeds"], masks=[m.cpu() for m in rec["masks"]],
           rois=[(i[: int(cn[0])].cpu(), cn.cpu(), n) for i, cn, n in rec["rois"]],
           objectness=torch.cat([a for a, _ in rec["maps"]]).cpu(), deltas=torch.cat([b for _, b in rec["maps"]]).cpu(),
           ins_set=rec["ins_set"], pooled=rec["pooled"], losses={k: float(v) for k, v in losses.items()},
           grads={n: p.grad.detach().cpu() for n, p in model.named_parameters()
                  if p.requires_grad and ("layer4.2.conv3" in n or "layer2.0.conv1" in n or "cls_score.weight" in n
                                          or "layer4.0.conv1" in n)})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
torch.save(out, os.path.join(ROOT, "gpurun_out", "aligned_dump.pt"))
print("saved", {k: (len(v) if isinstance(v, (list, dict)) else tuple(v.shape)) for k, v in out.items()})
