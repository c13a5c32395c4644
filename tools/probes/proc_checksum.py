import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from golden.cases import case_cfg  # noqa: E402
from golden.fill import fill_state_dict  # noqa: E402

device = torch.device("cuda", 0)
seed = 11
c = case_cfg("da_triplet_aligned")
c.merge_from_list(["MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", 0.0])
model = build_detection_model(c)
model.load_state_dict(fill_state_dict(model.state_dict(), seed))
model = model.to(device).train()
images, targets = make_batch(c, 3, 192, 320, seed=seed, device=device)
opt = make_optimizer(c, model)
opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
enable_overlapped_rpn_backward(model)
torch.manual_seed(seed)
losses = train_step(model, opt, images, targets)
torch.cuda.synchronize()
g = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
print("HASHSEED", os.environ.get("PYTHONHASHSEED"), " ".join("%s=%.7f" % (k[5:], float(v)) for k, v in sorted(losses.items())))
for n in ("backbone.body.layer2.0.conv1.weight", "roi_heads.box.feature_extractor.head.layer4.0.conv1.weight",
          "rpn.head.conv.weight", "roi_heads.box.predictor.cls_score.weight"):
    print("   %-62s norm %.9e  sum %.9e" % (n, float(g[n].double().norm()), float(g[n].double().sum())))
