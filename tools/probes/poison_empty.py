"""uninitialised-memory hunt: every torch.empty / empty_like / empty_strided / new_empty float buffer is filled with NaN
(integers with a large value) before use; a kernel that reads what nobody wrote then shows up as NaN / changed gradients"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

POISON = [False]


def poison(t):
    if POISON[0] and t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(0x3FFFFFF)
        elif t.dtype == torch.uint8:
            t.fill_(0xAB)
    return t


for name in ("empty", "empty_like", "empty_strided"):
    orig = getattr(torch, name)
    setattr(torch, name, (lambda o: (lambda *a, **k: poison(o(*a, **k))))(orig))
orig_new_empty = torch.Tensor.new_empty
torch.Tensor.new_empty = lambda self, *a, **k: poison(orig_new_empty(self, *a, **k))

from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from golden.cases import case_cfg  # noqa: E402
from golden.fill import fill_state_dict  # noqa: E402

device = torch.device("cuda", 0)


def run(case, overrides, flag, seed=11):
    c = case_cfg(case)
    c.merge_from_list(list(overrides))
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed))
    model = model.to(device).train()
    nimg = 3 if c.MODEL.DA_HEADS.TRIPLET_USE else 2
    images, targets = make_batch(c, nimg, 192, 320, seed=seed, device=device)
    opt = make_optimizer(c, model)
    opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
    enable_overlapped_rpn_backward(model)
    torch.manual_seed(seed)
    POISON[0] = flag
    losses = train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    POISON[0] = False
    return ({k: float(v) for k, v in losses.items()},
            {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad})


for case, ov in (("da_triplet_aligned", ("MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", 0.0)), ("da_triplet_aligned", ()),
                 ("da_img_only", ()), ("da_plain", ())):
    l0, g0 = run(case, ov, False)
    l1, g1 = run(case, ov, True)
    bad = [n for n, g in g1.items() if not torch.isfinite(g).all()]
    worst = max(float((g1[n].double() - g0[n].double()).norm()) / (float(g0[n].double().norm()) + 1e-30) for n in g0
                if n not in bad)
    print(case, ov, "losses equal:", all(abs(l0[k] - l1[k]) <= 1e-6 * max(1, abs(l0[k])) for k in l0),
          "NaN grads:", bad[:4], len(bad), "worst diff %.2e" % worst)
