"""End-to-end bring-up on the GPU: training steps at a small and the full BASELINE size with timings."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from da_detect_amd.config import cfg  # noqa: E402
from da_detect_amd.data.synthetic import make_batch  # noqa: E402
from da_detect_amd.engine.trainer import train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(yaml, nimg, H, W, steps):
    dev = torch.device("cuda:0")
    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, yaml))
    torch.manual_seed(0)
    model = build_detection_model(c).to(dev)
    model.train()
    opt = make_optimizer(c, model)
    opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
    images, targets = make_batch(c, nimg, H, W, seed=100, device=dev)
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.time()
        losses = train_step(model, opt, images, targets)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("%s %dx%d step %d: %.1f ms  %s" % (os.path.basename(yaml)[:40], H, W, i, dt * 1e3,
                                                 {k: round(float(v), 4) for k, v in losses.items()}), flush=True)


if __name__ == "__main__":
    run("configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml", 2, 256, 512, 3)
    run("configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml", 3, 256, 512, 3)
    run("configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_img_only.yaml", 2, 1024, 2048, 6)
