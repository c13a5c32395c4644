#!/usr/bin/env python
"""Training entry point with the flow of the reference's tools/train_net_triplet.py:54-220 on this package: build the
detector, the per-tensor SGD groups, the cosine schedule, the checkpointer, the source / target (/ auxiliary) loaders,
then engine.trainer.do_da_train.  One process per GPU:

    python tools/train_net_da.py --config-file configs/da_faster_rcnn/<yaml> \\
        --source ann.json,imgdir --target ann.json,imgdir [--auxiliary ann.json,imgdir] [KEY VALUE ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_net_da.py ...
    python tools/train_net_da.py --config-file <yaml> --synthetic 20        # no dataset: 20 seeded synthetic steps

Differences to the reference script, on purpose: DistributedDataParallel is replaced by the bucketed gradient reducer
attached to the fused optimizer (parallel/reducer.py); datasets are named by (annotation file, image root) pairs instead
of the path catalog; there is no periodic evaluation (engine.inference.inference produces the bbox.json records)."""
import argparse
import logging
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from da_detect_amd.config import cfg  # noqa: E402
from da_detect_amd.data.build import make_da_data_loaders  # noqa: E402
from da_detect_amd.engine.trainer import do_da_train, enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from da_detect_amd.solver.build import make_cosine_lr_scheduler  # noqa: E402
from da_detect_amd.utils.checkpoint import DetectronCheckpointer  # noqa: E402
from da_detect_amd.utils.comm import get_rank, synchronize  # noqa: E402


def setup_seed(seed):
    """train_net_triplet.py:46-51"""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def _pair(text):
    ann, root = text.split(",")
    return ann, root


def main():
    ap = argparse.ArgumentParser(description="DA Faster R-CNN training on MI355X")
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--source", type=_pair)
    ap.add_argument("--target", type=_pair)
    ap.add_argument("--auxiliary", type=_pair)
    ap.add_argument("--synthetic", type=int, default=0, help="run N steps on seeded synthetic batches instead of datasets")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE overrides of the yaml")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
        synchronize()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    logging.basicConfig(level=logging.INFO if get_rank() == 0 else logging.WARNING,
                        format="%(asctime)s %(name)s %(levelname)s: %(message)s")
    logger = logging.getLogger("maskrcnn_benchmark.trainer")

    setup_seed(100)
    device = torch.device("cuda", local_rank)
    model = build_detection_model(cfg).to(device)
    optimizer = make_optimizer(cfg, model)
    scheduler = make_cosine_lr_scheduler(cfg, optimizer)
    reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad])
    reducer.broadcast_parameters(0)
    optimizer.attach_reducer(reducer)

    output_dir = cfg.MODEL.OUTPUT_DIR
    checkpointer = DetectronCheckpointer(cfg, model, optimizer, scheduler, output_dir, save_to_disk=get_rank() == 0)
    weight = cfg.MODEL.WEIGHT
    arguments = {"iteration": 0}
    if weight and not weight.startswith("catalog://") and os.path.exists(weight):
        arguments.update(checkpointer.load(weight))
    else:
        logger.warning("MODEL.WEIGHT %r is not a local file: starting from the module initialisers", weight)

    if args.synthetic:
        from da_detect_amd.data.synthetic import make_batch

        if not (weight and os.path.exists(weight)):
            # no pretrained weights: the variance-preserving seeded init of bench.py — a ResNet with identity FrozenBN
            # statistics and the modules' own initialisers blows up within a few steps (R-101-FPN + DCN: NaN at step 2)
            import bench

            with torch.no_grad():
                bench.benchmark_init(model, 100)
        model.train()
        enable_overlapped_rpn_backward(model)
        n_img = 3 if cfg.MODEL.DA_HEADS.TRIPLET_USE else 2
        images, targets = make_batch(cfg, n_img, 608, 1216, seed=100 + get_rank(), device=device)
        for it in range(args.synthetic):
            losses = train_step(model, optimizer, images, targets, scheduler, it)
            if it % 5 == 0 or it == args.synthetic - 1:
                logger.info("iter %d  %s", it, "  ".join("%s %.4f" % (k, float(v)) for k, v in losses.items()))
        return

    specs = {k: v for k, v in (("source", args.source), ("target", args.target), ("auxiliary", args.auxiliary)) if v}
    assert "source" in specs and "target" in specs, "--source and --target are required (or --synthetic N)"
    loaders = make_da_data_loaders(cfg, specs, is_distributed=world > 1, start_iter=arguments["iteration"])
    do_da_train(model, loaders[0], loaders[1], optimizer, scheduler, checkpointer, device,
                cfg.SOLVER.CHECKPOINT_PERIOD, arguments, cfg=cfg,
                negative_data_loader=loaders[2] if len(loaders) > 2 else None, logger=logger)


if __name__ == "__main__":
    main()
