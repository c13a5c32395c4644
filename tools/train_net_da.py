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
from da_detect_amd.data.build import make_da_data_loaders, make_triplet_data_loader  # noqa: E402
from da_detect_amd.engine.trainer import do_da_train, enable_overlapped_rpn_backward, train_step  # noqa: E402
from da_detect_amd.modeling.detector import build_detection_model  # noqa: E402
from da_detect_amd.parallel.reducer import BucketedGradReducer  # noqa: E402
from da_detect_amd.solver import make_optimizer  # noqa: E402
from da_detect_amd.solver.build import make_cosine_lr_scheduler  # noqa: E402
from da_detect_amd.utils.checkpoint import DetectronCheckpointer  # noqa: E402
from da_detect_amd.utils.comm import get_rank, synchronize  # noqa: E402
from da_detect_amd.utils.metric_logger import MetricLogger  # noqa: E402


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
    ap.add_argument("--resume", action="store_true",
                    help="MODEL.WEIGHT is a checkpoint of THIS run: restore optimizer / scheduler state too and shorten "
                         "the loaders by its iteration (default: fine-tune — weights only, full schedule, as the "
                         "reference fork does with its optimizer restore commented out)")
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
    if output_dir and get_rank() == 0:
        os.makedirs(output_dir, exist_ok=True)       # train_net_triplet.py:311-312
    checkpointer = DetectronCheckpointer(cfg, model, optimizer, scheduler, output_dir, save_to_disk=get_rank() == 0)
    weight = cfg.MODEL.WEIGHT
    arguments = {"iteration": 0}
    try:
        extra = checkpointer.load(weight, load_optimizer=args.resume) if weight else {}
    except FileNotFoundError as e:
        if not args.synthetic:
            raise
        logger.warning("%s: starting from the module initialisers", e)
        extra = {}
    # the stored optimizer / scheduler state never travels into `arguments` (it would be written back, stale, by every
    # later checkpoint); a stored iteration only counts on --resume
    extra.pop("optimizer", None)
    extra.pop("scheduler", None)
    if not args.resume:
        extra.pop("iteration", None)
    arguments.update(extra)

    if args.synthetic:
        from da_detect_amd.data.synthetic import make_batch

        if not (weight and os.path.exists(weight)):   # catalog:// names that resolve were loaded above
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
    triplet = bool(cfg.MODEL.DA_HEADS.TRIPLET_USE)
    aligned = triplet and bool(cfg.MODEL.DA_HEADS.ALIGNMENT)
    if triplet and "auxiliary" not in specs:
        raise SystemExit("MODEL.DA_HEADS.TRIPLET_USE needs --auxiliary (the rainy / negative domain)")
    # --resume: the checkpoint holds the last COMPLETED iteration; the loaders and the loop continue behind it
    start = arguments["iteration"] + 1 if args.resume and "iteration" in extra else 0
    if aligned:
        # train_net_triplet.py:123-134: ALIGNMENT pools the three domains with the TARGET image's proposals, which is
        # only meaningful for index-aligned renderings of one scene -> ONE loader over aligned triplets
        source = negative = []
        positive = make_triplet_data_loader(cfg, specs, is_distributed=world > 1, start_iter=start)
    else:
        loaders = make_da_data_loaders(cfg, specs, is_distributed=world > 1, start_iter=start)
        source, positive = loaders[0], loaders[1]
        negative = loaders[2] if triplet else []
    do_da_train(model, source, positive, negative, None, optimizer, scheduler, checkpointer, device,
                cfg.SOLVER.CHECKPOINT_PERIOD, arguments, cfg, world > 1, MetricLogger(delimiter="  "),
                triplet_data_loading=triplet, triplet_data_aligned=aligned, start_iter=start)


if __name__ == "__main__":
    main()
