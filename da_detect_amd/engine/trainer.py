"""Training loop of the DA detector (reference: maskrcnn_benchmark/engine/trainer.py:41-336).

`train_step` is the body the reference repeats per iteration (trainer.py:228-242): forward -> sum of losses ->
zero_grad -> backward (gradient buckets all-reduced while it runs) -> fused SGD -> scheduler update.
`do_da_train` keeps the reference's calling convention for the source/target(/auxiliary) loaders; logging
host syncs happen every 20 iterations only (the reference syncs every iteration through meters.update)."""
import time

import torch
import torch.distributed as dist

from ..utils.comm import get_world_size


def reduce_loss_dict(loss_dict):
    """rank-0 average of the loss scalars for logging (trainer.py:41-63)"""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        vals = torch.stack([loss_dict[k] for k in names], dim=0)
        dist.reduce(vals, dst=0)
        if dist.get_rank() == 0:
            vals /= world
        return {k: v for k, v in zip(names, vals)}


def enable_overlapped_rpn_backward(model, flag=True):
    """opt the model into RPNModule.early_backward (see its docstring); only valid with train_step's schedule"""
    rpn = getattr(model, "rpn", None)
    if rpn is not None and getattr(model, "roi_heads", None):
        rpn.early_backward = bool(flag)
        da = getattr(model, "da_heads", None)
        if da and hasattr(da, "early_image_level"):
            da.early_backward = bool(flag)     # image-level DA loss + backward in front of the box head
    return model


def train_step(model, optimizer, images, targets, scheduler=None, iteration=0):
    """one optimizer step; returns the (un-synchronised) loss dict.  Gradients are cleared BEFORE the forward pass
    (the reference clears them after it, trainer.py:237 — equivalent) so that a model opted into
    enable_overlapped_rpn_backward may start accumulating during its forward."""
    optimizer.zero_grad()
    loss_dict = model(images, targets)
    losses = sum(loss for loss in loss_dict.values())
    losses.backward()
    optimizer.step()
    if scheduler is not None and hasattr(scheduler, "step_update"):
        scheduler.step_update(iteration)
    return loss_dict


def do_da_train(model, source_data_loader, target_data_loader, optimizer, scheduler, checkpointer, device,
                checkpoint_period, arguments, cfg=None, negative_data_loader=None, logger=None, log_period=20):
    """joint iteration over the source / target (/ auxiliary) loaders (trainer.py:150-336).  Each loader yields
    (ImageList, list[BoxList], ids); batches are concatenated source-first exactly as trainer.py:215-224."""
    model.train()
    enable_overlapped_rpn_backward(model)
    start_iter = arguments.get("iteration", 0)
    loaders = [source_data_loader, target_data_loader] + ([negative_data_loader] if negative_data_loader else [])
    max_iter = len(source_data_loader)
    t0 = time.time()
    for iteration, batches in enumerate(zip(*loaders), start_iter):
        arguments["iteration"] = iteration
        images = batches[0][0]
        targets = list(batches[0][1])
        for b in batches[1:]:
            images = images + b[0]
            targets = targets + list(b[1])
        images = images.to(device)
        targets = [t.to(device) for t in targets]
        loss_dict = train_step(model, optimizer, images, targets, scheduler, iteration)
        if iteration % log_period == 0 or iteration == max_iter - 1:
            reduced = reduce_loss_dict(loss_dict)
            total = float(sum(v for v in reduced.values()))
            if torch.isnan(torch.tensor(total)):
                if logger:
                    logger.critical("NaN encountered!")
                return
            if logger:
                logger.info("iter %d  loss %.4f  %s  lr %.6f  %.3f s/it", iteration, total,
                            "  ".join("%s %.4f" % (k, float(v)) for k, v in reduced.items()),
                            optimizer.param_groups[0]["lr"], (time.time() - t0) / max(iteration - start_iter + 1, 1))
        if checkpointer is not None and checkpoint_period > 0 and iteration % checkpoint_period == 0 and iteration > 0:
            checkpointer.save("model_{:07d}".format(iteration), **arguments)
    if checkpointer is not None:
        checkpointer.save("model_final", **arguments)
