"""Training loop of the DA detector (reference: maskrcnn_benchmark/engine/trainer.py:41-336).

`train_step` is the body the reference repeats per iteration (trainer.py:228-242): forward -> sum of losses ->
zero_grad -> backward (gradient buckets all-reduced while it runs) -> fused SGD -> scheduler update.
`do_da_train` / `do_train` have the reference's signatures (trainer.py:150-167 / 66-75) and batch conventions, so
tools/train_net_triplet.py:182-215 calls them unchanged.  What differs underneath: a DistributedDataParallel wrapper
is unwrapped and replaced by the bucketed gradient reducer attached to the fused optimizer (parallel/reducer.py —
DDP's one-backward-per-forward contract does not admit the overlapped RPN / DA backward); the per-iteration host
synchronisations of the reference (meters.update -> .item(), the NaN test) are deferred to the logging period."""
import datetime
import logging
import os
import time

import torch
import torch.distributed as dist

from ..utils import streams
from ..utils.comm import get_world_size, synchronize
from ..utils.metric_logger import MetricLogger


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


class WgradLaneTuner(object):
    """Picks, by measurement during the first iterations, whether the weight-gradient GEMMs of the narrow layers run on a
    second stream beside the data-gradient chain (utils.streams.WgradLane, `rows <= WGRAD_LANE_ROWS`).

    The same switch is worth +5% on one recipe and costs 19% on another (one MI355X, 1024x2048: image-level DA only,
    where the box head and the RPN head see half the rows, 20.9 -> 19.9 ms per step with the lane for GEMMs of up to
    17 000 rows; image + instance + consistency 29.4 -> 34.9 ms) — small tile grids leave CUs idle that a second GEMM
    fills, full ones are only slowed down.  So it is not a constant: every candidate runs `settle` untimed and `measure`
    timed iterations (ordinary training steps — the schedule changes the order of kernels, not a single result), and
    the lane is kept when it is at least MIN_GAIN faster than one stream.  DADET_WGRAD_LANE_ROWS / DADET_WGRAD_STREAM set by hand switch the tuner off."""

    CANDIDATES = (0, 17000)
    MIN_GAIN = 0.03

    def __init__(self, device, settle=2, measure=4):
        self.device = device
        self.settle, self.measure = settle, measure
        # Several ranks (round 5): every rank runs the same candidates for the same number of steps — the steps are
        # ordinary training steps with their gradient collectives, so the ranks stay in lockstep — and the decision is taken
        # on the MAXIMUM over ranks of each candidate's time (agreed_times: one all-reduce), i.e. on the time of the slowest
        # rank, which is the step time of the job: every rank keeps the same schedule, the one N = 1 would be measured
        # with.  Over RCCL only: over gloo (the one-GPU functional rig) collectives issued from the lane stream left the
        # process 10x slower for good (profiles/r03_gloo_two_ranks_one_gpu_lane_phases.json), which RCCL does not show
        # (profiles/r03_rccl_one_rank_first_contact.json).  DADET_TUNE_SCHEDULE_RANKS=0 keeps N > 1 on one stream.
        world = get_world_size()
        multi_ok = world == 1 or (torch.distributed.get_backend() == "nccl"
                                  and os.environ.get("DADET_TUNE_SCHEDULE_RANKS", "1") == "1")
        self.active = (device.type == "cuda" and "DADET_WGRAD_LANE_ROWS" not in os.environ
                       and not streams.WGRAD_OVERLAP and os.environ.get("DADET_TUNE_SCHEDULE", "1") == "1"
                       and multi_ok)
        self.times = {}
        self._cand = self._count = 0
        self._t0 = None

    def step_begin(self):
        if not self.active:
            return
        if self._count == 0:
            streams.join_wgrad_lane(self.device)           # nothing of the previous candidate is in flight on the lane
            streams.WGRAD_LANE_ROWS = self.CANDIDATES[self._cand]
        if self._count == self.settle:
            torch.cuda.synchronize(self.device)
            self._t0 = time.perf_counter()

    def step_end(self):
        if not self.active:
            return
        self._count += 1
        if self._count < self.settle + self.measure:
            return
        torch.cuda.synchronize(self.device)
        self.times[self.CANDIDATES[self._cand]] = (time.perf_counter() - self._t0) / self.measure
        self._cand, self._count = self._cand + 1, 0
        if self._cand == len(self.CANDIDATES):
            streams.join_wgrad_lane(self.device)
            # the default (first candidate: one GEMM stream) stays unless another one is CLEARLY faster: four timed steps
            # carry ~1% of noise, and a second stream makes per-kernel timings harder to read (bench.py's roofline line)
            self.times = agreed_times(self.times, self.device)
            streams.WGRAD_LANE_ROWS = self.choose(self.times)
            self.active = False

    @classmethod
    def choose(cls, times):
        """the candidate to keep: the default (first candidate: one GEMM stream) unless another is CLEARLY faster"""
        base = cls.CANDIDATES[0]
        best = min(times, key=times.get)
        if times[best] > (1.0 - cls.MIN_GAIN) * times[base]:
            best = base
        return best

    def close(self):
        """a run that ended before the measurement did: back to the default"""
        if self.active:
            streams.join_wgrad_lane(self.device)
            streams.WGRAD_LANE_ROWS = 0
            self.active = False

    def report(self):
        return {"wgrad_lane_rows": streams.WGRAD_LANE_ROWS,
                "tuned_ms_per_step": {str(k): round(v * 1e3, 3) for k, v in self.times.items()}}


def agreed_times(times, device):
    """candidate -> seconds per step as EVERY rank will see it: the maximum over the ranks (the slowest rank sets the step
    time of a data-parallel job), so that all ranks take the same schedule decision.  One rank: unchanged."""
    if get_world_size() == 1:
        return dict(times)
    keys = sorted(times)
    on = device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([times[k] for k in keys], dtype=torch.float64, device=on)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


def _prepare(model, optimizer, distributed):
    """the network that is actually stepped + its gradient reducer (see the module docstring).

    A DistributedDataParallel wrapper is taken off ONLY when this module's own reducer does (or will do) the gradient
    exchange, i.e. the optimizer is the fused one (`attach_reducer`).  With any other optimizer (the reference's
    signature allows torch.optim.SGD) the wrapper is the only gradient synchronisation there is: it stays, and the
    overlapped RPN / DA backward — several backward passes per forward, which DDP's contract does not admit — stays off."""
    wrapped = isinstance(model, torch.nn.parallel.DistributedDataParallel)
    own_reducer = hasattr(optimizer, "attach_reducer")
    if wrapped and not own_reducer:
        model.train()
        enable_overlapped_rpn_backward(model.module, False)
        return model
    net = _unwrap(model)
    if own_reducer and getattr(optimizer, "reducer", None) is None \
            and (distributed or get_world_size() > 1 or next(net.parameters()).is_cuda):
        from ..parallel.reducer import BucketedGradReducer

        reducer = BucketedGradReducer([p for p in net.parameters() if p.requires_grad])
        reducer.broadcast_parameters(0)
        optimizer.attach_reducer(reducer)
    if get_world_size() > 1 and getattr(optimizer, "reducer", None) is None:
        raise RuntimeError("world size %d without gradient synchronisation: pass the model wrapped in "
                           "DistributedDataParallel or use solver.make_optimizer (fused SGD + bucketed reducer)"
                           % get_world_size())
    net.train()
    enable_overlapped_rpn_backward(net)
    return net


def _log_line(logger, meters, iteration, max_iter, optimizer):
    eta = str(datetime.timedelta(seconds=int(meters.time.global_avg * (max_iter - iteration))))
    mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if torch.cuda.is_available() else 0.0
    logger.info(meters.delimiter.join(["eta: {eta}", "iter: {iter}", "{meters}", "lr: {lr:.6f}", "max mem: {memory:.0f}"])
                .format(eta=eta, iter=iteration, meters=str(meters), lr=optimizer.param_groups[0]["lr"], memory=mem))


def _update_meters(meters, loss_dict):
    reduced = reduce_loss_dict(loss_dict)
    vals = [v.detach() for v in reduced.values()]
    total = torch.stack([v.reshape(()) for v in vals]).sum() if vals else torch.zeros(())
    meters.update(loss=total, **{k: v.detach() for k, v in reduced.items()})
    return total


def _loss_is_nan(net, total, logger=None):
    """NaN in this step's summed loss, or — between two tests — in the parameters an earlier NaN step has since
    poisoned (SGD carries a NaN gradient into the weights for good; one small tensor is enough to see it).  Before that:
    the GEMMs' own non-finite guard (_C.check_nonfinite), which names the first launch that overflowed instead of a loss
    hundreds of kernels later (contraction mode 4: a per-tensor scale from too small a maximum).  Its FloatingPointError
    is caught HERE and logged with the launch's name: the caller then leaves through the same orderly exit as for a NaN
    loss (tuner closed, nothing written).  With more than one rank the verdict is all-reduced (max), so a rank whose peers
    overflowed does not go on into a gradient collective nobody else joins: every rank calls this at the same iterations."""
    overflow = None
    if total.is_cuda:
        from .. import _C

        try:
            _C.check_nonfinite()
        except FloatingPointError as e:
            overflow = str(e)
    bad = torch.isnan(total).any()
    probe = next((p for p in net.parameters() if p.requires_grad), None)
    if probe is not None:
        bad = bad | torch.isnan(probe.detach().sum())
    bad = bool(bad) or overflow is not None
    if overflow is not None:
        (logger or logging.getLogger("maskrcnn_benchmark.trainer")).critical("non-finite GEMM sums: %s" % overflow)
    if get_world_size() > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
        on = total.device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1.0 if bad else 0.0], device=on)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if not bad and float(flag) > 0:
            (logger or logging.getLogger("maskrcnn_benchmark.trainer")).critical(
                "another rank reported a NaN loss / non-finite GEMM sums: leaving with it")
        bad = bad or float(flag) > 0
    return bad


def do_train(model, data_loader, optimizer, scheduler, checkpointer, device, checkpoint_period, arguments):
    """single-domain loop (trainer.py:66-146): iteration counts from arguments["iteration"] + 1, `scheduler.step()`
    BEFORE the optimizer step, checkpoints every `checkpoint_period` and at the end"""
    logger = logging.getLogger("maskrcnn_benchmark.trainer")
    logger.info("Start training")
    meters = MetricLogger(delimiter="  ")
    max_iter = len(data_loader)
    start_iter = arguments["iteration"]
    net = _prepare(model, optimizer, get_world_size() > 1)
    tuner = WgradLaneTuner(torch.device(device))
    start_time = end = time.time()
    for iteration, (images, targets, _) in enumerate(data_loader, start_iter):
        data_time = time.time() - end
        iteration = iteration + 1
        arguments["iteration"] = iteration
        scheduler.step()
        images = images.to(device)
        targets = [t.to(device) for t in targets]
        tuner.step_begin()
        loss_dict = train_step(net, optimizer, images, targets)
        tuner.step_end()
        _update_meters(meters, loss_dict)
        meters.update(time=time.time() - end, data=data_time)
        end = time.time()
        if iteration % 20 == 0 or iteration == max_iter:
            _log_line(logger, meters, iteration, max_iter, optimizer)
        if checkpoint_period > 0 and iteration % checkpoint_period == 0:
            checkpointer.save("model_{:07d}".format(iteration), **arguments)
        if iteration == max_iter:
            checkpointer.save("model_final", **arguments)
    tuner.close()
    total = time.time() - start_time
    logger.info("Total training time: {} ({:.4f} s / it)".format(str(datetime.timedelta(seconds=total)),
                                                                 total / max(max_iter, 1)))


def _da_batches(source_data_loader, positive_target_data_loader, negative_target_data_loader, triplet_data_loading,
                triplet_data_aligned):
    """-> iterator over (ImageList, list[BoxList]) with the reference's batch layout: source images first, then the
    target (positive) and — for the triplet recipes — the auxiliary (negative) ones (trainer.py:186-226)"""
    if triplet_data_loading and triplet_data_aligned:
        # ONE loader over index-aligned triplets (data.build.make_data_loader_da): nine fields per batch
        for s_img, s_tgt, p_img, p_tgt, n_img, n_tgt, _, _, _ in positive_target_data_loader:
            yield s_img + p_img + n_img, list(s_tgt) + list(p_tgt) + list(n_tgt)
        return
    loaders = [source_data_loader, positive_target_data_loader]
    if triplet_data_loading:
        loaders.append(negative_target_data_loader)
    for parts in zip(*loaders):
        images, targets = parts[0][0], list(parts[0][1])
        for img, tgt, _ in parts[1:]:
            images = images + img
            targets = targets + list(tgt)
        yield images, targets


def do_da_train(model, source_data_loader, positive_target_data_loader, negative_target_data_loader, data_loader_val,
                optimizer, scheduler, checkpointer, device, checkpoint_period, arguments, cfg, distributed, meters,
                triplet_data_loading=True, triplet_data_aligned=True, start_iter=0):
    """joint source / target (/ auxiliary) loop with the reference's signature and conventions (trainer.py:150-336):
    the iteration index starts at 0 whatever `arguments` holds (trainer.py:177; `start_iter` — not in the reference —
    is for a true resume, tools/train_net_da.py --resume), max_iter = len(positive loader),
    `scheduler.step_update(iteration)` after the optimizer step, a checkpoint + `scheduler.step(epoch)` every
    `checkpoint_period` iterations except at 0, `model_final` at max_iter - 1, periodic evaluation on
    `data_loader_val[0]` when cfg.MODEL.EVAL_USE_IN_TRAINING.  NaN losses end the run (tested at the logging period
    and at every checkpoint instead of every iteration: the test is a host synchronisation)."""
    from .inference import inference

    logger = logging.getLogger("maskrcnn_benchmark.trainer")
    logger.info("Start training")
    if meters is None:
        meters = MetricLogger(delimiter="  ")
    if triplet_data_loading and cfg is not None and cfg.MODEL.DA_HEADS.ALIGNMENT and not triplet_data_aligned:
        raise ValueError("MODEL.DA_HEADS.ALIGNMENT pools all three domains with the TARGET image's proposals: it needs "
                         "the index-aligned triplet loader (make_data_loader_da / make_triplet_data_loader, "
                         "triplet_data_aligned=True); independent per-domain loaders feed it unrelated images")
    eval_in_training = bool(cfg.MODEL.EVAL_USE_IN_TRAINING) if cfg is not None else False
    max_iter = len(positive_target_data_loader)
    net = _prepare(model, optimizer, distributed)
    tuner, tuner_logged = WgradLaneTuner(torch.device(device)), False
    start_time = end = time.time()
    batches = _da_batches(source_data_loader, positive_target_data_loader, negative_target_data_loader,
                          triplet_data_loading, triplet_data_aligned)
    for iteration, (images, targets) in enumerate(batches, start_iter):
        data_time = time.time() - end
        arguments["iteration"] = iteration
        images = images.to(device)
        targets = [t.to(device) for t in targets]
        tuner.step_begin()
        loss_dict = train_step(net, optimizer, images, targets, scheduler, iteration)
        tuner.step_end()
        if tuner.times and not tuner.active and not tuner_logged:
            tuner_logged = True
            logger.info("weight-gradient lane: %s" % (tuner.report(),))
        total = _update_meters(meters, loss_dict)
        meters.update(time=time.time() - end, data=data_time)
        end = time.time()
        at_checkpoint = checkpoint_period > 0 and iteration % checkpoint_period == 0 and iteration != 0
        if iteration % 20 == 0 or iteration == max_iter:
            _log_line(logger, meters, iteration, max_iter, optimizer)
        # the NaN test is a host synchronisation: it runs at the logging period and BEFORE anything is written — a
        # checkpoint of poisoned weights would also retag `last_checkpoint` (the reference tests every iteration)
        if (iteration % 20 == 0 or at_checkpoint or iteration == max_iter - 1) and _loss_is_nan(net, total, logger):
            logger.critical("Loss is NaN, exiting...")
            tuner.close()
            return
        if at_checkpoint:
            checkpointer.save("model_{:07d}".format(iteration), **arguments)
            scheduler.step(int(iteration / checkpoint_period))
        if iteration == max_iter - 1:
            checkpointer.save("model_final", **arguments)
        if eval_in_training and at_checkpoint and data_loader_val is not None:
            synchronize()
            with torch.no_grad():
                inference(net, data_loader_val[0], dataset_name="[Validation]", iou_types=("bbox",),
                          box_only=False if cfg.MODEL.RETINANET_ON else cfg.MODEL.RPN_ONLY, device=cfg.MODEL.DEVICE,
                          expected_results=cfg.TEST.EXPECTED_RESULTS,
                          expected_results_sigma_tol=cfg.TEST.EXPECTED_RESULTS_SIGMA_TOL, output_folder=None)
            synchronize()
            net.train()
    tuner.close()
    total_time = time.time() - start_time
    logger.info("Total training time: {} ({:.4f} s / it)".format(str(datetime.timedelta(seconds=total_time)),
                                                                 total_time / max(max_iter, 1)))
