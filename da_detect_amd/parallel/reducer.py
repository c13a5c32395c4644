"""Data-parallel gradient reduction: flat gradient buckets + all-reduce overlapped with backward.

The reference wraps the model in torch DistributedDataParallel (reference: tools/train_net_triplet.py:83-88,
broadcast_buffers=False) — one process per GPU, bucketed NCCL all-reduce of 146 MB of fp32 gradients per step.
This module is the MI355X-side equivalent, built around the fused optimizer instead of around nn.Module:

  * every trainable parameter's .grad is a VIEW into a persistent flat fp32 bucket (same physical layout as
    the parameter, so the fused SGD kernel and RCCL both work on raw storage, and the optimizer's device-side
    pointer table never changes);
  * buckets are filled in reverse registration order (da heads -> box head -> rpn -> backbone), the order
    backward produces gradients; a post-accumulate hook counts arrivals and, when a bucket is complete, issues
    `all_reduce(async_op=True)` on it — RCCL runs it on its own stream while backward continues on the compute
    stream.  Priced for the step as it is since round 5 (backward ~6 ms at 1024 x 2048, not the > 100 ms of round 1):
    a ring all-reduce moves 2 (N - 1) / N x 146 MB per GPU and is bound by ONE xGMI link (~153 GB/s, 7 links per GPU,
    point to point) unless RCCL spreads rings over several links: 1.7 ms at N = 8 on one link, ~0.25 ms if all seven
    carry rings — between 4% and 28% of backward, so most of it can hide, but the LAST bucket cannot: it holds the
    first trainable layers (res2 / res3), completes when backward ends, and its collective (25 MB: 0.04 - 0.3 ms) plus
    everything behind it is an exposed tail.  Hence (round 6) the optimizer does not wait for ALL collectives and then
    update everything: finalize(per_bucket=...) hands each bucket over as its collective completes, the fused SGD
    updates that bucket's tensors, and the last bucket's all-reduce runs beside the earlier buckets' updates
    (solver/fused_sgd.py).  Bucket size: large enough to amortise a collective's launch latency (~20 us), small enough
    that the tail bucket is short — 25 MB by default like the reference's DDP;
  * parameters that receive no gradient in a step (e.g. the instance head when its loss weight is 0) are
    handled in `finalize()`: incomplete buckets are reduced there with their zero gradients, which is what
    DDP's find_unused_parameters achieves with a graph walk.  Collectives are issued in bucket order, so ONE
    bucket that never completes would push every later bucket's all-reduce to the end of backward — and the first
    bucket holds the DA heads, of which the image-level-only recipe never uses the instance head, and the triplet
    recipes never use the plain DA module.  After the first step the ranks therefore agree (one small all-reduce of a
    0/1 mask) on the parameters NO rank touched; from then on those are not waited for, their bucket goes out as soon
    as the used ones have arrived, and the overlap with backward is back.  A parameter classified unused that later
    does receive a gradient raises (the collective may already be on its way).
World size 1 keeps the flat views and skips communication.
"""
import torch
import torch.distributed as dist

from ..utils import streams


class BucketedGradReducer(object):
    def __init__(self, params, bucket_bytes=25 * 1024 * 1024, process_group=None, learn_unused=True,
                 always_communicate=False):
        self.params = [p for p in params if p.requires_grad]
        self.learn_unused = learn_unused
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # always_communicate: issue the bucket collectives also in a ONE-rank process group (a single-GPU box can then
        # drive RCCL itself — communicator set-up, stream ordering against the compute stream and the weight-gradient
        # lane, async work handles — through exactly the code path of N ranks; tests/test_multirank_gpu.py)
        self.communicate = self.world_size > 1 or (always_communicate and dist.is_available() and dist.is_initialized())
        self.buckets = []          # list of dict(flat=tensor, params=[...], pending=int, work=None)
        self._bucket_of = {}
        self._build(bucket_bytes)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._next = 0
        self._finalized = True
        self.touched = set()       # ids of the parameters that received a gradient since zero_grad()
        self.static_unused = None  # ids of the parameters no rank touched in the first step (None: not learned yet)
        self.mean_scale = 1.0      # what the buckets must still be multiplied by after finalize(mean=False)
        self._touched_any = frozenset()
        # communication evidence (record_comm(True); bench.py's N > 1 line, tests): per step — buckets whose collective
        # went out from a gradient hook (during backward) / only in finalize(), the GPU time the compute stream spent in
        # finalize() waiting for collectives (events on the compute stream around the waits: exposed communication), and the
        # collectives' own durations on RCCL's stream where the backend reports them (TORCH_NCCL_ENABLE_TIMING=1)
        self._record = False
        self._comm_steps = []
        self._step_rec = None

    def _build(self, bucket_bytes):
        cur, cur_bytes = [], 0
        groups = []
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        # ONE allocation for all buckets (each bucket a 256-byte aligned slice): zero_grad() is a single fill instead of
        # one per bucket, and the buckets stay separate tensors for the collectives
        layouts, total_all = [], 0
        for plist in groups:
            # 16-byte aligned slots so the optimizer's float4 path applies to every tensor
            offsets, total = [], 0
            for p in plist:
                offsets.append(total)
                total += (p.numel() + 3) // 4 * 4
            layouts.append((plist, offsets, total, total_all))
            total_all += (total + 63) // 64 * 64
        first = groups[0][0] if groups else None
        self._all = torch.zeros(total_all, dtype=first.dtype, device=first.device) if first is not None else None
        for plist, offsets, total, base in layouts:
            flat = self._all[base:base + total]
            b = dict(flat=flat, params=plist, pending=len(plist), work=None)
            for p, off in zip(plist, offsets):
                p.grad = flat[off:off + p.numel()].as_strided(p.size(), p.stride())
                self._bucket_of[id(p)] = b
            self.buckets.append(b)

    # -- per-step protocol: zero_grad() -> backward (hooks fire) -> finalize() -> optimizer.step() --------
    def zero_grad(self):
        unused = self.static_unused or ()
        # reduction passes still queued from a backward that no step() followed would be added into the NEXT step's
        # gradients (and pin ~1.5 GB of partial sums meanwhile): they belong to the gradients being cleared
        streams.discard_wgrad_reductions()
        if self._all is not None:
            self._all.zero_()
        for b in self.buckets:
            b["pending"] = sum(1 for p in b["params"] if id(p) not in unused)
            b["work"] = None
        self._next = 0
        self._finalized = False
        self.touched = set()
        if self._record:
            self._step_rec = dict(in_backward=0, in_finalize=0, works=[], wait=None, issued=[], backward_end=None)

    def record_comm(self, flag=True):
        self._record = bool(flag) and self.communicate
        self._comm_steps, self._step_rec = [], None

    def comm_summary(self, skip=0):
        """aggregate of the recorded steps (after `skip` leading ones).  Synchronises the device: call it outside timed
        regions.  allreduce_ms is None when the backend does not report durations."""
        steps = self._comm_steps[skip:]
        if not steps:
            return None
        if self.buckets and self.buckets[0]["flat"].is_cuda:
            torch.cuda.synchronize(self.buckets[0]["flat"].device)
        exposed, allred, have_dur = [], [], True
        for r in steps:
            exposed.append(r["wait"][0].elapsed_time(r["wait"][1]) if r["wait"] is not None else 0.0)
            tot = 0.0
            for w in r["works"]:
                try:
                    tot += float(w._get_duration())
                except Exception:       # noqa: BLE001 — gloo, timing disabled, older c10d
                    have_dur = False
                    break
            allred.append(tot)
        n = float(len(steps))
        out = {"backend": dist.get_backend(self.group) if dist.is_initialized() else None, "world": self.world_size,
               "buckets": len(self.buckets), "bucket_mb": round(max(b["flat"].numel() for b in self.buckets) * 4 / 2**20, 2),
               "grad_mb": round(self.grad_bytes() / 2**20, 2), "steps": len(steps),
               "buckets_issued_during_backward": sum(r["in_backward"] for r in steps) / n,
               "buckets_issued_in_finalize": sum(r["in_finalize"] for r in steps) / n,
               "exposed_ms": round(sum(exposed) / n, 4),
               "allreduce_ms": round(sum(allred) / n, 4) if have_dur else None}
        # per bucket: when its collective was issued, on the compute stream's time line, relative to the end of backward
        # (negative: that long before backward ended, i.e. that much room to hide behind it; ~0: the exposed tail)
        offs = [[r["backward_end"].elapsed_time(ev) for ev in r["issued"]] for r in steps
                if r.get("backward_end") is not None and len(r["issued"]) == len(self.buckets)]
        out["bucket_issue_offsets_ms"] = ([round(sum(o[i] for o in offs) / len(offs), 3) for i in range(len(self.buckets))]
                                          if offs else None)
        if have_dur and out["allreduce_ms"]:
            out["overlap_frac"] = round(max(0.0, 1.0 - out["exposed_ms"] / out["allreduce_ms"]), 4)
        else:
            out["overlap_frac"] = None
        out["note"] = ("exposed_ms = GPU time between finalize()'s first wait and its last one on the compute stream (what "
                       "the step pays for communication that backward did not hide); allreduce_ms = sum of the collectives' "
                       "durations on the backend's stream")
        return out

    def _launch_ready(self, force=False):
        """collectives must be issued in the same order on every rank: buckets are launched strictly by index,
        a ready bucket waits for its predecessors (which may only complete in finalize() on some ranks)"""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["pending"] > 0 and not force:
                return
            if self.communicate:
                # split weight gradients whose reduction pass was deferred (utils.streams.WgradLane.reduce_batch) must be
                # complete in this bucket before it goes out
                streams.flush_wgrad_reductions(b["flat"].device)
                if self._step_rec is not None and b["flat"].is_cuda:
                    ev = torch.cuda.Event(enable_timing=True)      # where on the compute stream's time line it went out
                    ev.record()
                    self._step_rec["issued"].append(ev)
                b["work"] = self._all_reduce(b["flat"])
                if self._step_rec is not None:
                    self._step_rec["in_finalize" if force else "in_backward"] += 1
                    self._step_rec["works"].append(b["work"])
            self._next += 1

    def _all_reduce(self, flat):
        """a collective is ordered after the stream it is issued from.  Weight gradients accumulated directly on the
        weight-gradient lane (utils.streams.run_into) must have landed, and so must the gradients autograd accumulated
        on the compute stream: the collective is issued FROM THE LANE after the lane waited for the compute stream's
        current position — the compute stream itself is not held up."""
        dev = flat.device
        if dev.type == "cuda" and streams.DIRECT_WGRAD and streams.lane_in_use():
            lane = streams.side_stream(dev, 2)
            lane.wait_event(torch.cuda.current_stream(dev).record_event())
            with torch.cuda.stream(lane):
                return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        self.touched.add(id(p))
        if self.static_unused and id(p) in self.static_unused:
            raise RuntimeError("BucketedGradReducer: a parameter of shape %s received a gradient after no rank had "
                               "touched it in the first step; its bucket is no longer waiting for it.  Build the "
                               "reducer with learn_unused=False for graphs that change between steps" %
                               (tuple(p.shape),))
        b["pending"] -= 1
        if b["pending"] == 0 and not self._finalized:
            self._launch_ready()

    def can_hand_over_buckets(self):
        """finalize(per_bucket=...) may be used: collectives are in flight and the set of parameters to update is known
        before they complete (from the second step on: the ranks have agreed on the unused parameters)"""
        return self.communicate and (self.world_size == 1 or self.static_unused is not None)

    def finalize(self, mean=True, per_bucket=None):
        """reduce buckets that never completed and wait for every collective.  mean=True turns the sums into means here
        (one multiply per bucket); mean=False leaves the SUMS in the buckets and the caller applies `mean_scale` itself —
        the fused optimizer folds it into the SGD kernel's gradient read (solver/fused_sgd.py), six launches fewer.
        per_bucket(i, bucket): called for every bucket, in order, as soon as the compute stream has been made to wait for
        THAT bucket's collective — work the callback queues runs beside the later buckets' collectives."""
        if self._finalized:
            return
        if self.buckets:
            streams.join_wgrad_lane(self.buckets[0]["flat"].device)
        rec = self._step_rec
        cuda = bool(self.buckets) and self.buckets[0]["flat"].is_cuda
        if rec is not None and cuda:
            rec["backward_end"] = torch.cuda.Event(enable_timing=True)
            rec["backward_end"].record()
        self._launch_ready(force=True)
        self.mean_scale = 1.0 if (mean or self.world_size == 1) else 1.0 / self.world_size
        if self.communicate:
            if rec is not None and cuda:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            for i, b in enumerate(self.buckets):
                if b["work"] is not None:
                    b["work"].wait()
                if per_bucket is not None:
                    if mean and self.world_size > 1:
                        b["flat"].mul_(1.0 / self.world_size)
                    per_bucket(i, b)
            if rec is not None and cuda:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                rec["wait"] = (e0, e1)
            if rec is not None:
                self._comm_steps.append(rec)
                self._step_rec = None
            if mean and self.world_size > 1 and per_bucket is None:
                for b in self.buckets:
                    b["flat"].mul_(1.0 / self.world_size)
        elif per_bucket is not None:
            for i, b in enumerate(self.buckets):
                per_bucket(i, b)
        self._finalized = True
        if self.communicate:
            if self.static_unused is None and self.learn_unused:
                self._learn_unused()
            elif self.static_unused is None and self.world_size > 1:
                self._agree_touched()

    def _agree_touched(self):
        """graphs that change between steps (learn_unused=False): the parameters ANY rank touched in this step, agreed on
        through one small all-reduce per step — what `update_ids` then reports on every rank alike"""
        dev = self.buckets[0]["flat"].device if self.buckets else torch.device("cpu")
        mask = torch.tensor([1.0 if id(p) in self.touched else 0.0 for p in self.params], dtype=torch.float32, device=dev)
        if mask.numel():
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        self._touched_any = frozenset(id(p) for p, m in zip(self.params, mask.tolist()) if m > 0.5)

    def update_ids(self):
        """ids of the parameters the optimizer must update after finalize().  One rank: those that received a gradient
        (torch.optim.SGD skips `grad is None`: no weight decay, no momentum step).  Several ranks: every parameter ANY
        rank touched — its averaged gradient is the same everywhere, so is its update (DDP with find_unused_parameters
        behaves the same); deciding by the LOCAL touched set would let weights, momentum and weight decay of a
        parameter used on rank A only drift apart between the ranks.
        Known difference from one rank (and from DDP, which leaves the gradient of a globally unused parameter None): with
        the learned `static_unused` mask, a parameter that was touched in the FIRST step and that no rank touches in some
        later step (the instance head in a step where no ROI is sampled) still takes weight decay and a momentum step on
        its zero gradient there.  learn_unused=False re-agrees every step and has no such case."""
        if self.world_size == 1:
            return self.touched
        if self.static_unused is not None:
            return frozenset(id(p) for p in self.params) - self.static_unused
        return self._touched_any

    def _learn_unused(self):
        """end of the first step: the parameters that NO rank touched (agreed on through one all-reduce of a mask)"""
        dev = self.buckets[0]["flat"].device if self.buckets else torch.device("cpu")
        mask = torch.tensor([0.0 if id(p) in self.touched else 1.0 for p in self.params], dtype=torch.float32, device=dev)
        if self.communicate and mask.numel():
            dist.all_reduce(mask, op=dist.ReduceOp.MIN, group=self.group)
        self.static_unused = frozenset(id(p) for p, m in zip(self.params, mask.tolist()) if m > 0.5)

    def broadcast_parameters(self, src=0):
        """rank-0 parameters to every rank at start-up (what DDP's constructor does; buffers are not
        broadcast, matching broadcast_buffers=False)"""
        if self.world_size > 1:
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)
            # a write through .data does not bump the tensors' autograd version: the caches of transposed / padded weights
            # (_C._TransposeCache, backbone.resnet._PaddedWeights) are keyed by (version, weight epoch)
            from .. import _C
            _C.bump_weight_epoch()

    def grad_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)
