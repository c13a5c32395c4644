#!/usr/bin/env python
"""bench.py — train images/sec of DA Faster R-CNN R-50-C4 on synthetic Cityscapes-shaped batches.

Workload (BASELINE.json configs[1]): configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_img_only.yaml — image-level
DA head only, 1 source + 1 target image of 1024 x 2048 per GPU per step (SOLVER.IMS_PER_BATCH = 2 * n_gpus),
256 ROIs sampled per image, SGD(momentum) step included.  One "step" = forward + backward (+ bucketed gradient
all-reduce when n_gpus > 1) + fused SGD over one such batch; inputs are resident in HBM before the timed region.
Weak scaling: every rank processes its own (source, target) pair.
Work the recipe never reads is not evaluated (DESIGN.md section 4 "Unread work"; the line's `elided` list says what,
`flop_per_step` says how many algorithmic FLOPs a step executes, `step_frac` = flop_per_step / time / ceiling).  After
the headline loop the other BASELINE recipes (configs[2] `da`, configs[3] `triplet`, configs[4] `fpn_dcn_da`: R-101-FPN +
DCN) are timed for a few steps each, every one in a process of its own, and reported under `other_workloads`; `resolutions`
holds `img_only` and `da` at the reference yaml's own training size (INPUT.MIN/MAX_SIZE_TRAIN 600 / 1200 -> 608 x 1216
padded).  With N > 1 ranks the line carries `comm` (bucket count, all-reduce time, exposed wait in finalize()).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line (metric / value / roofline / cpu_baseline ...).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ROCm runtime knob, read when the HIP runtime initialises: kernel arguments are written straight to device memory
# instead of being staged through host-coherent memory — lower launch latency for the ~480 launches of a step
# (measured on one box, 4 alternating runs each: 63.8 vs 62.8 images/s)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

YAML = "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_img_only.yaml"
HEIGHT, WIDTH, IMAGES_PER_GPU = 1024, 2048, 2
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
GEMM_MODES = {
    4: ("fp32 operands, scaled per tensor by a power of two, split into 2 fp16 terms, 3 x v_mfma_f32_32x32x16_f16 per "
        "K=16, fp32 accumulate (error against fp64 at production K: RMS <= 1.1x, max <= 1.5x of the exact-fp32 kernel's "
        "error, tests/test_ops_gpu.py::test_split_bf16_accuracy_at_production_k)", 3),
    0: ("exact fp32 MFMA (v_mfma_f32_32x32x2_f32)", 1),
    3: ("fp32 operands split into 3 bf16 terms, 6 x v_mfma_f32_32x32x16_bf16 per K=16, fp32 accumulate "
        "(error against fp64 at production K: RMS <= 1.1x, max <= 1.5x of the exact-fp32 kernel's error, "
        "tests/test_ops_gpu.py::test_split_bf16_accuracy_at_production_k)", 6),
    2: ("fp32 operands split into 2 bf16 terms, 3 x v_mfma_f32_32x32x16_bf16 per K=16 (~2^-16 products)", 3),
}


WORKLOADS = {
    # name: (yaml, overrides, images per GPU per step, description)
    "img_only": (YAML, [], 2, "R-50-C4, image-level DA head only, 1 source + 1 target"),
    "da": ("configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml", [], 2,
           "R-50-C4, image + instance DA heads + consistency, 1 source + 1 target"),
    "triplet": ("configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml", [],
                3, "R-50-C4 triplet recipe as shipped (AdvGRL, image triplet), source + foggy + rainy"),
    "triplet_aligned": ("configs/da_faster_rcnn/e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml",
                        ["MODEL.DA_HEADS.ALIGNMENT", True, "MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", 1.0], 3,
                        "R-50-C4 triplet recipe with ALIGNMENT (3 extra box-head passes) + instance triplet"),
    "fpn_dcn_da": ("configs/da_faster_rcnn/e2e_da_faster_rcnn_R_101_FPN_DCN_cityscapes_to_foggy_cityscapes.yaml", [],
                   2, "R-101-FPN + DCN (res3-5) + DA heads over the pyramid, 1 source + 1 target"),
}


def benchmark_init(model, seed):
    """seeded variance-preserving random init (no network for the MSRA R-50 pickle): He-normal conv weights,
    identity FrozenBN statistics, the last BN of every bottleneck scaled by 0.25 so that 16 un-normalised
    residual blocks keep O(1) activations.  Head initialisers are the reference's own (normal 0.01 / 0.001)."""
    from da_detect_amd.layers import FrozenBatchNorm2d
    from da_detect_amd.modeling.backbone.resnet import Bottleneck

    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, Bottleneck):
            conv2 = mod.conv2.conv if getattr(mod, "with_dcn", False) else mod.conv2   # DFConv2d wraps the 3x3
            convs = [mod.conv1, conv2, mod.conv3] + ([mod.downsample[0]] if mod.downsample is not None else [])
            for conv in convs:
                fan_in = conv.weight.shape[1] * conv.weight.shape[2] * conv.weight.shape[3]
                w = torch.randn(conv.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
                conv.weight.data.copy_(w)
            mod.bn3.weight.fill_(0.25)
    stem = model.backbone.body.stem
    stem.conv1.weight.data.copy_(torch.randn(stem.conv1.weight.shape, generator=g) * (2.0 / 147) ** 0.5 / 60.0)
    for mod in model.modules():
        if isinstance(mod, FrozenBatchNorm2d):
            mod._cache = None


def build(cfg_path, device, seed, overrides=()):
    from da_detect_amd.config import cfg
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.solver import make_optimizer

    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, cfg_path))
    if overrides:
        c.merge_from_list(list(overrides))
    torch.manual_seed(seed)
    model = build_detection_model(c)
    benchmark_init(model, seed)
    model = model.to(device)
    model.train()
    opt = make_optimizer(c, model)
    reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad])
    reducer.broadcast_parameters(0)
    opt.attach_reducer(reducer)
    return c, model, opt, reducer


def cpu_baseline(cfg_path, seed, budget_s=90.0):
    """the CPU oracle (oracle/model_ref.py: a torch-CPU fp32 restatement of the same training step, with the
    C restatements of NMS / ROIAlign) timed on the host cores on ONE step of the same 2 x 1024 x 2048 batch (~25 s on
    the GPU box's 64-core EPYC, plus a 2 s warm-up step)."""
    from oracle import model_ref

    return model_ref.timed_training_sample(os.path.join(ROOT, cfg_path), seed, HEIGHT, WIDTH, IMAGES_PER_GPU,
                                           benchmark_init, budget_s)


# the kernel family bracketed inside the timed region (mode 4): by summed time the largest one of the step
DOMINANT_FAMILY = "conv_big_kernel<256>"
# profiles/r05_gemm_lab.txt part 1: a register-only stream of v_mfma_f32_32x32x16_f16 on operands with the bit statistics of
# split fp32 data (leading / residual fp16 terms of gaussians) sustains 1625 TFLOP/s on this part — its power management
# lowers the clock under random operand bits (2464 on zeros) — i.e. 541.7 TFLOP/s ALGORITHMIC in the three-MFMA contraction.
# Reported next to the nominal ceiling, never instead of it.
F16_STREAM_ON_SPLIT_OPERANDS_TFLOPS = 1625.0
# kernel families of the mode-4 step for `roofline.families` (name fragments of the profiler's labels)
FAMILIES = [("fwd_dgrad_256x256", ("conv_big_kernel<256>",)), ("fwd_dgrad_256x128", ("conv_big128_kernel",)),
            ("wgrad_256x256", ("conv_wgrad_big_kernel", "conv_wgrad_big_group_kernel")), ("fwd_dgrad_128x128", ("conv_fwd_split_kernel", "conv_fwd_split_sk_kernel")),
            ("wgrad_128x128", ("conv_wgrad_split_kernel", "conv_wgrad_split_group_kernel")), ("weight_stationary_1x1", ("conv1x1_ws_kernel",))]


def pmc_traffic(kernel_name, gemm_mode):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes (profiles/r06_pmc_hbm_traffic.json; mode 3: r04_pmc_hbm_traffic.json,
    made by tools/profile_pmc.sh from this same bench command in that mode; counters cannot be read from inside the
    process).  The 128x128-tile family is launched in two forms with the same tile body — conv_fwd_split_kernel<2,2,M>
    and, where the tile grid leaves a partly empty last pass, conv_fwd_split_sk_kernel<M> (stream-K tail) — and the in-process
    timer brackets both under one name: the figure is the launch-weighted mean over both.  None when the summary does
    not cover the kernel / mode."""
    path = os.path.join(ROOT, "profiles", {3: "r04_pmc_hbm_traffic.json", 4: "r06_pmc_hbm_traffic.json"}.get(
        gemm_mode, "none"))
    if not os.path.exists(path):
        return None, None
    want = [kernel_name.replace(" ", "").rstrip(">")]      # "conv_fwd_split_kernel<2,2,3" also matches "...<2,2,3,0>"
    if want[0].endswith("<2,2,%d" % gemm_mode):
        want.append("conv_fwd_split_sk_kernel<%d" % gemm_mode)
    with open(path) as f:
        table = json.load(f)["kernels"]
    launches = total = 0.0
    for name, rec in table.items():
        if any(w in name.replace(" ", "") for w in want):
            launches += rec["launches"]
            total += rec["launches"] * rec["hbm_bytes_per_launch"]
    if not launches:
        return None, None
    return total / launches, os.path.relpath(path, ROOT)


def self_spawn(n):
    """one process per GPU on this node, rendezvous on 127.0.0.1 at a free port; rank 0 inherits stdout (the JSON
    line), the other ranks' stdout is dropped, every rank's stderr is inherited.  Returns the worst exit code."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and os.environ.get("DADET_BENCH_SHARE_GPU", "0") != "1":
        print("bench.py --gpus %d: only %d HIP device(s) visible (DADET_BENCH_SHARE_GPU=1 runs every rank on device 0 "
              "over gloo, a functional check only)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective: stop them (exact PIDs) instead of hanging
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            time.sleep(5.0)       # let the others fail on their own first (clean error messages)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                codes[i] = p.wait()
            break
        time.sleep(0.2)
    return max(abs(c) for c in codes)


def elided_work(c, model, targets):
    """what this build leaves out of a step of `model` on this batch because no loss reads it (same losses, same
    gradients: tests/test_default_path_gpu.py) — strings for the bench line"""
    import os as _os

    from da_detect_amd.modeling.elision import elision_enabled

    single_level = not c.MODEL.RPN.USE_FPN
    out = ["second pooler + box-head pass over the DA ROIs (the reference recomputes the first pass bit for bit)"]
    fx = getattr(getattr(model.roi_heads, "box", None), "feature_extractor", None)
    head = getattr(fx, "head", None)
    if head is not None and hasattr(head, "input_bin_stride") and head.input_bin_stride() > 1 \
            and _os.environ.get("DADET_ROI_SUBGRID", "1") == "1":
        out.append("3 of 4 ROIAlign bins in front of the stride-2 res5 head (never read)")
    if not elision_enabled():
        return out
    da = model.da_heads_triplet if getattr(model, "da_heads_triplet", False) else getattr(model, "da_heads", None)
    if da and not getattr(da, "needs_instance_features", True):
        out.append("box head (ROIAlign, res5, predictor; forward and backward) on the 256 target-domain ROIs: instance-level "
                   "loss weights are 0, every one of their gradient rows is zero")
    if single_level and model.rpn.early_backward:
        live = model._images_with_read_proposals(targets)
        n = len(targets)
        if live is not None and live < n:
            out.append("RPN head + proposal selection on %d of %d images (their proposals reach no loss)" % (n - live, n))
        out.append("RPN head backward on images without labelled anchors (target / auxiliary: identically zero)")
        if _os.environ.get("DADET_RPN_ROW_BACKWARD", "1") == "1":
            out.append("RPN head backward as dense GEMMs: run on the <= 256 sampled anchors' rows of the source image")
    if not single_level and model.rpn.early_backward and _os.environ.get("DADET_RPN_ROW_BACKWARD", "1") == "1":
        out.append("RPN head backward as dense GEMMs over the five pyramid maps: run on the sampled anchors' rows, level by level")
    return out


def run_workload(args, name, device, rank, world, steps, warmup, headline):
    """build the recipe, tune the schedule, warm up, time `steps` steps between barriers.  headline: with the dominant
    kernel bracketed in the timed region; always followed by an UNTIMED pass with every GEMM bracketed (single process
    only), from which flop_per_step and the all-GEMM rate come."""
    from da_detect_amd import _C
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.utils import streams
    from da_detect_amd.engine.trainer import WgradLaneTuner, enable_overlapped_rpn_backward, train_step

    yaml_path, overrides, images_per_gpu, workload_desc = WORKLOADS[name]
    height, width = (HEIGHT, WIDTH) if args.image_hw is None else [int(v) for v in args.image_hw.lower().split("x")]
    c, model, opt, reducer = build(yaml_path, device, seed=100, overrides=overrides)
    enable_overlapped_rpn_backward(model, not args.no_overlap)
    images, targets = make_batch(c, images_per_gpu, height, width, seed=100 + rank, device=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # schedule choice by measurement, as do_da_train does in its first iterations (untimed, before the warm-up)
    streams.join_wgrad_lane(device)
    streams.WGRAD_LANE_ROWS = int(os.environ.get("DADET_WGRAD_LANE_ROWS", "0"))
    tuner = WgradLaneTuner(device)
    while tuner.active:
        tuner.step_begin()
        train_step(model, opt, images, targets)
        tuner.step_end()
    for _ in range(warmup):
        loss_dict = train_step(model, opt, images, targets)
    if world > 1:
        reducer.record_comm(True)
    profiler = None
    if headline and not args.no_kernel_timing and rank == 0:
        # timed region: only the dominant kernel family (128x128-tile forward / data-gradient GEMM) is bracketed —
        # every event pair between two launches costs dispatch concurrency (measured: ~1 ms / step for all GEMMs)
        # (round 5: the 256 x 256-tile forward / data-gradient kernel, csrc/conv_big.hip, is the family with the most time)
        profiler = _C.KernelProfiler(pool=2 * 80 * steps, only=DOMINANT_FAMILY if args.gemm_mode == 4 else "<2,2")
    barrier()
    mem0 = torch.cuda.memory_stats(device) if os.environ.get("DADET_BENCH_MEMSTATS") else None
    t0 = time.perf_counter()
    for i in range(steps):
        # the brackets are not free (event pairs between launches cost dispatch concurrency: ~3% of the step when every
        # step is bracketed), so only every `--kernel-timing-every`-th timed step carries them
        _C.PROFILER = profiler if (profiler is not None and i % args.kernel_timing_every == 0) else None
        loss_dict = train_step(model, opt, images, targets)
    barrier()
    elapsed = time.perf_counter() - t0
    _C.PROFILER = None
    comm = reducer.comm_summary() if world > 1 else None
    reducer.record_comm(False)
    one_stream = None
    if headline and world == 1 and streams.lane_in_use() and not args.no_kernel_timing:
        # the same loop with the second GEMM stream off, so that a scaling curve whose N > 1 points kept one stream can be
        # read against a like-for-like N = 1 point
        saved = (streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS)
        streams.join_wgrad_lane(device)
        streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS = False, 0
        for _ in range(2):
            train_step(model, opt, images, targets)
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            train_step(model, opt, images, targets)
        barrier()
        one_stream = (time.perf_counter() - t1) / steps
        streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS = saved
    if mem0 is not None:       # allocator activity inside the timed region (diagnostics, stderr)
        mem1 = torch.cuda.memory_stats(device)
        print("memstats: " + ", ".join("%s %+d" % (k, mem1[k] - mem0[k]) for k in (
            "num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams",
            "reserved_bytes.all.current")) + ", reserved %.2f GB" % (mem1["reserved_bytes.all.current"] / 1e9),
            file=sys.stderr)
    exclusive = everything = None
    extra = extra_elapsed = 0
    if not args.no_kernel_timing and rank == 0 and world == 1:
        # extra UNTIMED passes (single process only: a lone rank must not enter the gradient all-reduce):
        # (1) all GEMM kernels bracketed, same schedule -> which kernel dominates, union-of-busy rate, FLOPs per step;
        # (2) headline only: the same kernels without a second GEMM stream beside them -> per-kernel durations that do
        #     not include a co-running weight-gradient / data-gradient kernel
        extra = max(2, min(steps, 5)) if headline else 2
        everything = _C.KernelProfiler()
        _C.PROFILER = everything
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(extra):
            train_step(model, opt, images, targets)
        torch.cuda.synchronize()
        extra_elapsed = time.perf_counter() - te
        if headline and streams.lane_in_use():
            saved = (streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS)
            streams.join_wgrad_lane(device)
            streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS = False, 0
            exclusive = _C.KernelProfiler()
            _C.PROFILER = exclusive
            for _ in range(extra):
                train_step(model, opt, images, targets)
            torch.cuda.synchronize()
            streams.WGRAD_OVERLAP, streams.WGRAD_LANE_ROWS = saved
        _C.PROFILER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ranks_in_sync = None
    if world > 1:
        # data-parallel invariant, checked after the timed region: every rank applied the same averaged gradients to
        # the same parameters, so the parameters are still identical bit for bit
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_in_sync = bool((lo == hi).all().item())
        if not ranks_in_sync and rank == 0:
            print("WARNING: parameters differ between ranks after %d steps (max checksum spread %.3e)" % (
                steps + warmup, float((hi - lo).abs().max())), file=sys.stderr, flush=True)
    streams.join_wgrad_lane(device)
    flop_per_step = None
    if everything is not None:
        flop_per_step = everything.union()[0] / extra
    return dict(name=name, yaml=yaml_path, overrides=list(overrides), images_per_gpu=images_per_gpu, desc=workload_desc,
                hw=(height, width), elapsed=elapsed, steps=steps, losses={k: float(v.detach()) for k, v in loss_dict.items()},
                schedule=tuner.report(), profiler=profiler, everything=everything, exclusive=exclusive, extra=extra,
                extra_elapsed=extra_elapsed, ranks_in_sync=ranks_in_sync, flop_per_step=flop_per_step,
                lane_overlap=streams.WGRAD_OVERLAP, lane_rows=streams.WGRAD_LANE_ROWS,
                elided=elided_work(c, model, targets), comm=comm, one_stream=one_stream)


def other_in_subprocess(args, name, image_hw=None):
    """`python bench.py --workload <name> --others none` with this run's settings, in a child process on the same GPU;
    -> the child's line reduced to the `other_workloads` record (None if the child failed: stderr says why)"""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", name, "--others", "none",
           "--steps", str(args.other_steps), "--warmup", str(max(3, args.warmup // 2)), "--no-cpu-baseline",
           "--gemm-mode", str(args.gemm_mode)]
    if image_hw is not None:
        cmd += ["--image-hw", image_hw]
    if args.no_kernel_timing:
        cmd.append("--no-kernel-timing")
    if args.no_overlap:
        cmd.append("--no-overlap")
    # a plain single-process run, whatever launched this one (under torch.distributed.run with one rank the rendezvous
    # variables would send the child to the parent's store)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                        "MASTER_PORT") and not k.startswith("TORCHELASTIC_")}
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=900, check=True, env=env).stdout.decode()
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    except Exception as exc:      # noqa: BLE001 — the headline line must still come out
        print("bench.py: the %s run failed: %r" % (name, exc), file=sys.stderr, flush=True)
        return None
    cfg_ = line.get("config", {})
    rec = {"workload": cfg_.get("workload"), "yaml": WORKLOADS[name][0], "images_per_step": WORKLOADS[name][2],
           "image_hw": cfg_.get("image_hw"),
           "steps": line["steps"], "ms_per_step": line["ms_per_step"], "images_per_s": line["value"],
           "schedule": cfg_.get("schedule"), "elided": line.get("elided"), "process": "its own (python bench.py "
           "--workload %s --others none --steps %d --warmup %d%s)" % (
               name, line["steps"], line["warmup"], "" if image_hw is None else " --image-hw " + image_hw)}
    if line.get("flop_per_step") is not None:
        rec["flop_per_step"] = line["flop_per_step"]
        rec["flop_unit"] = "TFLOP (algorithmic fp32, 2*MAC, all GEMM launches of one step per GPU)"
        rec["step_frac"] = line.get("step_frac")
        rec["all_gemm_frac"] = (line.get("roofline") or {}).get("gemm_streams", {}).get("all_gemm_frac")
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)   # steady state after >= 10 warm-up iterations
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing-every", type=int, default=4,
                    help="bracket the dominant kernel's launches with HIP events in every N-th timed step")
    ap.add_argument("--workload", default="img_only", choices=sorted(WORKLOADS),
                    help="img_only (default, BASELINE configs[1]) | da (configs[2]) | triplet (configs[3]) | "
                         "triplet_aligned | fpn_dcn_da (configs[4]); non-default workloads are extra measurements")
    ap.add_argument("--others", default=None,
                    help="comma-separated workloads timed after the headline loop and reported under other_workloads "
                         "(default: da,triplet,fpn_dcn_da next to the default headline, none otherwise; 'none' switches "
                         "it off)")
    ap.add_argument("--resolutions", default=None,
                    help="HxW at which img_only and da are timed again and reported under `resolutions` (default: 608x1216, "
                         "the reference yaml's own training size, next to the default headline; 'none' switches it off)")
    ap.add_argument("--other-steps", type=int, default=10)
    ap.add_argument("--image-hw", default=None, help="HxW of the synthetic images (default 1024x2048)")
    ap.add_argument("--no-overlap", action="store_true", help="keep the RPN backward inside the main backward pass")
    ap.add_argument("--gemm-mode", type=int, default=int(os.environ.get("DADET_GEMM_MODE", "4")),
                    help="3: fp32 operands as 3 bf16 terms, 6 bf16 MFMAs per K=16 (fp32-class accuracy, default); "
                         "0: exact fp32 MFMA; 2: 2-term split (~2^-16 products)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (what torch.distributed.run would do)
        raise SystemExit(self_spawn(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # DADET_BENCH_SHARE_GPU=1 (test rigs with fewer GPUs than ranks): every rank on device 0, gloo instead of RCCL
    share = os.environ.get("DADET_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the process group then keeps start / end events of every collective on RCCL's stream (Work._get_duration): the
        # `comm.allreduce_ms` of the line
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")
        dist.init_process_group(backend="gloo" if share else "nccl", init_method="env://")  # "nccl" is RCCL on ROCm
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from da_detect_amd import _C

    _C.set_gemm_mode(args.gemm_mode)
    desc, mfma_per_product = GEMM_MODES[args.gemm_mode]
    # peak for ALGORITHMIC flops: the fp32 pipe's peak in mode 0; in the split modes every fp32 product
    # costs `mfma_per_product` bf16 / fp16 MFMAs (both 2.5 PFLOP/s dense), so the algorithmic ceiling is that peak divided
    # by it: 833.3 TFLOP/s in mode 4 (three fp16 MFMAs), 416.7 in mode 3 (six bf16 MFMAs)
    peak = FP32_MFMA_PEAK_TFLOPS if args.gemm_mode == 0 else BF16_MFMA_PEAK_TFLOPS / mfma_per_product

    r = run_workload(args, args.workload, device, rank, world, args.steps, args.warmup, headline=True)
    if args.others is None:
        # N > 1 (the driver's scaling runs): the headline recipe only — the other recipes would be built as further models
        # of every rank's process, and a failure of one of them on one rank would leave the others waiting in a collective
        # with the headline line unprinted; `--others da,triplet,fpn_dcn_da` asks for them explicitly
        others = ["da", "triplet", "fpn_dcn_da"] if (args.workload == "img_only" and args.image_hw is None
                                                     and world == 1) else []
    else:
        others = [w for w in args.others.split(",") if w and w != "none"]
    other_results = {}
    isolate = world == 1 and os.environ.get("DADET_BENCH_OTHERS_IN_PROCESS") != "1"
    for name in others:
        # the other BASELINE recipes on the same GPU(s), same sizes, a few steps each: the headline recipe is the one that
        # loses the most work to `elided`, `da` / `triplet` lose the least; `fpn_dcn_da` is the one furthest from the roofline
        if isolate:
            # Single-process runs time each of them in a process of its own, as a training run of that recipe is.  Built
            # as the SECOND model of this process, `da` was measured 7 - 9% slower than alone (29.9 vs 27.3 ms per step on
            # one box, `triplet` 35.8 vs 32.9; the other order — `da` first, `img_only` second, `da` third — shows nothing,
            # and neither the allocator, the transposed-weight cache, the garbage collector, the stream creation order nor
            # GPU_MAX_HW_QUEUES changes it; under rocprofv3 the kernels take the same time: DESIGN.md section 7).
            # DADET_BENCH_OTHERS_IN_PROCESS=1 keeps them in this process.
            rec = other_in_subprocess(args, name)
            if rec is not None:
                other_results[name] = rec
            continue
        torch.cuda.empty_cache()
        o = run_workload(args, name, device, rank, world, args.other_steps, max(3, args.warmup // 2), headline=False)
        ms = o["elapsed"] / o["steps"] * 1e3
        rec = {"workload": o["desc"], "yaml": o["yaml"], "images_per_step": world * o["images_per_gpu"],
               "steps": o["steps"], "ms_per_step": round(ms, 3),
               "images_per_s": round(world * o["images_per_gpu"] * o["steps"] / o["elapsed"], 3),
               "schedule": o["schedule"], "elided": o["elided"], "process": "the headline's"}
        if o["flop_per_step"] is not None:
            rec["flop_per_step"] = round(o["flop_per_step"] / 1e12, 4)
            rec["flop_unit"] = "TFLOP (algorithmic fp32, 2*MAC, all GEMM launches of one step per GPU)"
            rec["step_frac"] = round(o["flop_per_step"] / 1e12 / (ms * 1e-3) / peak, 4)
            work, busy_ms = o["everything"].union()
            rec["all_gemm_frac"] = round(work / (busy_ms * 1e-3) / 1e12 / peak, 4)
        if o["ranks_in_sync"] is not None:
            rec["ranks_in_sync_after_run"] = o["ranks_in_sync"]
        other_results[name] = rec

    resolutions = {}
    default_run = args.workload == "img_only" and args.image_hw is None
    # (an explicit --others, e.g. the profiling scripts' `--others none`, also switches the default resolutions pass off)
    res_hw = ("608x1216" if (default_run and args.others is None) else "none") if args.resolutions is None else args.resolutions
    if res_hw != "none" and world == 1 and isolate:
        # SURVEY.md 8(d) "report both": the shipped yaml resizes Cityscapes to 600 x 1200 (608 x 1216 after padding to /32)
        for name in ("img_only", "da"):
            rec = other_in_subprocess(args, name, image_hw=res_hw)
            if rec is not None:
                resolutions["%s@%s" % (name, res_hw)] = rec

    if rank == 0:
        elapsed, steps = r["elapsed"], r["steps"]
        images_per_gpu = r["images_per_gpu"]
        height, width = r["hw"]
        profiler, everything, exclusive = r["profiler"], r["everything"], r["exclusive"]
        value = world * images_per_gpu * steps / elapsed
        ms_per_step = elapsed / steps * 1e3
        roofline = None
        kernels = {}
        if profiler is not None:
            kernels = profiler.summary()
            bracketed_steps = len(range(0, steps, args.kernel_timing_every))
            name = max(kernels, key=lambda k: kernels[k]["total_ms"])
            k = kernels[name]
            achieved = k["achieved"] / 1e12
            traffic, traffic_src = pmc_traffic(name, args.gemm_mode)
            shown = name + (" (+ its stream-K launch form conv_fwd_split_sk_kernel<%d>, same tile body)" % args.gemm_mode
                            if name.endswith("<2,2,%d>" % args.gemm_mode) else "")
            roofline = {"bound": "mfma", "kernel": shown, "achieved": round(achieved, 2),
                        "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch",
                        "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": round(k["bytes_per_launch"], 0),
                        "contraction": desc,
                        "executed_mfma_tflops": round(achieved * mfma_per_product, 1),
                        "vs_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 3),
                        "launches_per_step": k["launches"] / bracketed_steps,
                        "bracketed_steps": bracketed_steps,
                        "avg_launch_ms": round(k["avg_ms"], 4),
                        "gflop_per_launch": round(k["work_per_launch"] / 1e9, 3),
                        "share_of_step": round(k["total_ms"] / bracketed_steps / ms_per_step, 4)}
            if args.gemm_mode == 4:
                roofline["power_limited_mfma_stream"] = {
                    "executed_tflops": F16_STREAM_ON_SPLIT_OPERANDS_TFLOPS,
                    "algorithmic_tflops": round(F16_STREAM_ON_SPLIT_OPERANDS_TFLOPS / mfma_per_product, 1),
                    "frac_of_it": round(achieved * mfma_per_product / F16_STREAM_ON_SPLIT_OPERANDS_TFLOPS, 4),
                    "note": "a register-only v_mfma_f32_32x32x16_f16 stream on operands with the bit statistics of split "
                            "fp32 data sustains this on the part (2464 on zeros): tools/native/gemm_lab.hip part 1, "
                            "profiles/r05_gemm_lab.txt.  Context for `frac`, which stays against the nominal peak"}
                # continuity with rounds 1 - 3, whose contraction (mode 3) needed six bf16 MFMAs per fp32 product: the same
                # ALGORITHMIC rate against that mode's 416.7 TFLOP/s ceiling.  `frac` above is against the ceiling of the
                # contraction that actually runs (three fp16 MFMAs per product: 833.3).
                roofline["against_the_six_mfma_ceiling_of_mode_3"] = {
                    "peak": round(BF16_MFMA_PEAK_TFLOPS / 6, 1), "frac": round(achieved / (BF16_MFMA_PEAK_TFLOPS / 6), 4)}
            if everything is not None:
                work, busy_ms = everything.union()
                kernels = everything.summary()      # table of all GEMM kernels (extra pass, `extra` steps)
                dominant = max(kernels, key=lambda k: kernels[k]["total_ms"])
                roofline["gemm_streams"] = {
                    "note": ("forward/data-gradient and weight-gradient GEMMs run on two streams "
                             "(DADET_WGRAD_STREAM=1); `achieved` is per launch WHILE the other stream's kernel shares "
                             "the GPU; " if r["lane_overlap"] else
                             "weight-gradient GEMMs of layers with at most %d rows run on a second stream beside the "
                             "data-gradient chain (engine.trainer.WgradLaneTuner measured that faster for this recipe): "
                             "`achieved` is per launch WHILE such a kernel may share the GPU; exclusive_* = the same "
                             "launches with the second stream off; " % r["lane_rows"]
                             if r["lane_rows"] > 0 else
                             "one GEMM stream (default): a bracketed launch has the GPU to itself except for the "
                             "latency-bound side-stream kernels; ") +
                            "this block comes from extra untimed passes with every GEMM bracketed",
                    "dominant_kernel_all_bracketed": dominant,
                    "all_gemm_tflops_while_any_runs": round(work / (busy_ms * 1e-3) / 1e12, 2),
                    "all_gemm_frac": round(work / (busy_ms * 1e-3) / 1e12 / peak, 4),
                    "gemm_busy_share_of_step": round(busy_ms / (r["extra_elapsed"] * 1e3), 4)}
                # every GEMM family of the step (same extra pass): rate against the contraction's ceiling; PMC bytes per
                # launch over algorithmic bytes where the committed counter passes cover the family
                fams = {}
                # (from the pass with the second GEMM stream off when the tuner kept it: a family's rate, like `frac`, must
                # not include a co-running kernel)
                fam_src = exclusive.summary() if exclusive is not None else kernels
                for fam, frags in FAMILIES:
                    ks = [v for n_, v in fam_src.items() if any(f in n_ for f in frags)]
                    if not ks:
                        continue
                    ms_ = sum(v["total_ms"] for v in ks)
                    work_ = sum(v["work_per_launch"] * v["launches"] for v in ks)
                    launches_ = sum(v["launches"] for v in ks)
                    alg_ = sum(v["bytes_per_launch"] * v["launches"] for v in ks) / launches_
                    rec = {"ms_per_step": round(ms_ / r["extra"], 3), "launches_per_step": round(launches_ / r["extra"], 1),
                           "tflops": round(work_ / (ms_ * 1e-3) / 1e12, 1),
                           "frac": round(work_ / (ms_ * 1e-3) / 1e12 / peak, 4),
                           "algorithmic_bytes_per_launch": round(alg_, 0)}
                    tr_, _src = pmc_traffic(frags[0], args.gemm_mode)
                    if tr_ is not None:
                        rec["traffic_bytes_per_launch"] = round(tr_, 0)
                        rec["traffic_over_algorithmic"] = round(tr_ / alg_, 2)
                    fams[fam] = rec
                roofline["families"] = fams
            if exclusive is not None:
                ek = exclusive.summary().get(name)
                if ek:
                    # The headline `frac` is a KERNEL figure: it must not move with a per-box schedule decision.  When the
                    # tuner kept the second GEMM stream, the launches bracketed in the timed region shared the GPU with a
                    # weight-gradient GEMM; `achieved` / `frac` / `avg_launch_ms` then come from the extra pass that ran
                    # the same launches with the second stream off (HIP events, same process, right after the timed
                    # loop), and the under-contention numbers of the timed region move to gemm_streams.shared_*.
                    gs = roofline["gemm_streams"]
                    gs["shared_tflops"], gs["shared_frac"] = roofline["achieved"], roofline["frac"]
                    gs["shared_avg_launch_ms"] = roofline["avg_launch_ms"]
                    gs["exclusive_tflops"] = round(ek["achieved"] / 1e12, 2)
                    gs["exclusive_frac"] = round(ek["achieved"] / 1e12 / peak, 4)
                    ex = ek["achieved"] / 1e12
                    if "against_the_six_mfma_ceiling_of_mode_3" in roofline:
                        roofline["against_the_six_mfma_ceiling_of_mode_3"]["frac"] = round(ex / (BF16_MFMA_PEAK_TFLOPS / 6), 4)
                    roofline.update({"achieved": round(ex, 2), "frac": round(ex / peak, 4),
                                     "avg_launch_ms": round(ek["avg_ms"], 4),
                                     "executed_mfma_tflops": round(ex * mfma_per_product, 1),
                                     "vs_fp32_mfma_peak": round(ex / FP32_MFMA_PEAK_TFLOPS, 3),
                                     "measured_in": "extra untimed pass of %d steps with the second GEMM stream off (the "
                                                    "timed region's launches share the GPU: gemm_streams.shared_*)" % r["extra"]})
            elif roofline is not None:
                roofline["measured_in"] = "timed region (one GEMM stream: a bracketed launch has the GPU to itself)"
                if "gemm_streams" in roofline:
                    roofline["gemm_streams"]["exclusive_tflops"] = roofline["achieved"]
                    roofline["gemm_streams"]["exclusive_frac"] = roofline["frac"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline(YAML, seed=100) if args.workload == "img_only" and args.image_hw is None else None
                if cpu is not None:
                    cpu["sample"] = str(cpu.get("sample", "")) + (
                        "; ALL rows evaluated as the reference does (every head on every image, both box-head passes): "
                        "the GPU step leaves out the work listed under `elided`, so the ratio of the two values is not a "
                        "kernel-quality figure")
            except Exception as e:  # the baseline is reported, never required for the GPU number
                cpu = {"value": None, "unit": "images/s", "cores": min(os.cpu_count() or 1, 64), "kind": "port",
                       "sample": "failed: %r" % (e,)}
        rois = ("box head on the 256 source-domain ROIs (the target image's ROIs reach no loss), RPN head on the source "
                "image, RPN backward on <= 256 sampled anchor rows" if any("target-domain ROIs" in e for e in r["elided"])
                else "256 ROIs sampled per image, one box-head pass over all of them")
        line = {
            "metric": "train images/sec, DA-Faster-RCNN R-50 Cityscapes->Foggy",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded rand*255 - PIXEL_MEAN images, 8-20 seeded boxes/image, seeded random-init weights)",
            "config": {"workload": "configs/da_faster_rcnn %s %dx%d image per GPU per step, %s, fwd+bwd+SGD; work no loss "
                                   "reads is not evaluated (see `elided`, `flop_per_step`)"
                                   % (r["desc"], height, width, rois),
                       "yaml": r["yaml"], "overrides": r["overrides"], "global_batch": world * images_per_gpu,
                       "image_hw": [height, width],
                       "parallelism": "dp%d" % world, "schedule": r["schedule"]},
            "elided": r["elided"],
            "roofline": roofline, "cpu_baseline": cpu,
            "other_workloads": other_results,
            "resolutions": resolutions,
            "kernel_timing": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"], 3),
                                  "tflops": round(v["achieved"] / 1e12, 2)} for k, v in kernels.items()},
            "final_losses": {k: round(v, 5) for k, v in r["losses"].items()},
        }
        if r["flop_per_step"] is not None:
            # images/s counts images, not work: the FLOPs one step executes on one GPU (all GEMM launches, algorithmic
            # 2*MAC in fp32 terms) and the fraction of the contraction's ceiling the WHOLE step reaches with them
            line["flop_per_step"] = round(r["flop_per_step"] / 1e12, 4)
            line["flop_unit"] = "TFLOP per step per GPU (algorithmic fp32, 2*MAC, all GEMM launches; measured in an extra pass)"
            line["step_frac"] = round(r["flop_per_step"] / 1e12 / (ms_per_step * 1e-3) / peak, 4)
            line["reference_flop_per_step"] = {"value": 6.9, "unit": "TFLOP", "note": "the reference's dense evaluation "
                                               "of the same batch incl. its redundant second box-head pass (SURVEY.md 8d)"}
        if r["ranks_in_sync"] is not None:
            line["config"]["ranks_in_sync_after_run"] = r["ranks_in_sync"]
        if r["one_stream"] is not None:
            # the like-for-like N = 1 point for a run whose ranks (or whose rig: gloo) kept one GEMM stream — since round 5
            # the tuner's decision is agreed over the ranks (max of the candidates' times), over RCCL
            line["one_stream"] = {"ms_per_step": round(r["one_stream"] * 1e3, 3),
                                  "value": round(world * images_per_gpu / r["one_stream"], 3), "unit": "images/s",
                                  "note": "the same timed loop with the second GEMM stream off — the schedule an N > 1 "
                                          "run uses when its ranks' agreed tuner (or DADET_TUNE_SCHEDULE_RANKS=0) keeps "
                                          "one stream: config.schedule of that run says which; compare like with like"}
        elif world == 1:
            line["one_stream"] = {"ms_per_step": round(ms_per_step, 3), "value": round(value, 3), "unit": "images/s",
                                  "note": "the headline run already used one GEMM stream"}
        if r["comm"] is not None:
            line["comm"] = r["comm"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
