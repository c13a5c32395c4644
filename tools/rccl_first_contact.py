#!/usr/bin/env python
"""Multi-rank hardening on a one-GPU box.

  python tools/rccl_first_contact.py --backend nccl --world 1      # RCCL itself, one rank, the N-rank code path
  python tools/rccl_first_contact.py --backend gloo --world 2      # two ranks time-sharing cuda:0 (the round-2 rig)

Each rank runs the real model's training step (default schedule) in phases — lane off, lane on (weight-gradient GEMMs of
narrow layers on a second stream; the bucket collectives are then issued FROM that stream), lane off again — and prints
per-step wall times, allocator activity and the number of collectives that went out during backward.  With --backend
nccl --world 1 the reducer is built with always_communicate, so every bucket goes through ncclAllReduce on RCCL's own
stream with an async work handle, exactly as with N ranks.  The phase table answers the open item of round 2: "six steps
of lane + collectives left the process 10x slower for the rest of the run" (seen over gloo on one GPU only).
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def worker(rank, world, port, args, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    sys.argv = [sys.argv[0]]
    import bench
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.utils import streams

    yaml_path, overrides, images_per_gpu, _ = bench.WORKLOADS[args.workload]
    from da_detect_amd.config import cfg
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.solver import make_optimizer

    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, yaml_path))
    torch.manual_seed(100)
    model = build_detection_model(c)
    bench.benchmark_init(model, 100)
    model = model.to(dev).train()
    opt = make_optimizer(c, model)
    reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad], always_communicate=True)
    reducer.broadcast_parameters(0)
    opt.attach_reducer(reducer)
    enable_overlapped_rpn_backward(model)
    h, w = [int(v) for v in args.image_hw.split("x")]
    images, targets = make_batch(c, images_per_gpu, h, w, seed=100 + rank, device=dev)

    early = []
    orig_finalize = reducer.finalize

    def finalize(**kw):
        early.append(sum(1 for b in reducer.buckets if b["work"] is not None))
        return orig_finalize(**kw)

    reducer.finalize = finalize
    phases = []
    keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
    for label, rows in (("lane off", 0), ("lane on (<= 17000 rows)", 17000), ("lane off again", 0),
                        ("lane on again", 17000), ("lane off, end", 0)):
        streams.join_wgrad_lane(dev)
        streams.WGRAD_LANE_ROWS = rows
        times = []
        m0 = torch.cuda.memory_stats(dev)
        del early[:]
        for _ in range(args.steps):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            train_step(model, opt, images, targets)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        m1 = torch.cuda.memory_stats(dev)
        phases.append(dict(phase=label, ms_per_step=[round(t, 2) for t in times],
                           collectives_issued_during_backward=list(early), buckets=len(reducer.buckets),
                           allocator={k: m1[k] - m0[k] for k in keys},
                           reserved_gb=round(m1["reserved_bytes.all.current"] / 1e9, 2)))
    chk = torch.stack([p.detach().double().sum() for p in model.parameters()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    finite = bool(torch.isfinite(chk).all().item())
    out.put((rank, phases, bool((lo == hi).all().item()), finite))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--workload", default="img_only")
    ap.add_argument("--image-hw", default="1024x2048")
    args = ap.parse_args()
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=worker, args=(r, args.world, port, args, q)) for r in range(args.world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=1200) for _ in range(args.world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    print(json.dumps({"backend": args.backend, "world": args.world, "workload": args.workload,
                      "image_hw": args.image_hw, "ranks_in_sync": all(g[2] for g in got),
                      "parameters_finite": all(g[3] for g in got),
                      "ranks": {str(g[0]): g[1] for g in got}}, indent=1))
    raise SystemExit(max(abs(p.exitcode or 0) for p in procs))


if __name__ == "__main__":
    main()
