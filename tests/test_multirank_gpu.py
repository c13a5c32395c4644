"""The data-parallel path of the REAL model with world_size 2 (reference
tools/train_net_triplet.py:83-88, 304-309): two ranks share cuda:0 and exchange gradients over gloo (a functional rig
— one GPU is what the test box has; with one rank per GPU the only difference is the backend string "nccl" = RCCL).

Each rank runs the default training schedule (overlapped RPN backward, early image-level DA backward where the recipe
allows it, weight gradients accumulated straight into the reducer's flat buckets, fused SGD) on its own seeded batch.
Checked: (1) what every bucket holds after finalize() is the MEAN over ranks of what the ranks held locally when the
collective was issued; (2) every parameter received its gradient before its bucket was reduced (the local snapshot
equals the same step run single-process on that rank's batch); (3) after 3 steps the ranks' parameters are bit-identical
and differ from the initial ones."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case, seed, device):
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.solver import make_optimizer
    from golden.cases import case_cfg, fpn_dcn_da_cfg
    from golden.fill import fill_state_dict

    c = fpn_dcn_da_cfg() if case == "fpn_dcn_da" else case_cfg(case)
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed))
    model = model.to(device).train()
    return c, model, make_optimizer(c, model)


def _worker(rank, world, port, case, out):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rccl_alone = world == 1 and os.environ.get("DADET_TEST_RCCL_ONE_RANK") == "1"
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif rccl_alone:
        dist.init_process_group("nccl", rank=0, world_size=1)       # "nccl" is RCCL on ROCm
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step
    from da_detect_amd.parallel.reducer import BucketedGradReducer

    c, model, opt = _build(case, 3, dev)
    reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad], bucket_bytes=8 << 20,
                                  always_communicate=rccl_alone)
    reducer.broadcast_parameters(0)
    opt.attach_reducer(reducer)
    enable_overlapped_rpn_backward(model)
    nimg = 3 if c.MODEL.DA_HEADS.TRIPLET_USE else 2
    batch_rank = rank if world > 1 else int(os.environ.get("DADET_TEST_BATCH_RANK", "0"))
    images, targets = make_batch(c, nimg, 192, 320, seed=100 + batch_rank, device=dev)
    local = {}
    orig = reducer._all_reduce

    def spy(flat):
        local[flat.data_ptr()] = flat.detach().clone()      # what THIS rank holds when the collective is issued
        return orig(flat)

    reducer._all_reduce = spy
    p0 = [p.detach().clone() for p in reducer.params]
    torch.manual_seed(50 + batch_rank)                      # device sampler seeds: per-rank stream
    train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    # the fused optimizer leaves the SUMS in the buckets and folds 1 / world into the SGD kernel (reducer.mean_scale)
    after = [b["flat"].detach().clone().cpu() * reducer.mean_scale for b in reducer.buckets]
    if world > 1 or rccl_alone:
        snaps = [local[b["flat"].data_ptr()].cpu() for b in reducer.buckets]
    else:
        snaps = after                                        # single process: nothing is reduced
    reducer._all_reduce = orig
    early = []
    orig_finalize = reducer.finalize

    def finalize(**kw):
        # collectives already issued when backward is over: from the second step on the reducer no longer waits for the
        # parameters no rank used in the first (the instance head of the image-level-only recipe sits in bucket 0)
        early.append(sum(1 for b in reducer.buckets if b["work"] is not None))
        return orig_finalize(**kw)

    reducer.finalize = finalize
    for it in range(1, 3):
        torch.manual_seed(50 + batch_rank + 10 * it)
        train_step(model, opt, images, targets)
    torch.cuda.synchronize()
    if world > 1:
        assert reducer.static_unused is not None
        assert all(e == len(reducer.buckets) for e in early), (early, len(reducer.buckets), len(reducer.static_unused))
    if rccl_alone:      # the one-rank group learns the unused parameters like N ranks do: every collective overlaps backward
        assert reducer.static_unused is not None
        assert all(e == len(reducer.buckets) for e in early), (early, len(reducer.buckets))
    if world > 1 or rccl_alone:
        # (round 6) from the second step on the update is issued bucket by bucket, each behind its own collective
        assert opt._bucket_plan is not None, "the per-bucket update was not taken"
        assert sum(n for _, n, _ in opt._bucket_plan[3]) == len(reducer.update_ids())
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(p0, reducer.params))
    params = torch.cat([p.detach().reshape(-1).cpu() for p in reducer.params])
    out.put((rank, [s.numpy() for s in snaps], [a.numpy() for a in after], params.numpy(), moved,
             len(reducer.buckets), len(reducer.touched)))
    if world > 1 or rccl_alone:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, case, batch_rank=0, rccl_one_rank=False):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    os.environ["DADET_TEST_BATCH_RANK"] = str(batch_rank)
    os.environ["DADET_TEST_RCCL_ONE_RANK"] = "1" if rccl_one_rank else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("case", ["da_img_only", "da_plain", "fpn_dcn_da"])
def test_two_ranks_real_model_bucket_means_and_sync(device, case):
    import numpy as np

    r0, r1 = _run(2, case)
    n_buckets = r0[5]
    assert n_buckets >= 4 and r0[6] == r1[6] > 50
    differing = 0
    for b in range(n_buckets):
        mean = (r0[1][b].astype(np.float64) + r1[1][b].astype(np.float64)) / 2
        for r in (r0, r1):      # (1) bucket contents after finalize() == mean of the per-rank local gradients
            np.testing.assert_allclose(r[2][b], mean, rtol=1e-6, atol=1e-9 + 1e-6 * float(np.abs(mean).max()))
        differing += int(float(np.abs(r0[1][b] - r1[1][b]).max()) > 0)
    # the ranks saw different batches (a bucket of parameters outside the recipe's graph — the instance head when its
    # loss weight is 0 — is zero on both)
    assert differing >= n_buckets // 2, (differing, n_buckets)
    # (3) identical parameters on both ranks after 3 steps, and they moved
    assert np.array_equal(r0[3], r1[3]), "parameters differ between ranks"
    assert r0[4] == r1[4] and r0[4] > 50
    if case == "fpn_dcn_da":
        return      # (2) is schedule logic shared with the two recipes above; a third R-101 process costs 12 s of the suite
    # (2) nothing was reduced too early: rank 1's local snapshot == the same step run alone on rank 1's batch
    solo, = _run(1, case, batch_rank=1)
    floor = 1e-6
    for b in range(n_buckets):
        scale = float(np.abs(solo[2][b]).max())
        np.testing.assert_allclose(r1[1][b], solo[2][b], rtol=1e-4, atol=floor * scale + 1e-12)


def test_one_rank_over_rccl_takes_the_n_rank_path_and_changes_nothing(device):
    """RCCL first contact on a one-GPU box: a ONE-rank "nccl" process group with the reducer forced to communicate —
    communicator set-up, every bucket through ncclAllReduce on RCCL's stream with an async work handle, the optimizer
    waiting for the handles — must leave exactly the single-process result: same bucket contents, same parameters after
    3 steps (an all-reduce over one rank is the identity, the mean factor is 1)."""
    import numpy as np

    alone, = _run(1, "da_plain")
    rccl, = _run(1, "da_plain", rccl_one_rank=True)
    assert rccl[5] == alone[5] and rccl[6] == alone[6]
    # (two processes: the fp32 atomics of the image-level DA sums and of the RPN row scatter may differ in the last bit)
    for b in range(rccl[5]):
        scale = float(np.abs(alone[2][b]).max())
        tol = dict(rtol=1e-5, atol=1e-6 * scale + 1e-12)
        np.testing.assert_allclose(rccl[1][b], alone[2][b], err_msg="what went into bucket %d's collective" % b, **tol)
        assert np.array_equal(rccl[2][b], rccl[1][b]), "bucket %d changed in a one-rank all-reduce" % b
    # three steps later the two PROCESSES have drifted apart by what last-bit differences of the atomically summed terms
    # do to a discontinuous pipeline (a ReLU or an NMS decision flipping: measured 5e-5 on 2% of the parameters, values
    # ~0.02, lr 1e-3) — a sanity bound, not a bit comparison; the bucket comparison above is the exact one
    np.testing.assert_allclose(rccl[3], alone[3], rtol=1e-2, atol=3e-4, err_msg="parameters after 3 steps")


def _worker_production(rank, world, port, out):
    """one rank over RCCL at the production geometry: default 25 MB buckets, 512 x 1024 images, the weight-gradient lane ON
    (bucket collectives are then issued FROM the lane stream, parallel/reducer.py `_all_reduce`), comm recording on"""
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCH_NCCL_ENABLE_TIMING="1")
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward, train_step
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.utils import streams

    results = {}
    for lane_rows in (0, 17000):
        c, model, opt = _build("da_plain", 3, dev)
        reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad], always_communicate=True)
        opt.attach_reducer(reducer)
        enable_overlapped_rpn_backward(model)
        images, targets = make_batch(c, 2, 512, 1024, seed=100, device=dev)
        streams.join_wgrad_lane(dev)
        streams.WGRAD_LANE_ROWS = lane_rows
        reducer.record_comm(True)
        try:
            for it in range(4):
                torch.manual_seed(60 + it)
                train_step(model, opt, images, targets)
            torch.cuda.synchronize()
            comm = reducer.comm_summary(skip=1)
        finally:
            streams.join_wgrad_lane(dev)
            streams.WGRAD_LANE_ROWS = 0
        params = torch.cat([p.detach().reshape(-1).cpu() for p in reducer.params])
        results[lane_rows] = (comm, params.numpy(), [b["flat"].numel() * 4 for b in reducer.buckets])
    out.put(results)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_over_rccl_at_production_geometry_with_the_lane(device):
    """the collective-from-the-lane-stream path (parallel/reducer.py:_all_reduce) with RCCL at 25 MB buckets and 512 x 1024
    images, and the communication evidence bench.py prints for N > 1: every bucket's collective goes out during backward
    from the second step on, RCCL reports a duration for each, the exposed wait in finalize() is measured, and the lane
    changes no result (same parameters after four steps as with one GEMM stream, to the usual cross-run atomics noise)"""
    import numpy as np
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = ctx.Process(target=_worker_production, args=(0, 1, _free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(120)
    assert p.exitcode == 0
    for lane_rows, (comm, params, bucket_bytes) in res.items():
        assert comm["backend"] == "nccl" and comm["world"] == 1 and comm["steps"] == 3
        # 25 MB buckets; a tensor larger than that (the RPN's 3x3 conv: 36 MB) is a bucket of its own
        assert comm["buckets"] == len(bucket_bytes) >= 5 and 20.0 <= comm["bucket_mb"] <= 40.0, comm
        assert comm["buckets_issued_during_backward"] == comm["buckets"] and comm["buckets_issued_in_finalize"] == 0, comm
        assert comm["allreduce_ms"] is not None and comm["allreduce_ms"] > 0.0, comm
        assert comm["exposed_ms"] >= 0.0 and comm["overlap_frac"] is not None, comm
        print("one rank over RCCL, lane rows %d: %s" % (lane_rows, {k: v for k, v in comm.items() if k != "note"}))
    np.testing.assert_allclose(res[17000][1], res[0][1], rtol=1e-2, atol=3e-4)


def test_bench_py_with_two_ranks_prints_one_line_with_its_comm_block(device):
    """`python bench.py --gpus 2` as the driver's scaling runs start it — on this rig both ranks
    share device 0 over gloo (DADET_BENCH_SHARE_GPU=1): self-spawn, rendezvous, the timed loop with max-over-ranks timing,
    the comm evidence and the in-sync check; a trivial failure here would cost the first 8-GPU run.  Reference of the launch:
    tools/train_net_triplet.py:83-88, 304-309 (torch.distributed.launch, one process per GPU)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DADET_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--others", "none", "--no-cpu-baseline", "--image-hw", "512x1024"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's): %d" % len(lines)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["ranks_in_sync_after_run"] is True
    comm = d["comm"]
    assert comm["world"] == 2 and comm["buckets"] >= 2
    # every bucket's all-reduce goes out while backward still runs (nothing is left for finalize())
    assert comm["buckets_issued_during_backward"] == comm["buckets"], comm
