"""The N > 1 gradient path on CPU: two processes over gloo must end a step with identical, averaged gradients
in the flat buckets, including for a parameter that received no gradient on one rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from da_detect_amd.parallel.reducer import BucketedGradReducer

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    unused = torch.nn.Parameter(torch.ones(3))
    never = torch.nn.Parameter(torch.ones(5))          # touched by no rank, and it sits in the FIRST bucket
    params = list(model.parameters()) + [unused, never]
    red = BucketedGradReducer(params, bucket_bytes=256)  # several tiny buckets
    assert len(red.buckets) > 2 and red._bucket_of[id(never)] is red.buckets[0]
    red.broadcast_parameters(0)
    red.record_comm(True)
    early = []
    for step in range(3):
        red.zero_grad()
        x = torch.full((5, 8), float(rank + 1 + min(step, 1)))
        loss = model(x).pow(2).sum()
        if rank == 0:
            loss = loss + (unused * 2).sum()  # rank 1 never touches `unused`
        loss.backward()
        # collectives already issued while backward ran (before finalize() forces the rest)
        early.append(sum(1 for b in red.buckets if b["work"] is not None))
        red.finalize()
    # step 0: the first bucket waits for `never`, so nothing goes out before finalize().  Afterwards the ranks have
    # agreed that `never` is unused: on rank 0 every bucket completes during backward; rank 1 still waits for `unused`
    # (rank 0 touches it), whose bucket and its successors go out in finalize().
    assert red.static_unused == frozenset([id(never)]), red.static_unused
    assert early[0] == 0 and early[1] == early[2]
    # communication evidence: every bucket's collective is counted once per step, in backward or in finalize()
    comm = red.comm_summary(skip=1)
    assert comm["backend"] == "gloo" and comm["world"] == 2 and comm["buckets"] == len(red.buckets) and comm["steps"] == 2
    assert comm["buckets_issued_during_backward"] == early[1]
    assert comm["buckets_issued_during_backward"] + comm["buckets_issued_in_finalize"] == len(red.buckets)
    assert comm["exposed_ms"] == 0.0 and comm["allreduce_ms"] is None      # CPU tensors: no device events, no durations
    red.record_comm(False)
    assert early[1] == (len(red.buckets) if rank == 0 else red.buckets.index(red._bucket_of[id(unused)])), early
    # which parameters the optimizer updates: everything some rank touched — `unused` also on rank 1, which never
    # touched it (its averaged gradient is the same on both ranks, so must be its update); `never` on neither
    assert red.update_ids() == frozenset(id(p) for p in params[:-1]), "update set differs from 'touched by any rank'"
    assert (id(unused) in red.touched) == (rank == 0)
    # finalize(mean=False) leaves the sums and reports the factor the optimizer folds into its gradient read
    means = [b["flat"].clone() for b in red.buckets]
    red.zero_grad()
    loss = model(torch.full((5, 8), float(rank + 2))).pow(2).sum()
    if rank == 0:
        loss = loss + (unused * 2).sum()
    loss.backward()
    red.finalize(mean=False)
    assert red.mean_scale == 0.5
    for b, m in zip(red.buckets, means):
        torch.testing.assert_close(b["flat"] * red.mean_scale, m, rtol=0, atol=0)
    red.finalize()                      # a second call in the same step is a no-op
    assert red.mean_scale == 0.5
    for b in red.buckets:
        b["flat"].mul_(red.mean_scale)
    params = params[:-1]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([g.tolist() for g in gathered])
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g0, g1 = torch.tensor(got[0]), torch.tensor(got[1])
    assert torch.equal(g0, g1), "ranks disagree after the all-reduce"
    # reference: average of the two ranks' local gradients, computed in one process
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    grads = []
    for rank in range(2):
        model.zero_grad()
        model(torch.full((5, 8), float(rank + 2))).pow(2).sum().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    want = torch.cat([(grads[0] + grads[1]) / 2, torch.full((3,), 1.0)])  # unused: (2 + 0) / 2; `never` is not compared
    torch.testing.assert_close(g0, want, rtol=1e-5, atol=1e-6)


def _worker_dynamic(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from da_detect_amd.parallel.reducer import BucketedGradReducer

    a, b, c = (torch.nn.Parameter(torch.ones(4)) for _ in range(3))
    red = BucketedGradReducer([a, b, c], bucket_bytes=16, learn_unused=False)
    seen = []
    for step in range(2):
        red.zero_grad()
        loss = (a * 2).sum()
        if rank == step:                  # `b` is used by a different rank in each step, `c` by none
            loss = loss + (b * 3).sum()
        loss.backward()
        red.finalize()
        seen.append((sorted(i for i, p in enumerate((a, b, c)) if id(p) in red.update_ids()), b.grad.tolist()))
    out.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_update_set_is_agreed_per_step_when_graphs_change():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dynamic, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1], "ranks disagree on which parameters to update"
    for ids, gb in got[0]:
        assert ids == [0, 1] and gb == [1.5] * 4


def _tuner_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from da_detect_amd.engine.trainer import WgradLaneTuner, agreed_times

    # rank 0 alone would keep the lane (12% faster), rank 1 alone would not (lane 8% slower): the job's step time is the
    # slower rank's, so both must see {0: 10.5 ms, 17000: 10.8 ms} and both keep one stream
    mine = {0: 10.0e-3, 17000: 8.8e-3} if rank == 0 else {0: 10.5e-3, 17000: 10.8e-3}
    assert WgradLaneTuner.choose(mine) == (17000 if rank == 0 else 0)
    agreed = agreed_times(mine, torch.device("cpu"))
    assert agreed == {0: 10.5e-3, 17000: 10.8e-3}, agreed
    out[rank] = WgradLaneTuner.choose(agreed)
    # and when the lane wins on the slowest rank too, every rank keeps it
    mine = {0: 10.0e-3 + 1e-4 * rank, 17000: 9.0e-3 + 2e-4 * rank}
    out[10 + rank] = WgradLaneTuner.choose(agreed_times(mine, torch.device("cpu")))
    dist.destroy_process_group()


def test_the_schedule_tuner_decides_on_the_slowest_rank_and_all_ranks_agree():
    """engine.trainer.WgradLaneTuner with N > 1: one all-reduce (MAX) of the candidates' times,
    one decision — a rank that would choose differently on its own timings follows the job's"""
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_tuner_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] == 0
    assert out[10] == out[11] == 17000


def _nan_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from da_detect_amd.engine.trainer import _loss_is_nan

    net = torch.nn.Linear(3, 2)
    out[rank] = _loss_is_nan(net, torch.tensor(1.0))                                      # clean on both ranks
    out[10 + rank] = _loss_is_nan(net, torch.tensor(float("nan") if rank == 1 else 1.0))  # only rank 1 sees it
    dist.destroy_process_group()


def test_every_rank_leaves_when_one_rank_sees_a_nan():
    """engine.trainer._loss_is_nan with N > 1: the verdict of the NaN / non-finite-GEMM test is
    all-reduced, so the rank that saw nothing leaves with the one that did instead of blocking in the next collective"""
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_nan_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] is False and out[1] is False
    assert out[10] is True and out[11] is True


def _handover_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from da_detect_amd.parallel.reducer import BucketedGradReducer

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    red = BucketedGradReducer(list(model.parameters()), bucket_bytes=256)
    red.broadcast_parameters(0)
    seen = []
    for step in range(2):
        red.zero_grad()
        model(torch.full((5, 8), float(rank + 1))).pow(2).sum().backward()
        assert red.can_hand_over_buckets() == (step > 0)       # the first step still has to agree on the unused parameters
        # every bucket is handed over exactly once, in order, holding the SUM over ranks (mean=False) at that moment
        red.finalize(mean=False, per_bucket=lambda i, b: seen.append((step, i, b["flat"].clone())))
    mine = [(s, i) for s, i, _ in seen]
    assert mine == [(s, i) for s in range(2) for i in range(len(red.buckets))], mine
    assert red.mean_scale == 0.5
    out[rank] = [f.tolist() for _, _, f in seen]
    dist.destroy_process_group()


def test_buckets_are_handed_over_in_order_with_their_reduced_sums():
    """reducer.finalize(per_bucket=...) (round 6: the optimizer updates a bucket's tensors as soon as ITS collective is
    complete): every bucket exactly once, in bucket order, with the same reduced contents on both ranks"""
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_handover_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] and len(out[0]) >= 6
