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
    assert early[1] == (len(red.buckets) if rank == 0 else red.buckets.index(red._bucket_of[id(unused)])), early
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
