"""torch.distributed helpers with the reference's names (reference: maskrcnn_benchmark/utils/comm.py).
On ROCm builds of PyTorch the "nccl" backend is RCCL; collectives ride xGMI inside a node."""
import pickle

import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """barrier among all ranks (comm.py:34-45)"""
    if get_world_size() > 1:
        dist.barrier()


def _comm_device():
    return torch.device("cuda") if dist.get_backend() == "nccl" else torch.device("cpu")


def all_gather(data):
    """gather arbitrary picklable objects from every rank (comm.py:48-88): sizes first, then padded bytes"""
    world = get_world_size()
    if world == 1:
        return [data]
    dev = _comm_device()
    payload = torch.frombuffer(bytearray(pickle.dumps(data)), dtype=torch.uint8).to(dev)
    local = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if payload.numel() < mx:
        payload = torch.cat([payload, torch.zeros(mx - payload.numel(), dtype=torch.uint8, device=dev)])
    bufs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, payload)
    return [pickle.loads(b[:n].cpu().numpy().tobytes()) for b, n in zip(bufs, sizes)]


def reduce_dict(input_dict, average=True):
    """reduce a dict of scalars to rank 0 in sorted-key order (comm.py:91-117, engine/trainer.py:41-63)"""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return {k: v for k, v in zip(names, values)}
