"""Random draws of the training path (proposal sampling permutations, dropout masks).

The reference draws them from torch's global generator of the tensor's device
(reference: maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:57-58, F.dropout in
modeling/da_heads/da_heads.py:63,65).  Default here is the same (device generator).  For parity tests the
draws can be switched to the global CPU generator — then, for one `torch.manual_seed`, this package on the
GPU and the reference / oracle on the CPU consume identical random streams, because the calls are made in the
same order with the same sizes.
"""
import torch

_CPU_STREAM = False


def use_cpu_stream(flag=True):
    global _CPU_STREAM
    _CPU_STREAM = bool(flag)


def cpu_stream_enabled():
    return _CPU_STREAM


def randperm(n, device):
    if _CPU_STREAM:
        return torch.randperm(n).to(device)
    return torch.randperm(n, device=device)


def dropout_mask(shape, p, device):
    """multiplicative mask of F.dropout: bernoulli(1-p) / (1-p)"""
    if _CPU_STREAM:
        return torch.empty(shape, dtype=torch.float32).bernoulli_(1 - p).div_(1 - p).to(device)
    return torch.empty(shape, dtype=torch.float32, device=device).bernoulli_(1 - p).div_(1 - p)
