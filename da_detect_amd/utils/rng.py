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


_FALLBACK_COUNTER = [0]


def next_seed(device):
    """64-bit seed for a kernel that draws its own random keys (the fused proposal sampler).  Derived from the
    device generator's (seed, offset) and advancing that offset, so `torch.manual_seed` reproduces the draws exactly
    as it does for torch's own device-side random kernels."""
    try:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
    except Exception:   # generator without an offset (older torch): process-local counter
        seed, off = torch.initial_seed(), _FALLBACK_COUNTER[0]
        _FALLBACK_COUNTER[0] += 4
    z = (seed * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 32
    return z
