import torch


def cat(tensors, dim=0):
    """torch.cat that returns the element itself for a single-element list (modeling/utils.py:8-15)"""
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)
