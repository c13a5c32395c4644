"""Index samplers of the training loop (reference: maskrcnn_benchmark/data/samplers/{distributed,
grouped_batch_sampler,iteration_based_batch_sampler}.py).  Same index streams as the reference for the same seeds /
epochs (tests compare them on recorded sequences); pure host code."""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import BatchSampler, Sampler


class DistributedSampler(Sampler):
    """rank r sees the r-th contiguous slice of a permutation seeded by the epoch (distributed.py:40-60); the list
    is padded by wrapping around so that every rank gets ceil(N / world) indices"""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) / float(num_replicas)))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        indices += indices[: self.total_size - len(indices)]
        lo = self.num_samples * self.rank
        return iter(indices[lo: lo + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class GroupedBatchSampler(BatchSampler):
    """batches never mix groups (aspect-ratio bins); batches are emitted in the order in which their first element
    appears in the wrapped sampler's stream (grouped_batch_sampler.py:41-95)"""

    def __init__(self, sampler, group_ids, batch_size, drop_uneven=False):
        if not isinstance(sampler, Sampler):
            raise ValueError("sampler should be an instance of torch.utils.data.Sampler, but got sampler={}".format(sampler))
        self.sampler = sampler
        self.group_ids = torch.as_tensor(group_ids)
        assert self.group_ids.dim() == 1
        self.batch_size = batch_size
        self.drop_uneven = drop_uneven
        self.groups = torch.unique(self.group_ids).sort(0)[0].tolist()
        self._batches = None
        self._reuse = False

    def _prepare_batches(self):
        stream = list(self.sampler)
        position = {idx: pos for pos, idx in enumerate(stream)}
        gid = self.group_ids.tolist()
        batches = []
        for g in self.groups:                      # members of a group in stream order, cut into batches
            members = [idx for idx in stream if gid[idx] == g]
            batches += [members[i:i + self.batch_size] for i in range(0, len(members), self.batch_size)]
        batches.sort(key=lambda b: position[b[0]])
        if self.drop_uneven:
            batches = [b for b in batches if len(b) == self.batch_size]
        return batches

    def __iter__(self):
        if self._reuse:
            self._reuse = False
        else:
            self._batches = self._prepare_batches()
        return iter(self._batches)

    def __len__(self):
        if self._batches is None:
            self._batches = self._prepare_batches()
            self._reuse = True
        return len(self._batches)


class IterationBasedBatchSampler(BatchSampler):
    """re-iterates a batch sampler until num_iterations batches were produced, bumping the wrapped sampler's epoch
    to the current iteration count at every restart (iteration_based_batch_sampler.py:16-31)"""

    def __init__(self, batch_sampler, num_iterations, start_iter=0):
        self.batch_sampler = batch_sampler
        self.num_iterations = num_iterations
        self.start_iter = start_iter

    def __iter__(self):
        iteration = self.start_iter
        while iteration <= self.num_iterations:
            if hasattr(self.batch_sampler.sampler, "set_epoch"):
                self.batch_sampler.sampler.set_epoch(iteration)
            for batch in self.batch_sampler:
                iteration += 1
                if iteration > self.num_iterations:
                    break
                yield batch

    def __len__(self):
        return self.num_iterations
