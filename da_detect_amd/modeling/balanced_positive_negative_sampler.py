"""Fixed-size positive/negative sampler (reference:
maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:13-76).  Random permutations come from
`utils.rng` so the stream can be made device independent for parity tests."""
import torch

from ..utils import rng


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def __call__(self, matched_idxs, all_negative=None):
        """per image labels (-1 ignore, 0 negative, >0 positive) -> (list of pos masks, list of neg masks).
        all_negative[i] = True: the caller built image i's labels as all zeros (target-domain images, the DA ROI
        sample) — the index lists are then known without reading the labels back (no nonzero round trips); the two
        permutations are drawn exactly as in the general path (same sizes, same order)."""
        pos_idx, neg_idx = [], []
        for i, m in enumerate(matched_idxs):
            if all_negative is not None and all_negative[i]:
                n = m.numel()
                num_neg = min(n, self.batch_size_per_image)
                rng.randperm(0, m.device)
                perm2 = rng.randperm(n, m.device)[:num_neg]
                nm = torch.zeros_like(m, dtype=torch.bool)
                nm[perm2] = 1
                pos_idx.append(torch.zeros_like(m, dtype=torch.bool))
                neg_idx.append(nm)
                continue
            positive = torch.nonzero(m >= 1).squeeze(1)
            negative = torch.nonzero(m == 0).squeeze(1)
            num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
            perm1 = rng.randperm(positive.numel(), positive.device)[:num_pos]
            perm2 = rng.randperm(negative.numel(), negative.device)[:num_neg]
            pm = torch.zeros_like(m, dtype=torch.bool)
            nm = torch.zeros_like(m, dtype=torch.bool)
            pm[positive[perm1]] = 1
            nm[negative[perm2]] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx
