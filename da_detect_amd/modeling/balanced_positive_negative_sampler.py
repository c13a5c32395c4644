"""Fixed-size positive/negative sampler (reference:
maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:13-76).  Random permutations come from
`utils.rng` so the stream can be made device independent for parity tests."""
import torch

from ..utils import rng



def _nz(mask, size):
    """nonzero with a host-known result size (no device->host round trip); `_STATIC = False` restores nonzero()"""
    if _STATIC:
        return torch.nonzero_static(mask, size=size)
    return torch.nonzero(mask)


_STATIC = True

class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def __call__(self, matched_idxs, all_negative=None):
        """per image labels (-1 ignore, 0 negative, >0 positive) -> (list of pos masks, list of neg masks).
        all_negative[i] = True: the caller built image i's labels as all zeros (target-domain images, the DA ROI
        sample) — the index lists are then known without reading the labels back.  For the other images the
        positive / negative COUNTS of the whole batch come back in one host round trip and the index lists are built
        with nonzero_static (the reference's two nonzero() per image are a round trip each).  The two permutations
        per image are drawn exactly as in the reference (same sizes, same order).  `self.last_counts[i]` =
        (num_pos, num_neg) actually sampled."""
        n_img = len(matched_idxs)
        known = [bool(all_negative is not None and all_negative[i]) for i in range(n_img)]
        masks, counts = {}, {}
        pending = [i for i in range(n_img) if not known[i]]
        if pending:
            stacked = []
            for i in pending:
                m = matched_idxs[i]
                masks[i] = (m >= 1, m == 0)
                stacked += [masks[i][0].sum(), masks[i][1].sum()]
            flat = torch.stack(stacked).tolist()
            for j, i in enumerate(pending):
                counts[i] = (int(flat[2 * j]), int(flat[2 * j + 1]))
        pos_idx, neg_idx = [], []
        self.last_counts = []
        for i, m in enumerate(matched_idxs):
            if known[i]:
                n = m.numel()
                num_neg = min(n, self.batch_size_per_image)
                rng.randperm(0, m.device)
                perm2 = rng.randperm(n, m.device)[:num_neg]
                nm = torch.zeros_like(m, dtype=torch.bool)
                nm[perm2] = 1
                pos_idx.append(torch.zeros_like(m, dtype=torch.bool))
                neg_idx.append(nm)
                self.last_counts.append((0, num_neg))
                continue
            n_pos, n_neg = counts[i]
            positive = _nz(masks[i][0], size=n_pos).squeeze(1)
            negative = _nz(masks[i][1], size=n_neg).squeeze(1)
            num_pos = min(n_pos, int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(n_neg, self.batch_size_per_image - num_pos)
            perm1 = rng.randperm(n_pos, positive.device)[:num_pos]
            perm2 = rng.randperm(n_neg, negative.device)[:num_neg]
            pm = torch.zeros_like(m, dtype=torch.bool)
            nm = torch.zeros_like(m, dtype=torch.bool)
            pm[positive[perm1]] = 1
            nm[negative[perm2]] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
            self.last_counts.append((num_pos, num_neg))
        return pos_idx, neg_idx
