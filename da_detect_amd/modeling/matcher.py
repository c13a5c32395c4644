"""Ground-truth <-> prediction matcher (reference: maskrcnn_benchmark/modeling/matcher.py:5-112)."""
import torch


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        """[M gt, N pred] quality -> int64[N]: matched gt index, or -1 (below low) / -2 (between)."""
        if match_quality_matrix.numel() == 0:
            if match_quality_matrix.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        vals, matches = match_quality_matrix.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        matches[vals < self.low_threshold] = Matcher.BELOW_LOW_THRESHOLD
        matches[(vals >= self.low_threshold) & (vals < self.high_threshold)] = Matcher.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            # predictions that are the best match of some gt (ties included) keep their argmax (matcher.py:94-112)
            best_per_gt, _ = match_quality_matrix.max(dim=1)
            pairs = torch.nonzero(match_quality_matrix == best_per_gt[:, None])
            upd = pairs[:, 1]
            matches[upd] = all_matches[upd]
        return matches
