"""LossBase + ProgressiveScaling (API of the reference's packnet_sfm/losses/loss_base.py)."""
import numpy as np
import torch.nn as nn


class ProgressiveScaling:
    """Drops one loss scale each time training progress passes a multiple of `progressive_scaling` (0 = off)."""

    def __init__(self, progressive_scaling, num_scales=4):
        self.num_scales = num_scales
        if progressive_scaling > 0.0:
            steps = [progressive_scaling * (i + 1) for i in range(num_scales - 1)] + [1.0]
            self.thresholds = np.float32(steps)
        else:
            self.thresholds = None

    def __call__(self, progress):
        if self.thresholds is None:
            return self.num_scales
        return int(self.num_scales - np.searchsorted(self.thresholds, progress))


class LossBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._logs = {}
        self._metrics = {}

    @property
    def logs(self):
        return self._logs

    @property
    def metrics(self):
        return self._metrics

    def add_metric(self, key, val):
        self._metrics[key] = val.detach()
