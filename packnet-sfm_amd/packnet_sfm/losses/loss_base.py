"""LossBase + ProgressiveScaling (names of the reference's packnet_sfm/losses/loss_base.py)."""
import numpy as np
import torch.nn as nn

from packnet_sfm.utils.reporting import Reporting


class ProgressiveScaling:
    """Number of loss scales as training progresses: one scale fewer each time `progress` passes a multiple of
    `progressive_scaling` (0 disables the schedule)."""

    def __init__(self, progressive_scaling, num_scales=4):
        self.num_scales = num_scales
        self.thresholds = None
        if progressive_scaling > 0.0:
            self.thresholds = np.float32([progressive_scaling * (i + 1) for i in range(num_scales - 1)] + [1.0])

    def __call__(self, progress):
        if self.thresholds is None:
            return self.num_scales
        return int(self.num_scales - np.searchsorted(self.thresholds, progress))


class LossBase(Reporting, nn.Module):
    def __init__(self):
        nn.Module.__init__(self)

    logs = property(lambda self: self._report('logs'))
    metrics = property(lambda self: self._report('metrics'))

    def add_metric(self, key, val):
        self._record('metrics', key, val)
