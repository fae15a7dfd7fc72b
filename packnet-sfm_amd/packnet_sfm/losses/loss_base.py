"""LossBase + ProgressiveScaling (names of the reference's packnet_sfm/losses/loss_base.py)."""
import numpy as np
import torch.nn as nn

from packnet_sfm.utils.reporting import Reporting


class ProgressiveScaling:
    """Number of loss scales as a function of training progress.

    Reference quirk kept on purpose (losses/loss_base.py:21-44): the schedule is built as a numpy array but applied only
    `if is_list(...)`, which a numpy array is not -- so in the reference the number of scales NEVER decreases, whatever
    `progressive_scaling` is.  Identical training behaviour matters more than the apparent intent; `scheduled()` gives
    the intended value (one scale fewer each time `progress` passes a multiple of `progressive_scaling`)."""

    def __init__(self, progressive_scaling, num_scales=4):
        self.num_scales = num_scales
        self.thresholds = None
        if progressive_scaling > 0.0:
            self.thresholds = np.float32([progressive_scaling * (i + 1) for i in range(num_scales - 1)] + [1.0])

    def scheduled(self, progress):
        if self.thresholds is None:
            return self.num_scales
        return int(self.num_scales - np.searchsorted(self.thresholds, progress))

    def __call__(self, progress):
        return self.num_scales


class LossBase(Reporting, nn.Module):
    def __init__(self):
        nn.Module.__init__(self)

    logs = property(lambda self: self._report('logs'))
    metrics = property(lambda self: self._report('metrics'))

    def add_metric(self, key, val):
        self._record('metrics', key, val)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
