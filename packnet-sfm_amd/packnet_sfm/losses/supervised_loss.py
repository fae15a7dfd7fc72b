"""Supervised inverse-depth loss on a fused MI355X kernel.

Drop-in for the reference's packnet_sfm/losses/supervised_loss.py: `SupervisedLoss(supervised_method='sparse-l1',
supervised_num_scales=4, progressive_scaling=0.0, **kwargs)`, `forward(inv_depths, gt_inv_depth, return_logs, progress)`
-> {'loss': [1], 'metrics': {'supervised_loss'}}.  Per scale the reference gathers the valid pixels with boolean
indexing (a device->host sync) and runs several ATen reductions; here a scale is one streaming reduction kernel
(csrc/supervised.hip) and its backward one elementwise kernel.
"""
from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops
from packnet_sfm.losses.loss_base import LossBase, ProgressiveScaling
from packnet_sfm.utils.image import match_scales


def get_loss_method(supervised_method):
    """'sparse-l1', 'dense-berhu', ... -> (kernel method id, sparse flag); same suffix rule as the reference's
    get_loss_func (supervised_loss.py:70-84) and the 'sparse' prefix rule of calculate_loss (:141-146)."""
    for name in ('abs_rel', 'l1', 'mse', 'berhu', 'silog'):
        if supervised_method.endswith(name):
            return ops.SUP_METHODS[name], supervised_method.startswith('sparse')
    raise ValueError('Unknown supervised loss {}'.format(supervised_method))


class SupervisedLoss(LossBase):
    def __init__(self, supervised_method='sparse-l1', supervised_num_scales=4, progressive_scaling=0.0, **kwargs):
        super().__init__()
        self.method, self.sparse = get_loss_method(supervised_method)
        self.supervised_method = supervised_method
        self.n = supervised_num_scales
        self.progressive_scaling = ProgressiveScaling(progressive_scaling, self.n)

    @property
    def logs(self):
        return {'supervised_num_scales': self.n}

    def calculate_loss(self, inv_depths, gt_inv_depths):
        """Average over scales of the per-scale loss (masked to gt > 0 for the 'sparse-*' methods)."""
        return sum(HF.supervised_loss(inv_depths[i], gt_inv_depths[i], self.method, self.sparse)
                   for i in range(self.n)) / self.n

    def forward(self, inv_depths, gt_inv_depth, return_logs=False, progress=0.0):
        self.n = self.progressive_scaling(progress)
        gt_inv_depths = match_scales(gt_inv_depth, inv_depths, self.n, mode='nearest', align_corners=None)
        loss = self.calculate_loss(inv_depths, gt_inv_depths)
        self.add_metric('supervised_loss', loss)
        return {'loss': loss.unsqueeze(0), 'metrics': self.metrics}


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
