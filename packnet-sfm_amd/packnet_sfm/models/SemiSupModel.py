"""SemiSupModel: the self-supervised model plus a supervised inverse-depth term (contract of the reference's
packnet_sfm/models/SemiSupModel.py: `supervised_loss_weight` in (0, 1], batch key 'depth', merged metrics)."""
import torch

from packnet_sfm.losses.supervised_loss import SupervisedLoss
from packnet_sfm.models.model_utils import merge_outputs
from packnet_sfm.models.SelfSupModel import SelfSupModel, SfmModel
from packnet_sfm.utils.depth import depth2inv


class SemiSupModel(SelfSupModel):
    """
    supervised_loss_weight : float   w in (0, 1]: loss = (1 - w) * self-supervised + w * supervised;
                                     w == 1 is fully supervised and needs no pose network
    kwargs                           options of the two losses (MultiViewPhotometricLoss, SupervisedLoss)
    """

    def __init__(self, supervised_loss_weight=0.9, **kwargs):
        super().__init__(**kwargs)
        if not 0. < supervised_loss_weight <= 1.:
            raise AssertionError("Model requires (0, 1] supervision")
        self.supervised_loss_weight = supervised_loss_weight
        self._supervised_loss = SupervisedLoss(**kwargs)
        self._train_requirements.append('gt_depth')
        if self._fully_supervised:
            self._network_requirements.remove('pose_net')

    _fully_supervised = property(lambda self: self.supervised_loss_weight == 1)

    @property
    def logs(self):
        merged = dict(super().logs)
        merged.update(self._supervised_loss.logs)
        return merged

    def supervised_loss(self, inv_depths, gt_inv_depths, return_logs=False, progress=0.0):
        return self._supervised_loss(inv_depths, gt_inv_depths, return_logs=return_logs, progress=progress)

    def forward(self, batch, return_logs=False, progress=0.0):
        if not self.training:                           # evaluation: predictions only
            return SfmModel.forward(self, batch)
        w = self.supervised_loss_weight
        if self._fully_supervised:
            base = SfmModel.forward(self, batch)
            total = torch.zeros(1).type_as(batch['rgb'])
        else:
            base = SelfSupModel.forward(self, batch)
            total = (1.0 - w) * base['loss']
        supervised = self.supervised_loss(base['inv_depths'], depth2inv(batch['depth']), return_logs=return_logs,
                                          progress=progress)
        result = merge_outputs(base, supervised)
        result['loss'] = total + w * supervised['loss']
        return result


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
