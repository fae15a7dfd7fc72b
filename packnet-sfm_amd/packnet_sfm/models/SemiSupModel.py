"""SemiSupModel: SelfSupModel + a supervised inverse-depth loss (API of the reference's packnet_sfm/models/SemiSupModel.py)."""
import torch

from packnet_sfm.losses.supervised_loss import SupervisedLoss
from packnet_sfm.models.model_utils import merge_outputs
from packnet_sfm.models.SelfSupModel import SelfSupModel, SfmModel
from packnet_sfm.utils.depth import depth2inv


class SemiSupModel(SelfSupModel):
    """
    Parameters
    ----------
    supervised_loss_weight : float in (0, 1]; 1 = fully supervised (no pose network needed)
    kwargs : loss options of SelfSupModel and SupervisedLoss
    """

    def __init__(self, supervised_loss_weight=0.9, **kwargs):
        super().__init__(**kwargs)
        assert 0. < supervised_loss_weight <= 1., "Model requires (0, 1] supervision"
        self.supervised_loss_weight = supervised_loss_weight
        self._supervised_loss = SupervisedLoss(**kwargs)
        if self.supervised_loss_weight == 1:
            self._network_requirements.remove('pose_net')
        if self.supervised_loss_weight > 0:
            self._train_requirements.append('gt_depth')

    @property
    def logs(self):
        return {**super().logs, **self._supervised_loss.logs}

    def supervised_loss(self, inv_depths, gt_inv_depths, return_logs=False, progress=0.0):
        return self._supervised_loss(inv_depths, gt_inv_depths, return_logs=return_logs, progress=progress)

    def forward(self, batch, return_logs=False, progress=0.0):
        if not self.training:
            return SfmModel.forward(self, batch)
        if self.supervised_loss_weight == 1.:
            self_sup_output = SfmModel.forward(self, batch)
            loss = torch.tensor([0.]).type_as(batch['rgb'])
        else:
            self_sup_output = SelfSupModel.forward(self, batch)
            loss = (1.0 - self.supervised_loss_weight) * self_sup_output['loss']
        sup_output = self.supervised_loss(self_sup_output['inv_depths'], depth2inv(batch['depth']),
                                          return_logs=return_logs, progress=progress)
        loss = loss + self.supervised_loss_weight * sup_output['loss']
        return {'loss': loss, **merge_outputs(self_sup_output, sup_output)}
