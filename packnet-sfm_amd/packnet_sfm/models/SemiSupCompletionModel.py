"""SemiSupCompletionModel: depth prediction AND completion (contract of the reference's
packnet_sfm/models/SemiSupCompletionModel.py; configs/train_packnet_san_kitti.yaml): the depth network also receives
`input_depth`, and when it returns an RGB-D prediction that one is supervised too and its feature-consistency loss is added."""
import torch

from packnet_sfm.losses.supervised_loss import SupervisedLoss
from packnet_sfm.models.model_utils import merge_outputs
from packnet_sfm.models.SelfSupModel import SelfSupModel, SfmModel
from packnet_sfm.utils.depth import depth2inv


class SemiSupCompletionModel(SelfSupModel):
    def __init__(self, supervised_loss_weight=0.9, weight_rgbd=1.0, **kwargs):
        super().__init__(**kwargs)
        assert 0. < supervised_loss_weight <= 1., "Model requires (0, 1] supervision"
        self.supervised_loss_weight = supervised_loss_weight
        self._supervised_loss = SupervisedLoss(**kwargs)
        if self.supervised_loss_weight == 1:
            self._network_requirements.remove('pose_net')
        self._train_requirements.append('gt_depth')
        self._input_keys = ['rgb', 'input_depth', 'intrinsics']
        self.weight_rgbd = weight_rgbd

    @property
    def logs(self):
        return {**super().logs, **self._supervised_loss.logs}

    def supervised_loss(self, inv_depths, gt_inv_depths, return_logs=False, progress=0.0):
        return self._supervised_loss(inv_depths, gt_inv_depths, return_logs=return_logs, progress=progress)

    def forward(self, batch, return_logs=False, progress=0.0, **kwargs):
        if not self.training:
            return SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        w = self.supervised_loss_weight
        if w == 1.:
            out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
            loss = torch.tensor([0.]).type_as(batch['rgb'])
        else:
            out = SelfSupModel.forward(self, batch, return_logs=return_logs, progress=progress, **kwargs)
            loss = (1.0 - w) * out['loss']
        gt = depth2inv(batch['depth'])
        sup = self.supervised_loss(out['inv_depths'], gt, return_logs=return_logs, progress=progress)
        loss = loss + w * sup['loss']
        if 'inv_depths_rgbd' in out:
            sup2 = self.supervised_loss(out['inv_depths_rgbd'], gt, return_logs=return_logs, progress=progress)
            loss = loss + self.weight_rgbd * w * sup2['loss']
            if 'depth_loss' in out:
                loss = loss + out['depth_loss']
        return {'loss': loss, **merge_outputs(out, sup)}


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
