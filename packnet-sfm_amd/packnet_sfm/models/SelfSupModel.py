"""SelfSupModel: SfmModel + the multi-view photometric loss (API of the reference's
packnet_sfm/models/SelfSupModel.py)."""
from packnet_sfm.losses.multiview_photometric_loss import MultiViewPhotometricLoss
from packnet_sfm.models.model_utils import merge_outputs
from packnet_sfm.models.SfmModel import SfmModel


class SelfSupModel(SfmModel):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._photometric_loss = MultiViewPhotometricLoss(**kwargs)

    @property
    def logs(self):
        return {**super().logs, **self._photometric_loss.logs}

    def self_supervised_loss(self, image, ref_images, inv_depths, poses, intrinsics, return_logs=False, progress=0.0):
        return self._photometric_loss(image, ref_images, inv_depths, intrinsics, intrinsics, poses,
                                      return_logs=return_logs, progress=progress)

    def forward(self, batch, return_logs=False, progress=0.0):
        output = super().forward(batch, return_logs=return_logs)
        if not self.training:
            return output
        # the networks saw the (colour-jittered) rgb; the loss compares the un-jittered originals
        loss_output = self.self_supervised_loss(batch['rgb_original'], batch['rgb_context_original'], output['inv_depths'],
                                                output['poses'], batch['intrinsics'], return_logs=return_logs,
                                                progress=progress)
        return {'loss': loss_output['loss'], **merge_outputs(output, loss_output)}


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
