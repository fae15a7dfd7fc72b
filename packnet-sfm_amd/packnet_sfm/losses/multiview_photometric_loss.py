"""Self-supervised multi-view photometric loss on fused MI355X kernels.

Drop-in for the reference's packnet_sfm/losses/multiview_photometric_loss.py (same constructor keywords, same
`forward(image, context, inv_depths, K, ref_K, poses, return_logs, progress)` and the same
{'loss': [1], 'metrics': {'photometric_loss', 'smoothness_loss'}} result), but per scale the ~190 ATen ops of the
reference collapse into a handful of HIP launches (csrc/loss.hip):

    view synthesis   inv2depth -> reconstruct -> rigid transform -> project -> bilinear gather   (all J contexts)
    photometric      SSIM(3x3, reflect) + L1, automask candidates, per-pixel min/mean, pixel mean (one scalar)
    smoothness       edge-aware first differences, |.| means

and the backward pass mirrors them (hand-written derivatives, incl. the 12-float pose gradient).
"""
import torch

from packnet_sfm.geometry.camera_utils import scale_intrinsics
from packnet_sfm.hip import functional as HF
from packnet_sfm.losses.loss_base import LossBase, ProgressiveScaling
from packnet_sfm.utils.image import match_scales


class MultiViewPhotometricLoss(LossBase):
    """
    Parameters (as in the reference)
    ----------
    num_scales, ssim_loss_weight, occ_reg_weight (unused), smooth_loss_weight, C1, C2,
    photometric_reduce_op ('min' | 'mean'), disp_norm (unused), clip_loss, progressive_scaling,
    padding_mode, automask_loss
    """

    def __init__(self, num_scales=4, ssim_loss_weight=0.85, occ_reg_weight=0.1, smooth_loss_weight=0.1,
                 C1=1e-4, C2=9e-4, photometric_reduce_op='mean', disp_norm=True, clip_loss=0.5,
                 progressive_scaling=0.0, padding_mode='zeros', automask_loss=False, **kwargs):
        super().__init__()
        self.n = num_scales
        self.ssim_loss_weight = ssim_loss_weight
        self.occ_reg_weight = occ_reg_weight
        self.smooth_loss_weight = smooth_loss_weight
        self.C1 = C1
        self.C2 = C2
        self.photometric_reduce_op = photometric_reduce_op
        self.disp_norm = disp_norm
        self.clip_loss = clip_loss
        self.padding_mode = padding_mode
        self.automask_loss = automask_loss
        self.progressive_scaling = ProgressiveScaling(progressive_scaling, self.n)
        if self.automask_loss:
            assert self.photometric_reduce_op == 'min', \
                'For automasking only the min photometric_reduce_op is supported.'
        if photometric_reduce_op not in ('min', 'mean'):
            raise NotImplementedError('Unknown photometric_reduce_op: {}'.format(photometric_reduce_op))
        # configurations the fused kernels do not cover fail loudly instead of silently taking another path
        if padding_mode not in ('zeros', 'border', 'reflection'):
            raise ValueError('Unknown padding_mode {}'.format(padding_mode))
        if ssim_loss_weight < 0.0:
            raise ValueError('ssim_loss_weight must be >= 0')
        # (ssim_loss_weight == 0 with 'min' / clipping: the reference works on per-channel maps then (:205-219, :238-246) -- HF.photometric
        # routes those configurations to the L1 candidate kernels)

    @property
    def logs(self):
        return {'num_scales': self.n}

    def forward(self, image, context, inv_depths, K, ref_K, poses, return_logs=False, progress=0.0):
        """
        image [B,3,H,W]; context: list of J [B,3,H,W]; inv_depths: list of [B,1,h,w]; K, ref_K [B,3,3];
        poses: list of J Pose (target -> context).  Returns {'loss': [1], 'metrics': {...}}.
        """
        self.n = n = self.progressive_scaling(progress)
        H, W = image.shape[-2:]
        images = match_scales(image, inv_depths, n)
        refs_full = torch.stack(list(context), 0).contiguous()                        # [J,B,3,H,W]
        T = torch.stack([p.mat if hasattr(p, 'mat') else p for p in poses], 0).float()   # [J,B,4,4]
        K32, rK32 = K.float(), ref_K.float()
        reduce_op = HF.REDUCE_MIN if self.photometric_reduce_op == 'min' else HF.REDUCE_MEAN

        photometric, smoothness = [], []
        for i in range(n):
            h, w = inv_depths[i].shape[-2:]
            if (h, w) == (H, W):
                refs_i, Ki, rKi = refs_full, K32, rK32
            else:
                refs_i = torch.stack(match_scales_list(context, inv_depths[i]), 0).contiguous()
                s = w / float(W)
                Ki, rKi = scale_intrinsics(K32.clone(), s, s), scale_intrinsics(rK32.clone(), s, s)
            # view synthesis + photometric terms of the scale
            photometric.append(HF.warp_photometric(
                inv_depths[i], refs_i, images[i], Ki.contiguous(), rKi.contiguous(), T, self.ssim_loss_weight, self.C1, self.C2,
                bool(self.automask_loss), reduce_op, float(self.clip_loss), self.padding_mode))
        if self.smooth_loss_weight > 0.0:
            # (mean normalisation of the inverse depth, reference :269-271, fused into the kernels)
            smoothness = [HF.smoothness_norm(inv_depths[i], images[i]) for i in range(n)]
        # loss = mean_i photometric[i] + weight * mean_i (smoothness[i] / 2^i): the reference's Python sums of 0-dim tensors
        # (:248-252, :275-280, :337-338) as one launch over the 2n device scalars, same operation order
        loss, smoothness_loss, _photometric_only = HF.loss_combine(photometric, smoothness, self.smooth_loss_weight)
        # the reference adds the smoothness term IN PLACE (:337-338) into the tensor its 'photometric_loss' metric aliases, so
        # that metric reports the TOTAL loss once smoothness is enabled; kept for identical logs
        self.add_metric('photometric_loss', loss)
        if self.smooth_loss_weight > 0.0:
            self.add_metric('smoothness_loss', smoothness_loss)

        return {'loss': loss.unsqueeze(0), 'metrics': self.metrics}


def match_scales_list(images, target):
    """Each image of `images` resized (bilinear, align_corners=True) to the resolution of `target`."""
    return [match_scales(img, [target], 1)[0] for img in images]


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
