"""PoseNet (SfMLearner-style 7-layer CNN -> 6-DoF per context frame) on the gfx950 kernels.

Drop-in for the reference's packnet_sfm/networks/pose/PoseNet.py: same constructor, same `forward(image, context)`
-> [B, nb_ref_imgs, 6], same parameter names/shapes (conv{1..7}.0.{weight,bias} = stride-2 conv, conv{1..7}.1.* =
GroupNorm(16), pose_pred.*).  Each conv_gn block is: strided MFMA conv (csrc/conv2d.hip, stride 2, zero pad k//2) ->
fused GroupNorm(16)+ReLU (csrc/groupnorm.hip); the 1x1 head is the stride-1 MFMA conv.  torch.nn.Conv2d / GroupNorm
objects are parameter containers only (identical default initialisation order as the reference).
"""
import torch
import torch.nn as nn

from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops as _ops


class _ConvGNReLU(nn.Sequential):
    """nn.Sequential(Conv2d(stride 2), GroupNorm(16), ReLU) by name and parameters; HIP kernels by execution."""

    def __init__(self, in_planes, out_planes, kernel_size):
        super().__init__(
            nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, stride=2),
            nn.GroupNorm(16, out_planes),
            nn.ReLU(inplace=True))
        self._packed = HF.PackedConvWeight()

    def forward(self, x):
        conv, norm = self[0], self[1]
        y = HF.conv2d_stride2(x, conv.weight, conv.bias, self._packed)
        return HF.groupnorm_act(y, norm.weight, norm.bias, 16, norm.eps, _ops.ACT_RELU)


def conv_gn(in_planes, out_planes, kernel_size=3):
    return _ConvGNReLU(in_planes, out_planes, kernel_size)


class PoseNet(nn.Module):
    def __init__(self, nb_ref_imgs=2, rotation_mode='euler', **kwargs):
        super().__init__()
        self.nb_ref_imgs = nb_ref_imgs
        self.rotation_mode = rotation_mode
        widths = [16, 32, 64, 128, 256, 256, 256]
        kernels = [7, 5, 3, 3, 3, 3, 3]
        cin = 3 * (1 + nb_ref_imgs)
        for i, (c, k) in enumerate(zip(widths, kernels)):
            setattr(self, 'conv%d' % (i + 1), conv_gn(cin, c, kernel_size=k))
            cin = c
        self.pose_pred = nn.Conv2d(cin, 6 * nb_ref_imgs, kernel_size=1, padding=0)
        self._head_packed = HF.PackedConvWeight()
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, image, context):
        assert len(context) == self.nb_ref_imgs
        x = torch.cat([image] + list(context), 1)
        for i in range(7):
            x = getattr(self, 'conv%d' % (i + 1))(x)
        pose = HF.conv2d(x, self.pose_pred.weight, self.pose_pred.bias, self._head_packed)
        pose = pose.mean(3).mean(2)
        return 0.01 * pose.view(pose.size(0), self.nb_ref_imgs, 6)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
