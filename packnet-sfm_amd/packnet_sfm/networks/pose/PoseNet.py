"""PoseNet (SfMLearner-style 7-layer CNN -> 6-DoF per context frame).  Drop-in for the reference's
packnet_sfm/networks/pose/PoseNet.py (same names/shapes: conv{1..7}.{0,1}.*, pose_pred.*).

0.2 % of the step's FLOPs (0.89 of 411 GFLOP per image): round 1 runs it on stock PyTorch-ROCm ops; moving its
stride-2 convolutions onto the gfx950 MFMA conv kernel is listed as 'next' in SURVEY.md section 8(f) (N2)."""
import torch
import torch.nn as nn


def conv_gn(in_planes, out_planes, kernel_size=3):
    return nn.Sequential(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, stride=2),
        nn.GroupNorm(16, out_planes),
        nn.ReLU(inplace=True))


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
        pose = self.pose_pred(x).mean(3).mean(2)
        return 0.01 * pose.view(pose.size(0), self.nb_ref_imgs, 6)
