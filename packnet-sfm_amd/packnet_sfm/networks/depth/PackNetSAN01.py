"""PackNetSAN01 (PackNet-SAN, https://arxiv.org/abs/2103.16690) on MI355X kernels: the PackNet encoder/decoder with 32-channel
stem and 4 3-D feature maps, plus the sparse depth-completion branch whose features are injected into the skip connections.

Drop-in for the reference's packnet_sfm/networks/depth/PackNetSAN01.py: `PackNetSAN01(dropout=..., version='1A')`,
`net(rgb=..., input_depth=...)` -> {'inv_depths'[, 'inv_depths_rgbd', 'depth_loss']} (a LIST also in eval mode, :142-147),
parameter names `encoder.*`, `decoder.*`, `mconvs.*`, `weight`, `bias` (:160-180).  The dense part runs on the d=4 packing /
unpacking kernels that already serve PackNetSlim01; the sparse branch is networks/layers/minkowski_encoder.py.
"""
import torch
import torch.nn as nn

from packnet_sfm.hip import functional as HF
from packnet_sfm.networks.layers.minkowski_encoder import MinkowskiEncoder
from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, InvDepth, PackLayerConv3d, ResidualBlock, UnpackLayerConv3d


class Encoder(nn.Module):
    def __init__(self, version, in_channels, ni, n1, n2, n3, n4, n5, pack_kernel, num_blocks, num_3d_feat, dropout):
        super().__init__()
        self.version = version
        n = [n1, n2, n3, n4, n5]
        self.pre_calc = Conv2D(in_channels, ni, 5, 1)
        for i in range(5):
            setattr(self, 'pack%d' % (i + 1), PackLayerConv3d(n[i], pack_kernel[i], d=num_3d_feat))
        self.conv1 = Conv2D(ni, n1, 7, 1)
        for i in range(4):
            setattr(self, 'conv%d' % (i + 2), ResidualBlock(n[i], n[i + 1], num_blocks[i], 1, dropout=dropout))

    def forward(self, rgb):
        x = self.pre_calc(rgb)
        x1p = self.pack1(self.conv1(x))
        x2p = self.pack2(self.conv2(x1p))
        x3p = self.pack3(self.conv3(x2p))
        x4p = self.pack4(self.conv4(x3p))
        x5p = self.pack5(self.conv5(x4p))
        return x5p, [x, x1p, x2p, x3p, x4p]


class Decoder(nn.Module):
    def __init__(self, version, out_channels, ni, n1, n2, n3, n4, n5, unpack_kernel, iconv_kernel, num_3d_feat):
        super().__init__()
        self.version = version
        n = [n1, n2, n3, n4, n5]
        nin = [n1 + ni + out_channels, n2 + n1 + out_channels, n3 + n2 + out_channels, n4 + n3, n5 + n4]
        unpack_in = [n2, n3, n4, n5, n5]
        for j, i in enumerate((4, 3, 2, 1, 0)):           # registration order of the reference: unpack5..1, iconv5..1
            setattr(self, 'unpack%d' % (i + 1), UnpackLayerConv3d(unpack_in[i], n[i], unpack_kernel[j], d=num_3d_feat))
        for j, i in enumerate((4, 3, 2, 1, 0)):
            setattr(self, 'iconv%d' % (i + 1), Conv2D(nin[i], n[i], iconv_kernel[j], 1))
        self.unpack_disps = nn.PixelShuffle(2)            # parameter-free members kept for state-dict / repr parity
        self.unpack_disp4 = nn.Upsample(scale_factor=2, mode='nearest')
        self.unpack_disp3 = nn.Upsample(scale_factor=2, mode='nearest')
        self.unpack_disp2 = nn.Upsample(scale_factor=2, mode='nearest')
        for i in (3, 2, 1, 0):
            setattr(self, 'disp%d_layer' % (i + 1), InvDepth(n[i], out_channels=out_channels))

    def _merge(self, up, skip, disp=None):
        parts = (up, skip) if self.version == 'A' else (up + skip,)        # Conv2D folds the concatenation into its K loop
        if disp is not None:
            parts = parts + (HF.upsample_nearest(disp, scale_factor=2),)
        return parts if len(parts) > 1 else parts[0]

    def forward(self, x5p, skips):
        skip1, skip2, skip3, skip4, skip5 = skips
        iconv5 = self.iconv5(self._merge(self.unpack5(x5p), skip5))
        iconv4 = self.iconv4(self._merge(self.unpack4(iconv5), skip4))
        d4 = self.disp4_layer(iconv4)
        iconv3 = self.iconv3(self._merge(self.unpack3(iconv4), skip3, d4))
        d3 = self.disp3_layer(iconv3)
        iconv2 = self.iconv2(self._merge(self.unpack2(iconv3), skip2, d3))
        d2 = self.disp2_layer(iconv2)
        iconv1 = self.iconv1(self._merge(self.unpack1(iconv2), skip1, d2))
        d1 = self.disp1_layer(iconv1)
        return [d1, d2, d3, d4] if self.training else [d1]


class PackNetSAN01(nn.Module):
    """
    dropout : float     dropout on the residual shortcuts (0.5 in configs/train_packnet_san_kitti.yaml)
    version : str       'XY', Y = 'A' (concatenated skips) or 'B' (added)
    """

    def __init__(self, dropout=None, version=None, **kwargs):
        super().__init__()
        if version is None or version[1:] not in ('A', 'B'):
            raise ValueError('Unknown PackNet version {}'.format(version))
        self.version = version[1:]
        ni, n1, n2, n3, n4, n5 = 32, 32, 64, 128, 256, 512
        self.encoder = Encoder(self.version, 3, ni, n1, n2, n3, n4, n5, [5, 3, 3, 3, 3], [2, 2, 3, 3], 4, dropout)
        self.decoder = Decoder(self.version, 1, ni, n1, n2, n3, n4, n5, [3] * 5, [3] * 5, 4)
        self.mconvs = MinkowskiEncoder([n1, n2, n3, n4, n5], with_uncertainty=False)
        self.weight = nn.Parameter(torch.ones(5), requires_grad=True)
        self.bias = nn.Parameter(torch.zeros(5), requires_grad=True)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()

    def run_network(self, rgb, input_depth=None):
        """-> (inverse depth maps: 4 scales in training, 1 otherwise; the five fused feature maps)."""
        x5p, skips = self.encoder(rgb)
        if input_depth is not None:
            self.mconvs.prep(input_depth)
            for i in range(1, 5):       # skips[0] is the full-resolution stem output: it has no sparse counterpart
                skips[i] = skips[i] * self.weight[i - 1].view(1, 1, 1, 1) + self.mconvs(skips[i]) + self.bias[i - 1].view(1, 1, 1, 1)
            x5p = x5p * self.weight[4].view(1, 1, 1, 1) + self.mconvs(x5p) + self.bias[4].view(1, 1, 1, 1)
        return self.decoder(x5p, skips), skips + [x5p]

    def forward(self, rgb, input_depth=None, **kwargs):
        if not self.training:
            inv_depths, _ = self.run_network(rgb, input_depth)
            return {'inv_depths': inv_depths}
        inv_depths_rgb, feat_rgb = self.run_network(rgb)
        if input_depth is None:
            return {'inv_depths': inv_depths_rgb}
        inv_depths_rgbd, feat_rgbd = self.run_network(rgb, input_depth)
        # the RGB-only features are pulled towards the (detached) RGB-D ones (reference :226-229)
        loss = sum(((srgbd.detach() - srgb) ** 2).mean() for srgbd, srgb in zip(feat_rgbd, feat_rgb)) / len(feat_rgbd)
        return {'inv_depths': inv_depths_rgb, 'inv_depths_rgbd': inv_depths_rgbd, 'depth_loss': loss}


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
