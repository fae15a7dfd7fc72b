"""PackNet01 (3D packing/unpacking encoder-decoder, CVPR'20) on hand-written MI355X kernels.

Drop-in for the reference's packnet_sfm/networks/depth/PackNet01.py: resolved by name through
`load_class('PackNet01', ['packnet_sfm.networks.depth'])`, constructed as `PackNet01(dropout=..., version='1A')`,
called as `net(rgb=...)`, returns {'inv_depths': [4 scales]} in training and {'inv_depths': tensor} in eval
(reference :178-185), and owns exactly the reference's 216 parameter tensors under the same names.
"""
import torch
import torch.nn as nn

from packnet_sfm.hip import functional as HF
from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, InvDepth, PackLayerConv3d, ResidualBlock, UnpackLayerConv3d


class PackNet01(nn.Module):
    """
    Parameters
    ----------
    dropout : float
        Dropout on the residual shortcuts (0/None disables it)
    version : str
        'XY': X unused, Y = 'A' (skip connections concatenated) or 'B' (added)
    """

    # stem width ni, encoder widths n1..n5, number of 3-D feature maps of the packing blocks (PackNet01.py:32-37);
    # PackNetSlim01 overrides them (PackNetSlim01.py:33-39)
    STEM_WIDTH = 64
    WIDTHS = (64, 64, 128, 256, 512)
    NUM_3D_FEAT = 8

    def __init__(self, dropout=None, version=None, **kwargs):
        super().__init__()
        if version is None or version[1:] not in ('A', 'B'):
            raise ValueError('Unknown PackNet version {}'.format(version))
        self.version = version[1:]
        concat = self.version == 'A'
        ni, no = self.STEM_WIDTH, 1             # stem width, inverse-depth channels
        n = list(self.WIDTHS)                   # encoder widths n1..n5
        d3 = self.NUM_3D_FEAT
        blocks = [2, 2, 3, 3]
        pack_k = [5, 3, 3, 3, 3]
        if concat:
            dec_out = list(n)
            dec_in = [n[0] + ni + no, n[1] + n[0] + no, n[2] + n[1] + no, n[3] + n[2], n[4] + n[3]]
        else:
            dec_out = [n[0], n[1], n[2] // 2, n[3] // 2, n[4] // 2]
            dec_in = [n[0] + no, n[1] + no, n[2] // 2 + no, n[3] // 2, n[4] // 2]

        # registration order follows the reference so that identical seeds give identical initial weights
        self.pre_calc = Conv2D(3, ni, 5, 1)
        for i in range(5):
            setattr(self, 'pack%d' % (i + 1), PackLayerConv3d(n[i], pack_k[i], d=d3))
        self.conv1 = Conv2D(ni, n[0], 7, 1)
        for i in range(4):
            setattr(self, 'conv%d' % (i + 2), ResidualBlock(n[i], n[i + 1], blocks[i], 1, dropout=dropout))
        unpack_in = [n[1], n[2], n[3], n[4], n[4]]   # inputs of unpack1..unpack5
        for i in (4, 3, 2, 1, 0):
            setattr(self, 'unpack%d' % (i + 1), UnpackLayerConv3d(unpack_in[i], dec_out[i], 3, d=d3))
        for i in (4, 3, 2, 1, 0):
            setattr(self, 'iconv%d' % (i + 1), Conv2D(dec_in[i], n[i], 3, 1))
        for i in (3, 2, 1, 0):
            setattr(self, 'disp%d_layer' % (i + 1), InvDepth(n[i], out_channels=no))
        self.init_weights()

    def init_weights(self):
        """Xavier-uniform conv weights, zero conv biases (GroupNorm keeps (1, 0))."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()

    def _merge(self, up, skip, disp=None):
        """Input of an iconv block: cat(up, skip[, upsampled inverse depth]) (version 'A') or up + skip[, ...] ('B') -- returned as
        the TUPLE of its parts: Conv2D folds the concatenation into its K loop (reference :138-174 materialises it)."""
        parts = (up, skip) if self.version == 'A' else (up + skip,)
        if disp is not None:
            parts = parts + (HF.upsample_nearest(disp, scale_factor=2),)
        return parts if len(parts) > 1 else parts[0]

    def forward(self, rgb):
        """Inverse depth maps: list of 4 scales (training) or the full-resolution map (eval)."""
        # (round 5) every tensor with a second consumer -- the five skip connections and the three decoder features that also feed an
        # InvDepth head -- is read by that consumer through the gradient tap of the convolution that reads it first
        # (layers01._HipConv2d.forward_tap): its two gradients meet inside a backward-data launch, not in an elementwise sum
        x = self.pre_calc(rgb)
        c1, x_s = self.conv1.forward_tap(x)
        x1p = self.pack1(c1)
        c2, x1p_s = self.conv2.forward_tap(x1p)
        x2p = self.pack2(c2)
        c3, x2p_s = self.conv3.forward_tap(x2p)
        x3p = self.pack3(c3)
        c4, x3p_s = self.conv4.forward_tap(x3p)
        x4p = self.pack4(c4)
        c5, x4p_s = self.conv5.forward_tap(x4p)
        x5p = self.pack5(c5)

        iconv5 = self.iconv5(self._merge(self.unpack5(x5p), x4p_s))
        iconv4 = self.iconv4(self._merge(self.unpack4(iconv5), x3p_s))
        up3, iconv4_t = self.unpack3.forward_tap(iconv4)
        disp4 = self.disp4_layer(iconv4_t)
        iconv3 = self.iconv3(self._merge(up3, x2p_s, disp4))
        up2, iconv3_t = self.unpack2.forward_tap(iconv3)
        disp3 = self.disp3_layer(iconv3_t)
        iconv2 = self.iconv2(self._merge(up2, x1p_s, disp3))
        up1, iconv2_t = self.unpack1.forward_tap(iconv2)
        disp2 = self.disp2_layer(iconv2_t)
        iconv1 = self.iconv1(self._merge(up1, x_s, disp2))
        disp1 = self.disp1_layer(iconv1)

        if self.training:
            return {'inv_depths': [disp1, disp2, disp3, disp4]}
        return {'inv_depths': disp1}


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
