"""PackNet building blocks, MI355X edition.

Same class names, constructor signatures and parameter (state-dict) names as the reference's
packnet_sfm/networks/layers/packnet/layers01.py, so `PackNet01`, checkpoints and the dynamic loader keep working;
but every `forward` is a sequence of hand-written gfx950 kernels (packnet_sfm.hip.functional) instead of ATen ops:

    Conv2D            -> exact-fp32 MFMA implicit-GEMM conv  +  fused GroupNorm(16)+ELU          (ref :10-37)
    ResidualConv      -> 3 convs + GroupNorm/ELU with the residual add folded into the norm      (ref :40-72)
    InvDepth          -> conv + fused sigmoid/min_depth                                          (ref :98-122)
    PackLayerConv3d   -> space_to_depth, Conv3d(1->8) stencil, conv, GroupNorm+ELU               (ref :213-247)
    UnpackLayerConv3d -> conv, GroupNorm+ELU, Conv3d(1->8) stencil, depth_to_space               (ref :250-286)

torch.nn.Conv2d / Conv3d / GroupNorm objects are kept purely as *parameter containers* (identical names, shapes
and default initialisation order as the reference); their ATen forward is never called.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops as _ops


class _HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters, HIP forward (stride 1, 'same' zero padding k//2)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        if stride != 1:
            raise NotImplementedError('the gfx950 conv kernel implements stride 1 only (all PackNet01 convs)')
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride)
        self._packed = HF.PackedConvWeight()

    def forward(self, x):
        if isinstance(x, (tuple, list)):       # channel concatenation folded into the convolution (hip.functional.conv2d_cat)
            return HF.conv2d_cat(tuple(x), self.weight, self.bias, self._packed)
        return HF.conv2d(x, self.weight, self.bias, self._packed)

    def forward_tap(self, x):
        """(conv(x), x_tap): consumers of x other than this convolution read x_tap, and their gradient is added inside this
        convolution's backward-data launch instead of by an elementwise pass (hip.functional.conv2d_tap)."""
        return HF.conv2d_tap(x, self.weight, self.bias, self._packed)


class Conv2D(nn.Module):
    """2D convolution (zero 'same' padding) + GroupNorm(16) + ELU.  `x` may be a tuple of tensors standing for their channel
    concatenation (the decoder's skip connections): the concatenated tensor is then never materialised."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv_base = _HipConv2d(in_channels, out_channels, kernel_size, stride)
        self.normalize = nn.GroupNorm(16, out_channels)

    def forward(self, x):
        base, norm = self.conv_base, self.normalize
        # one autograd node whose forward and backward bodies are single calls into the block sequencer (HF.ConvGnActFn)
        return HF.conv2d_gn_act(x, base.weight, base.bias, norm.weight, norm.bias, base._packed, 16, norm.eps, _ops.ACT_ELU)

    def forward_tap(self, x):
        """(block(x), x_tap) for a single-tensor x -- see _HipConv2d.forward_tap."""
        base, norm = self.conv_base, self.normalize
        return HF.conv2d_gn_act_tap(x, base.weight, base.bias, norm.weight, norm.bias, base._packed, 16, norm.eps, _ops.ACT_ELU)


class ResidualConv(nn.Module):
    """Two Conv2D blocks plus a 1x1 shortcut; ELU(GroupNorm(main + shortcut))."""

    def __init__(self, in_channels, out_channels, stride, dropout=None):
        super().__init__()
        self.conv1 = Conv2D(in_channels, out_channels, 3, stride)
        self.conv2 = Conv2D(out_channels, out_channels, 3, 1)
        self.conv3 = _HipConv2d(in_channels, out_channels, 1, stride)
        self.normalize = nn.GroupNorm(16, out_channels)
        if dropout:
            # as the reference (layers01.py:64-65): the shortcut becomes Sequential(conv3, Dropout2d), so the state-dict
            # keys are conv3.0.weight / conv3.0.bias exactly like checkpoints trained with dropout
            self.conv3 = nn.Sequential(self.conv3, nn.Dropout2d(dropout))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # accept either key layout (conv3.* <-> conv3.0.*) so that a checkpoint trained with dropout loads into a model
        # built without it and vice versa
        has_seq = isinstance(self.conv3, nn.Sequential)
        for name in ('weight', 'bias'):
            plain, seq = prefix + 'conv3.' + name, prefix + 'conv3.0.' + name
            if has_seq and plain in state_dict and seq not in state_dict:
                state_dict[seq] = state_dict.pop(plain)
            elif not has_seq and seq in state_dict and plain not in state_dict:
                state_dict[plain] = state_dict.pop(seq)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x, tap=False):
        """tap=True: (out, x_tap) -- x_tap is what a further consumer of x (a decoder skip) reads, see _HipConv2d.forward_tap."""
        x_tap = x
        if HF.grad_taps() and isinstance(self.conv3, _HipConv2d):
            # x feeds conv1 and the shortcut (and, through x_tap, maybe a skip connection): chained taps -- conv1's backward-data adds
            # the skip's gradient, the shortcut's backward-data adds conv1's: no elementwise gradient sums (round 5)
            shortcut, xa = self.conv3.forward_tap(x)
            y1, x_tap = self.conv1.forward_tap(xa)
            main = self.conv2(y1)
        else:
            main = self.conv2(self.conv1(x))
            shortcut = self.conv3(x)
        out = HF.groupnorm_act(main, self.normalize.weight, self.normalize.bias, 16, self.normalize.eps, _ops.ACT_ELU,
                               res=shortcut)
        return (out, x_tap) if tap else out


class _ResidualSequence(nn.Sequential):
    """nn.Sequential of ResidualConv layers (same state-dict keys) that can hand out the gradient tap of its input."""

    def forward_tap(self, x):
        it = iter(self)
        y, x_tap = next(it)(x, tap=True)
        for m in it:
            y = m(y)
        return y, x_tap


def ResidualBlock(in_channels, out_channels, num_blocks, stride, dropout=None):
    """`num_blocks` ResidualConv layers in sequence."""
    blocks = [ResidualConv(in_channels if i == 0 else out_channels, out_channels, stride if i == 0 else 1, dropout=dropout)
              for i in range(num_blocks)]
    return _ResidualSequence(*blocks)


class InvDepth(nn.Module):
    """3x3 conv to `out_channels` followed by sigmoid(x) / min_depth."""

    def __init__(self, in_channels, out_channels=1, min_depth=0.5):
        super().__init__()
        self.min_depth = min_depth
        self.conv1 = _HipConv2d(in_channels, out_channels, 3, 1)

    def forward(self, x):
        c = self.conv1
        if c.out_channels == 1 and c.bias is not None:
            # one output channel: fused streaming kernel (csrc/invdepth.hip) instead of a 3 %-occupied MFMA tile
            return HF.invdepth_conv(x, c.weight, c.bias, self.min_depth)
        return HF.invdepth_act(c(x), self.min_depth)


def packing(x, r=2):
    """[B,C,H,W] -> [B,4C,H/2,W/2] space-to-depth (inverse of PixelShuffle(2))."""
    if r != 2:
        raise NotImplementedError('the gfx950 packing kernel implements r=2 only')
    return HF.space_to_depth(x)


class PackLayerConv2d(nn.Module):
    """Packing followed by a Conv2D back to `in_channels`."""

    def __init__(self, in_channels, kernel_size, r=2):
        super().__init__()
        self.r = r
        self.conv = Conv2D(in_channels * (r ** 2), in_channels, kernel_size, 1)

    def forward(self, x):
        return self.conv(packing(x, self.r))


class UnpackLayerConv2d(nn.Module):
    """Conv2D to r^2 * out_channels followed by depth-to-space."""

    def __init__(self, in_channels, out_channels, kernel_size, r=2):
        super().__init__()
        if r != 2:
            raise NotImplementedError('r=2 only')
        self.conv = Conv2D(in_channels, out_channels * (r ** 2), kernel_size, 1)

    def forward(self, x):
        return HF.depth_to_space(self.conv(x))


class PackLayerConv3d(nn.Module):
    """3D packing: space-to-depth, Conv3d(1 -> d) over (channel, y, x), then Conv2D back to `in_channels`.

    MI355X design: the reference materialises the 8x-expanded Conv3d output (pack1: 252 MB per image) and convolves it
    with a [C, 32C, k, k] kernel.  There is no non-linearity between the Conv3d and that Conv2d, so away from the image
    border the pair is ONE (k+2)x(k+2) convolution over the 4C packed channels with a composed kernel
    (functional.ComposePackWeightFn): 4.1x fewer flops for pack1, and the 8x tensor never exists.  Within k//2 pixels of
    the border the two differ (the reference zero-pads the Conv3d *output*), so that frame is computed with the
    original formula on four thin strips and stitched in -- the result equals the reference's up to fp32 summation order.
    `collapse`: 'auto' (use the composed kernel when it saves >= 30 % of the flops), True, or False.
    """

    collapse = 'auto'
    # round 6: the column strips of the collapsed form live TRANSPOSED ([2B, C, S, h]: rows of h pixels), with the (y, x) taps of both
    # kernels swapped -- conv(x^T, w^T) = conv(x, w)^T -- so that their convolutions and weight gradients run on a map as wide as the
    # row strips' instead of on rows of 2 - 5 pixels (the weight gradient of those fell back to the generic f32 kernel: 28 - 83 TFLOP/s)
    # (+0.5 % images/s, weight-gradient frac 0.366 -> 0.380 same-box: profiles/r06_ab_transposed_column_strips.txt; False = the old form)
    lr_transposed = True

    def __init__(self, in_channels, kernel_size, r=2, d=8):
        super().__init__()
        if r != 2 or d not in (4, 8):
            raise NotImplementedError('the gfx950 packing kernels implement r=2 and d in {4, 8} (PackNet01 / PackNetSlim01)')
        self.d = d
        self.conv = Conv2D(in_channels * (r ** 2) * d, in_channels, kernel_size, 1)
        self.conv3d = nn.Conv3d(1, d, kernel_size=(3, 3, 3), stride=(1, 1, 1), padding=(1, 1, 1))
        self._eff_packed = HF.PackedConvWeight(volatile=True)
        self._lr_packed = HF.PackedConvWeight()        # packed copies of the (y, x)-transposed Conv2d weight (column strips)

    def _use_collapsed(self, h, w):
        k = self.conv.kernel_size
        r = k // 2
        if h < 2 * r + 1 or w < 2 * r + 1 or self.collapse is False:
            return False
        if self.collapse is True:
            return True
        orig = float(self.d) * k * k * h * w
        strips = float(self.d) * k * k * (2 * (2 * r) * w + 2 * h * (2 * r))
        return (k + 2) ** 2 * h * w + strips <= 0.7 * orig

    def _conv_reference_form(self, P):
        """conv_base(Conv3d(P)) exactly as the reference composes it (materialises the 8x tensor)."""
        feats = HF.conv3d_1to8(P, self.conv3d.weight, self.conv3d.bias)
        return self.conv.conv_base(feats)

    def _conv_collapsed(self, P):
        base = self.conv.conv_base
        W2, b2, W3, b3 = base.weight, base.bias, self.conv3d.weight, self.conv3d.bias
        k = self.conv.kernel_size
        r, S = k // 2, 2 * (k // 2) + 1
        B, _, h, w = P.shape
        # interior: one (k+2)x(k+2) conv with the composed kernel; its bias is b2 + sum over ALL taps of W2 * b3
        W_eff, bias_eff = HF.compose_pack_params(W2, b2, W3, b3)
        # border frame (r pixels): original formula on strips of 2r+1 packed rows / columns (top+bottom and left+right
        # are batched together); only rows/cols whose Conv3d neighbourhood lies inside the strip are kept.  The three
        # helper Functions do the strip gather / select / paste without full-size zero-fills and adds in backward.
        lr_t = bool(self.lr_transposed)
        P_main, tb, lr = HF.pack_border_split(P, S, lr_t)
        y = HF.conv2d(P_main, W_eff, bias_eff, self._eff_packed)
        o_tb = base(HF.strip_select(HF.conv3d_1to8(tb, W3, b3), B, r, 2))        # [2B, C, 2r, w]
        if lr_t:
            # transposed space: strips [2B, C, S, h], Conv3d taps (dz, dx, dy), Conv2d taps (kx, ky); the result [2B, C, 2r, h] is pasted
            # through a transposed view (autograd carries the weight gradients back through the two transpose views)
            z = HF.conv3d_1to8(lr, W3.transpose(3, 4), b3)
            o_lr = HF.conv2d(HF.strip_select(z, B, r, 2), W2.transpose(2, 3), b2, self._lr_packed)
        else:
            o_lr = base(HF.strip_select(HF.conv3d_1to8(lr, W3, b3), B, r, 3))    # [2B, C, h, 2r]
        return HF.pack_border_paste(y, o_tb, o_lr, r, lr_t)

    def forward(self, x):
        P = HF.space_to_depth(x)
        y = self._conv_collapsed(P) if self._use_collapsed(P.shape[2], P.shape[3]) else self._conv_reference_form(P)
        norm = self.conv.normalize
        return HF.groupnorm_act(y, norm.weight, norm.bias, 16, norm.eps, _ops.ACT_ELU)


class UnpackLayerConv3d(nn.Module):
    """3D unpacking: Conv2D to out*r^2/d channels, Conv3d(1 -> d), then depth-to-space."""

    def __init__(self, in_channels, out_channels, kernel_size, r=2, d=8):
        super().__init__()
        if r != 2 or d not in (4, 8):
            raise NotImplementedError('the gfx950 unpacking kernels implement r=2 and d in {4, 8} (PackNet01 / PackNetSlim01)')
        self.conv = Conv2D(in_channels, out_channels * (r ** 2) // d, kernel_size, 1)
        self.conv3d = nn.Conv3d(1, d, kernel_size=(3, 3, 3), stride=(1, 1, 1), padding=(1, 1, 1))

    def forward(self, x):
        feats = HF.conv3d_1to8(self.conv(x), self.conv3d.weight, self.conv3d.bias)
        return HF.depth_to_space(feats)

    def forward_tap(self, x):
        """(unpack(x), x_tap) -- see _HipConv2d.forward_tap (the decoder feature also feeds an InvDepth head)."""
        y, x_tap = self.conv.forward_tap(x)
        return HF.depth_to_space(HF.conv3d_1to8(y, self.conv3d.weight, self.conv3d.bias)), x_tap


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
