"""Depth-completion encoder of PackNet-SAN on MI355X kernels.

Drop-in for the reference's packnet_sfm/networks/layers/minkowski_encoder.py (`MinkConv2D`, `MinkowskiEncoder`: same
constructor arguments, same `prep(depth)` / `forward(x)` protocol, same parameter names -- `layer3.0.kernel`
[k*k, in, out] like ME.MinkowskiConvolution, `layer3.1.bn.*` like ME.MinkowskiBatchNorm -- so checkpoints map one to one).

The reference runs this branch on MinkowskiEngine (third-party, NOT vendored in the reference and not installed here; the
Dockerfile builds it from git master, i.e. un-versioned).  Its operations are restated from the MinkowskiEngine 0.5
documentation on the dense-plus-mask representation of minkowski.py:
  ME.MinkowskiConvolution(k, stride 1, dimension 2, no bias): out[p] = sum_o W[o] . in[p + o*ts] over the ACTIVE neighbours, for
      active p only                         == mask * conv2d(features with zeros at inactive sites): the MFMA conv kernel;
  ME.MinkowskiMaxPooling(3, stride 2): output cell active iff one of its 2x2 input cells is; value = max over the active
      inputs of the 3x3 window centred on the cell's origin;
  ME.MinkowskiBatchNorm: BatchNorm1d over the active sites of the whole batch (running statistics, affine);
  ME.MinkowskiReLU, sparse + sparse on identical coordinates: elementwise.
Kernel offset order assumed for `kernel[i]`: ME's hyper-cube region iterator, first coordinate fastest:
i = (dy + k//2) + k * (dx + k//2).  PARITY OF THIS BRANCH IS UNPINNED (no MinkowskiEngine to run, the reference has no test
or golden vector for it): tests check it against an independent gather-based restatement of the same rules (oracle/).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from packnet_sfm.hip import functional as HF
from packnet_sfm.networks.layers.minkowski import GridSparse, densify_features, map_add_features, sparsify_depth


class MinkowskiConvolution(nn.Module):
    """Parameter container + masked MFMA conv: `kernel` [k*k, in, out] (MinkowskiEngine's layout), no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dimension=2):
        super().__init__()
        if stride != 1 or dimension != 2:
            raise NotImplementedError('stride 1, dimension 2 (all the SAN branch uses)')
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.kernel = nn.Parameter(torch.empty(kernel_size * kernel_size, in_channels, out_channels))
        with torch.no_grad():               # ME default: ME.utils.kaiming_normal_(kernel, mode='fan_out', nonlinearity='relu')
            self.kernel.normal_(0, (2.0 / (out_channels * kernel_size * kernel_size)) ** 0.5)
        self._packed = HF.PackedConvWeight(volatile=True)

    def dense_weight(self):
        k = self.kernel_size
        # kernel[i], i = ky + k*kx  ->  [out, in, ky, kx]
        return self.kernel.view(k, k, self.in_channels, self.out_channels).permute(3, 2, 1, 0).contiguous()

    def forward(self, x):
        y = HF.conv2d(x.F, self.dense_weight(), None, self._packed)
        return GridSparse(y * x.mask, x.mask, x.tensor_stride)


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm over the ACTIVE sites of the batch; parameters under `.bn` like ME.MinkowskiBatchNorm."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x):
        bn, m, f = self.bn, x.mask, x.F
        if self.training or not bn.track_running_stats:
            n = m.sum().clamp(min=1.0)
            mean = (f * m).sum((0, 2, 3)) / n
            var = (((f - mean.view(1, -1, 1, 1)) ** 2) * m).sum((0, 2, 3)) / n
            if bn.track_running_stats:
                with torch.no_grad():
                    mom = bn.momentum
                    bn.running_mean.mul_(1 - mom).add_(mom * mean)
                    bn.running_var.mul_(1 - mom).add_(mom * var * n / (n - 1).clamp(min=1.0))     # unbiased, like BatchNorm1d
                    bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        y = (f - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + bn.eps)
        y = y * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
        return GridSparse(y * m, m, x.tensor_stride)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return GridSparse(F.relu(x.F), x.mask, x.tensor_stride)


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride, dimension=2):
        super().__init__()
        if (kernel_size, stride, dimension) != (3, 2, 2):
            raise NotImplementedError('MaxPooling(3, 2, dimension=2) (all the SAN branch uses)')

    def forward(self, x):
        neg = torch.finfo(x.F.dtype).min
        f = torch.where(x.mask > 0, x.F, torch.full_like(x.F, neg))
        pooled = F.max_pool2d(f, kernel_size=3, stride=2, padding=1)
        mask = F.max_pool2d(x.mask, kernel_size=2, stride=2)
        return GridSparse(torch.where(mask > 0, pooled, torch.zeros_like(pooled)), mask, x.tensor_stride * 2)


class MinkConv2D(nn.Module):
    """Three parallel sparse conv stacks (1, 2 and 3 convolutions deep) summed, BatchNorm + ReLU (reference :10-88)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, with_uncertainty=False, add_rgb=False):
        super().__init__()
        if with_uncertainty:
            raise NotImplementedError('with_uncertainty=True is not used by PackNetSAN01 (PackNetSAN01.py:177)')
        C, k = out_planes, kernel_size
        self.layer3 = nn.Sequential(MinkowskiConvolution(in_planes, C * 2, k), MinkowskiBatchNorm(C * 2), MinkowskiReLU(),
                                    MinkowskiConvolution(C * 2, C * 2, k), MinkowskiBatchNorm(C * 2), MinkowskiReLU(),
                                    MinkowskiConvolution(C * 2, C, k))
        self.layer2 = nn.Sequential(MinkowskiConvolution(in_planes, C * 2, k), MinkowskiBatchNorm(C * 2), MinkowskiReLU(),
                                    MinkowskiConvolution(C * 2, C, k))
        self.layer1 = nn.Sequential(MinkowskiConvolution(in_planes, C, k))
        self.layer_final = nn.Sequential(MinkowskiBatchNorm(C), MinkowskiReLU())
        self.pool = None if stride == 1 else MinkowskiMaxPooling(3, stride)
        self.add_rgb, self.with_uncertainty = add_rgb, with_uncertainty

    def forward(self, x):
        if self.pool is not None:
            x = self.pool(x)
        x1, x2, x3 = self.layer1(x), self.layer2(x), self.layer3(x)
        return None, self.layer_final(GridSparse(x1.F + x2.F + x3.F, x.mask, x.tensor_stride))


class MinkowskiEncoder(nn.Module):
    """Depth-completion encoder: one MinkConv2D (stride-2 pooling in front) per feature level (reference :91-131)."""

    def __init__(self, channels, with_uncertainty=False, add_rgb=False):
        super().__init__()
        kernel_sizes = [5, 5] + [3] * (len(channels) - 1)
        self.mconvs = nn.ModuleList([MinkConv2D(1, channels[0], kernel_sizes[0], 2, with_uncertainty=with_uncertainty)])
        for i in range(len(channels) - 1):
            self.mconvs.append(MinkConv2D(channels[i], channels[i + 1], kernel_sizes[i + 1], 2, with_uncertainty=with_uncertainty))
        self.d = self.n = self.shape = 0
        self.with_uncertainty, self.add_rgb = with_uncertainty, add_rgb

    def prep(self, d):
        self.d = sparsify_depth(d)
        self.shape = d.shape
        self.n = 0

    def forward(self, x=None):
        _, self.d = self.mconvs[self.n](self.d)
        self.n += 1
        out = densify_features(self.d, self.shape)
        if self.add_rgb:
            self.d = map_add_features(x, self.d)
        return out
