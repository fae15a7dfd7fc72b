"""Depth-completion encoder of PackNet-SAN on MI355X kernels.

Drop-in for the reference's packnet_sfm/networks/layers/minkowski_encoder.py (`MinkConv2D`, `MinkowskiEncoder`: same
constructor arguments, same `prep(depth)` / `forward(x)` protocol, same parameter names -- `layer3.0.kernel`
[k*k, in, out] like ME.MinkowskiConvolution, `layer3.1.bn.*` like ME.MinkowskiBatchNorm -- so checkpoint KEYS and shapes map one to
one; whether the VALUES mean the same depends on the tap order below, which is unpinned: loading warns).

The reference runs this branch on MinkowskiEngine (third-party, NOT vendored in the reference and not installed here; the
Dockerfile builds it from git master, i.e. un-versioned).  Its operations are restated from the MinkowskiEngine 0.5
documentation on the active-site lists of minkowski.py (`SparseGrid`), each one a kernel of csrc/sparse.hip:
  ME.MinkowskiConvolution(k, stride 1, dimension 2, no bias): out[p] = sum_o W[o]^T . in[p + o*ts] over the ACTIVE neighbours, for
      active p only  -> gather-based implicit GEMM on the matrix cores (exact fp32), its backward-data and weight gradient;
  ME.MinkowskiMaxPooling(3, stride 2): output cell active iff one of its 2x2 input cells is; value = max over the active
      inputs of the 3x3 window centred on the cell's origin;
  ME.MinkowskiBatchNorm: BatchNorm1d over the feature rows of the active sites of the whole batch (running statistics, affine);
  ME.MinkowskiReLU, sparse + sparse on identical coordinates: elementwise on the rows.
Kernel offset order assumed for `kernel[i]`: ME's hyper-cube region iterator, first coordinate fastest:
i = (dy + k//2) + k * (dx + k//2).  PARITY AGAINST MinkowskiEngine ITSELF IS UNPINNED (it cannot be run here and the reference has no
test or golden vector for the branch); the rules above are pinned by hand-computed vectors (tests/test_sparse.py) on
oracle/minkowski_oracle.py, an independent coordinate-dictionary restatement, and the kernels are checked against that oracle.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from packnet_sfm.hip import functional as HF
from packnet_sfm.networks.layers.minkowski import (SparseGrid, densify_features, map_add_features, pool_coordinates,
                                                    sparsify_depth)


_WARNED = [False]


def _warn_unpinned_checkpoint(prefix):
    """a reference PackNet-SAN checkpoint loads into these modules without error, but the tap order of `kernel[i]` and
    the pooling window rule were never checked against MinkowskiEngine itself (it cannot be run here) -- say so, loudly, once."""
    if not _WARNED[0]:
        _WARNED[0] = True
        import warnings
        warnings.warn('packnet_sfm (MI355X): loading MinkowskiEngine-layout weights into %s*: the kernel tap order (i = (dy + k//2) + k * '
                      '(dx + k//2)) and the MaxPooling(3, 2) window rule are restated from the MinkowskiEngine documentation and pinned '
                      'by hand-computed vectors only -- parity against MinkowskiEngine itself is UNPINNED.  Validate the sparse branch '
                      'of a checkpoint trained with the reference before relying on it.' % prefix, RuntimeWarning, stacklevel=3)


class MinkowskiConvolution(nn.Module):
    """`kernel` [k*k, in, out] (MinkowskiEngine's layout), no bias; the sparse convolution kernel of csrc/sparse.hip."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dimension=2):
        super().__init__()
        if stride != 1 or dimension != 2:
            raise NotImplementedError('stride 1, dimension 2 (all the SAN branch uses)')
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.kernel = nn.Parameter(torch.empty(kernel_size * kernel_size, in_channels, out_channels))
        with torch.no_grad():               # ME default: ME.utils.kaiming_normal_(kernel, mode='fan_out', nonlinearity='relu')
            self.kernel.normal_(0, (2.0 / (out_channels * kernel_size * kernel_size)) ** 0.5)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix + 'kernel' in state_dict:
            _warn_unpinned_checkpoint(prefix)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x):
        return x.with_features(HF.sparse_conv(x.F, self.kernel, x.neighbors(self.kernel_size), x.count, self.kernel_size))


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm over the feature rows of the ACTIVE sites of the batch; parameters under `.bn` like ME.MinkowskiBatchNorm."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x):
        bn, f, rows = self.bn, x.F, x.row_mask
        if self.training or not bn.track_running_stats:
            n = x.count.to(torch.float32).clamp(min=1.0)            # device scalar: no host round trip
            mean = f.sum(0) / n                                      # rows past count are zero
            var = (((f - mean) ** 2) * rows).sum(0) / n
            if bn.track_running_stats:
                with torch.no_grad():
                    mom = bn.momentum
                    bn.running_mean.mul_(1 - mom).add_(mom * mean)
                    bn.running_var.mul_(1 - mom).add_(mom * var * n / (n - 1).clamp(min=1.0))     # unbiased, like BatchNorm1d
                    bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        y = ((f - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias) * rows
        return x.with_features(y)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x.with_features(F.relu(x.F))


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride, dimension=2):
        super().__init__()
        if (kernel_size, stride, dimension) != (3, 2, 2):
            raise NotImplementedError('MaxPooling(3, 2, dimension=2) (all the SAN branch uses)')

    def forward(self, x):
        out = pool_coordinates(x)
        out.F = HF.sparse_maxpool(x.F, x.imap, out.sites, out.count, out.cap, x.h, x.w)
        return out


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
        return None, self.layer_final(x.with_features(x1.F + x2.F + x3.F))


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
