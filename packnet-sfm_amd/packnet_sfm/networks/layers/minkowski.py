"""Sparse <-> dense helpers of the depth-completion branch (API of the reference's packnet_sfm/networks/layers/minkowski.py,
which builds MinkowskiEngine SparseTensors).

MI355X design: the sparse tensors of this branch live on the regular pixel grid of a feature level, so they are kept as a
DENSE feature map plus an occupancy mask (`GridSparse`): every MinkowskiEngine operation the branch uses then becomes a
masked dense operation that runs on the existing MFMA / streaming kernels (see minkowski_encoder.py) -- no coordinate hash
maps, no gather / scatter kernel maps, regular tiles.  LiDAR occupancy is ~5 % at full resolution but every stride-2 pooling
level roughly triples it, so from the third level on the dense form is also the cheaper one.
"""
import torch


class GridSparse:
    """features [B,C,h,w] (zero at inactive sites), mask [B,1,h,w] in {0,1}, tensor_stride (pixels of the input image per
    cell) -- what a MinkowskiEngine SparseTensor with 2-D coordinates on this grid represents."""

    def __init__(self, features, mask, tensor_stride=1):
        self.F, self.mask, self.tensor_stride = features, mask, tensor_stride

    @property
    def num_active(self):
        return int(self.mask.sum())


def sparsify_depth(x):
    """[B,1,H,W] depth map -> GridSparse holding the range values of the valid (> 0) pixels (reference :33-57)."""
    mask = (x > 0).to(x.dtype)
    return GridSparse(x * mask, mask, 1)


def sparsify_features(x):
    """Dense feature map as a fully occupied GridSparse (reference :8-30)."""
    return GridSparse(x, torch.ones_like(x[:, :1]), 1)


def densify_features(x, shape):
    """GridSparse -> dense [B,C,H/stride,W/stride], zeros where nothing is stored (reference :60-83)."""
    B, _, H, W = shape
    s = x.tensor_stride
    assert tuple(x.F.shape[2:]) == (H // s, W // s), (x.F.shape, shape, s)
    return x.F * x.mask


def map_add_features(x, s):
    """Add the dense features `x` to the sparse ones at the active sites (reference :116-136)."""
    return GridSparse((s.F + x) * s.mask, s.mask, s.tensor_stride)
