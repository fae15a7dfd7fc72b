"""Sparse <-> dense helpers of the depth-completion branch (API of the reference's packnet_sfm/networks/layers/minkowski.py,
which builds MinkowskiEngine SparseTensors).

MI355X design (csrc/sparse.hip): a sparse tensor of this branch lives on the regular pixel grid of a feature level.  It is kept
as the ascending list of its ACTIVE cells plus one contiguous feature row per site (`SparseGrid`); every operation of the branch
-- convolution, stride-2 max pooling, densify, the dense-feature pick-up -- is a HIP kernel over that list, so the cost follows
the ~5 % LiDAR occupancy instead of the image size (round 2 ran the branch dense-plus-mask on the MFMA conv kernels: 11 ms of a
28.6 ms PackNetSAN01 forward + backward).  The coordinate map is built on the device (three-pass compaction, neighbour tables);
the only host round trip is the site count of the input depth map, which sizes every buffer of the branch (the reference's
MinkowskiEngine manages its coordinate hash maps on the host for every tensor).
"""
import torch

from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops


class SparseGrid:
    """Active cells of a [B, h, w] grid: `sites` (int32 [cap], ascending linear cell index), `imap` (int32 [B*h*w], row or -1),
    `count` (int32 [1], on the device), feature rows `F` [cap, C] (zero past count) and `tensor_stride` (input pixels per cell) --
    what a MinkowskiEngine SparseTensor with 2-D coordinates on this grid holds."""

    def __init__(self, B, h, w, tensor_stride, sites, imap, count, feats, shared=None):
        self.B, self.h, self.w, self.tensor_stride = B, h, w, tensor_stride
        self.sites, self.imap, self.count, self.F = sites, imap, count, feats
        self._shared = shared if shared is not None else {}      # neighbour tables / row mask, shared by tensors on the same coordinates

    @property
    def cap(self):
        return self.sites.shape[0]

    def with_features(self, feats):
        return SparseGrid(self.B, self.h, self.w, self.tensor_stride, self.sites, self.imap, self.count, feats, self._shared)

    def neighbors(self, ks):
        key = ('nbr', ks)
        if key not in self._shared:
            self._shared[key] = ops.sparse_neighbors(self.imap, self.sites, self.count, self.cap, self.h, self.w, ks)
        return self._shared[key]

    @property
    def row_mask(self):
        """[cap, 1] float: 1 for the rows that hold a site."""
        if 'rows' not in self._shared:
            self._shared['rows'] = (torch.arange(self.cap, device=self.sites.device, dtype=torch.int32) < self.count).to(torch.float32).unsqueeze(1)
        return self._shared['rows']

    @property
    def num_active(self):
        return int(self.count.item())

    @property
    def mask(self):
        """Dense occupancy [B, 1, h, w] (tests / debugging)."""
        return (self.imap >= 0).to(torch.float32).view(self.B, 1, self.h, self.w)


def _round_cap(n):
    return max(32, (int(n) + 31) // 32 * 32)


def sparsify_depth(x):
    """[B,1,H,W] depth map -> SparseGrid holding the range values of the valid (> 0) pixels (reference :33-57)."""
    B, _, H, W = x.shape
    x = x.contiguous()
    imap, sites, count = ops.sparse_compact(x)
    cap = _round_cap(min(int(count.item()), B * H * W))       # the ONE host round trip of the branch: sizes every level's buffers
    if cap > sites.numel():                                    # (tiny grids: the 32-row granule exceeds the cell count)
        sites = torch.cat([sites, sites.new_zeros(cap - sites.numel())])
    sites = sites[:cap].contiguous()
    feats = ops.sparse_gather(x.view(B, 1, H * W), sites, count, cap)
    return SparseGrid(B, H, W, 1, sites, imap, count, feats)


def sparsify_features(x):
    """Dense feature map as a fully occupied SparseGrid (reference :8-30)."""
    B, C, H, W = x.shape
    ones = torch.ones((B * H * W,), dtype=torch.float32, device=x.device)
    imap, sites, count = ops.sparse_compact(ones)
    cap = _round_cap(B * H * W)
    if cap > sites.numel():
        sites = torch.cat([sites, sites.new_zeros(cap - sites.numel())])
    return SparseGrid(B, H, W, 1, sites, imap, count, HF.sparse_gather(x, imap, sites, count, cap))


def pool_coordinates(x):
    """Coordinates of ME.MinkowskiMaxPooling(3, 2) on `x`: the cells floor(c / 2) of the twice coarser grid (no features yet)."""
    mask = ops.sparse_pool_cells(x.imap, x.B, x.h, x.w)
    h2, w2 = (x.h + 1) // 2, (x.w + 1) // 2                      # odd grids: ceil, as floor(c / 2) of the last row / column needs
    cap = min(x.cap, _round_cap(x.B * h2 * w2))                # a coarse cell has at least one fine site: never more rows than before
    imap, sites, count = ops.sparse_compact(mask, cap=cap)
    return SparseGrid(x.B, h2, w2, x.tensor_stride * 2, sites, imap, count, None)


def densify_features(x, shape):
    """SparseGrid -> dense [B,C,H/stride,W/stride], zeros where nothing is stored (reference :60-83)."""
    B, _, H, W = shape
    s = x.tensor_stride
    assert (x.h, x.w) == (-(-H // s), -(-W // s)), ((x.h, x.w), shape, s)
    return HF.sparse_densify(x.F, x.imap, x.sites, x.count, x.B, x.h, x.w)


def map_add_features(x, s):
    """Add the dense features `x` to the sparse ones at the active sites (reference :116-136)."""
    return s.with_features(s.F + HF.sparse_gather(x, s.imap, s.sites, s.count, s.cap))
