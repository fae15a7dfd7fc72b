"""PackNetSlim01: PackNet01 with a 32-channel stem / first stage and 4 (instead of 8) 3-D feature maps in every packing
and unpacking block, on the same MI355X kernels (the Conv3d stencils are built for 4 and 8 feature maps).

Drop-in for the reference's packnet_sfm/networks/depth/PackNetSlim01.py (`ni, n1 = 32`, `num_3d_feat = 4`, :33-39):
same class name, constructor `PackNetSlim01(dropout=..., version='1A')`, `net(rgb=...)` contract and parameter names.
"""
from packnet_sfm.networks.depth.PackNet01 import PackNet01


class PackNetSlim01(PackNet01):
    STEM_WIDTH = 32
    WIDTHS = (32, 64, 128, 256, 512)
    NUM_3D_FEAT = 4


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
