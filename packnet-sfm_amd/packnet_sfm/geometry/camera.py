"""Pinhole camera with the public surface of the reference's packnet_sfm/geometry/camera.py (K, Tcw, fx/fy/cx/cy, Twc,
Kinv, scaled, reconstruct, project).

The training step never goes through reconstruct()/project(): MultiViewPhotometricLoss hands intrinsics and poses to the
fused view-synthesis kernel (csrc/loss.hip), which does lift -> rigid transform -> project -> gather per pixel in
registers.  These two methods serve evaluation / visualisation code; they are written as per-pixel affine maps
(broadcast multiply-adds on the u, v pixel coordinates) rather than batched matrix products over a materialised
homogeneous grid."""
import torch
import torch.nn as nn

from packnet_sfm.geometry.camera_utils import scale_intrinsics
from packnet_sfm.geometry.pose import Pose

_FRAMES = ('c', 'w')


def _pixel_axes(H, W, like):
    """u = 0..W-1 as [1,1,1,W] and v = 0..H-1 as [1,1,H,1] (un-normalised pixel centres, utils/image.py:249-250)."""
    u = torch.arange(W, device=like.device, dtype=like.dtype).view(1, 1, 1, W)
    v = torch.arange(H, device=like.device, dtype=like.dtype).view(1, 1, H, 1)
    return u, v


class Camera(nn.Module):
    def __init__(self, K, Tcw=None):
        super().__init__()
        self.K = K
        self.Tcw = Tcw if Tcw is not None else Pose.identity(len(K))

    def __len__(self):
        return self.K.shape[0]

    def to(self, *args, **kwargs):
        self.K, self.Tcw = self.K.to(*args, **kwargs), self.Tcw.to(*args, **kwargs)
        return self

    # ---- intrinsics --------------------------------------------------------------------------------------------------
    def _entry(self, r, c):
        return self.K[:, r, c]

    fx = property(lambda self: self._entry(0, 0))
    fy = property(lambda self: self._entry(1, 1))
    cx = property(lambda self: self._entry(0, 2))
    cy = property(lambda self: self._entry(1, 2))

    @property
    def Kinv(self):
        """Inverse intrinsics the way the reference forms them (camera.py:72-80): a copy of K whose focal and principal
        entries are replaced by 1/f and -c/f; every other entry (zero skew, the bottom row) is carried over."""
        inv = self.K.clone()
        for axis in (0, 1):
            f, c = self.K[:, axis, axis], self.K[:, axis, 2]
            inv[:, axis, axis] = f.reciprocal()
            inv[:, axis, 2] = -1. * c / f
        return inv

    @property
    def Twc(self):
        return self.Tcw.inverse()

    def scaled(self, x_scale, y_scale=None):
        y_scale = x_scale if y_scale is None else y_scale
        if (x_scale, y_scale) == (1., 1.):
            return self
        return Camera(scale_intrinsics(self.K.clone(), x_scale, y_scale), Tcw=self.Tcw)

    # ---- lifting / projection ----------------------------------------------------------------------------------------
    def reconstruct(self, depth, frame='w'):
        """[B,1,H,W] depth -> [B,3,H,W] points, X = (Kinv [u, v, 1]^T) * depth, in the camera or world frame."""
        if frame not in _FRAMES:
            raise ValueError('Unknown reference frame {}'.format(frame))
        B, C, H, W = depth.shape
        assert C == 1
        u, v = _pixel_axes(H, W, depth)
        ki = self.Kinv.to(depth.dtype).view(B, 3, 3, 1, 1)
        rays = ki[:, :, 0] * u + ki[:, :, 1] * v + ki[:, :, 2]          # [B,3,H,W]
        Xc = rays * depth
        return Xc if frame == 'c' else self.Twc @ Xc

    def project(self, X, frame='w'):
        """[B,3,H,W] points -> [B,H,W,2] sampling coordinates in [-1, 1] (align_corners=True convention)."""
        if frame not in _FRAMES:
            raise ValueError('Unknown reference frame {}'.format(frame))
        B, C, H, W = X.shape
        assert C == 3
        Xc = X if frame == 'c' else self.Tcw @ X
        k = self.K.to(X.dtype).view(B, 3, 3, 1, 1)
        p = k[:, :, 0] * Xc[:, 0:1] + k[:, :, 1] * Xc[:, 1:2] + k[:, :, 2] * Xc[:, 2:3]      # K @ Xc per pixel
        z = p[:, 2].clamp(min=1e-5)
        x_n = 2 * (p[:, 0] / z) / (W - 1) - 1.
        y_n = 2 * (p[:, 1] / z) / (H - 1) - 1.
        return torch.stack((x_n, y_n), dim=-1)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
