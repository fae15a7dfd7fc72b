"""Pinhole camera.  API of the reference's packnet_sfm/geometry/camera.py (K, Tcw, scaled, reconstruct, project).

The training hot path never calls reconstruct()/project() separately: MultiViewPhotometricLoss hands intrinsics and
poses to the fused view-synthesis kernel (csrc/loss.hip).  The two methods are kept for API completeness (they are
thin tensor algebra used by evaluation / visualisation code) and run as ordinary torch ops on whatever device the
inputs live on."""
import torch
import torch.nn as nn

from packnet_sfm.geometry.camera_utils import scale_intrinsics
from packnet_sfm.geometry.pose import Pose
from packnet_sfm.utils.image import image_grid


class Camera(nn.Module):
    def __init__(self, K, Tcw=None):
        super().__init__()
        self.K = K
        self.Tcw = Pose.identity(len(K)) if Tcw is None else Tcw

    def __len__(self):
        return len(self.K)

    def to(self, *args, **kwargs):
        self.K = self.K.to(*args, **kwargs)
        self.Tcw = self.Tcw.to(*args, **kwargs)
        return self

    @property
    def fx(self):
        return self.K[:, 0, 0]

    @property
    def fy(self):
        return self.K[:, 1, 1]

    @property
    def cx(self):
        return self.K[:, 0, 2]

    @property
    def cy(self):
        return self.K[:, 1, 2]

    @property
    def Twc(self):
        return self.Tcw.inverse()

    @property
    def Kinv(self):
        Kinv = self.K.clone()
        Kinv[:, 0, 0] = 1. / self.fx
        Kinv[:, 1, 1] = 1. / self.fy
        Kinv[:, 0, 2] = -1. * self.cx / self.fx
        Kinv[:, 1, 2] = -1. * self.cy / self.fy
        return Kinv

    def scaled(self, x_scale, y_scale=None):
        if y_scale is None:
            y_scale = x_scale
        if x_scale == 1. and y_scale == 1.:
            return self
        return Camera(scale_intrinsics(self.K.clone(), x_scale, y_scale), Tcw=self.Tcw)

    def reconstruct(self, depth, frame='w'):
        """[B,1,H,W] depth -> [B,3,H,W] points in the camera ('c') or world ('w') frame."""
        B, C, H, W = depth.shape
        assert C == 1
        grid = image_grid(B, H, W, depth.dtype, depth.device, normalized=False)
        Xc = (self.Kinv.bmm(grid.view(B, 3, -1))).view(B, 3, H, W) * depth
        if frame == 'c':
            return Xc
        if frame == 'w':
            return self.Twc @ Xc
        raise ValueError('Unknown reference frame {}'.format(frame))

    def project(self, X, frame='w'):
        """[B,3,H,W] points -> [B,H,W,2] normalised image coordinates."""
        B, C, H, W = X.shape
        assert C == 3
        if frame == 'c':
            Xc = self.K.bmm(X.view(B, 3, -1))
        elif frame == 'w':
            Xc = self.K.bmm((self.Tcw @ X).view(B, 3, -1))
        else:
            raise ValueError('Unknown reference frame {}'.format(frame))
        Z = Xc[:, 2].clamp(min=1e-5)
        Xnorm = 2 * (Xc[:, 0] / Z) / (W - 1) - 1.
        Ynorm = 2 * (Xc[:, 1] / Z) / (H - 1) - 1.
        return torch.stack([Xnorm, Ynorm], dim=-1).view(B, H, W, 2)
