"""GenericCamera: the Neural-Ray-Surface camera model (per-pixel ray directions instead of a pinhole matrix).

Drop-in for the reference's packnet_sfm/geometry/camera_generic.py (`GenericCamera(R, Tcw)`, `.Twc`, `reconstruct(depth, frame)`,
`project(X, progress, downsample, frame)` -> grid [B,H,W,2] for F.grid_sample).  `project` is the hot part: the reference
builds a [3, H*W, 1681] tensor of candidate ray vectors per call (camera_generic.py:127-183); here the gather, the logits,
the temperature softmax and the coordinate expectation are one HIP kernel forward and two backward (csrc/nrs.hip), and only
the two bilinear resamplings around it (the reference's F.interpolate calls, :150-152,165-167,197-201) stay torch ops.
Like the reference (`.squeeze()` at :171,176,183) it handles one camera at a time (B == 1).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from packnet_sfm.geometry.pose import Pose
from packnet_sfm.hip import functional as HF


class GenericCamera(nn.Module):
    def __init__(self, R, Tcw=None):
        """R: ray surface [B,3,H,W]; Tcw: camera -> world Pose."""
        super().__init__()
        self.ray_surface = R
        self.Tcw = Pose.identity(1) if Tcw is None else Tcw

    def to(self, *args, **kwargs):
        self.ray_surface = self.ray_surface.to(*args, **kwargs)
        self.Tcw = self.Tcw.to(*args, **kwargs)
        return self

    @property
    def Twc(self):
        """World -> camera transformation (inverse of Tcw)."""
        return self.Tcw.inverse()

    def reconstruct(self, depth, frame='w'):
        """P(x, y) = d(x, y) * r(x, y) (reference :53-84)."""
        B, C, H, W = depth.shape
        assert C == 1
        Xc = self.ray_surface * depth[0].unsqueeze(0)
        if frame == 'c':
            return Xc
        if frame == 'w':
            return self.Twc @ Xc
        raise ValueError('Unknown reference frame {}'.format(frame))

    @staticmethod
    def temperature(progress, min_temp=1e-8, start_temp=0.0001, constant=0.1):
        """Annealed softmax temperature (reference :108-111,185-186)."""
        return max(min_temp, start_temp / math.exp(constant * progress))

    def project(self, X, progress, downsample=True, frame='c'):
        """3-D points [1,3,H,W] -> normalised sampling grid [1,H,W,2] ((x, y) order, as F.grid_sample wants)."""
        B, C, H, W = X.shape
        assert C == 3
        if B != 1:
            raise NotImplementedError('one camera per call (the reference squeezes the batch dimension, camera_generic.py:171-183)')
        ray_surface = self.ray_surface
        if frame == 'w':
            X = self.Tcw @ X
        direction = X
        if downsample:
            H, W = int(H / 2.), int(W / 2.)
            ray_surface = F.interpolate(ray_surface, mode='bilinear', scale_factor=0.5, align_corners=True)
            direction = F.interpolate(direction, mode='bilinear', scale_factor=0.5, align_corners=True)
        direction = direction / torch.norm(direction, dim=1, keepdim=True)
        coords = HF.nrs_project(direction[0], ray_surface[0], self.temperature(progress))      # [H,W,2] (row, col)
        Xnorm = 2 * coords[:, :, 0] / (H - 1) - 1.
        Ynorm = 2 * coords[:, :, 1] / (W - 1) - 1.
        if downsample:
            Xnorm = F.interpolate(Xnorm[None, None], mode='bilinear', scale_factor=2.0, align_corners=True)[0, 0]
            Ynorm = F.interpolate(Ynorm[None, None], mode='bilinear', scale_factor=2.0, align_corners=True)[0, 0]
            H, W = H * 2, W * 2
        return torch.stack([Ynorm, Xnorm], dim=-1).view(B, H, W, 2)
