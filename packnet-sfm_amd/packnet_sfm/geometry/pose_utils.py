"""Rigid-transform helpers (tiny [B,4,4] algebra; stays in torch so autograd reaches the pose network).
API of the reference's packnet_sfm/geometry/pose_utils.py."""
import numpy as np
import torch


def _rot(axis, angle):
    """Batch of elementary rotations about `axis` ('x' | 'y' | 'z'); angle: [B]."""
    c, s = torch.cos(angle), torch.sin(angle)
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    rows = {'x': (o, z, z, z, c, -s, z, s, c),
            'y': (c, z, s, z, o, z, -s, z, c),
            'z': (c, -s, z, s, c, z, z, z, o)}[axis]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def euler2mat(angle):
    """[B,3] euler angles (x, y, z) -> [B,3,3] with R = Rx @ Ry @ Rz."""
    return _rot('x', angle[:, 0]).bmm(_rot('y', angle[:, 1])).bmm(_rot('z', angle[:, 2]))


def pose_vec2mat(vec, mode='euler'):
    """[B,6] (tx, ty, tz, rx, ry, rz) -> [B,3,4]."""
    if mode is None:
        return vec
    if mode != 'euler':
        raise ValueError('Rotation mode not supported {}'.format(mode))
    return torch.cat([euler2mat(vec[:, 3:]), vec[:, :3].unsqueeze(-1)], dim=2)


def invert_pose(T):
    """Inverse of a batch of [B,4,4] rigid transforms."""
    Rt = T[:, :3, :3].transpose(-2, -1)
    t = -Rt.bmm(T[:, :3, 3:])
    out = torch.eye(4, device=T.device, dtype=T.dtype).repeat(len(T), 1, 1)
    out[:, :3, :3] = Rt
    out[:, :3, 3:] = t
    return out


def invert_pose_numpy(T):
    """Inverse of one [4,4] numpy rigid transform."""
    out = np.copy(T)
    R, t = T[:3, :3], T[:3, 3]
    out[:3, :3], out[:3, 3] = R.T, -np.matmul(R.T, t)
    return out


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
