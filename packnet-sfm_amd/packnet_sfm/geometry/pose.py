"""Pose: a batch of [4,4] rigid transforms.  API of the reference's packnet_sfm/geometry/pose.py."""
import torch

from packnet_sfm.geometry.pose_utils import invert_pose, pose_vec2mat


class Pose:
    def __init__(self, mat):
        assert tuple(mat.shape[-2:]) == (4, 4)
        if mat.dim() == 2:
            mat = mat.unsqueeze(0)
        assert mat.dim() == 3
        self.mat = mat

    def __len__(self):
        return len(self.mat)

    @classmethod
    def identity(cls, N=1, device=None, dtype=torch.float):
        return cls(torch.eye(4, device=device, dtype=dtype).repeat([N, 1, 1]))

    @classmethod
    def from_vec(cls, vec, mode):
        """[B,6] pose vector -> Pose (bottom row [0,0,0,1])."""
        if mode == 'euler' and vec.dtype == torch.float32 and vec.dim() == 2 and vec.shape[1] == 6:
            from packnet_sfm.hip import functional as HF      # one launch each way instead of ~85 tiny ATen kernels
            return cls(HF.pose_vec2mat44(vec))
        top = pose_vec2mat(vec, mode)
        bottom = torch.zeros((len(vec), 1, 4), device=vec.device, dtype=vec.dtype)
        bottom[:, 0, 3] = 1.0
        return cls(torch.cat([top, bottom], dim=1))

    @property
    def shape(self):
        return self.mat.shape

    def item(self):
        return self.mat

    def repeat(self, *args, **kwargs):
        self.mat = self.mat.repeat(*args, **kwargs)
        return self

    def inverse(self):
        return Pose(invert_pose(self.mat))

    def to(self, *args, **kwargs):
        self.mat = self.mat.to(*args, **kwargs)
        return self

    def transform_pose(self, pose):
        """self * pose"""
        assert tuple(pose.shape[-2:]) == (4, 4)
        return Pose(self.mat.bmm(pose.item()))

    def transform_points(self, points):
        """[B,3,H,W] points -> R @ points + t"""
        assert points.shape[1] == 3
        B, _, H, W = points.shape
        out = self.mat[:, :3, :3].bmm(points.reshape(B, 3, -1)) + self.mat[:, :3, 3:]
        return out.view(B, 3, H, W)

    def __matmul__(self, other):
        if isinstance(other, Pose):
            return self.transform_pose(other)
        if isinstance(other, torch.Tensor):
            if other.shape[1] == 3 and other.dim() in (3, 4):
                return self.transform_points(other)
            raise ValueError('Unknown tensor dimensions {}'.format(other.shape))
        raise NotImplementedError()
