"""Pose: a batch of 4x4 rigid transforms (names of the reference's packnet_sfm/geometry/pose.py).

`Pose.from_vec(vec, 'euler')` -- the only constructor on the training path -- is ONE launch of the fused pose kernel
(csrc/elementwise.hip) instead of ~85 tiny ATen ops; everything else is host-side bookkeeping on [B,4,4] tensors."""
import torch

from packnet_sfm.geometry.pose_utils import invert_pose, pose_vec2mat


def _is_transform_stack(t):
    return torch.is_tensor(t) and t.dim() == 3 and tuple(t.shape[-2:]) == (4, 4)


class Pose:
    def __init__(self, mat):
        if torch.is_tensor(mat) and mat.dim() == 2:
            mat = mat[None]
        assert _is_transform_stack(mat), 'Pose expects [B,4,4] (or a single [4,4]) matrices'
        self.mat = mat

    # ---- constructors ------------------------------------------------------------------------------------------------
    @classmethod
    def identity(cls, N=1, device=None, dtype=torch.float):
        return cls(torch.eye(4, device=device, dtype=dtype).expand(N, 4, 4).clone())

    @classmethod
    def from_vec(cls, vec, mode):
        """[B,6] = (tx, ty, tz, rx, ry, rz) -> Pose; the last row of every matrix is (0, 0, 0, 1)."""
        fused = mode == 'euler' and vec.dtype == torch.float32 and vec.dim() == 2 and vec.shape[1] == 6
        if fused:
            from packnet_sfm.hip import functional as HF
            return cls(HF.pose_vec2mat44(vec))
        rt = pose_vec2mat(vec, mode)                                     # [B,3,4]
        last = rt.new_tensor([0., 0., 0., 1.]).expand(len(vec), 1, 4)
        return cls(torch.cat((rt, last), dim=1))

    # ---- container protocol ------------------------------------------------------------------------------------------
    def __len__(self):
        return self.mat.shape[0]

    shape = property(lambda self: self.mat.shape)

    def item(self):
        return self.mat

    def repeat(self, *args, **kwargs):
        self.mat = self.mat.repeat(*args, **kwargs)
        return self

    def to(self, *args, **kwargs):
        self.mat = self.mat.to(*args, **kwargs)
        return self

    # ---- algebra -----------------------------------------------------------------------------------------------------
    def inverse(self):
        return Pose(invert_pose(self.mat))

    def transform_pose(self, pose):
        """Composition self * pose."""
        other = pose.item()
        assert _is_transform_stack(other)
        return Pose(torch.bmm(self.mat, other))

    def transform_points(self, points):
        """R @ p + t for a [B,3,H,W] (or [B,3,N]) point map."""
        assert points.shape[1] == 3
        flat = points.flatten(2)                                         # [B,3,N]
        moved = torch.baddbmm(self.mat[:, :3, 3:], self.mat[:, :3, :3], flat)
        return moved.view_as(points)

    def __matmul__(self, other):
        if isinstance(other, Pose):
            return self.transform_pose(other)
        if not torch.is_tensor(other):
            raise NotImplementedError()
        if other.dim() in (3, 4) and other.shape[1] == 3:
            return self.transform_points(other)
        raise ValueError('Unknown tensor dimensions {}'.format(other.shape))


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
