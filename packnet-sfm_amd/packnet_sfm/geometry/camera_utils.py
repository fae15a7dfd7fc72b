"""Intrinsics helpers + view synthesis.  API of the reference's packnet_sfm/geometry/camera_utils.py."""
import torch

from packnet_sfm.hip import functional as HF


def construct_K(fx, fy, cx, cy, dtype=torch.float, device=None):
    return torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=dtype, device=device)


def scale_intrinsics(K, x_scale, y_scale):
    """In-place rescale of [...,3,3] intrinsics (pixel-centre convention of the reference)."""
    K[..., 0, 0] *= x_scale
    K[..., 1, 1] *= y_scale
    K[..., 0, 2] = (K[..., 0, 2] + 0.5) * x_scale - 0.5
    K[..., 1, 2] = (K[..., 1, 2] + 0.5) * y_scale - 0.5
    return K


def view_synthesis(ref_image, depth, ref_cam, cam, mode='bilinear', padding_mode='zeros'):
    """Warp `ref_image` into the view of `cam` given that view's depth map -- ONE fused gfx950 kernel
    (reconstruct -> rigid transform -> project -> bilinear gather) instead of the reference's ~25 ATen ops.
    The fused kernel takes inverse depth; 1/depth here is the exact inverse of inv2depth for depth >= 1e-6."""
    if mode != 'bilinear':
        raise NotImplementedError('the gfx950 view-synthesis kernel implements bilinear sampling')
    assert depth.size(1) == 1
    T = ref_cam.Tcw.mat.bmm(cam.Twc.mat)          # target camera -> world -> reference camera
    warped = HF.view_synthesis(1.0 / depth, ref_image.unsqueeze(0), cam.K.float(), ref_cam.K.float(), T.unsqueeze(0),
                               padding_mode)
    return warped[0]


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
