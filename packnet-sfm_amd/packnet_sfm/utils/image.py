"""Image-tensor helpers on the training path (API subset of the reference's packnet_sfm/utils/image.py).
Pure data movement / resampling; the arithmetic that matters lives in the HIP kernels."""
from functools import lru_cache

import torch
import torch.nn.functional as funct

from packnet_sfm.utils.misc import same_shape


def flip_lr(image):
    assert image.dim() == 4, 'You need to provide a [B,C,H,W] image to flip'
    return torch.flip(image, [3])


def gradient_x(image):
    return image[:, :, :, :-1] - image[:, :, :, 1:]


def gradient_y(image):
    return image[:, :, :-1, :] - image[:, :, 1:, :]


def interpolate_image(image, shape, mode='bilinear', align_corners=True):
    if len(shape) > 2:
        shape = shape[-2:]
    if same_shape(image.shape[-2:], shape):
        return image
    return funct.interpolate(image, size=shape, mode=mode, align_corners=align_corners)


def interpolate_scales(images, shape=None, mode='bilinear', align_corners=False):
    """Resize every image of a list to `shape` (default: the first image's)."""
    if shape is None:
        shape = images[0].shape
    if len(shape) > 2:
        shape = shape[-2:]
    if mode == 'nearest':
        # integer up-sampling factors of device maps (every predicted scale brought to full resolution): the gfx950 kernel, and the
        # full-resolution map itself instead of a copy of it
        from packnet_sfm.hip.functional import upsample_nearest
        return [upsample_nearest(image, size=shape) for image in images]
    return [funct.interpolate(image, shape, mode=mode, align_corners=align_corners) for image in images]


def match_scales(image, targets, num_scales, mode='bilinear', align_corners=True):
    """`image` resized to each of the first `num_scales` target resolutions; where the resolution already matches, the
    tensor itself is returned (no copy)."""
    def at_scale(target):
        if same_shape(image.shape[-2:], target.shape[-2:]):
            return image
        return interpolate_image(image, target.shape, mode=mode, align_corners=align_corners)
    return [at_scale(t) for t in targets[:num_scales]]


def _axis(n, normalized, dtype, device):
    lo, hi = (-1., 1.) if normalized else (0., float(n - 1))
    return torch.linspace(lo, hi, n, device=device, dtype=dtype)


@lru_cache(maxsize=None)
def meshgrid(B, H, W, dtype, device, normalized=False):
    """x and y coordinate planes, each [B,H,W] (pixel centres 0..W-1 / 0..H-1, or [-1, 1] when `normalized`)."""
    xs = _axis(W, normalized, dtype, device).view(1, 1, W).expand(B, H, W)
    ys = _axis(H, normalized, dtype, device).view(1, H, 1).expand(B, H, W)
    return xs.contiguous(), ys.contiguous()


@lru_cache(maxsize=None)
def image_grid(B, H, W, dtype, device, normalized=False):
    """Homogeneous pixel grid [B,3,H,W] = (x, y, 1)."""
    xs, ys = meshgrid(B, H, W, dtype, device, normalized=normalized)
    return torch.stack((xs, ys, torch.ones_like(xs)), dim=1)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
