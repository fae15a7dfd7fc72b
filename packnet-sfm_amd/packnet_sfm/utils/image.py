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
    return [funct.interpolate(image, shape, mode=mode, align_corners=align_corners) for image in images]


def match_scales(image, targets, num_scales, mode='bilinear', align_corners=True):
    """One copy of `image` per target resolution (the same tensor when the resolution already matches)."""
    out = []
    for i in range(num_scales):
        tshape = targets[i].shape
        if same_shape(image.shape[-2:], tshape[-2:]):
            out.append(image)
        else:
            out.append(interpolate_image(image, tshape, mode=mode, align_corners=align_corners))
    return out


@lru_cache(maxsize=None)
def meshgrid(B, H, W, dtype, device, normalized=False):
    if normalized:
        xs = torch.linspace(-1, 1, W, device=device, dtype=dtype)
        ys = torch.linspace(-1, 1, H, device=device, dtype=dtype)
    else:
        xs = torch.linspace(0, W - 1, W, device=device, dtype=dtype)
        ys = torch.linspace(0, H - 1, H, device=device, dtype=dtype)
    ys, xs = torch.meshgrid([ys, xs], indexing='ij')
    return xs.repeat([B, 1, 1]), ys.repeat([B, 1, 1])


@lru_cache(maxsize=None)
def image_grid(B, H, W, dtype, device, normalized=False):
    xs, ys = meshgrid(B, H, W, dtype, device, normalized=normalized)
    return torch.stack([xs, ys, torch.ones_like(xs)], dim=1)
