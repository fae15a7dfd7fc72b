"""TEST INFRASTRUCTURE -- CPU oracle of the training input pipeline: the reference's train_transforms
(/root/reference/packnet_sfm/datasets/transforms.py:11-41) executed with the REAL Pillow of this image.

The reference calls torchvision (pinned torchvision==0.9.1 in docker/Dockerfile:88-89; not installed here) on PIL images.
For PIL inputs torchvision 0.9.1's functional_pil does nothing but call Pillow, restated here call for call:
  transforms.Resize(shape, ANTIALIAS)       -> img.resize(shape[::-1], Image.LANCZOS)            (augmentations.py:101-125)
  adjust_brightness / _contrast / _saturation-> ImageEnhance.Brightness / Contrast / Color(img).enhance(f)   (functional_pil.py)
  adjust_hue                                 -> HSV split, np.uint8 add of uint8(hue_factor*255) with wrap, merge, back
  transforms.ToTensor                        -> uint8 HWC -> float32 CHW / 255
and the random draws are the reference's own (augmentations.py:254-337: random.random, 4 x random.uniform, random.shuffle).
Pinned: tests/test_input_pipeline.py checks the product kernels against THIS (i.e. against Pillow 12.2 itself), exhaustively
over all 2^24 colours for the HSV round trip.  Only tests / smoke may import this module."""
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance


def resize_image(img, shape):
    return img.resize((shape[1], shape[0]), Image.LANCZOS)


def adjust_hue(img, hue_factor):
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError('hue_factor ({}) is not in [-0.5, 0.5].'.format(hue_factor))
    h, s, v = img.convert('HSV').split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over='ignore'):
        np_h += np.array(hue_factor * 255).astype(np.uint8)
    return Image.merge('HSV', (Image.fromarray(np_h, 'L'), s, v)).convert('RGB')


def random_color_jitter_transform(parameters):
    brightness, contrast, saturation, hue = parameters
    bf = random.uniform(max(0, 1 - brightness), 1 + brightness)
    cf = random.uniform(max(0, 1 - contrast), 1 + contrast)
    sf = random.uniform(max(0, 1 - saturation), 1 + saturation)
    hf = random.uniform(-hue, hue)
    all_transforms = [lambda im: ImageEnhance.Brightness(im).enhance(bf), lambda im: ImageEnhance.Contrast(im).enhance(cf),
                      lambda im: ImageEnhance.Color(im).enhance(sf), lambda im: adjust_hue(im, hf)]
    random.shuffle(all_transforms)

    def composed(im):
        for t in all_transforms:
            im = t(im)
        return im
    return composed


def to_tensor(img):
    return torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)


CROP_CASES = [  # (config value, (H, W)): the forms the reference's parse_crop_borders accepts (utils/misc.py:77-146)
    ((), (41, 70)), ((5, 32, 3, 64), (41, 70)), ((-36, 0, -67, 0), (41, 70)), ((-36, -4, 3, -3), (41, 70)),
    ((0.5, 20, 0.5, 40), (41, 70)), ((0.25, 10, 3, 64), (41, 70)), ((5, 3), (41, 70)), ((-5, -3), (41, 70)), ((-5, 3), (41, 70)),
    ((-352, 0, 0.5, 1216), (375, 1242)),
]


def parse_crop_borders(borders, shape):
    """Restatement of the reference's utils/misc.py:77-146 (config crop value + image (H, W) -> (left, top, right, bottom)).
    Pinned against the reference itself for CROP_CASES by oracle/pin_against_reference.py."""
    if len(borders) == 0:                                        # :98-99
        return 0, 0, shape[1], shape[0]
    b = list(borders)
    if len(b) == 4:                                              # :103-122  (y, height, x, width) -> [x, y, width, height]
        b = [b[2], b[0], b[3], b[1]]
        for lo, hi, size in ((0, 2, shape[1]), (1, 3, shape[0])):
            if isinstance(b[lo], int):                           # regular crop: negative start from the far edge, extent <= 0 too
                if b[lo] < 0:
                    b[lo] += size
                b[hi] += size if b[hi] <= 0 else b[lo]
            else:                                                # centre crop: fraction of the image, extent centred on it
                c, half = b[lo] * size, b[hi] / 2
                b[lo], b[hi] = int(c - half), int(c + half)
    elif len(b) == 2:                                            # :124-137  (y, x) -> [x, y]
        b = [b[1], b[0]]
        if isinstance(b[0], int):
            b = (max(0, b[0]), max(0, b[1]), shape[1] + min(0, b[0]), shape[0] + min(0, b[1]))
        else:
            cw, ch, half = b[0] * shape[1], b[0] * shape[0], b[1] / 2
            b = (int(cw - half), int(ch - half), int(cw + half), int(ch + half))
    else:
        raise NotImplementedError('Crop tuple must have 2 or 4 values.')
    assert 0 <= b[0] < b[2] <= shape[1] and 0 <= b[1] < b[3] <= shape[0], 'Crop borders {} are invalid'.format(b)   # :141-144
    return tuple(b)


def train_transforms(sample, image_shape, jittering, crop_train_borders=()):
    """sample: {'rgb': PIL, 'rgb_context': [PIL], 'intrinsics': np [3,3]} -> as the reference's train_transforms (image keys;
    datasets/transforms.py:10-39: crop borders are the CONFIG value, resolved per image through parse_crop_borders :26-28)."""
    sample = dict(sample)
    if len(crop_train_borders) > 0:
        b = parse_crop_borders(crop_train_borders, sample['rgb'].size[::-1])
        K = np.copy(sample['intrinsics']); K[0, 2] -= b[0]; K[1, 2] -= b[1]
        sample['intrinsics'] = K
        sample['rgb'] = sample['rgb'].crop(b)
        sample['rgb_context'] = [k.crop(b) for k in sample['rgb_context']]
    if len(image_shape) > 0:
        ow, oh = sample['rgb'].size
        K = np.copy(sample['intrinsics']); K[0] *= image_shape[1] / ow; K[1] *= image_shape[0] / oh
        sample['intrinsics'] = K
        sample['rgb'] = resize_image(sample['rgb'], image_shape)
        sample['rgb_context'] = [resize_image(k, image_shape) for k in sample['rgb_context']]
    sample['rgb_original'] = sample['rgb'].copy()
    sample['rgb_context_original'] = [k.copy() for k in sample['rgb_context']]
    if len(jittering) > 0 and random.random() < 1.0:
        t = random_color_jitter_transform(jittering[:4])
        # augmentations.py:266-277: the optional 3x4 'color' matrix, drawn after the jitter transform, applied after it
        matrix = None
        if len(jittering) > 4 and jittering[4] > 0:
            matrix = (random.uniform(1. - jittering[4], 1 + jittering[4]), 0, 0, 0,
                      0, random.uniform(1. - jittering[4], 1 + jittering[4]), 0, 0,
                      0, 0, random.uniform(1. - jittering[4], 1 + jittering[4]), 0)
        sample['rgb'] = t(sample['rgb'])
        sample['rgb_context'] = [t(k) for k in sample['rgb_context']]
        if matrix is not None:
            sample['rgb'] = sample['rgb'].convert('RGB', matrix)
            sample['rgb_context'] = [k.convert('RGB', matrix) for k in sample['rgb_context']]
    for key in ('rgb', 'rgb_original'):
        sample[key] = to_tensor(sample[key])
    for key in ('rgb_context', 'rgb_context_original'):
        sample[key] = [to_tensor(k) for k in sample[key]]
    return sample
