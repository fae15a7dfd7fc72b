"""Training input pipeline on the device: resize (Lanczos) -> duplicate -> colour jitter -> ToTensor, for a BATCH of decoded
uint8 frames that already sit in HBM.

Drop-in for what `train_transforms` (reference packnet_sfm/datasets/transforms.py:11-41) does per sample on the host with
PIL / torchvision (datasets/augmentations.py: resize_sample :101-180, duplicate_sample :228-252, colorjitter_sample /
random_color_jitter_transform :254-337, to_tensor_sample :185-226): same keys in, same keys out ('rgb', 'rgb_context',
'rgb_original', 'rgb_context_original', 'intrinsics'), same random draws from Python's `random` in the same order (one
`random.random()` for the jitter probability, four `random.uniform` factors, one `random.shuffle` of the four operations per
sample), and BIT-IDENTICAL pixels: the kernels of csrc/augment.hip restate Pillow's integer arithmetic (tests pin them against
PIL itself).  The frames of a batch share one launch per stage; crop borders are index arithmetic on the uint8 tensor.

Not covered (the reference's host path stays available through the merged package): depth-map resizing
(resize_depth_preserve), the optional 3x4 'color' matrix of `jittering[4]`, PIL images as input.
"""
import math
import random
import struct

import numpy as np
import torch

from packnet_sfm.hip import ops
from packnet_sfm.utils.misc import parse_crop_borders

_PRECISION_BITS = 32 - 8 - 2          # libImaging/Resample.c


def _lanczos(x):
    def sinc(v):
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    if -3.0 <= x < 3.0:
        return sinc(x) * sinc(x / 3)
    return 0.0


_COEFF_CACHE = {}


def lanczos_coefficients(in_size, out_size):
    """(kk int32 [out, ksize], bounds int32 [out, 2]) exactly as Pillow's precompute_coeffs + normalize_coeffs_8bpc compute
    them for Image.resize(..., Image.LANCZOS) without a box (libImaging/Resample.c)."""
    key = (in_size, out_size)
    if key in _COEFF_CACHE:
        return _COEFF_CACHE[key]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    _COEFF_CACHE[key] = (kk, bounds)
    return kk, bounds


def resize_frames(frames, shape):
    """frames: uint8 [N, H, W, 3] on the device -> uint8 [N, shape[0], shape[1], 3]; == PIL's img.resize((W, H), LANCZOS) per
    frame (horizontal pass, then vertical pass)."""
    N, H, W, _ = frames.shape
    oH, oW = shape
    out = frames.contiguous()
    if W != oW:
        kk, bounds = lanczos_coefficients(W, oW)
        out = ops.resample8(out, torch.from_numpy(kk).to(out.device), torch.from_numpy(bounds).to(out.device), oW, axis=1)
    if H != oH:
        kk, bounds = lanczos_coefficients(H, oH)
        out = ops.resample8(out, torch.from_numpy(kk).to(out.device), torch.from_numpy(bounds).to(out.device), oH, axis=0)
    return out


def draw_jitter(parameters, prob=1.0):
    """One sample's colour-jitter decision, consuming Python's RNG exactly like colorjitter_sample +
    random_color_jitter_transform (augmentations.py:254-337) -> packed 56-byte record for the kernel (order, four factors, hue, enable flag, 3-entry colour scale: include/pnsfm.h)."""
    if not (random.random() < prob):
        return ops.jitter_record()
    brightness, contrast, saturation, hue = parameters[:4]
    factors = [random.uniform(max(0, 1 - brightness), 1 + brightness),
               random.uniform(max(0, 1 - contrast), 1 + contrast),
               random.uniform(max(0, 1 - saturation), 1 + saturation),
               random.uniform(-hue, hue)]
    order = [0, 1, 2, 3]
    random.shuffle(order)
    hue_factor = factors[3]
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError('hue_factor ({}) is not in [-0.5, 0.5].'.format(hue_factor))
    # torchvision F_pil.adjust_hue: np_h (uint8) += np.uint8(hue_factor * 255), wrapping
    hue_add = int(np.array(hue_factor * 255).astype(np.uint8))
    fac = [np.float32(factors[o]) if o < 3 else np.float32(0.0) for o in order]
    # the 3x4 'color' matrix of jittering[4] (augmentations.py:266-270): three more uniform draws AFTER the jitter transform's own,
    # a per-channel gain applied last with PIL's Image.convert('RGB', matrix) arithmetic (csrc/augment.hip: color_scale8)
    color = None
    if len(parameters) > 4 and parameters[4] > 0:
        color = [random.uniform(1. - parameters[4], 1 + parameters[4]) for _ in range(3)]
    return ops.jitter_record(order, [float(f) for f in fac], hue_add, 1, color)


class DeviceTrainTransform:
    """
    Parameters (as `get_transforms('train', ...)` of the reference)
    ----------
    image_shape : (H, W) or ()        output resolution
    jittering : (brightness, contrast, saturation, hue[, color]) or ()
    crop_train_borders : the reference's config value -- (y, height, x, width) or (y, x), negative / float forms included --
                         resolved against every incoming frame size by utils.misc.parse_crop_borders exactly like
                         train_transforms does (datasets/transforms.py:26-29), or ()
    """

    def __init__(self, image_shape=(), jittering=(), crop_train_borders=(), jitter_prob=1.0):
        self.image_shape = tuple(image_shape)
        self.jittering = tuple(jittering)
        self.crop_spec = tuple(crop_train_borders)
        self.jitter_prob = jitter_prob
        if self.crop_spec and len(self.crop_spec) not in (2, 4):
            raise NotImplementedError('Crop tuple must have 2 or 4 values.')

    def _geometry(self, frames, borders):
        if borders:
            l, t, r, b = borders
            frames = frames[:, t:b, l:r]
        if self.image_shape:
            frames = resize_frames(frames, self.image_shape)
        return frames.contiguous()

    def __call__(self, sample):
        """sample: {'rgb': uint8 [B,H,W,3], 'rgb_context': [uint8 [B,H,W,3], ...], 'intrinsics': [B,3,3]} on the device."""
        out = dict(sample)
        rgb = sample['rgb']
        B, H0, W0, _ = rgb.shape
        ctx = list(sample.get('rgb_context', []))
        # geometry: every frame of the batch (target + contexts) in one launch per pass
        borders = parse_crop_borders(self.crop_spec, (H0, W0)) if self.crop_spec else ()
        allf = self._geometry(torch.cat([rgb] + ctx, 0) if ctx else rgb, borders)
        if 'intrinsics' in sample:
            K = sample['intrinsics'].clone()
            if borders:
                K[:, 0, 2] -= borders[0]
                K[:, 1, 2] -= borders[1]
            if self.image_shape:
                hc = (borders[3] - borders[1]) if borders else H0
                wc = (borders[2] - borders[0]) if borders else W0
                K[:, 0] *= self.image_shape[1] / wc
                K[:, 1] *= self.image_shape[0] / hc
            out['intrinsics'] = K
        # colour: one record per SAMPLE, shared by its target and context frames (augmentations.py:276-291)
        if self.jittering:
            recs = [draw_jitter(self.jittering, self.jitter_prob) for _ in range(B)]
        else:
            recs = [ops.jitter_record()] * B
        rec_all = b''.join(recs * (1 + len(ctx)))
        records = torch.frombuffer(bytearray(rec_all), dtype=torch.uint8).to(allf.device)
        jit, orig = ops.jitter_totensor(allf, records, want_original=True)
        out['rgb'], out['rgb_original'] = jit[:B], orig[:B]
        if ctx:
            out['rgb_context'] = [jit[B * (i + 1):B * (i + 2)] for i in range(len(ctx))]
            out['rgb_context_original'] = [orig[B * (i + 1):B * (i + 2)] for i in range(len(ctx))]
        return out
