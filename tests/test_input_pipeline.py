"""Device-side input pipeline (csrc/augment.hip, packnet_sfm/datasets/device_transforms.py) against the reference's
train_transforms executed with the real Pillow (oracle/augment_oracle.py): byte work, so the bar is BIT-EXACT.
CPU: the same kernel sources on the host emulator; GPU: the gfx950 build."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import augment_oracle as AO


def _frames(B, H, W, seed, ncontext=2):
    rng = np.random.default_rng(seed)
    # smooth-ish content + noise: exercises clipping in the Lanczos overshoot and every hue sextant
    base = rng.integers(0, 256, (B * (1 + ncontext), H // 4 + 1, W // 4 + 1, 3), dtype=np.uint8)
    up = np.stack([np.asarray(Image.fromarray(b).resize((W, H), Image.BILINEAR)) for b in base])
    noise = rng.integers(-40, 41, up.shape)
    fr = np.clip(up.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    return fr[:B], [fr[B * (i + 1):B * (i + 2)] for i in range(ncontext)]


def _run_case(device, B, H, W, shape, jitter, borders, seed):
    from packnet_sfm.datasets.device_transforms import DeviceTrainTransform
    rgb, ctx = _frames(B, H, W, seed)
    K = np.array([[0.58 * W, 0, 0.5 * W], [0, 1.92 * H, 0.5 * H], [0, 0, 1]], dtype=np.float64)
    random.seed(100 + seed)
    ref = [AO.train_transforms({'rgb': Image.fromarray(rgb[b]), 'rgb_context': [Image.fromarray(c[b]) for c in ctx],
                                'intrinsics': K.copy()}, shape, jitter, borders) for b in range(B)]
    random.seed(100 + seed)
    t = DeviceTrainTransform(shape, jitter, borders)
    out = t({'rgb': torch.from_numpy(rgb).to(device), 'rgb_context': [torch.from_numpy(c).to(device) for c in ctx],
             'intrinsics': torch.from_numpy(np.stack([K] * B)).to(device)})
    for b in range(B):
        for key in ('rgb', 'rgb_original'):
            assert torch.equal(out[key][b].cpu(), ref[b][key]), '%s differs from PIL (sample %d)' % (key, b)
        for key in ('rgb_context', 'rgb_context_original'):
            for j in range(len(ctx)):
                assert torch.equal(out[key][j][b].cpu(), ref[b][key][j]), '%s[%d] differs from PIL (sample %d)' % (key, j, b)
        np.testing.assert_allclose(out['intrinsics'][b].cpu().numpy(), ref[b]['intrinsics'], rtol=1e-12)
    assert float((out['rgb'] - out['rgb_original']).abs().max()) > 0 or not jitter


CASES = [  # B, H, W, image_shape, jittering, crop borders, seed
    (2, 37, 124, (19, 64), (0.2, 0.2, 0.2, 0.05), (), 0),          # KITTI-like 2x downscale + the YAML's jitter
    (1, 24, 40, (48, 56), (0.9, 0.9, 0.9, 0.5), (), 1),             # upscale; extreme factors (extrapolating blends, full hue range)
    (2, 40, 64, (), (0.3, 0.0, 0.4, 0.0), (), 2),                   # no resize; zero-width ranges
    (2, 41, 70, (16, 32), (), (5, 32, 3, 64), 3),                   # crop (y, height, x, width) + resize, no jitter
    (1, 41, 70, (16, 32), (0.2, 0.2, 0.2, 0.05), (-36, -4, 0.5, 40), 5),   # negative start / extent, float centre crop
    (1, 41, 70, (), (), (-5, 3), 6),                                # two-value form
    (1, 30, 50, (30, 25), (0.2, 0.2, 0.2, 0.05), (), 4),            # one axis only
    (2, 37, 124, (19, 64), (0.2, 0.2, 0.2, 0.05, 0.3), (), 7),      # + the 3x4 'color' matrix of jittering[4] (augmentations.py:266-277)
    (1, 24, 40, (), (0.0, 0.0, 0.0, 0.0, 0.9), (), 8),              # the colour gains alone, large enough to clip at 255
]


@pytest.mark.parametrize('case', CASES)
def test_device_train_transform_emulated(emulated_kernels, case):
    _run_case('cpu', *case)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES + [(4, 375, 1242, (192, 640), (0.2, 0.2, 0.2, 0.05), (), 9)])
def test_device_train_transform_gpu(case):
    _run_case('cuda', *case)


def test_parse_crop_borders_matches_oracle():
    """utils.misc.parse_crop_borders (what DeviceTrainTransform resolves `crop_train_borders` with) against the oracle's
    restatement of the reference's rules, itself pinned against the reference by oracle/pin_against_reference.py."""
    from packnet_sfm.utils.misc import parse_crop_borders
    for spec, shape in AO.CROP_CASES:
        assert tuple(parse_crop_borders(spec, shape)) == tuple(AO.parse_crop_borders(spec, shape)), (spec, shape)
    import os
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'crop.pt'), weights_only=False)
    assert len(gold) == len(AO.CROP_CASES)
    for spec, shape, ref in gold:           # the reference's own answers (oracle/pin_against_reference.py crop)
        assert tuple(parse_crop_borders(spec, shape)) == tuple(ref) == tuple(AO.parse_crop_borders(spec, shape)), (spec, shape)
    for bad in ((5, 32, 3, 640), (50, 3), (1, 2, 3)):
        with pytest.raises((AssertionError, NotImplementedError)):
            parse_crop_borders(bad, (41, 70))


def test_hsv_round_trip_all_colours(emulated_kernels):
    """The hue path (RGB -> HSV, H + delta mod 256, HSV -> RGB) over ALL 2^24 colours of a 4096x4096 image, for two hue
    shifts, against PIL's own conversions."""
    import struct
    from packnet_sfm.hip import ops
    g = np.arange(256, dtype=np.uint8)
    allrgb = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(1, 4096, 4096, 3)[:, ::1, ::8]   # 2^21 colours on the emulator
    for hue_add in (0, 37):
        rec = ops.jitter_record((3, -1, -1, -1), (0.0, 1.0, 1.0, 1.0), hue_add, 1)
        out, _ = ops.jitter_totensor(torch.from_numpy(np.ascontiguousarray(allrgb)), torch.frombuffer(bytearray(rec), dtype=torch.uint8),
                                     want_original=False)
        h, s, v = Image.fromarray(allrgb[0]).convert('HSV').split()
        np_h = np.array(h, dtype=np.uint8) + np.uint8(hue_add)
        ref = np.asarray(Image.merge('HSV', (Image.fromarray(np_h, 'L'), s, v)).convert('RGB'))
        got = (out[0].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()
        assert (got != ref).sum() == 0, '%d mismatching bytes at hue_add %d' % (int((got != ref).sum()), hue_add)


@pytest.mark.gpu
def test_hsv_round_trip_all_colours_gpu():
    import struct
    from packnet_sfm.hip import ops
    g = np.arange(256, dtype=np.uint8)
    allrgb = np.ascontiguousarray(np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(1, 4096, 4096, 3))
    for hue_add in (0, 37, 200):
        rec = ops.jitter_record((3, -1, -1, -1), (0.0, 1.0, 1.0, 1.0), hue_add, 1)
        out, _ = ops.jitter_totensor(torch.from_numpy(allrgb).cuda(), torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda(),
                                     want_original=False)
        h, s, v = Image.fromarray(allrgb[0]).convert('HSV').split()
        np_h = np.array(h, dtype=np.uint8) + np.uint8(hue_add)
        ref = np.asarray(Image.merge('HSV', (Image.fromarray(np_h, 'L'), s, v)).convert('RGB'))
        got = (out[0].permute(1, 2, 0) * 255).round().to(torch.uint8).cpu().numpy()
        assert (got != ref).sum() == 0
