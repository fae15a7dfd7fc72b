"""CPU: the kernel SOURCES of packnet-sfm_amd/csrc, compiled for the host with tests/emu/hipemu.h (fibers + emulated
MFMA/shuffles), reproduce the reference goldens.  This checks tiling, halo, fragment-layout and reduction index math of
every kernel without a GPU; the `-m gpu` tests run the same cases on the real gfx950 build."""
import pytest
import torch

import parity_cases as P


@pytest.mark.parametrize('name', ['conv2d_k3', 'conv2d_k5', 'conv2d_k7'])
def test_conv2d_block(emulated_kernels, name):
    P.case_conv2d_block(name, 'cpu')


def test_residual_conv(emulated_kernels):
    P.case_residual_conv('cpu')


def test_packing_invdepth(emulated_kernels):
    P.case_packing('cpu')
    P.case_invdepth('cpu')


def test_space_to_depth_channel_slice(emulated_kernels):
    """space_to_depth on a channel slice of a wider tensor (what torch.cat's backward hands to PixelShuffle's backward)."""
    import torch.nn.functional as F
    from packnet_sfm.hip import ops
    wide = torch.randn(3, 7, 4, 6, generator=torch.Generator().manual_seed(2))
    sl = wide[:, 2:5]
    assert not sl.is_contiguous()
    assert torch.equal(ops.space_to_depth(sl), F.pixel_unshuffle(sl, 2))
    assert torch.equal(ops.space_to_depth(wide), F.pixel_unshuffle(wide, 2))
    tr = wide.transpose(2, 3)                       # not a channel slice: falls back to a contiguous copy
    assert torch.equal(ops.space_to_depth(tr), F.pixel_unshuffle(tr, 2))


@pytest.mark.parametrize('shape', [(2, 3, 4, 16), (1, 5, 6, 8), (2, 4, 2, 24), (1, 2, 4, 12), (3, 1, 2, 4)])
def test_space_to_depth_depth_to_space_16_byte_forms(emulated_kernels, shape):
    """The float4 forms of the two shuffles (csrc/pack3d.hip: W % 8 == 0 for space_to_depth, W % 4 == 0 for depth_to_space, 16-byte
    aligned tensors) and the per-element fallbacks (W = 12 / 4, a channel slice at an odd offset) vs torch's pixel_unshuffle /
    pixel_shuffle -- bit-exact, data movement only."""
    import torch.nn.functional as F
    from packnet_sfm.hip import ops
    B, C, H, W = shape
    x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(sum(shape)))
    y = ops.space_to_depth(x)
    assert torch.equal(y, F.pixel_unshuffle(x, 2))
    assert torch.equal(ops.depth_to_space(y), x)
    z = torch.randn(B, 4 * C, H, W, generator=torch.Generator().manual_seed(1 + sum(shape)))
    assert torch.equal(ops.depth_to_space(z), F.pixel_shuffle(z, 2))
    wide = torch.randn(B, C + 3, H, W, generator=torch.Generator().manual_seed(2 + sum(shape)))
    for lo in (1, 2):                   # a slice that starts 1 / 2 planes in: aligned or not depending on H * W
        sl = wide[:, lo:lo + C]
        assert torch.equal(ops.space_to_depth(sl), F.pixel_unshuffle(sl, 2))


def test_unpack(emulated_kernels):
    P.case_unpack('cpu')


def test_pack_k3(emulated_kernels):
    P.case_pack('pack_k3', 'cpu')


def test_pack_unpack_d4(emulated_kernels):
    """d = 4 3-D feature maps (PackNetSlim01): reference-form packing block and unpacking block vs the reference golden."""
    P.case_pack_d4('pack_d4_k3', 'cpu')
    P.case_unpack_d4('cpu')


def test_compose_pack_weight(emulated_kernels):
    """Kernel composition used by the collapsed packing block (and its gradients) vs the oracle's formula.
    (The full collapsed block is checked against the reference golden in the -m gpu tests; too slow to emulate.)"""
    from oracle import packnet_oracle as O
    from packnet_sfm.hip import functional as HF
    g = torch.Generator().manual_seed(9)
    C, D, k = 3, 6, 3
    W2 = torch.randn(C, 8 * D, k, k, generator=g)
    W3 = torch.randn(8, 1, 3, 3, 3, generator=g)
    a, b = W2.clone().requires_grad_(True), W3.clone().requires_grad_(True)
    ar, br = W2.clone().requires_grad_(True), W3.clone().requires_grad_(True)
    We = HF.compose_pack_weight(a, b)
    Wr = O.compose_pack_weight(ar, br)
    P.check(We, Wr, 1e-5, 'W_eff')
    go = torch.randn(Wr.shape, generator=g)
    We.backward(go)
    Wr.backward(go)
    P.check(a.grad, ar.grad, 1e-5, 'dW2')
    P.check(b.grad, br.grad, 1e-5, 'dW3')
    # the composed kernel reproduces conv2d(conv3d(x)) away from the border (bias-free)
    x = torch.randn(1, D, 9, 10, generator=g)
    ref = torch.nn.functional.conv2d(O.conv3d_1to8(x, W3, torch.zeros(8)), W2, padding=k // 2)
    col = torch.nn.functional.conv2d(x, Wr.detach(), padding=k // 2 + 1)
    r = k // 2
    P.check(col[:, :, r:-r, r:-r], ref[:, :, r:-r, r:-r], 1e-5, 'interior equality')


@pytest.mark.parametrize('name', ['loss_default', 'loss_multires_mean', 'loss_clip_min', 'loss_clip_mean', 'loss_border',
                                  'loss_reflection', 'loss_l1_only', 'loss_l1_min', 'loss_l1_clip_min', 'loss_l1_clip_mean',
                                  'loss_l1_min_noauto'])
def test_loss(emulated_kernels, name):
    P.case_loss(name, 'cpu')


def test_region_ops_batched_windows(emulated_kernels):
    P.case_region_ops('cpu')


def test_upsample_nearest(emulated_kernels):
    P.case_upsample_nearest('cpu')


def test_loss_combine(emulated_kernels):
    P.case_loss_combine('cpu')


def test_compose_pack_params(emulated_kernels):
    P.case_compose_pack_params('cpu')


@pytest.mark.parametrize('scale', [1.0, 1e-7])
def test_smoothness_norm_fused(emulated_kernels, scale):
    """hip.functional.smoothness_norm (mean normalisation of the inverse depth fused into the smoothness kernels, round 4) against
    the reference's formula on torch autograd (multiview_photometric_loss.py:255-285, utils/depth.py:165-198): value and gradient,
    the normalisation's own gradient path included; scale 1e-7 makes the per-sample mean fall under the 1e-6 clamp (no gradient
    through the mean then).  An upstream gradient != 1 checks the device-scalar path."""
    from packnet_sfm.hip import functional as HF
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 9, 21
    d = ((0.2 + torch.rand(B, 1, H, W, generator=g)) * scale).requires_grad_(True)
    img = torch.rand(B, 3, H, W, generator=g)

    def ref(dd):
        n = dd / dd.mean(2, True).mean(3, True).clamp(min=1e-6)
        gx = n[:, :, :, :-1] - n[:, :, :, 1:]
        gy = n[:, :, :-1, :] - n[:, :, 1:, :]
        wx = torch.exp(-(img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, True))
        wy = torch.exp(-(img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, True))
        return (gx * wx).abs().mean() + (gy * wy).abs().mean()

    lr = ref(d) * 0.37
    lr.backward()
    gr = d.grad.clone()
    d.grad = None
    lh = HF.smoothness_norm(d, img) * 0.37
    lh.backward()
    P.check(lh, lr.detach(), 1e-5, 'smoothness_norm value')
    P.check(d.grad, gr, 2e-5, 'smoothness_norm gradient')


@pytest.mark.parametrize('direct_a', [6, 5, 4, 3, 2, 1, 0])
@pytest.mark.parametrize('shape', [(1, 4, 8, 8, 32, 3), (2, 3, 5, 6, 20, 3), (1, 6, 4, 4, 32, 7), (2, 20, 70, 5, 7, 1),
                                   (1, 96, 64, 6, 20, 3), (1, 40, 33, 9, 32, 5), (1, 24, 40, 10, 32, 7), (2, 129, 16, 5, 24, 3),
                                   (2, 32, 64, 4, 40, 1), (1, 48, 33, 12, 40, 1), (2, 3, 40, 9, 64, 5), (1, 3, 24, 6, 32, 5)])
def test_conv2d_raw(emulated_kernels, shape, direct_a):
    """Raw C-ABI conv entry points vs torch: 2-D tiles, linear tiles, odd channels, split-K, every kernel size; every
    variant of the forward/backward-data kernel: f32 MFMA (0 patch through registers, 1 patch by LDS-DMA, 2 fully pipelined)
    and the split-bf16 arithmetic (3 one patch buffer, 4 two, 5 whole kernel rows per stage, 6 three workgroups per CU; shapes with < 16 K-channels or
    fall through to the f32 kernels there; 1x1 layers run the split kernels since round 3, on 32-wide rows of the flattened map
    when H*W is a multiple of 32).  The last two shapes are the depth networks' stem (3 channels, 5x5, 32-multiple width): variant 0 runs
    conv2d_stem5_kernel there (ragged tile rows, 40 / 24 output channels: padded M tiles)."""
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    _lib.get().pnsfm_set_conv_math(1 if direct_a >= 3 else 0)
    _lib.get().pnsfm_set_conv_variant(direct_a)
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wf, wb = ops.conv2d_pack(w)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    P.check(ops.conv2d_forward(x, wf, b, Cout, ks), yr, 1e-5, 'fwd')
    P.check(ops.conv2d_backward_data(dy, wb, Cin, ks), xr.grad, 1e-5, 'dgrad')
    dw, db = ops.conv2d_backward_weight(x, dy, ks)
    P.check(dw, wr.grad, 1e-5, 'wgrad')
    P.check(db, br.grad, 1e-5, 'dbias')


_PIN_CFGS = [(2, 3, 0, 1), (2, 3, 0, 2), (2, 4, 0, 2), (2, 5, 1, 1), (1, 4, 1, 3), (2, 0, 0, 2), (2, 2, 1, 1), (1, 6, 0, 1), (2, 6, 1, 2),
             (2, 7, 0, 1), (1, 7, 0, 2), (2, 7, 1, 3), (1, 7, 1, 1)]       # 7: ping-pong workgroup (conv2d_bx3pp.h)


# every configuration on the 3x3 shape; on the 5x5 / 7x7 shapes the ping-pong ones and every other one of the rest (CPU-suite time)
@pytest.mark.parametrize('shape,cfg', [(sh, c) for i, sh in enumerate([(1, 48, 64, 9, 32, 3), (1, 40, 64, 20, 24, 5), (1, 32, 40, 8, 32, 7)])
                                       for j, c in enumerate(_PIN_CFGS) if i == 0 or c[1] == 7 or (i + j) % 2 == 0])
def test_conv2d_pinned_tilings(emulated_kernels, shape, cfg):
    """Configurations the un-tuned heuristics never pick for small test shapes (two pixel tiles per wave, narrow M tiles,
    K splits) pinned through pnsfm_tune_set -- what the autotuner does on the GPU: cfg = (NT, variant, narrow-M, K-split)."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    NT, variant, narrow, split = cfg
    bx3 = variant >= 3
    lib.pnsfm_set_conv_math(1 if bx3 else 0)
    B, Cin, Cout, H, W, ks = shape
    for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
        key = (ctypes.c_int * 7)(kind + 10 + (100 if bx3 else 0), B, K, M, H, W, ks)
        assert lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8), split) == 0
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wf, wb = ops.conv2d_pack(w)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    P.check(ops.conv2d_forward(x, wf, b, Cout, ks), yr, 1e-5, 'fwd')
    P.check(ops.conv2d_backward_data(dy, wb, Cin, ks), xr.grad, 1e-5, 'dgrad')
    lib.pnsfm_set_conv_variant(0)      # clears the pinned entries


_RB_CFGS = [(1, 3, 0, 1, 1), (2, 3, 0, 1, 1), (2, 4, 1, 2, 1), (1, 3, 0, 1, 2), (2, 5, 0, 1, 2), (2, 3, 1, 2, 2),
            (1, 7, 0, 1, 1), (2, 7, 1, 2, 1), (2, 7, 0, 1, 2)]
_RB_SHAPES = [(1, 32, 64, 12, 48, 3), (2, 48, 40, 7, 40, 3), (2, 16, 32, 6, 20, 5), (1, 32, 32, 9, 80, 7),
              (1, 16, 32, 18, 64, 7), (1, 32, 16, 10, 32, 5)]


# every configuration on the first 3x3 and the first 7x7 shape; a rotating third of them on the other four (CPU-suite time)
@pytest.mark.parametrize('shape,cfg', [(sh, c) for i, sh in enumerate(_RB_SHAPES) for j, c in enumerate(_RB_CFGS)
                                       if i in (0, 3) or (i + j) % 3 == 0])
def test_conv2d_rect_and_band_tiles(emulated_kernels, shape, cfg):
    """Tile modes of the split-bf16 kernels for maps whose width is not a multiple of 32 (24x80, 12x40, 6x20 in PackNet01), pinned
    like the autotuner does: cfg = (NT, variant, narrow-M, K-split, tile mode) with tile mode 1 = 16-wide rectangles (16 x 8*NT),
    2 = bands of whole rows (W x floor(128*NT / W)); ragged last tiles in both directions, heights below the tile height.  The last
    two shapes are 32-multiple widths, where the 16-wide rectangles are offered to 5x5 / 7x7 layers only (a pinned band falls back)."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    NT, variant, narrow, split, tm = cfg
    lib.pnsfm_set_conv_math(1)
    B, Cin, Cout, H, W, ks = shape
    if tm == 2 and W > 128 * NT:
        pytest.skip('a row does not fit the tile')
    for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
        key = (ctypes.c_int * 7)(kind + 10 + 100, B, K, M, H, W, ks)
        assert lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8) | (tm << 9), split) == 0
    g = torch.Generator().manual_seed(sum(shape) + tm)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wf, wb = ops.conv2d_pack(w)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    P.check(ops.conv2d_forward(x, wf, b, Cout, ks), yr, 1e-5, 'fwd')
    P.check(ops.conv2d_backward_data(dy, wb, Cin, ks), xr.grad, 1e-5, 'dgrad')
    lib.pnsfm_set_conv_variant(0)      # clears the pinned entries


@pytest.mark.parametrize('shape,split', [((1, 3, 64, 8, 64, 5), 1), ((2, 3, 64, 6, 40, 5), 3), ((1, 3, 40, 9, 136, 5), 4), ((3, 3, 24, 4, 8, 5), 2),
                                         ((2, 3, 32, 12, 64, 5), 100), ((1, 3, 70, 5, 72, 5), 2)])
def test_conv2d_wgrad_stem_kernel(emulated_kernels, shape, split):
    """The stem's weight gradient on its own split-bf16 kernel (conv2d.hip: conv2d_wgrad_stem5_kernel -- 3 input channels, 5x5) vs torch,
    the pixel split pinned through pnsfm_tune_set: direct stores (one split) and slabs, ragged tiles in both directions (6 / 9 / 5 rows
    of 4-row tiles, 40 / 136 / 8 / 72 columns of 64-column tiles), both wave layouts (<= 32 channels: four pixel parts; more: two co
    tiles x two parts), padded co tiles (40, 24, 70 channels), more splits than tiles, several images."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_conv_math(1)
    B, Cin, Cout, H, W, ks = shape
    key = (ctypes.c_int * 7)(2 + 10, B, Cin, Cout, H * W, W, ks)
    assert lib.pnsfm_tune_set(key, split, 0) == 0
    g = torch.Generator().manual_seed(sum(shape) + split)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dw, db = ops.conv2d_backward_weight(x, dy, ks)
    out = (ctypes.c_int * 8)()
    tiles = B * -(-H // 4) * -(-W // 64)
    assert lib.pnsfm_conv2d_last_config(out) == 0 and out[0] == 105 and out[4] == -(-tiles // -(-tiles // min(split, tiles))), list(out)
    P.check(dw, wr.grad, 1e-5, 'wgrad (stem)')
    P.check(db, br.grad, 1e-5, 'dbias (stem)')
    lib.pnsfm_set_conv_math(0)           # the generic f32 kernel on the same data: the two agree to fp32 round-off
    dw0, db0 = ops.conv2d_backward_weight(x, dy, ks)
    lib.pnsfm_set_conv_math(1)
    P.check(dw, dw0, 1e-5, 'wgrad (stem) vs generic')
    lib.pnsfm_set_wgrad_variant(-1)      # clears the pinned entry


def _flat32(H, W):
    """launch_conv tiles a 1x1 layer whose H*W is a multiple of 32 (and W is not) as 32-wide rows: the tuning key carries those."""
    return ((H * W) // 32, 32) if (H * W) % 32 == 0 and W % 32 != 0 else (H, W)


@pytest.mark.parametrize('shape,cfg', [((2, 32, 64, 4, 40, 1), (1, 0, 1)), ((2, 32, 64, 4, 40, 1), (2, 0, 1)), ((1, 48, 33, 12, 40, 1), (1, 0, 1)),
                                       ((1, 48, 33, 12, 40, 1), (2, 1, 1)), ((2, 20, 70, 5, 7, 1), (1, 0, 1)), ((2, 20, 70, 5, 7, 1), (2, 1, 1)),
                                       ((1, 256, 64, 6, 20, 1), (1, 0, 3)), ((1, 256, 64, 6, 20, 1), (2, 0, 16)), ((3, 80, 96, 9, 32, 1), (2, 0, 2)),
                                       ((1, 144, 40, 17, 19, 1), (1, 1, 4)), ((1, 64, 64, 8, 64, 1), (2, 0, 1))])
def test_conv1x1_lds_free_kernel(emulated_kernels, shape, cfg):
    """The LDS-free 1x1 kernel (csrc/conv2d_bx3_1x1.h, tuner variant 8) vs torch, forward and backward-data, pinned as the autotuner
    would: cfg = (NT, narrow-M, K split).  Ragged last pixel tiles (160 / 35 / 323 pixels per image), a ragged last K chunk (20, 48,
    144 = 9 chunks over 4 splits: 3 + 3 + 3), padded M tiles (33, 70, 40 channels), every wave tile and prefetch depth (D = 2, 3, 4),
    K splits with fewer chunks than the prefetch depth, several images."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    NT, narrow, split = cfg
    lib.pnsfm_set_conv_math(1)
    B, Cin, Cout, H, W, ks = shape
    Hk, Wk = _flat32(H, W)
    for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
        key = (ctypes.c_int * 7)(kind + 10 + 100, B, K, M, Hk, Wk, ks)
        assert lib.pnsfm_tune_set(key, NT | (8 << 4) | (narrow << 8), split) == 0
    g = torch.Generator().manual_seed(sum(shape) + NT)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wf, wb = ops.conv2d_pack(w)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    out = (ctypes.c_int * 8)()
    P.check(ops.conv2d_forward(x, wf, b, Cout, ks), yr, 1e-5, 'fwd (1x1, no LDS)')
    assert lib.pnsfm_conv2d_last_config(out) == 0 and out[0] == 8 and out[1] == NT, list(out)
    P.check(ops.conv2d_backward_data(dy, wb, Cin, ks), xr.grad, 1e-5, 'dgrad (1x1, no LDS)')
    assert lib.pnsfm_conv2d_last_config(out) == 0 and out[0] == 8, list(out)
    lib.pnsfm_set_conv_variant(0)      # clears the pinned entries


@pytest.mark.parametrize('shape', [(1, 64, 64, 4, 32, 3), (2, 33, 70, 5, 16, 3), (1, 130, 20, 9, 8, 3), (2, 16, 96, 3, 64, 1),
                                   (1, 40, 64, 6, 24, 5), (3, 17, 31, 7, 40, 1), (2, 64, 64, 8, 32, 3)])
def test_conv2d_wgrad_tap_major(emulated_kernels, shape):
    """The tap-major weight-gradient kernel (csrc/conv2d_wgrad2.hip) vs torch: every fragment width (32 / 16 / 8 columns),
    k in {1, 3, 5}, channel counts that do not fill the 64 x 64 tile, heights that do not fill the tile rows, and a pixel
    split (atomics into a zero-filled buffer) -- forced through pnsfm_set_wgrad_variant(1)."""
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_wgrad_variant(1)
    try:
        B, Cin, Cout, H, W, ks = shape
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
        b = torch.randn(Cout, generator=g)
        xr, wr, br = x.clone(), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, padding=ks // 2)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        dw, db = ops.conv2d_backward_weight(x, dy, ks)
        P.check(dw, wr.grad, 1e-5, 'wgrad (tap-major)')
        P.check(db, br.grad, 1e-5, 'dbias (tap-major)')
    finally:
        lib.pnsfm_set_wgrad_variant(-1)


@pytest.mark.parametrize('shape', [(1, 64, 64, 4, 32, 3), (2, 33, 70, 5, 16, 3), (1, 130, 20, 9, 8, 3), (2, 16, 96, 3, 64, 3),
                                   (1, 40, 64, 6, 24, 5), (1, 32, 40, 6, 40, 7), (1, 48, 129, 5, 16, 3), (2, 40, 24, 6, 20, 3),
                                   (1, 16, 32, 9, 4, 5), (1, 32, 16, 3, 80, 3), (2, 16, 96, 3, 64, 1), (3, 17, 31, 7, 40, 1),
                                   (1, 64, 64, 4, 40, 1), (2, 40, 24, 6, 20, 1)])
def test_conv2d_wgrad_split_bf16(emulated_kernels, shape):
    """The split-bf16 weight-gradient kernel (csrc/conv2d_wgrad3.hip) vs torch: k in {1, 3, 5, 7} (even and odd operand shifts; 1x1
    maps flattened to 32-wide rows where exact),
    one and two ci tiles per wave, 1 / 2 / 4 co tiles per workgroup, 32- and 16-column tiles, widths that are not a multiple
    of 8 (per-element masking: 20, 4), heights that do not fill the 4 tile rows, odd channel counts, single-split (direct
    stores) and pixel-split (two-stage reduction) launches -- the library default under the split arithmetic, pinned here."""
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_conv_math(1)
    lib.pnsfm_set_wgrad_variant(2)
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone(), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dw, db = ops.conv2d_backward_weight(x, dy, ks)
    P.check(dw, wr.grad, 1e-5, 'wgrad (split-bf16)')
    P.check(db, br.grad, 1e-5, 'dbias (split-bf16)')


@pytest.mark.parametrize('cfg', [(1, 1, 1), (1, 1, 2), (2, 1, 1), (2, 2, 2), (3, 1, 4), (1, 1, 9), (2, 1, 12), (2, 1, 10)])
@pytest.mark.parametrize('shape', [(1, 64, 128, 8, 32, 3), (2, 48, 160, 5, 40, 3), (1, 32, 100, 6, 24, 5)])
def test_conv2d_wgrad_split_bf16_pinned(emulated_kernels, shape, cfg):
    """wgrad3 configurations the autotuner explores on the GPU, pinned through pnsfm_tune_set: cfg = (pixel split, ci tiles per
    wave NT, co tiles per workgroup WM; WM | 8 = the build whose register budget lets three workgroups share a CU) -- WM below the
    layer's maximum turns waves into extra pixel shares (LDS reduction)."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_conv_math(1)
    split, NT, WM = cfg
    B, Cin, Cout, H, W, ks = shape
    key = (ctypes.c_int * 7)(2 + 10 + 100, B, Cin, Cout, H * W, W, ks)
    assert lib.pnsfm_tune_set(key, split, 2 | (NT << 4) | (WM << 6)) == 0
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dw, db = ops.conv2d_backward_weight(x, dy, ks)
    P.check(dw, wr.grad, 1e-5, 'wgrad (split-bf16, pinned)')
    P.check(db, br.grad, 1e-5, 'dbias (split-bf16, pinned)')
    lib.pnsfm_set_wgrad_variant(-1)      # clears the pinned entry


# every configuration on the first two shapes, a rotating 3 of 5 on the others (CPU-suite time)
@pytest.mark.parametrize('shape,cfg', [(sh, c) for i, sh in enumerate([(1, 64, 64, 8, 32, 3), (2, 48, 70, 5, 40, 3), (2, 40, 24, 6, 20, 3), (1, 33, 129, 7, 80, 3),
                                                                      (1, 16, 32, 9, 4, 3), (1, 130, 20, 9, 8, 3)])
                                       for j, c in enumerate([(1, 1, 0, 0), (2, 2, 4, 0), (3, 1, 5, 0), (2, 1, 0, 6), (5, 2, 0, 4)])
                                       if i < 2 or (i + j) % 5 in (0, 2, 3)])
def test_conv2d_wgrad_nine_taps(emulated_kernels, shape, cfg):
    """The nine-taps-per-workgroup 3x3 weight gradient on the 16x16x32 MFMA (csrc/conv2d_wgrad4.hip) vs torch, pinned through
    pnsfm_tune_set (variant 3): cfg = (pixel split, ci tiles per workgroup, tile width in 8-pixel groups, tile rows; 0 = the
    library's choice).  Widths of 3 / 4 / 5 groups incl. ragged last tiles (W = 80 with 32-column tiles), W % 8 == 4 (masked half
    groups: 20, 4), one-group images, 6-row tiles whose last k-step is partly empty, heights that do not fill the tile rows,
    odd channel counts, direct stores (one split) and the two-stage reduction."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_conv_math(1)
    split, WCI, TG, TR = cfg
    B, Cin, Cout, H, W, ks = shape
    if W <= 24:
        TG = 0                      # narrow images have one legal width (3 groups)
    elif TR == 6:
        TR = 0                      # 6-row tiles exist for 3-group tiles only
    key = (ctypes.c_int * 7)(2 + 10 + 100, B, Cin, Cout, H * W, W, ks)
    assert lib.pnsfm_tune_set(key, split, 3 | ((WCI | (TG << 4) | (TR << 8)) << 4)) == 0
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dw, db = ops.conv2d_backward_weight(x, dy, ks)
    P.check(dw, wr.grad, 1e-5, 'wgrad (nine taps)')
    P.check(db, br.grad, 1e-5, 'dbias (nine taps)')
    lib.pnsfm_set_wgrad_variant(-1)      # clears the pinned entry


@pytest.mark.parametrize('nf', [8, 4])
@pytest.mark.parametrize('shape', [(1, 5, 4, 6), (2, 13, 3, 5), (1, 40, 2, 3), (1, 5, 4, 8), (2, 13, 3, 4), (1, 9, 5, 12)])
def test_conv3d_raw(emulated_kernels, shape, nf):
    """3x3x3 1->8 stencil: forward, data gradient (column-sliding kernel, ragged run lengths along d; W % 4 == 0 runs the
    four-outputs-per-thread form) and weight/bias gradient (register accumulation + LDS block reduction) vs the oracle on small
    odd volumes."""
    from oracle import packnet_oracle as O
    from packnet_sfm.hip import functional as HF
    B, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    p = torch.randn(B, D, H, W, generator=g)
    w3 = 0.3 * torch.randn(nf, 1, 3, 3, 3, generator=g)
    b3 = torch.randn(nf, generator=g)
    pr, wr, br = (t.clone().requires_grad_(True) for t in (p, w3, b3))
    pd, wd, bd = (t.clone().requires_grad_(True) for t in (p, w3, b3))
    yr = O.conv3d_1to8(pr, wr, br)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    y = HF.conv3d_1to8(pd, wd, bd)
    y.backward(dy)
    P.check(y, yr, 1e-5, 'conv3d fwd')
    P.check(pd.grad, pr.grad, 1e-5, 'conv3d dgrad')
    P.check(wd.grad, wr.grad, 1e-5, 'conv3d wgrad')
    P.check(bd.grad, br.grad, 1e-5, 'conv3d dbias')


@pytest.mark.parametrize('nf', [8, 4])
@pytest.mark.parametrize('run', [8, 4, 2])
@pytest.mark.parametrize('shape', [(1, 19, 3, 5), (2, 8, 2, 70), (1, 33, 4, 6), (1, 3, 5, 7)])
def test_conv3d_dgrad_column_kernel(emulated_kernels, monkeypatch, shape, run, nf):
    """Data gradient of the 3x3x3 stencil on the 8- / 4- / 2-plane column kernel (conv3d_dgrad_col_kernel: the outputs of a run stay
    in registers, one pass per feature) -- what every volume of the training step runs; small volumes reach it through
    PNSFM_CONV3D_LEN.  Ragged last runs, several runs per column, rows that end inside a wave, D < run."""
    from oracle import packnet_oracle as O
    from packnet_sfm.hip import ops
    monkeypatch.setenv('PNSFM_CONV3D_LEN', str(run))
    B, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + run)
    p = torch.randn(B, D, H, W, generator=g).requires_grad_(True)
    w3 = 0.3 * torch.randn(nf, 1, 3, 3, 3, generator=g)
    yr = O.conv3d_1to8(p, w3, torch.zeros(nf))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    P.check(ops.conv3d_backward_data(dy, w3), p.grad, 1e-5, 'conv3d dgrad (run %d)' % run)


@pytest.mark.parametrize('nf', [8, 4])
@pytest.mark.parametrize('variant', ['0', '34'])
@pytest.mark.parametrize('shape', [(1, 40, 2, 3), (2, 13, 3, 70), (1, 5, 4, 6)])
def test_conv3d_wgrad_variants(emulated_kernels, monkeypatch, shape, variant, nf):
    """Weight / bias gradient of the 3x3x3 stencil on both builds of the kernel (PNSFM_CONV3D_WGRAD_RING: 0 = round 3's, default =
    three planes in flight, centre loads + lane exchange, packed register pairs): several runs per column with a ragged last one,
    rows that end inside a wave, two images."""
    from oracle import packnet_oracle as O
    from packnet_sfm.hip import ops
    monkeypatch.setenv('PNSFM_CONV3D_WGRAD_RING', variant)
    B, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + int(variant))
    p = torch.randn(B, D, H, W, generator=g)
    w3 = (0.3 * torch.randn(nf, 1, 3, 3, 3, generator=g)).requires_grad_(True)
    b3 = torch.randn(nf, generator=g).requires_grad_(True)
    yr = O.conv3d_1to8(p, w3, b3)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    dw, db = ops.conv3d_backward_weight(p, dy)
    P.check(dw, w3.grad, 1e-5, 'conv3d wgrad (variant %s)' % variant)
    P.check(db, b3.grad, 1e-5, 'conv3d dbias (variant %s)' % variant)


@pytest.mark.parametrize('shape', [(2, 5, 7, 9), (1, 19, 4, 70), (1, 64, 3, 5)])
def test_invdepth_conv_raw(emulated_kernels, shape):
    """Fused InvDepth head (one output channel): ragged channel quarters / channel groups, pixel tails, vs torch."""
    import torch.nn.functional as F
    from packnet_sfm.hip import functional as HF
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, generator=g)
    w = 0.2 * torch.randn(1, C, 3, 3, generator=g)
    b = torch.randn(1, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    xd, wd, bd = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.sigmoid(F.conv2d(xr, wr, br, padding=1)) / 0.5
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    y = HF.invdepth_conv(xd, wd, bd, 0.5)
    y.backward(dy)
    P.check(y, yr, 1e-5, 'invdepth fwd')
    P.check(xd.grad, xr.grad, 1e-5, 'invdepth dx')
    P.check(wd.grad, wr.grad, 1e-5, 'invdepth dw')
    P.check(bd.grad, br.grad, 1e-5, 'invdepth db')


@pytest.mark.parametrize('shape,R', [((2, 5, 7, 9), 4), ((1, 19, 13, 70), 8), ((1, 8, 9, 130), 4), ((2, 6, 16, 64), 8)])
def test_invdepth_conv_strip_kernel_is_bit_identical(emulated_kernels, monkeypatch, shape, R):
    """Round 5: the strip form of the InvDepth forward kernel (R rows x 64 columns per block, neighbours by wave shifts, XCD-ranged
    block order) keeps the per-pixel accumulation order of the 64-pixel kernel: equal bits on ragged strips / rows / channel quarters."""
    from packnet_sfm.hip import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + R)
    x = torch.randn(B, C, H, W, generator=g)
    w = 0.2 * torch.randn(1, C, 3, 3, generator=g)
    b = torch.randn(1, generator=g)
    monkeypatch.setenv('PNSFM_INVDEPTH_STRIP', '0')
    y0 = ops.invdepth_conv_forward(x, w, b, 0.5)
    monkeypatch.setenv('PNSFM_INVDEPTH_STRIP', str(R))
    y1 = ops.invdepth_conv_forward(x, w, b, 0.5)
    assert torch.equal(y0, y1)
    ref = torch.sigmoid(torch.nn.functional.conv2d(x, w, b, padding=1)) / 0.5
    P.check(y1, ref, 1e-5, 'invdepth strip fwd')


def test_supervised_loss(emulated_kernels):
    P.case_supervised_loss('cpu')


def test_semisup_model_plumbing(emulated_kernels):
    """SemiSupModel with supervised_loss_weight = 1 (no pose network): loss == SupervisedLoss on the depth net's output,
    metrics merged, eval returns the plain SfmModel output.  A stub depth network keeps the emulation fast."""
    from oracle import packnet_oracle as O
    from packnet_sfm.models.SemiSupModel import SemiSupModel

    class StubDepth(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.7))

        def forward(self, rgb):
            d = 0.1 + self.w * rgb.mean(1, keepdim=True)
            scales = [d, d[:, :, ::2, ::2], d[:, :, ::4, ::4], d[:, :, ::8, ::8]]
            return {'inv_depths': scales if self.training else d}

    g = torch.Generator().manual_seed(5)
    model = SemiSupModel(supervised_loss_weight=1.0, supervised_method='sparse-l1', supervised_num_scales=4,
                         flip_lr_prob=0.0, upsample_depth_maps=True)
    assert 'pose_net' not in model.network_requirements and 'gt_depth' in model.train_requirements
    model.add_depth_net(StubDepth())
    batch = {'rgb': torch.rand(2, 3, 16, 24, generator=g),
             'depth': 5.0 * torch.rand(2, 1, 16, 24, generator=g) * (torch.rand(2, 1, 16, 24, generator=g) > 0.5).float()}
    model.train()
    out = model(batch)
    assert set(out['metrics']) == {'supervised_loss'} and out['poses'] is None
    inv = [t.detach() for t in out['inv_depths']]
    gt_inv = 1. / batch['depth'].clamp(min=1e-6)
    gt_inv[batch['depth'] <= 0] = 0.
    P.check(out['loss'], O.supervised_loss(inv, gt_inv, 'sparse-l1', 4).reshape(1), 1e-5, 'semi-sup loss')
    out['loss'].sum().backward()
    assert model.depth_net.w.grad is not None and torch.isfinite(model.depth_net.w.grad)
    model.eval()
    with torch.no_grad():
        ev = model(batch)
    assert torch.is_tensor(ev['inv_depths']) and 'loss' not in ev


def test_side_stream_bookkeeping(emulated_kernels):
    """The per-pass use counts behind the weight-gradient side stream (hip/functional.py:_WgradStream): recorded only for
    nodes that will be back-propagated, a parameter used twice or a non-leaf weight is never left in flight, and the
    end-of-backward callback drops everything -- also after a no_grad evaluation or a graph that is never
    back-propagated.  (On CPU tensors no stream is involved; the decisions are the same code.)"""
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    WS = HF._WgradStream
    WS._uses.clear()
    decisions = []
    orig = WS.side_ok.__func__

    def spy(cls, *params):
        ok = orig(cls, *params)
        decisions.append(ok)
        return ok
    WS.side_ok = classmethod(spy)
    try:
        torch.manual_seed(0)
        m = Conv2D(4, 16, 3, 1)
        x = torch.randn(1, 4, 6, 8)
        with torch.no_grad():
            m(x)
        assert not WS._uses, 'evaluation under no_grad must not be counted'
        m(x).sum().backward()                              # weight used once: may stay in flight
        assert decisions == [True] and not WS._uses
        decisions.clear()
        m(x).sum().backward()                              # .grad already defined (gradient accumulation): AccumulateGrad
        assert decisions == [False] and not WS._uses       # adds on the compute stream -> the node must wait itself
        decisions.clear()
        m.zero_grad(set_to_none=True)
        (m(x).sum() + m(2 * x).sum()).backward()           # the same weight used twice in one graph: wait in the node
        assert decisions == [False, False] and not WS._uses
        decisions.clear()
        m.zero_grad(set_to_none=True)
        m(x)                                               # graph that is never back-propagated leaves a stale count ...
        assert WS._uses
        m(x).sum().backward()                              # ... which only makes the next pass conservative, then clears
        assert decisions == [False] and not WS._uses
        decisions.clear()
        m.zero_grad(set_to_none=True)
        m(x).sum().backward()
        assert decisions == [True]
        decisions.clear()
        m.zero_grad(set_to_none=True)
        h = m.conv_base.weight.register_hook(lambda g: g * 1.0)   # a tensor hook runs on the compute stream
        m(x).sum().backward()
        assert decisions == [False]
        h.remove()
        decisions.clear()
        m.zero_grad(set_to_none=True)

        class Boom(torch.autograd.Function):               # a backward pass that raises never reaches its callback ...
            @staticmethod
            def forward(ctx, t):
                return t.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError('boom')
        with pytest.raises(RuntimeError):
            Boom.apply(m(x)).sum().backward()
        m.zero_grad(set_to_none=True)
        decisions.clear()
        m(x).sum().backward()                              # ... the next pass still queues its own (graph-task id differs)
        assert WS._cb_task is None and not WS._uses
        decisions.clear()
        m.zero_grad(set_to_none=True)
        w_eff = m.conv_base.weight * 2.0                   # a non-leaf weight is consumed by compute-stream kernels
        HF.conv2d(x, w_eff, m.conv_base.bias, HF.PackedConvWeight(volatile=True)).sum().backward()
        assert decisions == [False] and not WS._uses
    finally:
        WS.side_ok = classmethod(orig)


def test_pose_vec2mat(emulated_kernels):
    """Pose.from_vec on the fused kernel vs the oracle's euler2mat composition (forward and gradient)."""
    from oracle import packnet_oracle as O
    from packnet_sfm.geometry.pose import Pose
    g = torch.Generator().manual_seed(4)
    v = torch.randn(5, 6, generator=g)
    a, b = v.clone().requires_grad_(True), v.clone().requires_grad_(True)
    T = Pose.from_vec(a, 'euler').mat
    Tr = O.pose_vec2mat44(b)
    P.check(T, Tr, 1e-6, 'pose matrix')
    go = torch.randn(5, 4, 4, generator=g)
    T.backward(go)
    Tr.backward(go)
    P.check(a.grad, b.grad, 1e-5, 'd pose vector')


def test_adam_matches_torch(emulated_kernels):
    from packnet_sfm.hip import ops
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2e-4)
    p, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad * 2.0, m, v, 2e-4, 0.9, 0.999, 1e-8, 0.0, 0.5, step)
    P.check(p, p_ref, 1e-6, 'adam')


def test_flat_adam_matches_torch_adam(emulated_kernels):
    """FlatAdam (one adam_kernel launch per group on a flat buffer) == torch.optim.Adam, two groups with different lr."""
    from packnet_sfm.rccl.flat_adam import FlatAdam
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    net_b = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    net_b.load_state_dict(net_a.state_dict())
    ref = torch.optim.Adam([{'params': net_a[0].parameters(), 'lr': 1e-2}, {'params': net_a[2].parameters(), 'lr': 3e-3}])
    opt = FlatAdam([{'params': list(net_b[0].parameters()), 'lr': 1e-2}, {'params': list(net_b[2].parameters()), 'lr': 3e-3}])
    x = torch.randn(5, 6)
    for _ in range(4):
        ref.zero_grad()
        opt.zero_grad()
        net_a(x).pow(2).sum().backward()
        net_b(x).pow(2).sum().backward()
        ref.step()
        opt.step()
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        P.check(pb, pa, 1e-5, 'flat adam parameter')
    assert set(net_b.state_dict().keys()) == set(net_a.state_dict().keys())
    # optimizer state in torch.optim.Adam's layout, both directions (the reference's checkpoints store optimizer.state_dict())
    sd = opt.state_dict()
    assert set(sd.keys()) == {'state', 'param_groups'} and len(sd['state']) == 4 and [g['params'] for g in sd['param_groups']] == [[0, 1], [2, 3]]
    rsd = ref.state_dict()
    for i in range(4):
        P.check(sd['state'][i]['exp_avg'], rsd['state'][i]['exp_avg'], 1e-5, 'exp_avg %d' % i)
        P.check(sd['state'][i]['exp_avg_sq'], rsd['state'][i]['exp_avg_sq'], 1e-5, 'exp_avg_sq %d' % i)
        assert float(sd['state'][i]['step']) == float(rsd['state'][i]['step']) == 4.0
    net_c = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    net_c.load_state_dict(net_a.state_dict())
    opt_c = FlatAdam([{'params': list(net_c[0].parameters()), 'lr': 1e-2}, {'params': list(net_c[2].parameters()), 'lr': 3e-3}])
    opt_c.load_state_dict(rsd)                  # resume from a TORCH Adam checkpoint ...
    ref2 = torch.optim.Adam([{'params': net_a[0].parameters(), 'lr': 1e-2}, {'params': net_a[2].parameters(), 'lr': 3e-3}])
    ref2.load_state_dict(sd)                    # ... and torch Adam from a FlatAdam checkpoint
    for o, net in ((opt_c, net_c), (ref2, net_a), (opt, net_b)):
        o.zero_grad()
        net(x).pow(2).sum().backward()
        o.step()
    for pa, pb, pc in zip(net_a.parameters(), net_b.parameters(), net_c.parameters()):
        P.check(pc, pb, 1e-5, 'resumed from torch state')
        P.check(pa, pb, 1e-5, 'torch resumed from FlatAdam state')
    opt.param_groups[0]['lr'] = 5e-3            # an LR scheduler writes param_groups[i]['lr']: picked up by the next step
    w0 = net_b[0].weight.detach().clone()
    opt.zero_grad(); net_b(x).pow(2).sum().backward(); opt.step()
    moved = float((net_b[0].weight.detach() - w0).abs().max())
    assert 3e-3 < moved <= 5e-3 * 1.2, moved     # ~lr * m/sqrt(v): the new lr (was 1e-2) took effect


# (every variant on the small shapes; the 256- / 512-channel ones -- channels per group 16 / 32 -- on the two kernels that ship)
@pytest.mark.parametrize('shape,variant', [(sh, v) for sh in [(2, 32, 64, 5, 32, 3), (1, 16, 128, 6, 20, 3), (1, 20, 64, 7, 40, 5)] for v in (3, 6, 7, 0)] +
                         [(sh, v) for sh in [(1, 48, 256, 5, 24, 3), (1, 32, 512, 4, 32, 1)] for v in (3, 7)])
def test_conv_gn_act_fused_block(emulated_kernels, shape, variant):
    """The Conv2D block as one autograd node (hip.functional.ConvGnActFn; since round 6 its two bodies are single calls into the block
    sequencer, csrc/seq/pnsfm_seq.cpp) against the two-node form (conv, then GroupNorm + activation) and against torch: output and every
    gradient.  Channels per group 4 / 8 / 16 / 32, ragged tiles, the ping-pong kernel (7), three workgroups per CU (6), the f32 kernels (0)."""
    import ctypes
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, functional as HF, ops
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    bx3 = variant >= 3 and Cin >= 16
    lib.pnsfm_set_conv_math(1 if variant >= 3 else 0)
    lib.pnsfm_set_conv_variant(variant if variant < 7 else 3)
    if variant == 7 and bx3:
        for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
            key = (ctypes.c_int * 7)(kind + 10 + 100, B, K, M, H, W, ks)
            assert lib.pnsfm_tune_set(key, 1 | (7 << 4), 1) == 0
    g = torch.Generator().manual_seed(sum(shape) + variant)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    dout = torch.randn(B, Cout, H, W, generator=g)
    res = {}
    for fused in (True, False):
        HF.set_conv_gn_fuse(fused)
        leaves = [t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta)]
        out = HF.conv2d_gn_act(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], HF.PackedConvWeight(), 16, 1e-5, ops.ACT_ELU)
        out.backward(dout)
        res[fused] = [out.detach()] + [t.grad for t in leaves]
    HF.set_conv_gn_fuse(True)
    ref = [t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta)]
    o = F.elu(F.group_norm(F.conv2d(ref[0], ref[1], ref[2], padding=ks // 2), 16, ref[3], ref[4], 1e-5))
    o.backward(dout)
    names = ('out', 'dx', 'dw', 'db', 'dgamma', 'dbeta')
    for n, a, c, r in zip(names, res[True], res[False], [o.detach()] + [t.grad for t in ref]):
        # (the conv bias in front of a GroupNorm has a mathematically zero gradient: round-off in every implementation)
        if n == 'db':
            continue
        P.check(a, c, 2e-5, n + ' fused vs two nodes')
        P.check(a, r, 1e-4, n + ' vs torch')
    lib.pnsfm_set_conv_variant(0)


def test_flat_adam_fused_tail(emulated_kernels):
    P.case_flat_adam_fused_tail('cpu')


def test_flat_adam_gradient_slots(emulated_kernels):
    """The conv weight-gradient kernel writes straight into FlatAdam's gradient arena (hip.functional.register_grad_slots):
    after backward the parameter's .grad IS the arena view (no gather copy), a parameter used twice falls back to the normal
    path, and three steps equal torch.optim.Adam on an identical replica."""
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    from packnet_sfm.rccl.flat_adam import FlatAdam
    torch.manual_seed(2)
    a, b = Conv2D(4, 32, 3, 1), Conv2D(4, 32, 3, 1)     # 2 channels per GroupNorm group: the conv bias has a real gradient
    b.load_state_dict(a.state_dict())
    ref = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = FlatAdam([{'params': list(b.parameters()), 'lr': 1e-2}])
    g = opt.param_groups[0]
    x = torch.randn(2, 4, 6, 8)
    for step in range(3):
        ref.zero_grad(); opt.zero_grad()
        a(x).pow(2).mean().backward()
        b(x).pow(2).mean().backward()
        w = b.conv_base.weight
        assert w.grad.data_ptr() == opt.grad_view(g, w).data_ptr(), 'conv weight gradient was not written into the arena'
        assert b.conv_base.bias.grad.data_ptr() == opt.grad_view(g, b.conv_base.bias).data_ptr()
        P.check(w.grad, a.conv_base.weight.grad, 1e-5, 'slot gradient')
        ref.step(); opt.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        P.check(pb, pa, 1e-5, 'parameter after 3 steps')
    opt.zero_grad()
    (b(x).pow(2).mean() + b(2 * x).pow(2).mean()).backward()          # weight used twice: gradients must ACCUMULATE
    ref.zero_grad()
    (a(x).pow(2).mean() + a(2 * x).pow(2).mean()).backward()
    P.check(b.conv_base.weight.grad, a.conv_base.weight.grad, 1e-5, 'shared-use gradient')
    opt._slots.remove()


@pytest.mark.parametrize('variant', [0, 2, 3])
@pytest.mark.parametrize('shape', [(1, 9, 16, 8, 64, 7), (2, 6, 32, 12, 40, 5), (1, 16, 8, 3, 10, 3), (1, 4, 8, 6, 20, 3),
                                   (1, 32, 24, 6, 20, 3)])
def test_conv2d_stride2(emulated_kernels, shape, variant):
    """PoseNet's stride-2 convs (even and odd input sizes): strided forward / weight-gradient kernels and the
    zero-upsample + stride-1 backward-data path vs torch."""
    import torch.nn.functional as F
    from packnet_sfm.hip import _lib, functional as HF
    _lib.get().pnsfm_set_conv_math(1 if variant >= 3 else 0)
    _lib.get().pnsfm_set_conv_variant(variant)
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=2, padding=(ks - 1) // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xh, wh, bh = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = HF.conv2d_stride2(xh, wh, bh, HF.PackedConvWeight())
    assert y.shape == yr.shape
    y.backward(dy)
    P.check(y, yr, 1e-5, 'fwd')
    P.check(xh.grad, xr.grad, 1e-5, 'dgrad')
    P.check(wh.grad, wr.grad, 1e-5, 'wgrad')
    P.check(bh.grad, br.grad, 1e-5, 'dbias')
    _lib.get().pnsfm_set_conv_variant(0)


def test_batched_repack_after_the_optimizer_step(emulated_kernels):
    """hip.functional.repack_all (one pnsfm_conv2d_pack_table launch for every registered conv weight, called by
    FlatAdam.step): after the step the caches are FRESH (the next forward does not re-pack), the packed buffers are
    bit-identical to what the per-layer packer writes for the updated weights (3x3, 5x5, 7x7 and 1x1 layers, ragged channel
    counts), a layer whose input needs no gradient (no backward-data buffer) or that has fewer than 16 channels stays on the
    lazy path, and PNSFM_PACK_BATCH=0 switches the batched launch off."""
    import os
    from packnet_sfm.hip import _lib, functional as HF, ops
    from packnet_sfm.networks.layers.packnet.layers01 import _HipConv2d
    from packnet_sfm.rccl.flat_adam import FlatAdam
    _lib.get().pnsfm_set_conv_math(1)
    torch.manual_seed(4)
    stem = _HipConv2d(3, 16, 5)                      # Cin = 3: f32 layout, never in the table
    layers = [_HipConv2d(16, 40, 3), _HipConv2d(40, 32, 5), _HipConv2d(32, 48, 7), _HipConv2d(48, 17, 1)]
    net = torch.nn.Sequential(stem, *layers)
    opt = FlatAdam([{'params': list(net.parameters()), 'lr': 1e-2}])
    x = torch.randn(1, 3, 6, 8)

    def fwd_bwd():
        opt.zero_grad()
        net(x).pow(2).mean().backward()

    fwd_bwd()
    caches = [l._packed for l in layers]
    assert all(c.wp_fwd is not None and c.wp_bwd is not None for c in caches)
    opt.step()
    n_lazy = [0]
    real_pack = ops.conv2d_pack

    def counting_pack(*a, **k):
        n_lazy[0] += 1
        return real_pack(*a, **k)

    ops.conv2d_pack = counting_pack
    try:
        for l, c in zip(layers, caches):
            w = l.weight
            assert c.key_fwd == HF.PackedConvWeight.key_of(w) and c.key_bwd == c.key_fwd, 'cache not stamped by repack_all'
            f, b = real_pack(w.detach().contiguous())
            assert torch.equal(f, c.wp_fwd) and torch.equal(b, c.wp_bwd), 'batched pack differs from the per-layer packer'
        fwd_bwd()
        assert n_lazy[0] == 1, 'only the stem (f32 layout, no backward-data buffer) may re-pack lazily, saw %d' % n_lazy[0]
        os.environ['PNSFM_PACK_BATCH'] = '0'
        opt.step()
        n_lazy[0] = 0
        fwd_bwd()
        assert n_lazy[0] == 1 + len(layers)
    finally:
        ops.conv2d_pack = real_pack
        os.environ.pop('PNSFM_PACK_BATCH', None)
        opt._slots.remove()


def test_batched_repack_follows_the_arithmetic_mode(emulated_kernels):
    """the device pack table is built for ONE arithmetic mode (which weights it covers, and the split-bf16 layout it
    writes).  Switching hip.functional.set_conv_math between two FlatAdam steps must rebuild it: after the switch every packed
    buffer equals the per-layer packer's output for the mode in force, both directions of the switch."""
    from packnet_sfm.hip import _lib, functional as HF, ops
    from packnet_sfm.networks.layers.packnet.layers01 import _HipConv2d
    from packnet_sfm.rccl.flat_adam import FlatAdam
    torch.manual_seed(6)
    layers = [_HipConv2d(16, 32, 3), _HipConv2d(32, 24, 3)]
    net = torch.nn.Sequential(*layers)
    opt = FlatAdam([{'params': list(net.parameters()), 'lr': 1e-2}])
    x = torch.randn(1, 16, 6, 8, requires_grad=True)

    def fwd_bwd():
        opt.zero_grad()
        y = net(x)
        y.pow(2).mean().backward()
        return y.detach().clone()

    def check_packed(tag):
        for l in layers:
            c = l._packed
            # (packed into CLONES of the cache's buffers: bytes beyond the mode's layout are don't-care and stay equal)
            f, b = ops.conv2d_pack(l.weight.detach().contiguous(), c.wp_fwd.clone(), c.wp_bwd.clone())
            assert torch.equal(f, c.wp_fwd), tag + ': forward pack is stale / in the wrong layout'
            assert torch.equal(b, c.wp_bwd), tag + ': backward pack is stale / in the wrong layout'

    try:
        for first, second in (('bx3', 'f32'), ('f32', 'bx3')):
            HF.set_conv_math(first)
            fwd_bwd()
            opt.step()                       # builds the table under `first`
            HF.set_conv_math(second)
            fwd_bwd()                        # lazy re-pack in the new layout (new buffers)
            opt.step()                       # the batched launch must not replay the old table
            y = fwd_bwd()                    # (weights the new mode's table does not cover re-pack lazily here)
            check_packed('%s -> %s' % (first, second))
            ref = torch.nn.functional.conv2d(torch.nn.functional.conv2d(x.detach(), layers[0].weight, layers[0].bias, padding=1),
                                             layers[1].weight, layers[1].bias, padding=1)
            P.check(y, ref, 1e-5, 'forward after the mode switch')
    finally:
        HF.set_conv_math('bx3')
        opt._slots.remove()


def test_flat_adam_skips_parameters_without_gradient(emulated_kernels):
    """torch.optim.Adam skips a parameter whose .grad is None (value and moments frozen); FlatAdam's flat launch must leave such
    parameters untouched too (they used to decay on stale momentum)."""
    from packnet_sfm.rccl.flat_adam import FlatAdam
    torch.manual_seed(4)
    net_a = torch.nn.ModuleList([torch.nn.Linear(5, 7), torch.nn.Linear(5, 3)])
    net_b = torch.nn.ModuleList([torch.nn.Linear(5, 7), torch.nn.Linear(5, 3)])
    net_b.load_state_dict(net_a.state_dict())
    ref = torch.optim.Adam(net_a.parameters(), lr=1e-2, weight_decay=1e-2)
    opt = FlatAdam([{'params': list(net_b.parameters()), 'lr': 1e-2, 'weight_decay': 1e-2}])
    x = torch.randn(4, 5)
    for step in range(4):
        ref.zero_grad(); opt.zero_grad()
        use_second = step in (0, 3)                     # the second layer has no gradient on steps 1 and 2
        for net in (net_a, net_b):
            out = net[0](x).pow(2).sum() + (net[1](x).pow(2).sum() if use_second else 0.0)
            out.backward()
        frozen = [p.detach().clone() for p in net_b[1].parameters()]
        ref.step(); opt.step()
        if not use_second:
            for p, q in zip(net_b[1].parameters(), frozen):
                assert torch.equal(p.detach(), q), 'a parameter without a gradient moved'
        for pa, pb in zip(net_a[0].parameters(), net_b[0].parameters()):
            P.check(pb, pa, 1e-5, 'always-used parameter, step %d' % step)
    sd = opt.state_dict()['state']
    rsd = ref.state_dict()['state']
    for i in (2, 3):                                    # moments of the sometimes-unused layer: frozen while unused, like torch
        P.check(sd[i]['exp_avg'], rsd[i]['exp_avg'], 1e-5, 'exp_avg %d' % i)
        P.check(sd[i]['exp_avg_sq'], rsd[i]['exp_avg_sq'], 1e-5, 'exp_avg_sq %d' % i)


def test_grad_slots_die_with_their_parameters(emulated_kernels):
    """hip.functional._GRAD_SLOTS is keyed by id(parameter): entries must vanish with the parameter / optimizer, a newer
    optimizer over the same parameters must win, and a stale id must never match another tensor."""
    import gc
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    from packnet_sfm.rccl.flat_adam import FlatAdam
    base = len(HF._GRAD_SLOTS)
    m = Conv2D(4, 32, 3, 1)
    opt1 = FlatAdam([{'params': list(m.parameters()), 'lr': 1e-2}])
    w = m.conv_base.weight
    assert HF._slot_of(w).data_ptr() == opt1.grad_view(opt1.param_groups[0], w).data_ptr()
    opt2 = FlatAdam([{'params': list(m.parameters()), 'lr': 1e-2}])       # re-registration: the newer arena wins ...
    assert HF._slot_of(w).data_ptr() == opt2.grad_view(opt2.param_groups[0], w).data_ptr()
    del opt1
    gc.collect()                                                          # ... and the older optimizer's finalizer leaves it alone
    assert HF._slot_of(w).data_ptr() == opt2.grad_view(opt2.param_groups[0], w).data_ptr()
    x = torch.randn(1, 4, 6, 8)
    m(x).pow(2).mean().backward()
    assert w.grad.data_ptr() == opt2.grad_view(opt2.param_groups[0], w).data_ptr()
    key = id(w)
    ent = HF._GRAD_SLOTS[key]
    other = torch.nn.Parameter(torch.zeros(32, 4, 3, 3))
    HF._GRAD_SLOTS[id(other)] = ent                                       # simulate id reuse: an entry whose weakref is another tensor
    assert HF._slot_of(other) is None
    del HF._GRAD_SLOTS[id(other)]
    del w, ent, opt2, m
    gc.collect()
    assert len(HF._GRAD_SLOTS) == base, 'slots outlived their parameters'


def test_conv2d_bx3_edge_inputs_emulated(emulated_kernels):
    """Same contract as tests/test_gpu_round3.py::test_conv2d_bx3_edge_inputs on the host-emulated kernel sources: values the
    split cannot represent (+-inf, NaN, |x| >= 3.39e38) poison exactly their receptive field with non-finite outputs."""
    from packnet_sfm.hip import ops
    B, Cin, Cout, H, W, ks = 1, 16, 32, 6, 32, 3
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    wf, _ = ops.conv2d_pack(w)
    clean = ops.conv2d_forward(x, wf, None, Cout, ks)
    P.check(clean, torch.nn.functional.conv2d(x, w, padding=1), 1e-5, 'clean run')
    for bad in (float('inf'), float('-inf'), float('nan'), 3.4e38):
        xb = x.clone()
        xb[0, 5, 2, 11] = bad
        y = ops.conv2d_forward(xb, wf, None, Cout, ks)
        touched = torch.zeros(H, W, dtype=torch.bool)
        touched[1:4, 10:13] = True
        assert not torch.isfinite(y[0][:, touched]).any(), bad
        assert torch.equal(y[0][:, ~touched], clean[0][:, ~touched]), bad


@pytest.mark.parametrize('shape', [(3, 16, 2, 4), (2, 32, 4, 6), (3, 48, 5, 7), (1, 16, 64, 80), (2, 64, 12, 40), (5, 32, 24, 80)])
@pytest.mark.parametrize('act,use_res', [(1, False), (2, True), (0, True)])
def test_groupnorm_two_launch_form(emulated_kernels, shape, act, use_res):
    """GroupNorm(16) + activation (+ residual) forward AND backward vs torch on shapes that exercise every work split of
    csrc/groupnorm.hip: one lane per row with surplus rows in the workgroup (2x4 maps), scalar loads (5x7), rows cut into chunks
    (64x80), rows of 128/256 lanes (barrier-based row sums), many channels per group -- since round 3 the apply kernels add the
    partial slots themselves (no gn_finish / gn_bwd_group launches)."""
    from packnet_sfm.hip import _lib
    _lib.get().pnsfm_set_gn_fused(0)          # the two-launch kernels (the large maps' path), also on the slabs the one-launch form takes
    P.case_groupnorm('cpu', shape, act, use_res)


# slab = (C / 16) * H * W floats: one float4 per thread of a 256-thread workgroup up to 16 per thread of a 1024-thread one; parameter
# blocks with 64 .. 1024 threads per channel and surplus rows (C = 48: three channels per group)
@pytest.mark.parametrize('shape', [(3, 16, 2, 4), (2, 32, 4, 6), (2, 64, 12, 40), (5, 32, 24, 80), (2, 48, 6, 20), (1, 16, 96, 320),
                                   (3, 128, 12, 40), (1, 32, 130, 128)])
@pytest.mark.parametrize('act,use_res', [(1, False), (2, True), (0, True)])
def test_groupnorm_one_launch_form(emulated_kernels, shape, act, use_res):
    """Round 6 (csrc/groupnorm.hip: gn_fused_fwd_kernel / gn_fused_bwd_kernel): a workgroup owns a (sample, group) slab -- statistics,
    normalisation and activation in ONE launch, backward with slab blocks (dx) and channel blocks (dgamma, dbeta) in one launch --
    forward AND backward vs torch, the tolerances of the two-launch form."""
    from packnet_sfm.hip import _lib
    assert _lib.get().pnsfm_set_gn_fused(1) in (0, 1)
    P.case_groupnorm('cpu', shape, act, use_res)


def _check_conv2d_cat(device, B, Cs, Cout, H, W, ks, seed):
    """conv(cat(xs)) folded into the K loop (hip.functional.conv2d_cat) vs F.conv2d on the concatenated tensor: output, the gradient of
    every input tensor, weight and bias gradients."""
    import torch.nn.functional as F
    from packnet_sfm.hip import functional as HF
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, c, H, W, generator=g) for c in Cs]
    w = torch.randn(Cout, sum(Cs), ks, ks, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xr = [t.clone().requires_grad_(True) for t in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(torch.cat(xr, 1), wr, br, padding=ks // 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xh = [t.clone().to(device).requires_grad_(True) for t in xs]
    wh, bh = w.clone().to(device).requires_grad_(True), b.clone().to(device).requires_grad_(True)
    y = HF.conv2d_cat(tuple(xh), wh, bh, HF.PackedConvWeight())
    y.backward(dy.to(device))
    P.check(y, yr, 1e-5, 'fwd')
    for i, (a, r) in enumerate(zip(xh, xr)):
        P.check(a.grad, r.grad, 1e-5, 'd input %d' % i)
    P.check(wh.grad, wr.grad, 2e-5, 'wgrad')
    P.check(bh.grad, br.grad, 2e-5, 'dbias')


@pytest.mark.parametrize('case', [(2, (32, 32, 1), 16, 6, 20, 3, 0), (1, (64, 32), 40, 5, 8, 3, 1), (1, (16, 5), 32, 4, 32, 5, 2),
                                  (1, (32, 64, 3), 8, 3, 16, 3, 3), (1, (8, 8), 16, 4, 8, 3, 4), (1, (64, 1), 32, 6, 24, 3, 5),
                                  (2, (32, 33), 16, 5, 16, 3, 6)])
def test_conv2d_cat_multi_source(emulated_kernels, case):
    """The decoder's cat(unpacked, skip[, upsampled disparity]) as a multi-source K loop: three and two tensors, a ragged last tensor, a
    first tensor of 16 channels (forward folds, the weight gradient falls back to one concatenation: 32-channel granule), and a shape
    outside the envelope (8-channel tensors: plain torch.cat path); two tensors whose SECOND is ragged (PackNet01 version '1B':
    (up + skip, upsampled inverse depth) = 64 + 1 channels) -- the end of the channels is not a boundary, the multi-source weight
    gradient takes them (round 4: the guard used to demand a 32-channel boundary there and the caller fell back silently)."""
    _check_conv2d_cat('cpu', *case)
    if case[1] in ((64, 1), (32, 33)):
        from packnet_sfm.hip import ops
        assert ops.conv2d_cat_wgrad_supported(list(case[1]), case[2], case[3], case[4], case[5], B=case[0])


@pytest.mark.parametrize('shape,cfg', [((2, 48, 64, 9, 32, 3), c) for c in [(2, 3, 0, 1), (1, 4, 1, 3), (2, 7, 0, 1), (1, 7, 1, 2), (2, 6, 0, 1)]] +
                         [((2, 40, 33, 5, 24, 1), (2, 7, 0, 1)), ((2, 40, 33, 5, 24, 1), (1, 4, 1, 3)),
                          ((1, 32, 40, 8, 32, 7), (1, 7, 1, 2)), ((1, 32, 40, 8, 32, 7), (2, 3, 0, 1))])
def test_conv2d_backward_data_addend(emulated_kernels, shape, cfg):
    """Round 5: dx = backward-data + addend in the launch's epilogue (un-split) / in the second stage of a K-split launch, for a dense
    addend and for a channel slice of a wider tensor: the bits of the separate elementwise sum."""
    import ctypes
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    NT, variant, narrow, split = cfg
    lib.pnsfm_set_conv_math(1)
    B, Cin, Cout, H, W, ks = shape
    key = (ctypes.c_int * 7)(1 + 10 + 100, B, Cout, Cin, H, W, ks)
    assert lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8), split) == 0
    g = torch.Generator().manual_seed(sum(shape) + sum(cfg))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    _, wb = ops.conv2d_pack(w)
    dy = torch.randn(B, Cout, H, W, generator=g)
    wide = torch.randn(B, Cin + 7, H, W, generator=g)
    try:
        plain = ops.conv2d_backward_data(dy, wb, Cin, ks)
        for addend in (wide[:, :Cin].contiguous(), wide[:, 5:5 + Cin], wide[:, 5:5 + Cin].permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)):
            got = ops.conv2d_backward_data(dy, wb, Cin, ks, addend=addend)
            assert torch.equal(got, plain + addend)
    finally:
        lib.pnsfm_set_conv_variant(0)      # clears the pinned entries


def test_gradient_taps_match_the_plain_graph(emulated_kernels):
    """Round 5: ResidualConv / Conv2D / UnpackLayerConv3d with gradient taps (the input's other consumers' gradient added inside the
    convolution's backward-data launch) against the same modules on the plain autograd graph: outputs equal, every gradient close
    (the three-consumer case associates the sum differently), and no zero tensor is materialised for an unused tap."""
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers.packnet.layers01 import ResidualBlock, Conv2D, UnpackLayerConv3d
    torch.manual_seed(3)
    blk, c2d, unp = ResidualBlock(32, 32, 2, 1), Conv2D(32, 32, 3, 1), UnpackLayerConv3d(32, 32, 3)
    x0 = torch.randn(1, 32, 6, 8)
    res = []
    for taps in (True, False):
        HF.set_grad_taps(taps)
        try:
            for m in (blk, c2d, unp):
                m.zero_grad()
            x = x0.clone().requires_grad_(True)
            h = x * 1.0                                     # a non-leaf producer, as in the network
            y, h_s = blk.forward_tap(h)                     # encoder stage + its skip
            z, y_t = c2d.forward_tap(y)                     # tensor with two consumers
            u, z_t = unp.forward_tap(z)
            loss = (u * u).sum() + (h_s * 0.3).sum() + (y_t * y_t).sum() * 0.1 + z_t.sum() * 0.2
            loss.backward()
            res.append((loss.detach(), x.grad.clone(), [p.grad.clone() for m in (blk, c2d, unp) for p in m.parameters()]))
        finally:
            HF.set_grad_taps(True)
    (l1, g1, p1), (l0, g0, p0) = res
    assert torch.equal(l1, l0)
    P.check(g1, g0, 1e-5, 'd input')
    for a, b in zip(p1, p0):
        P.check(a, b, 1e-5, 'parameter gradient', floor=1e-6)
    # an unused tap costs nothing: the block alone (tap output dropped) still trains
    x = x0.clone().requires_grad_(True)
    y, _ = blk.forward_tap(x * 1.0)
    y.sum().backward()
    assert x.grad is not None


def test_groupnorm_one_launch_form_other_group_counts(emulated_kernels):
    """Four workgroups per slab with a slab count that is NOT a multiple of 8 (3 samples x 4 groups: the parts of a slab fall back to
    consecutive blocks), and one channel per group."""
    from packnet_sfm.hip import _lib
    _lib.get().pnsfm_set_gn_fused(1)
    P.case_groupnorm('cpu', (3, 16, 64, 64), 1, True, G=4)
    P.case_groupnorm('cpu', (2, 8, 8, 16), 2, False, G=8)


def test_block_sequencer_equals_python_bodies(emulated_kernels):
    """The block sequencer (csrc/seq/pnsfm_seq.cpp) bound to the EMULATED kernels against the pure-Python bodies of hip/functional.py on a
    stack of real blocks (Conv2D with a multi-source input, ResidualConv with taps, collapsed PackLayerConv3d, UnpackLayerConv3d): the
    same launches, so outputs and every gradient agree bit for bit."""
    from packnet_sfm.hip import _seq
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, ResidualConv, PackLayerConv3d, UnpackLayerConv3d
    torch.manual_seed(7)
    blocks = [Conv2D(32, 32, 3, 1), ResidualConv(32, 32, 1), PackLayerConv3d(32, 3), UnpackLayerConv3d(32, 32, 3)]
    x0 = torch.randn(1, 16, 12, 32)
    x1 = torch.randn(1, 16, 12, 32)
    assert _seq.get() is not None
    res = {}
    try:
        for on in (False, True):
            _seq.set_enabled(on)
            for m in blocks:
                for p in m.parameters():
                    p.grad = None
            a, b = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
            y = blocks[0]((a, b))
            y = blocks[1](y)
            y = blocks[3](blocks[2](y))
            (y * torch.linspace(0.5, 1.5, y.numel()).view_as(y)).sum().backward()
            res[on] = [y.detach().clone(), a.grad.clone(), b.grad.clone()] + [p.grad.clone() for m in blocks for p in m.parameters()]
    finally:
        _seq.set_enabled(True)
    assert len(res[False]) == len(res[True])
    for i, (u, v) in enumerate(zip(res[False], res[True])):
        assert torch.equal(u, v), 'tensor %d differs between the Python bodies and the sequencer (max |d| %.3e)' % (i, float((u - v).abs().max()))
