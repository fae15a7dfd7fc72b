"""GPU (MI355X): round-6 parity pins.

The ping-pong workgroup (csrc/conv2d_bx3pp.h, tuner variant 7) pinned ON THE DEVICE against the oracle's convolution.  The host
emulator (tests/test_kernels_emulated.py) runs workgroups serially and cannot see what is new in this kernel: the barrier in front
of the last tap's MFMAs, the in-place overwrite of a group's single patch buffer, lgkmcnt-only barriers with LDS-DMA in flight,
zero-filled ragged stages, the idle second group of an odd tile count.  Every case pins the configuration through pnsfm_tune_set and
asserts with pnsfm_conv2d_last_config that variant 7 is what ran (a pin that does not fit a shape falls back silently).
Reference op: nn.Conv2d of packnet_sfm/networks/layers/packnet/layers01.py:28-36 (+ its autograd backward-data)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import parity_cases as P

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from packnet_sfm.hip import _lib
    assert _lib.get().pnsfm_build_target() == b'gfx950'
    assert _lib.REQUIRE_CUDA


def _cfg(NT, variant, narrow=0, tm=0):
    return NT | (variant << 4) | (narrow << 8) | (tm << 9)


def _pin(lib, kind, B, K, M, H, W, ks, v0, split):
    key = (ctypes.c_int * 7)(kind + 10 + 100, B, K, M, H, W, ks)
    assert lib.pnsfm_tune_set(key, v0, split) == 0


def _last(lib):
    out = (ctypes.c_int * 8)()
    assert lib.pnsfm_conv2d_last_config(out) == 0
    return dict(zip(('variant', 'NT', 'MT', 'G', 'split', 'tm', 'blocks', 'lds'), list(out)))


def _data(shape, seed_extra=0):
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape) + seed_extra)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    return x, w, b, dy


# (shape, (NT, narrow, tile mode), K-split forward, K-split backward-data).  The first five are the shipped database's own variant-7
# decisions at BASELINE.json configs[1] (csrc/tuned_gfx950.db); then an odd number of pixel tiles (group 1 of the last workgroup
# idles), a 7x7 whose 49 taps leave a ragged last stage at G = 3 on a map with ragged tiles, the 32-row M tile and a 1-round launch.
PP_CASES = [
    ((4, 64, 64, 192, 640, 7), (2, 0, 1), 1, 1),        # conv1 -- 626 1 in the database
    ((4, 256, 64, 96, 320, 7), (2, 0, 1), 1, 1),        # pack1 collapsed 7x7
    ((4, 8192, 256, 12, 40, 3), (2, 0, 2), 16, 1),      # pack4.conv, K split 16 -- 1138 16
    ((4, 512, 128, 24, 80, 5), (2, 0, 2), 8, 2),        # pack3 collapsed 5x5, split 8 -- 1138 8
    ((4, 129, 64, 192, 640, 3), (2, 0, 0), 1, 1),       # iconv1 (129 K-channels: a nearly empty ninth chunk)
    ((4, 256, 256, 24, 80, 3), (1, 1, 1), 1, 1),        # conv4 body -- 881 1 (NT 1, 32-row tiles, rectangles)
    ((3, 64, 64, 36, 96, 3), (2, 0, 0), 1, 1),          # 3 x ceil(36/8) x 3 = 45 pixel tiles: odd -> the last workgroup's group 1 idles
    ((1, 64, 64, 40, 80, 3), (2, 0, 1), 1, 1),          # 5 x 3 = 15 rectangle tiles: odd
    ((1, 48, 64, 28, 72, 7), (2, 0, 1), 1, 1),          # 7x7, G = 3: 16 full stages + 1 ragged; ragged tiles in both directions
    ((2, 112, 64, 13, 40, 5), (2, 0, 2), 3, 2),         # 5x5 row bands, ragged last band, K split 3 of 7 chunks (3 + 3 + 1)
    ((2, 512, 512, 12, 40, 3), (2, 0, 2), 8, 8),        # conv5 body
]


@pytest.mark.parametrize('case', PP_CASES, ids=lambda c: 'x'.join(map(str, c[0])))
def test_pingpong_kernel_vs_cpu_oracle(case):
    """Forward AND backward-data of the ping-pong kernel against the oracle's convolution at 2e-5 (the tolerance of the other
    variants in test_gpu_parity.py::test_conv2d_vs_cpu_oracle), on the configurations the shipped database takes."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    shape, (NT, narrow, tm), split_f, split_b = case
    B, Cin, Cout, H, W, ks = shape
    x, w, b, dy = _data(shape)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    yr.backward(dy)
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    wf, wb = ops.conv2d_pack(wd)
    try:
        _pin(lib, 0, B, Cin, Cout, H, W, ks, _cfg(NT, 7, narrow, tm), split_f)
        _pin(lib, 1, B, Cout, Cin, H, W, ks, _cfg(NT, 7, narrow, tm), split_b)
        y = ops.conv2d_forward(xd, wf, bd, Cout, ks)
        c = _last(lib)
        assert c['variant'] == 7 and c['NT'] == NT and c['split'] == split_f, c
        dx = ops.conv2d_backward_data(dyd, wb, Cin, ks)
        c = _last(lib)
        assert c['variant'] == 7 and c['split'] == split_b, c
        P.check(y, yr, 2e-5, 'fwd (ping-pong)')
        P.check(dx, xr.grad, 2e-5, 'dgrad (ping-pong)')
    finally:
        lib.pnsfm_set_conv_variant(3)      # clears the pinned entries


@pytest.mark.parametrize('channels', [(64, 128, 1), (64, 64, 1), (32, 32, 0)])
def test_pingpong_kernel_multi_source(channels):
    """The decoder's concatenations folded into the K loop (pnsfm_conv2d_forward_cat: iconv3 = cat(unpack3, skip3, up(disp4)),
    PackNet01.py:150-168) through the ping-pong kernel against conv(cat(...)) on the CPU."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    C = [c for c in channels if c]
    Cin, Cout, B, H, W, ks = sum(C), 128, 2, 48, 160, 3
    g = torch.Generator().manual_seed(Cin)
    xs = [torch.randn(B, c, H, W, generator=g) for c in C]
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g)
    yr = F.conv2d(torch.cat(xs, 1), w, b, padding=1)
    wf, _ = ops.conv2d_pack(w.to(DEV), want_bwd=False)
    try:
        _pin(lib, 0, B, Cin, Cout, H, W, ks, _cfg(2, 7), 2)
        y = ops.conv2d_forward_cat([t.to(DEV) for t in xs], wf, b.to(DEV), Cout, ks)
        c = _last(lib)
        assert c['variant'] == 7 and c['split'] == 2, c
        P.check(y, yr, 2e-5, 'fwd cat (ping-pong)')
    finally:
        lib.pnsfm_set_conv_variant(3)


@pytest.mark.parametrize('shape,pp', [((1, 64, 64, 48, 160, 7), (2, 0, 1, 1)), ((1, 64, 64, 48, 160, 7), (2, 0, 1, 4)),
                                      ((1, 2048, 64, 24, 80, 5), (2, 0, 2, 4)), ((4, 512, 512, 6, 20, 3), (1, 1, 2, 4))])
def test_pingpong_error_vs_fp64(shape, pp):
    """The fp64 error study of test_conv2d_bx3_error_vs_fp64 for variant 7, with both arithmetics pinned to the SAME K split: the
    error of either kernel follows the length of its fp32 accumulation chain (profiles/r06_pp_err_probe.txt: 5.8e-7 / 2.6e-7 /
    2.1e-7 of sum |x||w| for the f32-MFMA kernel of the 7x7 shape at K split 1 / 2 / 4, 4.9e-7 / 2.4e-7 / 2.0e-7 for every split-bf16
    variant), so a comparison across different splits says nothing about the arithmetic.  At equal split the six-product arithmetic
    must stay within 1.25x of the f32 instruction's error (measured 0.55-0.94x) and below 1e-6 outright; plain bf16 sits at ~4e-4."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    NT, narrow, tm, split = pp
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))      # mixed magnitudes
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    dy = torch.randn(B, Cout, H, W, generator=g)
    y64 = F.conv2d(x.double(), w.double(), padding=ks // 2)
    ymag = F.conv2d(x.double().abs(), w.double().abs(), padding=ks // 2)
    dx64 = F.conv_transpose2d(dy.double(), w.double(), padding=ks // 2)
    dxmag = F.conv_transpose2d(dy.double().abs(), w.double().abs(), padding=ks // 2)
    err = {}
    try:
        for mode in ('f32', 'pp'):
            HF.set_conv_math('f32' if mode == 'f32' else 'bx3')
            for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
                key = (ctypes.c_int * 7)(kind + 10 + (100 if mode == 'pp' else 0), B, K, M, H, W, ks)
                v0 = _cfg(NT, 7, narrow, tm) if mode == 'pp' else _cfg(1, 0)
                assert lib.pnsfm_tune_set(key, v0, split) == 0
            wf, wb = ops.conv2d_pack(w.to(DEV))
            y = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks).cpu().double()
            c = _last(lib)
            assert c['variant'] == (7 if mode == 'pp' else 0) and c['split'] == split, c
            dx = ops.conv2d_backward_data(dy.to(DEV), wb, Cin, ks).cpu().double()
            c = _last(lib)
            assert c['variant'] == (7 if mode == 'pp' else 0) and c['split'] == split, c
            err[mode] = (float(((y - y64).abs() / ymag).max()), float(((dx - dx64).abs() / dxmag).max()))
    finally:
        HF.set_conv_math('bx3')
        lib.pnsfm_set_conv_variant(0)
        lib.pnsfm_set_conv_variant(3)
    print('K split %d: max |err| / sum|a||b|  (fwd, dgrad):  f32 MFMA %.2e %.2e   ping-pong %.2e %.2e   [2^-24 = 5.96e-08]'
          % ((split,) + err['f32'] + err['pp']))
    for i in range(2):
        assert err['pp'][i] <= max(1.25 * err['f32'][i], 1.5e-7), err
        assert err['pp'][i] <= 1e-6, err


@pytest.mark.parametrize('shape,NT,tm,split', [((4, 64, 64, 96, 320, 3), 2, 0, 1), ((2, 64, 64, 48, 160, 7), 2, 1, 1),
                                                ((4, 512, 128, 24, 80, 5), 2, 2, 8), ((2, 256, 256, 24, 80, 3), 1, 1, 1)])
def test_pingpong_bit_identical_to_single_tile_kernel(shape, NT, tm, split):
    """conv2d_bx3pp.h's header: "bit-identical results for the same chunk order" -- a pixel tile accumulates the same piece products
    in the same order (chunks ascending, taps ascending, l-m-h pieces) as conv2d_bx3_kernel does with the same (NT, tile mode,
    K-split): torch.equal between variant 7 and variant 3 on the device."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    x, w, b, dy = _data(shape, 3)
    xd, bd, dyd = x.to(DEV), b.to(DEV), dy.to(DEV)
    wf, wb = ops.conv2d_pack(w.to(DEV))
    narrow = 1 if NT == 1 else 0
    out = {}
    try:
        for variant in (3, 7):
            _pin(lib, 0, B, Cin, Cout, H, W, ks, _cfg(NT, variant, narrow, tm), split)
            _pin(lib, 1, B, Cout, Cin, H, W, ks, _cfg(NT, variant, narrow, tm), split)
            y = ops.conv2d_forward(xd, wf, bd, Cout, ks)
            assert _last(lib)['variant'] == variant, _last(lib)
            dx = ops.conv2d_backward_data(dyd, wb, Cin, ks)
            assert _last(lib)['variant'] == variant, _last(lib)
            out[variant] = (y, dx)
    finally:
        lib.pnsfm_set_conv_variant(3)
    assert torch.equal(out[3][0], out[7][0]), 'forward: max |d| %.3e' % float((out[3][0] - out[7][0]).abs().max())
    assert torch.equal(out[3][1], out[7][1]), 'backward-data: max |d| %.3e' % float((out[3][1] - out[7][1]).abs().max())


@pytest.mark.parametrize('shape,NT,tm,split', [((4, 64, 64, 192, 640, 7), 2, 1, 1), ((4, 8192, 256, 12, 40, 3), 2, 2, 16),
                                                ((3, 64, 64, 36, 96, 3), 2, 0, 1)])
def test_pingpong_repeated_launches_are_bit_identical(shape, NT, tm, split):
    """Races are intermittent: 50 launches of one shape (other kernels of the library in between, so that the workgroups meet
    different neighbours and LDS contents) must all return the bits of the first."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    x, w, b, dy = _data(shape, 5)
    xd, bd = x.to(DEV), b.to(DEV)
    wf, _ = ops.conv2d_pack(w.to(DEV), want_bwd=False)
    noise = torch.randn(2, 64, 48, 160, device=DEV)
    nwf, _ = ops.conv2d_pack(torch.randn(64, 64, 3, 3, device=DEV), want_bwd=False)
    try:
        _pin(lib, 0, B, Cin, Cout, H, W, ks, _cfg(NT, 7, 0, tm), split)
        y0 = ops.conv2d_forward(xd, wf, bd, Cout, ks)
        assert _last(lib)['variant'] == 7
        bad = 0
        for it in range(50):
            if it % 3 == 0:
                ops.conv2d_forward(noise, nwf, None, 64, 3)          # leaves other data in LDS / other weights in L2
            y = ops.conv2d_forward(xd, wf, bd, Cout, ks)
            bad += int(not torch.equal(y, y0))
        assert bad == 0, '%d of 50 launches differ from the first' % bad
    finally:
        lib.pnsfm_set_conv_variant(3)
    # and the first launch is right
    P.check(y0, F.conv2d(x, w, b, padding=ks // 2), 2e-5, 'fwd')


# ------------------------------------------------------------------------------------------------ GroupNorm, one launch per direction
GN_SHAPES = [(4, 256, 24, 80), (4, 128, 48, 160), (4, 512, 12, 40), (4, 512, 6, 20), (4, 32, 96, 320), (4, 64, 24, 80), (2, 512, 24, 80),
             (4, 16, 96, 320), (4, 256, 3, 10), (3, 48, 6, 20)]


@pytest.mark.parametrize('fused', [1, 0])
@pytest.mark.parametrize('shape', GN_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('act,use_res', [(1, False), (1, True), (2, False)])
def test_groupnorm_one_launch_form_gpu(shape, act, use_res, fused):
    """csrc/groupnorm.hip, round 6: slabs of <= 64 K floats run statistics + normalise + activation in ONE launch and the whole
    backward in ONE launch (slab blocks: dx; channel blocks: dgamma / dbeta).  Both forms (fused = 1 / 0) against torch's group_norm
    on the CPU at the real PackNet01 / PoseNet layer shapes below the 96x320 level (reference layers01.py:31-37, 61-72;
    networks/pose/PoseNet.py:28-34)."""
    from packnet_sfm.hip import _lib
    lib = _lib.get()
    prev = lib.pnsfm_set_gn_fused(fused)
    try:
        P.case_groupnorm(DEV, shape, act, use_res, tol=2e-5)
    finally:
        lib.pnsfm_set_gn_fused(prev)


def test_groupnorm_one_launch_form_is_deterministic_and_close_to_the_two_launch_form():
    """Two runs of the one-launch kernels return the same bits; against the two-launch kernels only the fp64 summation ORDER differs."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    g = torch.Generator().manual_seed(11)
    B, C, H, W = 4, 256, 24, 80
    x, res, dy = (torch.randn(B, C, H, W, generator=g).to(DEV) for _ in range(3))
    ga, be = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    out = {}
    prev = lib.pnsfm_set_gn_fused(1)
    try:
        for fused in (1, 1, 0):
            lib.pnsfm_set_gn_fused(fused)
            y, mean, rstd = ops.groupnorm_act_forward(x, res, ga, be, 16, 1e-5, ops.ACT_ELU)
            dx, dga, dbe = ops.groupnorm_act_backward(dy, x, res, ga, be, mean, rstd, 16, ops.ACT_ELU)
            out.setdefault(fused, []).append((y, mean, rstd, dx, dga, dbe))
    finally:
        lib.pnsfm_set_gn_fused(prev)
    for a, b in zip(out[1][0], out[1][1]):
        assert torch.equal(a, b)
    for a, b, name in zip(out[1][0], out[0][0], ('y', 'mean', 'rstd', 'dx', 'dgamma', 'dbeta')):
        P.check(a, b, 2e-6, 'one-launch vs two-launch ' + name)


# ------------------------------------------------------------------------------------------------ block sequencer
def test_block_sequencer_is_bit_identical_to_the_python_bodies():
    """csrc/seq/pnsfm_seq.cpp holds the bodies of the hot autograd nodes (Conv2D block, plain / multi-source convolution, GroupNorm with
    residual, region_ops) as single C++ calls over the C ABI; hip/functional.py keeps a pure-Python body for each (PNSFM_SEQ=0).  Same
    launches in the same per-stream order: loss and every gradient of the golden training step agree BIT FOR BIT between the two."""
    from packnet_sfm.hip import _seq
    from test_gpu_parity import _selfsup, _step_batch
    from test_gpu_round4 import _grads_of_one_step
    fx = dict(P.golden('step')['step_flip0'])
    model, dn, pn = _selfsup(DEV, fx)
    batch = _step_batch(fx)
    assert _seq.get() is not None, 'the sequencer extension must be loaded on the GPU box'
    _grads_of_one_step(model, batch, False)             # autotuning
    try:
        _seq.set_enabled(False)
        assert _seq.get() is None
        l0, g0 = _grads_of_one_step(model, batch, False)
        _seq.set_enabled(True)
        for rep in range(2):
            l1, g1 = _grads_of_one_step(model, batch, False)
            assert torch.equal(l0, l1), (l0.item(), l1.item())
            bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
            assert not bad and g0.keys() == g1.keys(), bad[:3]
    finally:
        _seq.set_enabled(True)


@pytest.mark.parametrize('k', [3, 5])
def test_collapsed_pack_transposed_column_strips(k):
    """Round 6: the column strips of the collapsed PackLayerConv3d are stored transposed and convolved with (y, x)-swapped kernels
    (layers01.py: lr_transposed).  Forward and every gradient against the un-transposed form of the SAME module (which the reference
    goldens pin: test_pack_golden) on a map with distinct height and width."""
    from packnet_sfm.networks.layers.packnet.layers01 import PackLayerConv3d
    torch.manual_seed(11)
    m = PackLayerConv3d(16, k).to(DEV)
    m.collapse = True
    x = torch.randn(2, 16, 28, 72, device=DEV, requires_grad=True)
    res = {}
    for form in (False, True):
        m.lr_transposed = form
        for p in m.parameters():
            p.grad = None
        x.grad = None
        y = m(x)
        (y * torch.linspace(0.5, 1.5, y.numel(), device=DEV).view_as(y)).sum().backward()
        res[form] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    y0, dx0, g0 = res[False]
    y1, dx1, g1 = res[True]
    P.check(y1, y0, 1e-5, 'forward')
    P.check(dx1, dx0, 5e-5, 'dx')
    gmax = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        P.check(g1[n], g0[n], 1e-4, 'd' + n, floor=0.05 * gmax)


# ---- the 3-channel 5x5 stem kernel (conv2d.hip: conv2d_stem5_kernel; variant 0 of the f32 kernels on 32-multiple widths)
@pytest.mark.parametrize('NT', [1, 2])
@pytest.mark.parametrize('shape', [(4, 3, 64, 192, 640, 5), (2, 3, 32, 96, 320, 5), (1, 3, 40, 45, 96, 5), (3, 3, 64, 6, 32, 5)])
def test_stem_kernel_vs_cpu_oracle_and_generic_kernel(shape, NT):
    """The depth networks' first layer (PackNet01.py:42 `Conv2D(3, 64, 5, 1)`) on its own kernel: against the oracle's convolution at
    2e-5 (full size, PackNetSlim01's 32 channels, 40 channels = a padded 32-row M tile with 45 rows = ragged tile rows, a map of a single
    tile row) and BIT-identical to the generic f32 kernel (variant 1), whose non-zero terms it adds in the same order."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    x, w, b, _ = _data(shape, 3)
    yr = F.conv2d(x, w, b, padding=ks // 2)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    key = (ctypes.c_int * 7)(10, B, Cin, Cout, H, W, ks)
    lib.pnsfm_set_autotune(0)
    try:
        wf, _wb = ops.conv2d_pack(wd)
        assert lib.pnsfm_tune_set(key, _cfg(NT, 0), 1) == 0
        y0 = ops.conv2d_forward(xd, wf, bd, Cout, ks)
        c = _last(lib)
        MT = 2 if Cout > 32 else 1
        assert c['variant'] == 0 and c['NT'] == NT and c['lds'] == 4 * (76 * 32 * MT + 3 * (4 * NT + 4) * 36), c      # the stem kernel's LDS image
        assert lib.pnsfm_tune_set(key, _cfg(NT, 1), 1) == 0
        y1 = ops.conv2d_forward(xd, wf, bd, Cout, ks)
        assert _last(lib)['variant'] == 1
    finally:
        lib.pnsfm_set_conv_variant(0)
        lib.pnsfm_set_conv_variant(3)
        lib.pnsfm_set_autotune(1)
    P.check(y0, yr, 2e-5, 'stem fwd')
    assert torch.equal(y0, y1)


# ---- the LDS-free 1x1 kernel (csrc/conv2d_bx3_1x1.h, tuner variant 8): the shortcuts of the residual blocks (layers01.py:57-60)
def _flat32(H, W):
    return ((H * W) // 32, 32) if (H * W) % 32 == 0 and W % 32 != 0 else (H, W)       # launch_conv's 32-wide rows of a 1x1 layer


# (shape, (NT, narrow M), K split): the shipped database's decisions at BASELINE.json configs[1] / [2], then ragged pixel tiles and
# K chunks, padded M tiles, K splits (with fewer chunks per split than the prefetch depth)
C1_CASES = [
    ((4, 256, 256, 24, 80, 1), (1, 0), 1), ((4, 512, 512, 12, 40, 1), (1, 1), 1), ((4, 64, 64, 96, 320, 1), (2, 0), 1),
    ((4, 128, 256, 24, 80, 1), (1, 0), 1), ((2, 256, 256, 48, 160, 1), (2, 0), 1), ((2, 64, 64, 192, 640, 1), (2, 0), 1),
    ((3, 48, 33, 13, 19, 1), (2, 0), 1), ((2, 144, 70, 9, 35, 1), (1, 1), 4), ((4, 512, 512, 12, 40, 1), (2, 0), 16),
    ((1, 20, 40, 7, 5, 1), (1, 0), 1),
]


@pytest.mark.parametrize('case', C1_CASES, ids=lambda c: 'x'.join(map(str, c[0])) + '-s%d' % c[2])
def test_conv1x1_kernel_vs_cpu_oracle_and_lds_kernel(case):
    """Forward and backward-data (the latter also with an addend in the epilogue: the gradient taps of the residual blocks) of the
    LDS-free 1x1 kernel against the oracle's convolution at 2e-5, BIT-identical to conv2d_bx3_kernel (variant 3) at the same K split
    -- same six piece products per chunk in the same order -- and bit-identical over 20 repeated launches."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    shape, (NT, narrow), split = case
    B, Cin, Cout, H, W, ks = shape
    Hk, Wk = _flat32(H, W)
    x, w, b, dy = _data(shape, 8)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, b)
    yr.backward(dy)
    g = torch.Generator().manual_seed(5)
    add = torch.randn(B, Cin, H, W, generator=g)
    xd, wd, bd, dyd, addd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV), add.to(DEV)
    HF.set_conv_math('bx3')
    lib.pnsfm_set_autotune(0)
    out = {}
    try:
        wf, wb = ops.conv2d_pack(wd)
        for variant in (8, 3):
            _pin(lib, 0, B, Cin, Cout, Hk, Wk, ks, NT | (variant << 4) | (narrow << 8), split)
            _pin(lib, 1, B, Cout, Cin, Hk, Wk, ks, NT | (variant << 4) | (narrow << 8), split)
            y = ops.conv2d_forward(xd, wf, bd, Cout, ks)
            c = _last(lib)
            assert c['variant'] == variant and c['NT'] == NT, c
            dx = ops.conv2d_backward_data(dyd, wb, Cin, ks)
            assert _last(lib)['variant'] == variant
            dxa = ops.conv2d_backward_data(dyd, wb, Cin, ks, addend=addd)
            out[variant] = (y, dx, dxa)
            if variant == 8:
                for _ in range(20):
                    assert torch.equal(ops.conv2d_forward(xd, wf, bd, Cout, ks), y)
    finally:
        lib.pnsfm_set_conv_variant(3)
        lib.pnsfm_set_autotune(1)
    y, dx, dxa = out[8]
    P.check(y, yr, 2e-5, '1x1 fwd')
    P.check(dx, xr.grad, 2e-5, '1x1 dgrad')
    P.check(dxa, xr.grad + add, 2e-5, '1x1 dgrad + addend')
    for got, ref, what in zip(out[8], out[3], ('fwd', 'dgrad', 'dgrad + addend')):
        assert torch.equal(got, ref), what


@pytest.mark.parametrize('shape,split', [((4, 3, 64, 192, 640, 5), 256), ((2, 3, 32, 96, 320, 5), 512), ((1, 3, 40, 45, 72, 5), 7), ((3, 3, 64, 8, 64, 5), 1)])
def test_stem_weight_gradient_kernel_gpu(shape, split):
    """The stem's weight gradient on conv2d_wgrad_stem5_kernel (split-bf16 arithmetic, 16 pixels per k-step): against the oracle at 5e-5,
    against fp64 with the error bound of the other split-bf16 weight gradients (<= 16 * 2^-24 of sum |dy||x|), the same bits from
    repeated launches; pnsfm_conv2d_last_config proves the kernel ran.  Reference: the autograd weight gradient of PackNet01.py:42."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    x, w, b, dy = _data(shape, 4)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv2d(x, wr, br, padding=ks // 2).backward(dy)
    xd, dyd = x.to(DEV), dy.to(DEV)
    ref64 = torch.nn.grad.conv2d_weight(xd.double(), (Cout, Cin, ks, ks), dyd.double(), padding=ks // 2)
    mag = torch.nn.grad.conv2d_weight(xd.double().abs(), (Cout, Cin, ks, ks), dyd.double().abs(), padding=ks // 2)
    HF.set_conv_math('bx3')
    lib.pnsfm_set_autotune(0)
    try:
        key = (ctypes.c_int * 7)(12, B, Cin, Cout, H * W, W, ks)
        assert lib.pnsfm_tune_set(key, split, 0) == 0
        dw, db = ops.conv2d_backward_weight(xd, dyd, ks)
        assert _last(lib)['variant'] == 105
        for _ in range(10):
            dw2, db2 = ops.conv2d_backward_weight(xd, dyd, ks)
            assert torch.equal(dw, dw2) and torch.equal(db, db2)
    finally:
        lib.pnsfm_set_wgrad_variant(-1)
        lib.pnsfm_set_autotune(1)
    P.check(dw, wr.grad, 5e-5, 'stem wgrad')
    P.check(db, br.grad, 5e-5, 'stem dbias')
    assert float(((dw.double() - ref64).abs() / mag).max()) <= 16 * 2.0 ** -24
