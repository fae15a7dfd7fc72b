"""GPU (MI355X): round-4 properties of the training step.

  (a) bit-reproducibility: since round 4 no kernel on the self-supervised step meets partial sums with atomics (K-split forward /
      backward-data, pixel-split weight gradients, Conv3d weight gradient, InvDepth head, pose gradient all finish in a second,
      fixed-order stage), so two executions of one step from one state must agree BIT FOR BIT in the loss and in every gradient --
      at the golden step's size and at BASELINE.json's 192x640 batch 4, where the K-split and pixel-split layers are the real ones;
  (b) the K-split forward / backward-data path against an fp64 convolution at a real split layer shape.
Reference semantics: packnet_sfm/models/SelfSupModel.py:63-97 (the step), trainers/horovod_trainer.py:85-93 (backward + optimizer)."""
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


def _grads_of_one_step(model, batch, flip):
    for p in model.parameters():
        p.grad = None
    model._flip_override = flip
    out = model(batch, progress=0.0)
    model._flip_override = None
    out['loss'].backward()
    torch.cuda.synchronize()
    return out['loss'].detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _assert_bitwise_equal_runs(model, batch, flips, what):
    _grads_of_one_step(model, batch, False)            # autotuning (which times candidates) happens here, not between the runs
    for flip in flips:
        l0, g0 = _grads_of_one_step(model, batch, flip)
        for rep in range(2):
            l1, g1 = _grads_of_one_step(model, batch, flip)
            assert torch.equal(l0, l1), '%s: loss differs between two runs of one step (flip=%s): %r vs %r' % (what, flip, l0.item(), l1.item())
            assert g0.keys() == g1.keys()
            bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
            assert not bad, '%s: %d gradients differ between two runs of one step (flip=%s), e.g. %s (max |d| %.3e)' % (
                what, len(bad), flip, bad[0], float((g0[bad[0]] - g1[bad[0]]).abs().max()))


def test_deterministic_step_golden_size():
    """Two executions of the golden training step (PackNet01 + PoseNet + loss, forward + backward) are bit-identical."""
    from test_gpu_parity import _selfsup, _step_batch
    fx = dict(P.golden('step')['step_flip0'])
    model, dn, pn = _selfsup(DEV, fx)
    _assert_bitwise_equal_runs(model, _step_batch(fx), (False, True), 'golden-size step')


def test_deterministic_step_full_size():
    """The same at BASELINE.json configs[1]: 192x640, batch 4 -- the shapes whose low-resolution layers really split K (pack4 /
    pack5, conv4 / conv5) and whose weight gradients really split pixels."""
    import sys
    import os
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
    import bench
    model = bench.build_model(torch.device(DEV))
    batch = bench.synthetic_batch(4, 192, 640, 1234, DEV)
    _assert_bitwise_equal_runs(model, batch, (False,), '192x640 batch-4 step')


@pytest.mark.parametrize('shape', [(4, 512, 512, 12, 40, 3), (2, 1024, 256, 6, 20, 3)])
@pytest.mark.parametrize('split', [2, 5])
def test_conv2d_split_k_two_stage_vs_fp64(shape, split):
    """K-split forward / backward-data (partial outputs in the stream's scratch buffer, conv_splitk_reduce_kernel adds them in
    split order, bias included) at a real low-resolution layer shape, pinned through pnsfm_tune_set: error against an fp64
    convolution in the class of the un-split kernel (1e-6 of sum |x||w|), and bit-identical between two launches."""
    import ctypes
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    dy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    wf, wb = ops.conv2d_pack(w)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    dx64 = F.conv_transpose2d(dy.double(), w.double(), padding=ks // 2)
    scale_y = F.conv2d(x.abs().double(), w.abs().double(), padding=ks // 2).max()
    scale_dx = F.conv_transpose2d(dy.abs().double(), w.abs().double(), padding=ks // 2).max()
    try:
        for kind, K, M in ((0, Cin, Cout), (1, Cout, Cin)):
            key = (ctypes.c_int * 7)(kind + 10 + 100, B, K, M, H, W, ks)
            assert lib.pnsfm_tune_set(key, 2 | (3 << 4), split) == 0
        y = ops.conv2d_forward(x, wf, b, Cout, ks)
        dx = ops.conv2d_backward_data(dy, wb, Cin, ks)
        assert float((y.double() - y64).abs().max() / scale_y) < 1e-6
        assert float((dx.double() - dx64).abs().max() / scale_dx) < 1e-6
        assert torch.equal(y, ops.conv2d_forward(x, wf, b, Cout, ks))
        assert torch.equal(dx, ops.conv2d_backward_data(dy, wb, Cin, ks))
    finally:
        lib.pnsfm_set_conv_variant(3)      # clears the pinned entries


def test_region_ops_batched_windows_gpu():
    """pnsfm_region_ops on the device (the collapsed packing block's strip plumbing, one launch per autograd Function)."""
    P.case_region_ops(DEV)


def test_branch_stream_is_bit_identical():
    """The pose network on the second compute stream (models/SfmModel.py, hip/functional.py: branch_stream) against everything on
    one stream: the same kernels in the same per-stream order, so loss and every gradient must agree bit for bit -- anything else
    is a missing stream dependency."""
    from packnet_sfm.hip import functional as HF
    from test_gpu_parity import _selfsup, _step_batch
    fx = dict(P.golden('step')['step_flip0'])
    model, dn, pn = _selfsup(DEV, fx)
    batch = _step_batch(fx)
    _grads_of_one_step(model, batch, False)             # autotuning
    try:
        HF.set_branch_stream(False)
        l0, g0 = _grads_of_one_step(model, batch, False)
        HF.set_branch_stream(True)
        assert HF.branch_stream(batch['rgb']) is not None
        for rep in range(3):
            l1, g1 = _grads_of_one_step(model, batch, False)
            assert torch.equal(l0, l1), (l0.item(), l1.item())
            bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
            assert not bad and g0.keys() == g1.keys(), bad[:3]
    finally:
        HF.set_branch_stream(False)


def test_upsample_nearest_gpu():
    """Nearest up-sampling kernels (every predicted scale brought to full resolution; the inverse depth handed to the next iconv
    block) vs torch's interpolate and its gradient."""
    P.case_upsample_nearest(DEV)


def test_loss_combine_gpu():
    """The scalar tail of the multi-view photometric loss as one launch each way: bit-equal to the Python sums of 0-dim tensors."""
    P.case_loss_combine(DEV)


def test_compose_pack_params_gpu():
    """Composed packing convolution: kernel, bias and all four parameter gradients from one node vs the oracle's formula."""
    P.case_compose_pack_params(DEV)


@pytest.mark.parametrize('hw', [(8, 26), (6, 14)])
def test_collapsed_pack_on_maps_smaller_than_two_strips(hw):
    """for S <= h < 2S (S = 2(k//2)+1 packed rows) the leading and the trailing border strip of the collapsed packing
    block overlap, so their gradient adds into dP must not share a region_ops launch (one launch = single-writer destinations).
    collapse=True against the reference form (layers01.py:239-247) of the SAME module, forward and every gradient."""
    from packnet_sfm.networks.layers.packnet.layers01 import PackLayerConv3d
    torch.manual_seed(3)
    for k in (3, 5):
        m = PackLayerConv3d(16, k).to(DEV)
        x = torch.randn(2, 16, hw[0], hw[1], device=DEV, requires_grad=True)    # packed map: hw / 2 (k=3: S=3, k=5: S=5)
        if hw[0] // 2 < 2 * (k // 2) + 1:
            continue
        res = {}
        for form in (False, True):
            m.collapse = form
            for p in m.parameters():
                p.grad = None
            x.grad = None
            y = m(x)
            (y * torch.linspace(0.5, 1.5, y.numel(), device=DEV).view_as(y)).sum().backward()
            res[form] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
        y0, dx0, g0 = res[False]
        y1, dx1, g1 = res[True]
        assert float((y0 - y1).abs().max()) <= 1e-4 * float(y0.abs().max())
        assert float((dx0 - dx1).abs().max()) <= 5e-4 * float(dx0.abs().max()), (k, hw)
        gmax = max(float(v.abs().max()) for v in g0.values())
        for n in g0:    # (floor: the conv bias in front of the GroupNorm has a mathematically zero gradient -- round-off in both forms)
            assert float((g0[n] - g1[n]).abs().max()) <= 5e-4 * max(float(g0[n].abs().max()), 1e-2 * gmax), (k, hw, n)


@pytest.mark.parametrize('shape', [(4, 64, 192, 640), (4, 64, 96, 320), (2, 64, 384, 1280), (1, 19, 13, 70)])
def test_invdepth_conv_strip_kernel_is_bit_identical(monkeypatch, shape):
    """Round 5: strip form of the InvDepth forward kernel (what 192x640 / 96x320 run) vs the 64-pixel kernel: equal bits."""
    from packnet_sfm.hip import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (0.2 * torch.randn(1, C, 3, 3, generator=g)).cuda()
    b = torch.randn(1, generator=g).cuda()
    monkeypatch.setenv('PNSFM_INVDEPTH_STRIP', '0')
    y0 = ops.invdepth_conv_forward(x, w, b, 0.5)
    for mode in ('1', '4', '8'):
        monkeypatch.setenv('PNSFM_INVDEPTH_STRIP', mode)
        assert torch.equal(y0, ops.invdepth_conv_forward(x, w, b, 0.5)), mode
    ref = torch.sigmoid(F.conv2d(x.double(), w.double(), b.double(), padding=1)) / 0.5
    P.check(y0, ref.float(), 1e-5, 'invdepth fwd vs fp64')


@pytest.mark.parametrize('shape', [(4, 64, 64, 192, 640, 7), (4, 64, 64, 96, 320, 1), (4, 512, 512, 12, 40, 3), (2, 48, 33, 19, 40, 3)])
def test_conv2d_backward_data_addend(shape):
    """Round 5: dx = backward-data + addend inside the launch (epilogue or K-split second stage, whatever the tuned configuration of
    the shape is): the bits of the separate elementwise sum, for a dense addend and for a channel slice of a wider tensor."""
    from packnet_sfm.hip import ops
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1).cuda()
    _, wb = ops.conv2d_pack(w)
    dy = torch.randn(B, Cout, H, W, generator=g).cuda()
    wide = torch.randn(B, Cin + 65, H, W, generator=g).cuda()
    plain = ops.conv2d_backward_data(dy, wb, Cin, ks)
    for addend in (wide[:, :Cin].contiguous(), wide[:, 64:64 + Cin]):
        assert torch.equal(ops.conv2d_backward_data(dy, wb, Cin, ks, addend=addend), plain + addend)


def test_gradient_taps_match_the_plain_graph_packnet01():
    """PackNet01 forward + backward with the gradient taps (default) against the plain autograd graph (PNSFM_GRAD_TAPS=0 path)."""
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.depth.PackNet01 import PackNet01
    torch.manual_seed(5)
    net = PackNet01(dropout=None, version='1A').cuda().train()
    rgb = torch.rand(1, 3, 64, 128, device='cuda')
    res = []
    for taps in (True, False):
        HF.set_grad_taps(taps)
        try:
            net.zero_grad()
            out = net(rgb)['inv_depths']
            loss = sum((d * d).mean() * (i + 1) for i, d in enumerate(out))
            loss.backward()
            res.append(([d.detach().clone() for d in out], {n: p.grad.clone() for n, p in net.named_parameters()}))
        finally:
            HF.set_grad_taps(True)
    (o1, g1), (o0, g0) = res
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        P.check(g1[n], g0[n], 2e-4, n, floor=1e-3 * gmax)


def test_flat_adam_fused_tail_gpu():
    """Round 5: the two-launch optimizer tail (Adam + re-pack from registers) against the flat update + batched re-pack on the GPU:
    parameters, moments, step counters and both packed images bit for bit (same case as the emulated test)."""
    P.case_flat_adam_fused_tail('cuda')
