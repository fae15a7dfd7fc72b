"""GPU (MI355X): the gfx950 kernels, called through the C ABI, against
  (1) the REFERENCE's goldens (tests/golden/*.pt),
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (3) at BASELINE.json's full size (batch 4, 192x640): the oracle's math executed by stock PyTorch-ROCm ops on the same
      device, plus size-independent properties (pack/unpack round trip, conv linearity, identity warp, SSIM(x,x)=1).
Tolerances are fp32: 1e-4 of max for activations, 1e-3 for depth / gradients (north-star: depth within 1e-3 rel)."""
import os
import random

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
    assert _lib.get().pnsfm_build_target() == b'gfx950'      # the native library is what runs, or we fail loudly
    assert _lib.REQUIRE_CUDA


# ------------------------------------------------------------------------------------------- (1) reference goldens
@pytest.mark.parametrize('name', ['conv2d_k3', 'conv2d_k5', 'conv2d_k7'])
def test_conv2d_block_golden(name):
    P.case_conv2d_block(name, DEV)


def test_residual_conv_golden():
    P.case_residual_conv(DEV)


def test_packing_invdepth_golden():
    P.case_packing(DEV)
    P.case_invdepth(DEV)


@pytest.mark.parametrize('name', ['pack_k3', 'pack_k5'])
@pytest.mark.parametrize('collapse', [False, True])
def test_pack_golden(name, collapse):
    P.case_pack(name, DEV, collapse=collapse)


@pytest.mark.parametrize('collapse', [False, True])
@pytest.mark.parametrize('name', ['pack_d4_k3', 'pack_d4_k5'])
def test_pack_d4_golden(name, collapse):
    P.case_pack_d4(name, DEV, collapse)


def test_unpack_d4_golden():
    P.case_unpack_d4(DEV)


def test_packnetslim01_golden():
    P.case_packnetslim01(DEV)


def test_packnet01_version_1B_golden():
    """Skip connections added instead of concatenated (PackNet01.py:46-52,142-176)."""
    P.case_packnetslim01(DEV, variant='1B')


def test_dropout_and_eval_contract():
    """dropout > 0 drops whole shortcut channels in training only (layers01.py:64-65); eval returns ONE tensor and is
    dropout-free (PackNet01.py:178-185)."""
    from packnet_sfm.networks.depth.PackNetSlim01 import PackNetSlim01
    torch.manual_seed(0)
    net = PackNetSlim01(dropout=0.5, version='1A').to(DEV)
    rgb = torch.rand(1, 3, 64, 96, device=DEV)
    net.train()
    a = net(rgb=rgb)['inv_depths']
    b = net(rgb=rgb)['inv_depths']
    assert isinstance(a, list) and len(a) == 4
    assert not torch.equal(a[0], b[0])                      # two different dropout masks
    net.eval()
    with torch.no_grad():
        e1, e2 = net(rgb=rgb)['inv_depths'], net(rgb=rgb)['inv_depths']
    assert torch.is_tensor(e1) and tuple(e1.shape) == (1, 1, 64, 96)
    P.check(e1, e2, 1e-5, 'eval is dropout-free (equal up to the fp32 atomics order of split-K layers)')
    assert any(k.startswith('conv2.0.conv3.0.') for k in net.state_dict())   # reference key layout with dropout


def test_supervised_loss_golden():
    P.case_supervised_loss(DEV)


def test_unpack_golden():
    P.case_unpack(DEV)


@pytest.mark.parametrize('name', ['loss_default', 'loss_multires_mean', 'loss_clip_min', 'loss_clip_mean', 'loss_border',
                                  'loss_reflection', 'loss_l1_only', 'loss_l1_min', 'loss_l1_clip_min', 'loss_l1_clip_mean',
                                  'loss_l1_min_noauto'])
def test_loss_golden(name):
    P.case_loss(name, DEV)


def test_packnet01_golden():
    P.case_packnet01(DEV)


def test_posenet_golden():
    P.case_posenet(DEV)


def _selfsup(device, fx):
    from oracle import packnet_oracle as O
    from packnet_sfm.models.SelfSupModel import SelfSupModel
    from packnet_sfm.networks.depth.PackNet01 import PackNet01
    from packnet_sfm.networks.pose.PoseNet import PoseNet
    sd = O.init_params(O.packnet01_param_shapes('1A'), seed=fx['depth_seed'])
    psd = O.init_params(O.posenet_param_shapes(2), seed=fx['pose_seed'])
    psd['pose_pred.bias'] = fx['pose_pred_bias'].clone()
    model = SelfSupModel(**fx['loss_kwargs'], clip_loss=0.0, flip_lr_prob=1.0 if fx['flip'] else 0.0,
                         upsample_depth_maps=True, rotation_mode='euler')
    dn, pn = PackNet01(dropout=0.0, version='1A'), PoseNet(nb_ref_imgs=2)
    dn.load_state_dict(sd)
    pn.load_state_dict(psd)
    model.add_depth_net(dn)
    model.add_pose_net(pn)
    return model.to(device).train(), dn, pn


@pytest.mark.parametrize('key', ['step_flip0', 'step_flip1'])
def test_training_step_golden(key):
    """Full SelfSupModel step (PackNet01 + PoseNet + loss, fwd + bwd) vs the reference's loss and gradients."""
    fx = P.golden('step')[key]
    model, dn, pn = _selfsup(DEV, fx)
    batch = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in fx['batch'].items()}
    random.seed(0)
    out = model(batch, progress=0.0)
    P.check(out['loss'], fx['loss'], 1e-4, 'loss')
    P.check(out['metrics']['smoothness_loss'], fx['smoothness_loss'], 1e-3, 'smoothness')
    d = 1.0 / out['inv_depths'][0].clamp(min=1e-6)
    dref = 1.0 / fx['inv_depth0'].clamp(min=1e-6)
    P.check(d, dref, 1e-3, 'depth (north-star 1e-3 rel)')
    out['loss'].backward()
    named = [('depth_net.' + n, p) for n, p in dn.named_parameters()] + [('pose_net.' + n, p) for n, p in pn.named_parameters()]
    gmax = max(fx['grad_norms'].values())
    worst = 0.0
    for n, p in named:
        ref = fx['grad_norms'][n]
        got = float(p.grad.norm())
        # biases in front of a GroupNorm have zero true gradient: compare on the scale of the layer's real gradients
        tol = 1e-2 * max(ref, 1e-4 * gmax)
        assert abs(got - ref) <= tol, 'grad norm %s: %.6e vs reference %.6e' % (n, got, ref)
        worst = max(worst, abs(got - ref) / max(ref, 1e-4 * gmax))
    print('worst relative grad-norm deviation vs reference: %.2e' % worst)


# ------------------------------------------------------------------------------------------- (2) CPU oracle, seeded
CONV_SHAPES = [  # (B, Cin, Cout, H, W, k): real PackNet01 layer shapes at reduced batch / resolution
    (1, 3, 64, 48, 160, 5),       # pre_calc
    (1, 64, 64, 48, 160, 7),      # conv1
    (1, 2048, 64, 24, 80, 5),     # pack1.conv   (W=80 -> linear tiles)
    (2, 129, 64, 48, 160, 3),     # iconv1 (odd Cin)
    (2, 193, 128, 12, 40, 3),     # iconv3
    (1, 64, 32, 24, 80, 3),       # unpack1.conv (Cout 32)
    (1, 256, 1, 24, 80, 3),       # disp4 head (Cout 1)
    (2, 4096, 128, 6, 20, 3),     # pack3.conv at low res (split-K)
    (4, 512, 512, 6, 20, 3),      # conv5 stage
    (2, 256, 512, 12, 40, 1),     # residual shortcut 1x1
]


@pytest.fixture
def conv_variant(request):
    """Pin the forward/backward-data kernel variant with the autotuner off, so that every variant is exercised, not only the
    ones the tuner happens to pick.  f32 MFMA: 0 register-staged patch, 1 double-buffered LDS-DMA patch, 2 fully pipelined
    (patch and weight slabs by LDS-DMA).  Split-bf16 arithmetic (conv2d_bx3.h; the default): 3 one patch buffer, 4 two,
    5 two + a kernel row of weights per stage."""
    from packnet_sfm.hip import _lib, functional as HF
    lib = _lib.get()
    lib.pnsfm_set_autotune(0)
    HF.set_conv_math('bx3' if request.param >= 3 else 'f32')
    lib.pnsfm_set_conv_variant(request.param)
    yield request.param
    HF.set_conv_math('bx3')
    lib.pnsfm_set_conv_variant(0)
    lib.pnsfm_set_conv_variant(3)
    lib.pnsfm_set_autotune(1)


@pytest.mark.parametrize('conv_variant', [0, 1, 2, 3, 4, 5], indirect=True)
@pytest.mark.parametrize('shape', CONV_SHAPES)
def test_conv2d_vs_cpu_oracle(shape, conv_variant):
    from packnet_sfm.hip import ops
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)          # the oracle's conv (oracle/packnet_oracle.py conv2d_gn_elu)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    wf, wb = ops.conv2d_pack(wd)
    P.check(ops.conv2d_forward(xd, wf, bd, Cout, ks), yr, 2e-5, 'fwd')
    P.check(ops.conv2d_backward_data(dyd, wb, Cin, ks), xr.grad, 2e-5, 'dgrad')
    dw, db = ops.conv2d_backward_weight(xd, dyd, ks)
    P.check(dw, wr.grad, 5e-5, 'wgrad')
    P.check(db, br.grad, 5e-5, 'dbias')


@pytest.mark.parametrize('shape', [(1, 64, 64, 48, 160, 7), (1, 2048, 64, 24, 80, 5), (4, 512, 512, 6, 20, 3)])
def test_conv2d_bx3_error_vs_fp64(shape):
    """The split-bf16 arithmetic is not a reduced-precision mode: against an fp64 convolution its forward / backward-data error
    (scaled by sum |x||w|, the quantity rounding errors are proportional to) stays in the class of the f32-MFMA kernels on the
    same data -- within 2x of theirs (measured 1.0-1.7x: both are dominated by the fp32 accumulation inside the matrix pipe,
    not by the 2^-26 of dropped piece products) and below 8 * 2^-24 outright; plain bf16 operands would sit at ~4e-4."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    B, Cin, Cout, H, W, ks = shape
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
        for mode in ('f32', 'bx3'):
            HF.set_conv_math(mode)
            wf, wb = ops.conv2d_pack(w.to(DEV))
            y = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks).cpu().double()
            dx = ops.conv2d_backward_data(dy.to(DEV), wb, Cin, ks).cpu().double()
            err[mode] = (float(((y - y64).abs() / ymag).max()), float(((dx - dx64).abs() / dxmag).max()))
    finally:
        HF.set_conv_math('bx3')
    print('max |err| / sum|a||b|  (fwd, dgrad):  f32 MFMA %.2e %.2e   split-bf16 %.2e %.2e   [2^-24 = 5.96e-08]'
          % (err['f32'] + err['bx3']))
    for i in range(2):
        assert err['bx3'][i] <= max(2.0 * err['f32'][i], 1.5e-7), err
        assert err['bx3'][i] <= 8 * 2.0 ** -24, err


WGRAD2_SHAPES = [  # (B, Cin, Cout, H, W, k): shapes the tap-major weight-gradient kernel supports (W % 8 == 0, k in 1/3/5)
    (2, 64, 64, 48, 160, 3),      # conv2 stage, 32-column fragments
    (2, 129, 64, 48, 160, 3),     # iconv1 (odd Cin: third ci tile nearly empty)
    (2, 256, 256, 24, 80, 3),     # conv4 stage, 16-column fragments
    (4, 512, 512, 12, 40, 3),     # conv5 stage, 8-column fragments, H not a multiple of the tile rows
    (2, 256, 512, 12, 40, 1),     # residual shortcut 1x1
    (1, 512, 128, 24, 80, 5),     # pack3.conv collapsed (5x5: one kernel row per workgroup)
    (1, 64, 32, 24, 80, 3),       # unpack1.conv (Cout 32: half-empty co tile)
]


@pytest.mark.parametrize('shape', [sh for sh in WGRAD2_SHAPES if sh[5] != 1] +
                         [(1, 64, 64, 48, 160, 7), (2, 64, 256, 24, 80, 7), (1, 2048, 64, 8, 80, 5), (4, 512, 512, 6, 20, 3), (2, 4096, 128, 6, 20, 3),
                          (8, 2048, 64, 96, 4, 5)])
def test_conv2d_wgrad_split_bf16_vs_cpu_oracle(shape):
    """csrc/conv2d_wgrad3.hip pinned (autotuner off) vs the oracle's conv weight/bias gradient -- same tolerance as the f32
    kernels."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    lib.pnsfm_set_autotune(0)
    HF.set_conv_math('bx3')
    lib.pnsfm_set_wgrad_variant(2)
    try:
        B, Cin, Cout, H, W, ks = shape
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
        b = torch.randn(Cout, generator=g)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv2d(x, wr, br, padding=ks // 2)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        dw, db = ops.conv2d_backward_weight(x.to(DEV), dy.to(DEV), ks)
        P.check(dw, wr.grad, 5e-5, 'wgrad (split-bf16)')
        P.check(db, br.grad, 5e-5, 'dbias (split-bf16)')
    finally:
        lib.pnsfm_set_wgrad_variant(-1)
        lib.pnsfm_set_autotune(1)


@pytest.mark.parametrize('shape', WGRAD2_SHAPES)
def test_conv2d_wgrad_tap_major_vs_cpu_oracle(shape):
    """csrc/conv2d_wgrad2.hip forced on (autotuner off) vs the oracle's conv weight/bias gradient."""
    from packnet_sfm.hip import _lib, ops
    lib = _lib.get()
    lib.pnsfm_set_autotune(0)
    lib.pnsfm_set_wgrad_variant(1)
    try:
        B, Cin, Cout, H, W, ks = shape
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
        b = torch.randn(Cout, generator=g)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv2d(x, wr, br, padding=ks // 2)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        dw, db = ops.conv2d_backward_weight(x.to(DEV), dy.to(DEV), ks)
        P.check(dw, wr.grad, 5e-5, 'wgrad (tap-major)')
        P.check(db, br.grad, 5e-5, 'dbias (tap-major)')
    finally:
        lib.pnsfm_set_wgrad_variant(-1)
        lib.pnsfm_set_autotune(1)


def test_groupnorm_vs_cpu_oracle():
    from packnet_sfm.hip import functional as HF, ops
    g = torch.Generator().manual_seed(5)
    for (B, C, H, W, act) in [(2, 64, 48, 160, ops.ACT_ELU), (4, 512, 6, 20, ops.ACT_ELU), (2, 32, 24, 80, ops.ACT_RELU)]:
        x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
        r = torch.randn(B, C, H, W, generator=g)
        ga, be = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        dy = torch.randn(B, C, H, W, generator=g)
        xr, rr, gr, br = (t.clone().requires_grad_(True) for t in (x, r, ga, be))
        z = F.group_norm(xr + rr, 16, gr, br, eps=1e-5)
        yr = F.elu(z) if act == ops.ACT_ELU else F.relu(z)
        yr.backward(dy)
        xd, rd, gd, bd = (t.to(DEV).requires_grad_(True) for t in (x, r, ga, be))
        y = HF.groupnorm_act(xd, gd, bd, 16, 1e-5, act, res=rd)
        y.backward(dy.to(DEV))
        P.check(y, yr, 2e-5, 'gn fwd')
        P.check(xd.grad, xr.grad, 2e-4, 'gn dx')
        P.check(rd.grad, rr.grad, 2e-4, 'gn dres')
        P.check(gd.grad, gr.grad, 2e-4, 'gn dgamma')
        P.check(bd.grad, br.grad, 2e-4, 'gn dbeta')


def test_conv3d_vs_cpu_oracle():
    from oracle import packnet_oracle as O
    from packnet_sfm.hip import functional as HF
    g = torch.Generator().manual_seed(6)
    for (B, D, H, W) in [(2, 256, 24, 80), (1, 2048, 6, 20), (2, 32, 48, 160)]:
        p = torch.randn(B, D, H, W, generator=g)
        w3 = 0.3 * torch.randn(8, 1, 3, 3, 3, generator=g)
        b3 = torch.randn(8, generator=g)
        pr, wr, br = (t.clone().requires_grad_(True) for t in (p, w3, b3))
        yr = O.conv3d_1to8(pr, wr, br)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        pd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (p, w3, b3))
        y = HF.conv3d_1to8(pd, wd, bd)
        y.backward(dy.to(DEV))
        P.check(y, yr, 1e-5, 'conv3d fwd')
        P.check(pd.grad, pr.grad, 1e-5, 'conv3d dgrad')
        P.check(wd.grad, wr.grad, 1e-4, 'conv3d wgrad')
        P.check(bd.grad, br.grad, 1e-4, 'conv3d dbias')


@pytest.mark.parametrize('opt_kw', [dict(fused=True), dict(foreach=True), dict()])
def test_packed_weights_follow_the_optimizer(opt_kw):
    """The conv layers keep MFMA-packed copies of their weights; they must be refreshed after EVERY optimizer step, also
    for fused optimizers that do not bump tensor._version.  Three Adam steps of a Conv2D block (HIP) vs the same block in
    the oracle's math under an identical optimizer."""
    from oracle import packnet_oracle as O
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    torch.manual_seed(3)
    m = Conv2D(8, 32, 3, 1).to(DEV)      # 2 channels per GroupNorm group: the conv bias has a real (non-zero) gradient
    sd = {'l.' + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = torch.randn(2, 8, 24, 32, device=DEV)
    tgt = torch.randn(2, 32, 24, 32, device=DEV)
    oa = torch.optim.Adam(m.parameters(), lr=1e-2, **opt_kw)
    ob = torch.optim.Adam(list(sd.values()), lr=1e-2, **opt_kw)
    for step in range(3):
        la = ((m(x) - tgt) ** 2).mean()
        lb = ((O.conv2d_gn_elu(x, sd, 'l', 3) - tgt) ** 2).mean()
        P.check(la, lb, 1e-5, 'loss at step %d' % step)
        oa.zero_grad(); ob.zero_grad()
        la.backward(); lb.backward()
        oa.step(); ob.step()
    for k, v in m.state_dict().items():
        P.check(v, sd['l.' + k], 1e-4, 'parameter ' + k)
    P.check(m(x), O.conv2d_gn_elu(x, sd, 'l', 3), 1e-4, 'forward after 3 steps')


def test_adam_vs_torch():
    from packnet_sfm.hip import ops
    g = torch.Generator().manual_seed(3)
    n = 1 << 20
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2e-4)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, (grad * 8.0).to(DEV), m, v, 2e-4, 0.9, 0.999, 1e-8, 0.0, 0.125, step)
    P.check(p, p_ref, 1e-6, 'adam')


def test_flat_adam_and_trainer_loop():
    """HorovodTrainer.fit (RCCL facade, world size 1) driving FlatAdam on a toy module: parameters follow torch.optim.Adam."""
    import types
    from packnet_sfm.rccl.flat_adam import FlatAdam
    from packnet_sfm.trainers.horovod_trainer import HorovodTrainer

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))
            self.current_epoch = 0
            self.config = types.SimpleNamespace(datasets=types.SimpleNamespace(
                train=types.SimpleNamespace(batch_size=4), validation=types.SimpleNamespace(batch_size=4)))
            g = torch.Generator().manual_seed(1)
            self.data = [{'x': torch.randn(4, 6, generator=g), 'y': torch.randn(4, 1, generator=g), 'name': 'a'} for _ in range(5)]
            self.losses = []

        def configure_optimizers(self):
            self.optimizer = FlatAdam([{'params': list(self.net.parameters()), 'lr': 1e-2}])
            self.scheduler = types.SimpleNamespace(step=lambda: None)

        def train_dataloader(self):
            return types.SimpleNamespace(sampler=None, __iter__=None, data=self.data)

        def val_dataloader(self):
            return []

        def training_step(self, batch, i):
            return {'loss': ((self.net(batch['x']) - batch['y']) ** 2).mean().unsqueeze(0)}

        def training_epoch_end(self, outputs):
            self.losses.append(float(torch.stack([o['loss'] for o in outputs]).mean()))

        def validation_epoch_end(self, outputs):
            return {}

    class Loader(list):
        sampler = None

    toy = Toy()
    toy.train_dataloader = lambda: Loader(toy.data)
    ref = Toy()
    ref_opt = torch.optim.Adam(ref.net.parameters(), lr=1e-2)
    trainer = HorovodTrainer(max_epochs=3)
    trainer.fit(toy)
    for _ in range(3):
        for b in ref.data:
            ref_opt.zero_grad()
            ((ref.net(b['x']) - b['y']) ** 2).mean().backward()
            ref_opt.step()
    for a, b in zip(toy.net.parameters(), ref.net.parameters()):
        P.check(a, b, 1e-4, 'trainer+FlatAdam parameter')
    assert toy.losses[-1] < toy.losses[0]


def _step_batch(fx):
    return {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in fx['batch'].items()}


def test_wgrad_side_stream_gradient_accumulation():
    """Two backward() calls without zero_grad (gradient accumulation) and a zero_grad(set_to_none=False) step: weight
    gradients with the side stream on must equal the single-stream order (AccumulateGrad adds on the compute
    stream when .grad is already defined).  A stack of the real blocks under a SMOOTH loss (the photometric loss re-routes
    pixels on 1e-7 input differences), so the comparison is tight."""
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, PackLayerConv3d, ResidualConv, UnpackLayerConv3d
    torch.manual_seed(5)
    net = torch.nn.Sequential(Conv2D(3, 32, 5, 1), PackLayerConv3d(32, 5), ResidualConv(32, 64, 1), UnpackLayerConv3d(64, 32, 3),
                              Conv2D(32, 16, 3, 1)).to(DEV)
    x = torch.randn(2, 3, 64, 96, device=DEV)
    tgt = torch.randn(2, 16, 64, 96, device=DEV)

    def loss():
        return ((net(x) - tgt) ** 2).mean()
    grads = {}
    was = HF._WgradStream.enabled
    try:
        loss().backward()                                               # autotune outside the comparison
        for side in (False, True):
            HF.set_wgrad_stream(side)
            net.zero_grad(set_to_none=True)
            for _ in range(2):
                loss().backward()                                       # accumulates into defined .grad the second time
            g2 = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
            net.zero_grad(set_to_none=False)                            # zero-filled, still defined
            loss().backward()
            torch.cuda.synchronize()
            grads[side] = (g2, {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    finally:
        HF.set_wgrad_stream(was)
    for k in (0, 1):
        gmax = max(float(v.abs().max()) for v in grads[False][k].values())
        for n, g in grads[False][k].items():
            P.check(grads[True][k][n], g, 2e-5, 'accumulated grad %s (pass %d)' % (n, k), floor=1e-2 * gmax)
    gmax = float(max(v.abs().max() for v in grads[True][1].values()))
    for n, g in grads[True][1].items():
        # (a conv bias in front of a GroupNorm has a mathematically ZERO gradient: what is compared there is summation noise,
        # ~1e-7 of the largest gradient, whose order changes with the atomics of the K-split kernels)
        P.check(grads[True][0][n], 2.0 * g, 5e-5, 'two accumulated passes == 2 x one pass: ' + n, floor=1e-2 * gmax)


def test_trainer_fit_on_selfsup_model():
    """a18: HorovodTrainer.fit (RCCL facade) drives the REAL SelfSupModel (PackNet01 + PoseNet on the HIP kernels) through
    a ModelWrapper-shaped module for one epoch of 3 steps; parameters equal a hand-written zero_grad/forward/backward/Adam
    loop on an identical replica."""
    import types
    from packnet_sfm.trainers.horovod_trainer import HorovodTrainer
    fx = dict(P.golden('step')['step_flip0'])
    cpu_batch = fx['batch']

    class Wrapper(torch.nn.Module):                 # the surface of the reference's ModelWrapper that the trainer touches
        def __init__(self):
            super().__init__()
            self.model, self.dn, self.pn = _selfsup('cpu', fx)
            self.init = [p.detach().clone() for p in self.dn.parameters()]
            self.current_epoch = 0
            self.config = types.SimpleNamespace(datasets=types.SimpleNamespace(
                train=types.SimpleNamespace(batch_size=1), validation=types.SimpleNamespace(batch_size=1)))
            self.losses = []

        def configure_optimizers(self):             # model_wrapper.py:128-166: two Adam groups + StepLR
            self.optimizer = torch.optim.Adam([{'name': 'Depth', 'params': list(self.dn.parameters()), 'lr': 2e-4},
                                               {'name': 'Pose', 'params': list(self.pn.parameters()), 'lr': 2e-4}])
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=30, gamma=0.5)

        def train_dataloader(self):
            return Loader([cpu_batch] * 3)

        def val_dataloader(self):
            return []

        def training_step(self, batch, i):
            out = self.model(batch, progress=0.0)
            return {'loss': out['loss'], 'metrics': out['metrics']}

        def training_epoch_end(self, outputs):
            self.losses = [float(o['loss']) for o in outputs]

        def validation_epoch_end(self, outputs):
            return {}

    class Loader(list):
        sampler = None

    random.seed(3)
    w = Wrapper()
    HorovodTrainer(max_epochs=1).fit(w)
    ref_model, rdn, rpn = _selfsup(DEV, fx)
    opt = torch.optim.Adam([{'params': list(rdn.parameters()), 'lr': 2e-4}, {'params': list(rpn.parameters()), 'lr': 2e-4}])
    batch = _step_batch(fx)
    random.seed(3)                     # the same flip draws as the trainer's run
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        out = ref_model(batch, progress=0.0)
        out['loss'].backward()
        opt.step()
        ref_losses.append(float(out['loss'].detach()))
    # step 0 sees identical parameters: round-off only.  Later steps: this loss re-routes pixels on 1e-7 differences and Adam
    # amplifies them, so the sequence is held to 1 %.
    assert abs(w.losses[0] - ref_losses[0]) <= 1e-5 * abs(ref_losses[0]), (w.losses, ref_losses)
    P.check(torch.tensor(w.losses), torch.tensor(ref_losses), 1e-2, 'trainer losses')
    # (no element-wise parameter comparison: three Adam steps of +-2e-4 on noise-level gradients differ between any two runs)
    assert next(w.dn.parameters()).is_cuda
    moved = sum(float((a.detach().cpu() - b0).abs().max()) > 0 for (n, a), b0 in zip(w.dn.named_parameters(), w.init))
    assert moved >= 100, 'the optimizer did not update the depth network (%d tensors moved)' % moved


# ------------------------------------------------------------------------------------------- (3) full size
def _full_batch(B=4, H=192, W=640, seed=1234):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import bench
    return bench.synthetic_batch(B, H, W, seed, DEV)


def test_full_size_properties():
    """BASELINE.json configs[1] size (batch 4, 192x640): properties that need no CPU oracle."""
    from packnet_sfm.hip import functional as HF, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 64, 192, 640, generator=g).to(DEV)
    # pack / unpack data movement round trip, and equality with the torch definition of packing
    s = ops.space_to_depth(x)
    assert torch.equal(ops.depth_to_space(s), x)
    assert torch.equal(s, F.pixel_unshuffle(x, 2))
    # conv linearity: conv(a*x1 + x2) == a*conv(x1) + conv(x2)  (bias off)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    wf, _ = ops.conv2d_pack(w, want_bwd=False)
    x2 = torch.randn(4, 64, 192, 640, generator=g).to(DEV)
    lhs = ops.conv2d_forward(1.5 * x + x2, wf, None, 64, 3)
    rhs = 1.5 * ops.conv2d_forward(x, wf, None, 64, 3) + ops.conv2d_forward(x2, wf, None, 64, 3)
    P.check(lhs, rhs, 1e-5, 'conv linearity')
    # identity pose => the warp returns the context image itself; SSIM(x, x) = 1 => zero photometric loss
    batch = _full_batch()
    img = batch['rgb']
    K = batch['intrinsics'].float()
    inv = (0.05 + torch.rand(4, 1, 192, 640, generator=g)).to(DEV)
    T = torch.eye(4, device=DEV).repeat(1, 4, 1, 1)
    warped = HF.view_synthesis(inv, img.unsqueeze(0), K, K, T)
    P.check(warped[0], img, 2e-4, 'identity warp')   # K*Kinv round trip: ~4e-5 px at W=640
    loss = HF.photometric(warped, img.unsqueeze(0), img, 0.85, 1e-4, 9e-4, False, HF.REDUCE_MIN)
    assert float(loss) < 1e-5
    assert float(HF.smoothness(torch.ones_like(inv), img)) == 0.0


def _full_size_step(B, H, W, ref_device=DEV):
    """Training step (fwd + loss + bwd) at a BASELINE.json size through the HIP kernels vs the oracle's math run by stock
    PyTorch-ROCm ops on the same MI355X: loss, depth (north-star bound 1e-3 rel), per-parameter gradient norm AND
    per-parameter gradient direction (relative L2 error of every gradient tensor).  Returns what the caller needs for the
    eager-baseline timing."""
    from oracle import packnet_oracle as O
    from packnet_sfm.models.SelfSupModel import SelfSupModel
    from packnet_sfm.networks.depth.PackNet01 import PackNet01
    from packnet_sfm.networks.pose.PoseNet import PoseNet
    import bench
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    batch = _full_batch(B, H, W)
    sd = O.init_params(O.packnet01_param_shapes('1A'), seed=42)
    psd = O.init_params(O.posenet_param_shapes(2), seed=43)
    psd['pose_pred.bias'] = torch.tensor([2., 0.5, -1., 0.3, -0.2, 0.1, -2., -0.5, 1., -0.3, 0.2, -0.1])
    kw = {k: bench.LOSS_DEFAULTS[k] for k in ('num_scales', 'ssim_loss_weight', 'smooth_loss_weight', 'C1', 'C2',
                                               'photometric_reduce_op', 'automask_loss')}
    model = SelfSupModel(**{**bench.LOSS_DEFAULTS, 'flip_lr_prob': 0.0})
    dn, pn = PackNet01(dropout=0.0, version='1A'), PoseNet(nb_ref_imgs=2)
    dn.load_state_dict(sd)
    pn.load_state_dict(psd)
    model.add_depth_net(dn)
    model.add_pose_net(pn)
    model = model.to(DEV).train()
    out = model(batch, progress=0.0)
    out['loss'].backward()
    # reference: the oracle's math on stock PyTorch ops -- on the same MI355X (MIOpen / ATen), or on the host cores
    sdd = {k: v.to(ref_device).requires_grad_(True) for k, v in sd.items()}
    psdd = {k: v.to(ref_device).requires_grad_(True) for k, v in psd.items()}
    rbatch = batch if ref_device == DEV else {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in batch.items()}
    if ref_device == 'cpu':
        torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = O.selfsup_forward(sdd, psdd, rbatch, flip=False, **kw)
    ref['loss'].sum().backward()
    ref = {k: (v.to(DEV) if torch.is_tensor(v) else ([t.to(DEV) for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else v))
           for k, v in ref.items()}
    gd = {k: v.grad.to(DEV) for k, v in sdd.items()}
    gp = {k: v.grad.to(DEV) for k, v in psdd.items()}
    P.check(out['loss'], ref['loss'], 2e-4, 'loss')
    d, dref = 1.0 / out['inv_depths'][0].clamp(min=1e-6), 1.0 / ref['inv_depths'][0].clamp(min=1e-6)
    P.check(d, dref, 1e-3, 'depth')
    # BASELINE.json metric, second half: depth abs_rel of ours vs the reference math (utils/depth.py:275 abs_rel form)
    abs_rel = float(((d - dref).abs() / dref).mean())
    worst_rel = float(((d - dref).abs() / dref).max())
    print('%dx%d b%d: depth abs_rel vs reference: mean %.3e, worst pixel %.3e' % (H, W, B, abs_rel, worst_rel))
    assert abs_rel <= 1e-3 and worst_rel <= 1e-3
    gmax = max(float(v.norm()) for v in gd.values())
    worst = (0.0, '')
    for n, p in dn.named_parameters():
        gref = gd[n]
        r = float(gref.norm())
        got = float(p.grad.norm())
        assert abs(got - r) <= 2e-2 * max(r, 1e-4 * gmax), 'grad norm %s: %.6e vs %.6e' % (n, got, r)
        # direction, not only length: ||g - g_ref|| / ||g_ref|| per tensor.  The floor covers gradients that are
        # mathematically zero (conv biases in front of a GroupNorm): pure round-off in both implementations.  2e-2: the
        # fp32 MIOpen reference is itself ~2.5e-3 off on the K = 147 456 pack5 weight gradient.
        rel = float((p.grad - gref).norm()) / max(r, 1e-3 * gmax)
        if rel > worst[0]:
            worst = (rel, n)
        assert rel <= 2e-2, 'grad direction %s: relative L2 error %.3e' % (n, rel)
    for n, p in pn.named_parameters():
        gref = gp[n]
        pmax = max(float(v.norm()) for v in gp.values())
        rel = float((p.grad - gref).norm()) / max(float(gref.norm()), 1e-3 * pmax)
        assert rel <= 2e-2, 'pose grad %s: relative L2 error %.3e' % (n, rel)
    print('worst per-tensor gradient relative L2 error: %.3e (%s)' % worst)
    assert torch.isfinite(out['loss']).all()
    # keep the margins of this test visible (copied to profiles/rNN_full_size_parity.json by the GPU session)
    import json
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        f = os.path.join(out_dir, 'full_size_parity.json')
        try:
            rec = json.load(open(f))
        except Exception:
            rec = {}
        rec['%dx%d_b%d_vs_%s' % (H, W, B, 'cpu_oracle' if ref_device == 'cpu' else 'same_device_eager')] = {
            'loss': float(out['loss']), 'loss_reference': float(ref['loss'].sum()),
            'depth_abs_rel_mean': abs_rel, 'depth_rel_worst_pixel': worst_rel, 'depth_bound': 1e-3,
            'worst_per_tensor_grad_rel_l2': worst[0], 'worst_tensor': worst[1], 'grad_bound': 2e-2}
        with open(f, 'w') as fh:
            json.dump(rec, fh, indent=1)
    return model, batch, sdd, psdd, kw, abs_rel, worst_rel


def test_full_size_step_384x1280_vs_cpu_oracle():
    """BASELINE.json configs[2] shape: batch 2 per GPU at 384x1280 (4x the pixels of configs[1]) against the oracle on the
    host cores (~20 s of CPU work; the stock PyTorch-ROCm ops took MIOpen 10 minutes to find kernels for these shapes)."""
    _full_size_step(2, 384, 1280, ref_device='cpu')


def test_full_size_step_vs_same_device_reference():
    """Batch 4, 192x640 training step through the HIP kernels vs the oracle's math run by stock PyTorch-ROCm ops on the
    same MI355X (loss, depth, gradient norms and directions)."""
    from oracle import packnet_oracle as O
    model, batch, sdd, psdd, kw, abs_rel, worst_rel = _full_size_step(4, 192, 640)
    # like-for-like GPU baseline: the same math through stock PyTorch-ROCm eager ops (MIOpen / ATen) on
    # this MI355X, forward + backward, vs our step's forward + backward.  Reported, not asserted.
    import json, os, time

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def eager_step():
        for v in list(sdd.values()) + list(psdd.values()):
            v.grad = None
        O.selfsup_forward(sdd, psdd, batch, flip=False, **kw)['loss'].sum().backward()

    def hip_step():
        model.zero_grad(set_to_none=True)
        model(batch, progress=0.0)['loss'].backward()

    t_eager, t_hip = timed(eager_step), timed(hip_step)
    rec = {'workload': 'fwd+loss+bwd, batch 4, 192x640, fp32 (no optimizer step in either leg)',
           'pytorch_rocm_eager_images_per_sec': round(4 / t_eager, 2), 'hip_path_images_per_sec': round(4 / t_hip, 2),
           'speedup': round(t_eager / t_hip, 2), 'depth_abs_rel_vs_reference': abs_rel, 'depth_worst_rel': worst_rel}
    print('eager baseline:', json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'eager_baseline.json'), 'w') as f:
            json.dump(rec, f)
