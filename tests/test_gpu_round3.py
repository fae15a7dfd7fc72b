"""GPU (MI355X): parity of the code paths the bench actually runs since round 2 ("parity holes").

  (a) the split-bf16 weight-gradient kernel (csrc/conv2d_wgrad3.hip) against an fp64 weight gradient at the REAL reduction
      lengths of the step (K = B*H*W up to 491 520 pixels), next to the f32-MFMA kernels on the same data;
  (b) FlatAdam with the conv weight gradients written straight into its gradient arena (hip.functional.register_grad_slots)
      on a stack of real PackNet blocks, against torch.optim.Adam on plain gradients;
  (c) the same step with the gradient reducer forced on (1-rank RCCL group, collectives on the side stream);
  (d) what the split-bf16 arithmetic does with inputs outside its envelope (+-inf, >= 3.39e38, NaN, denormal-range values).
Reference semantics: packnet_sfm/models/model_wrapper.py:128-166 (optimizer), trainers/horovod_trainer.py:46-48,92-93."""
import os
import socket

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


# ------------------------------------------------------------------------------------------------ (a) wgrad3 vs fp64
def _wgrad_fp64(x, dy, ks):
    """dW[co][ci][ky][kx] = sum_{b,y,x} dY * X(shifted) in float64, and the same sum over |dY| |X| (the quantity rounding errors
    scale with), by unfold + matmul per image (rocBLAS dgemm: an implementation that shares nothing with the kernels under test)."""
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    dw = torch.zeros(Cout, Cin * ks * ks, dtype=torch.float64, device=x.device)
    mag = torch.zeros_like(dw)
    cstep = max(1, (1 << 28) // (H * W * ks * ks))              # <= 2 GiB of unfolded fp64 columns at a time
    for b in range(B):
        dyb = dy[b].double().reshape(Cout, H * W)
        for c0 in range(0, Cin, cstep):
            c1 = min(Cin, c0 + cstep)
            cols = F.unfold(x[b:b + 1, c0:c1].double(), ks, padding=ks // 2)[0]        # [(c1-c0)*k*k, H*W]
            dw[:, c0 * ks * ks:c1 * ks * ks] += dyb @ cols.t()
            mag[:, c0 * ks * ks:c1 * ks * ks] += dyb.abs() @ cols.abs().t()
            del cols
    return dw.view(Cout, Cin, ks, ks), mag.view(Cout, Cin, ks, ks)


WGRAD_REAL_SHAPES = [  # (B, Cin, Cout, H, W, k): weight gradients of the 192x640 batch-4 step at their real reduction length
    (4, 64, 64, 192, 640, 7),       # conv1: K = 491 520 pixels, pixel-split launch + two-stage reduction
    (4, 256, 256, 24, 80, 3),       # conv4 stage
    (4, 16384, 512, 6, 20, 3),      # pack5.conv (reference form): W % 8 == 4 -> masked variant, K = 480
    (8, 2048, 64, 4, 320, 5),       # pack1 border strips (5x5, batched top+bottom)
    (4, 129, 64, 192, 640, 3),      # iconv1: ragged Cin
]


@pytest.mark.parametrize('shape', WGRAD_REAL_SHAPES)
def test_conv2d_wgrad3_error_vs_fp64_real_K(shape):
    """max |dW - dW64| / sum |dY||X| of the split-bf16 weight gradient stays in the class of the f32-MFMA weight-gradient
    kernel on the same data (within 2x of it, or below 1.5e-7 outright) at the reduction lengths the training step has."""
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape) + 3)
    x = (torch.randn(B, Cin, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cin, 1, 1, generator=g))).to(DEV)
    dy = (torch.randn(B, Cout, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cout, 1, 1, generator=g))).to(DEV)
    dw64, mag = _wgrad_fp64(x, dy, ks)
    db64 = dy.double().sum((0, 2, 3))
    dbmag = dy.double().abs().sum((0, 2, 3))
    err = {}
    lib.pnsfm_set_autotune(0)
    try:
        for name, variant in (('f32', 0), ('bx3', 2)):
            HF.set_conv_math('bx3')
            lib.pnsfm_set_wgrad_variant(variant)
            dw, db = ops.conv2d_backward_weight(x, dy, ks)
            err[name] = (float(((dw.double() - dw64).abs() / mag).max()), float(((db.double() - db64).abs() / dbmag).max()))
    finally:
        lib.pnsfm_set_wgrad_variant(-1)
        lib.pnsfm_set_autotune(1)
    print('wgrad %s  max|err|/sum|dy||x|:  f32 MFMA %.2e (dbias %.2e)   split-bf16 %.2e (dbias %.2e)   [2^-24 = 5.96e-08]'
          % (shape, err['f32'][0], err['f32'][1], err['bx3'][0], err['bx3'][1]))
    assert err['bx3'][0] <= max(2.0 * err['f32'][0], 1.5e-7), err
    assert err['bx3'][0] <= 16 * 2.0 ** -24, err
    assert err['bx3'][1] <= max(2.0 * err['f32'][1], 1.5e-7), err


@pytest.mark.parametrize('cfg', [(0, 1), (0, 2), (5, 2)])
@pytest.mark.parametrize('shape', [sh for sh in WGRAD_REAL_SHAPES if sh[5] == 3] + [(4, 512, 512, 12, 40, 3), (4, 64, 64, 96, 320, 3)])
def test_conv2d_wgrad_nine_taps_error_vs_fp64_real_K(shape, cfg):
    """The nine-taps 3x3 weight gradient (csrc/conv2d_wgrad4.hip, 16x16x32 MFMA) at the training step's reduction lengths: same
    fp64 error bound as the one-row kernel, and the two kernels agree to fp32 round-off.  cfg = (tile width in groups, ci tiles
    per workgroup); the pixel split is the one two workgroups per CU ask for."""
    import ctypes
    from packnet_sfm.hip import _lib, ops, functional as HF
    lib = _lib.get()
    B, Cin, Cout, H, W, ks = shape
    TG, WCI = cfg
    if W <= 24:
        TG = 0
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = (torch.randn(B, Cin, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cin, 1, 1, generator=g))).to(DEV)
    dy = (torch.randn(B, Cout, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cout, 1, 1, generator=g))).to(DEV)
    dw64, mag = _wgrad_fp64(x, dy, ks)
    db64 = dy.double().sum((0, 2, 3))
    dbmag = dy.double().abs().sum((0, 2, 3))
    HF.set_conv_math('bx3')
    base = -(-Cin // (16 * WCI)) * -(-Cout // (32 * (4 // WCI)))
    split = max(1, 512 // base)
    key = (ctypes.c_int * 7)(2 + 10 + 100, B, Cin, Cout, H * W, W, ks)
    lib.pnsfm_set_autotune(0)
    try:
        lib.pnsfm_set_wgrad_variant(2)
        dw3, db3 = ops.conv2d_backward_weight(x, dy, ks)
        lib.pnsfm_set_wgrad_variant(-1)
        assert lib.pnsfm_tune_set(key, split, 3 | ((WCI | (TG << 4)) << 4)) == 0
        dw, db = ops.conv2d_backward_weight(x, dy, ks)
    finally:
        lib.pnsfm_set_wgrad_variant(-1)
        lib.pnsfm_set_autotune(1)
    e9 = float(((dw.double() - dw64).abs() / mag).max())
    e3 = float(((dw3.double() - dw64).abs() / mag).max())
    eb = float(((db.double() - db64).abs() / dbmag).max())
    print('wgrad %s cfg %s split %d  max|err|/sum|dy||x|: nine taps %.2e (dbias %.2e)  one row %.2e' % (shape, cfg, split, e9, eb, e3))
    assert e9 <= max(2.0 * e3, 1.5e-7) and e9 <= 16 * 2.0 ** -24
    assert eb <= 1.5e-7
    assert float(((dw - dw3).abs() / mag.float()).max()) <= max(e9 + e3, 3e-7) * 1.05      # each within its own error of fp64


# ------------------------------------------------------------------------ (b) FlatAdam, gradients produced inside the arena
class _BlockStack(torch.nn.Module):
    """Conv2D -> ResidualConv -> PackLayerConv3d (collapsed form) -> UnpackLayerConv3d -> Conv2D: every kind of conv weight the
    PackNet01 step hands to the optimizer (leaf conv weights, the packing block's composed kernel, Conv3d stencils, GroupNorm)."""

    def __init__(self, C=32):
        super().__init__()
        from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, PackLayerConv3d, ResidualConv, UnpackLayerConv3d
        self.stem = Conv2D(16, C, 5, 1)
        self.res = ResidualConv(C, C, 1)
        self.pack = PackLayerConv3d(C, 3)
        self.pack.collapse = True
        self.unpack = UnpackLayerConv3d(C, C, 3)
        self.head = Conv2D(2 * C, 16, 3, 1)

    def forward(self, x):
        a = self.res(self.stem(x))
        u = self.unpack(self.pack(a))
        return self.head(torch.cat((u, a), 1))


def _arena_range(opt):
    out = []
    for g in opt.param_groups:
        out.append((g['_grad'].data_ptr(), g['_grad'].data_ptr() + g['_grad'].numel() * 4))
    return out


def _noise_only(net):
    """Names of conv biases feeding a GroupNorm with ONE channel per group: their gradient is mathematically zero, both runs see
    pure round-off, and Adam turns the sign of that noise into +-lr steps (they do not influence the output)."""
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    return {n + '.conv_base.bias' for n, m in net.named_modules() if isinstance(m, Conv2D) and m.normalize.num_channels == m.normalize.num_groups}


def _stack_inputs():
    g = torch.Generator().manual_seed(5)
    return torch.randn(2, 16, 32, 64, generator=g).to(DEV), torch.randn(2, 16, 32, 64, generator=g).to(DEV)


def _mirrored_steps(make_opt, steps=3, lr=2e-3):
    """`net` trains with the optimizer under test (FlatAdam with arena gradient slots, optionally behind the gradient reducer);
    `rep` is an identical replica of plain parameters stepped by torch.optim.Adam.  Every step
      1. both run forward + backward on the same batch from the same parameters: the gradients `net` produced INSIDE its arena
         (pointer check) must equal the freshly allocated gradients of `rep` -- same kernels, so only the summation-order noise
         of split-K atomics separates them;
      2. `rep` is then handed net's gradient VALUES, both optimizers step, and the parameters must agree to fp32 round-off of the
         Adam formula.  (Feeding both optimizers the same numbers is what makes 1e-6 meaningful: Adam divides by sqrt(v), so two
         runs whose gradients differ by 1e-7 of the tensor scale drift apart by ~1e-4 on the elements whose gradient is ~0 --
         measured on this very stack -- which says nothing about either optimizer.)"""
    x, tgt = _stack_inputs()
    torch.manual_seed(11)
    net = _BlockStack().to(DEV).train()
    torch.manual_seed(11)
    rep = _BlockStack().to(DEV).train()
    opt = make_opt(net, lr)
    inner = getattr(opt, '_opt', opt)
    ropt = torch.optim.Adam(rep.parameters(), lr=lr)
    skip = _noise_only(net)
    assert skip == {'unpack.conv.conv_base.bias', 'head.conv_base.bias'}, skip
    ranges = _arena_range(inner)
    for step in range(steps):
        opt.zero_grad()
        ropt.zero_grad()
        loss = ((net(x) - tgt) ** 2).mean()
        rloss = ((rep(x) - tgt) ** 2).mean()
        assert abs(float(loss) - float(rloss)) <= 1e-5 * abs(float(rloss)), (step, float(loss), float(rloss))
        loss.backward()
        rloss.backward()
        if hasattr(opt, 'synchronize'):
            opt.synchronize()               # 1-rank RCCL group: the buckets have been all-reduced (AVG over one rank) in place
        gscale = max(float(q.grad.abs().max()) for q in rep.parameters())
        n_in = 0
        for (name, p), q in zip(net.named_parameters(), rep.parameters()):
            P.check(p.grad, q.grad, 2e-5, 'step %d gradient of %s' % (step, name), floor=0.05 * gscale)
            if p.dim() == 4 and not name.startswith('pack.conv.conv_base'):
                # a Conv2d weight that is a leaf used once: the wgrad kernel wrote it straight into its arena slice
                # (pack.conv.conv_base.weight reaches the optimizer through the composed kernel, a non-leaf: gathered instead)
                view = inner.grad_view([g for g in inner.param_groups if id(p) in g['_offs']][0], p)
                assert any(lo <= p.grad.data_ptr() < hi for lo, hi in ranges) and p.grad.data_ptr() == view.data_ptr(), \
                    '%s: gradient not produced inside the arena' % name
                n_in += 1
            q.grad = p.grad.detach().clone()
        assert n_in >= 6, n_in
        if hasattr(opt, 'skip_synchronize'):
            with opt.skip_synchronize():
                opt.step()
        else:
            opt.step()
        ropt.step()
        for (name, p), q in zip(net.named_parameters(), rep.parameters()):
            P.check(p, q, 1e-5, 'step %d parameter %s' % (step, name), floor=1e-2)
    torch.cuda.synchronize()
    return net, opt


def test_flat_adam_arena_slots_on_real_blocks():
    """Bench default path: FlatAdam + register_grad_slots on a stack of real blocks (Conv2D, ResidualConv, collapsed
    PackLayerConv3d, UnpackLayerConv3d) -- wgrad kernel -> arena slice -> adam_flat_kernel -- against torch.optim.Adam."""
    from packnet_sfm.rccl.flat_adam import FlatAdam
    _mirrored_steps(lambda n, lr: FlatAdam([{'params': list(n.parameters()), 'lr': lr}]))


def test_flat_adam_slots_released_with_the_optimizer():
    """gradient slots must not outlive their optimizer -- a second model whose parameters reuse the ids of a
    collected one must not find (and write into) the dead optimizer's arena."""
    import gc
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.rccl.flat_adam import FlatAdam
    n_before = len(HF._GRAD_SLOTS)
    net, opt = _mirrored_steps(lambda n, lr: FlatAdam([{'params': list(n.parameters()), 'lr': lr}]), steps=1)
    assert len(HF._GRAD_SLOTS) > n_before
    del net, opt
    gc.collect()
    assert len(HF._GRAD_SLOTS) <= n_before, 'gradient slots of a collected optimizer are still registered'
    x, tgt = _stack_inputs()
    torch.manual_seed(11)
    ref = _BlockStack().to(DEV).train()
    ((ref(x) - tgt) ** 2).mean().backward()
    for p in ref.parameters():
        assert p.grad is not None and HF._slot_of(p) is None


# ------------------------------------------------------------------------------- (c) the reducer path, 1-rank RCCL group
def test_forced_collectives_step_equals_plain_step():
    """hvd.DistributedOptimizer(FlatAdam, force_collectives=True) in a 1-rank RCCL group -- buckets = slices of the gradient
    arena, all-reduce (ReduceOp.AVG) on the side stream from post-accumulate hooks, join before the update: after synchronize()
    the arena holds the same gradients the plain path produces, and the update equals torch.optim.Adam on them (reference:
    horovod_trainer.py:46-48,92-93)."""
    import torch.distributed as dist
    from packnet_sfm.rccl import hvd
    from packnet_sfm.rccl.flat_adam import FlatAdam
    created = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
        created = True
    try:
        assert dist.get_backend() == 'nccl'

        def make(n, lr):
            opt = hvd.DistributedOptimizer(FlatAdam([{'params': list(n.parameters()), 'lr': lr}]),
                                           named_parameters=n.named_parameters(), compression=hvd.Compression.none,
                                           bucket_bytes=64 << 10, force_collectives=True)
            assert opt._reducer.force and len(opt._reducer.buckets) >= 3
            return opt
        net, opt = _mirrored_steps(make)
        assert opt._reducer._launched == len(opt._reducer.buckets)      # every bucket went through the collective
    finally:
        if created:
            dist.destroy_process_group()


# --------------------------------------------------------------------- (d) inputs outside the split arithmetic's envelope
def test_conv2d_bx3_edge_inputs():
    """What the split-bf16 kernels do outside their envelope, pinned:
      * +-inf, NaN, |x| >= 3.39e38 (bf16 rounding overflows): every output whose receptive field holds such a value is NON-FINITE
        (the residual x - bf16(x) is inf - inf) -- never a plausible finite number -- and every other output is bit-identical to
        the clean run;
      * tiny magnitudes (1e-30 ... 1e-37, where the low pieces reach the bf16 denormal range): the error against fp64 stays
        below 2^-7 of sum|x||w| even if the matrix pipe flushes denormal pieces; measured value printed."""
    from packnet_sfm.hip import ops, functional as HF
    assert HF.get_conv_math() == 'bx3'
    B, Cin, Cout, H, W, ks = 1, 32, 32, 16, 32, 3
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.1
    wf, wb = ops.conv2d_pack(w.to(DEV))
    clean = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks).cpu()
    assert torch.isfinite(clean).all()
    for bad in (float('inf'), float('-inf'), float('nan'), 3.4e38, -3.4e38):
        xb = x.clone()
        xb[0, 5, 7, 11] = bad
        y = ops.conv2d_forward(xb.to(DEV), wf, None, Cout, ks).cpu()
        touched = torch.zeros(H, W, dtype=torch.bool)
        touched[6:9, 10:13] = True
        assert not torch.isfinite(y[0][:, touched]).any(), 'value %r produced finite outputs inside its receptive field' % bad
        assert torch.equal(y[0][:, ~touched], clean[0][:, ~touched]), 'value %r leaked outside its receptive field' % bad
    # largest magnitude inside the envelope: exact like any other value
    xm = x.clone()
    xm[0, 5, 7, 11] = 3.3e38
    wm = w.clone() * 1e-3
    wfm, _ = ops.conv2d_pack(wm.to(DEV))
    y = ops.conv2d_forward(xm.to(DEV), wfm, None, Cout, ks).cpu().double()
    y64 = F.conv2d(xm.double(), wm.double(), padding=1)
    mag = F.conv2d(xm.double().abs(), wm.double().abs(), padding=1)
    assert float(((y - y64).abs() / mag).max()) <= 8 * 2.0 ** -24
    for scale in (1e-30, 1e-34, 1e-36):
        xs = x * scale
        y = ops.conv2d_forward(xs.to(DEV), wf, None, Cout, ks).cpu().double()
        y64 = F.conv2d(xs.double(), w.double(), padding=1)
        mag = F.conv2d(xs.double().abs(), w.double().abs(), padding=1)
        e = float(((y - y64).abs() / mag).max())
        print('inputs scaled by %.0e: max |err| / sum|x||w| = %.2e' % (scale, e))
        assert e <= 2.0 ** -7, (scale, e)


# -------------------------------------------------------------------------- (e) two data-parallel ranks (RCCL when possible)
def test_two_rank_gradient_averaging():
    """tests/rccl_two_ranks.py under torch.distributed.run, 2 ranks: averaged gradients == the single-process full batch, the
    parameters after two steps too, replicas bit-identical.  With >= 2 devices this is RCCL over xGMI (one device per rank) and the
    step is also timed with the side-stream overlap on / off; on a 1-GPU box the two ranks share the device and the same code
    runs over gloo with host-staged buckets (functional rehearsal)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(here, 'rccl_two_ranks.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, 'two-rank run failed:\nSTDOUT:\n%s\nSTDERR:\n%s' % (r.stdout[-3000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    print(line)
    assert res['ok']
    if torch.cuda.device_count() >= 2:
        assert res['backend'] == 'nccl' and res['timing'] is not None


# ------------------------------------------------------------------------------------- (f) the decoder's concatenations
@pytest.mark.parametrize('case', [(2, (64, 64, 1), 64, 48, 160, 3, 0), (4, (128, 128, 1), 128, 24, 80, 3, 1), (2, (512, 512), 512, 12, 40, 3, 2),
                                  (1, (64, 64, 1), 64, 192, 640, 3, 3), (2, (64, 1), 64, 48, 160, 3, 4), (2, (128, 33), 64, 24, 80, 3, 5)])
def test_conv2d_cat_multi_source_gpu(case):
    """iconv1 .. iconv5 of PackNet01 read cat(unpacked, skip[, upsampled inverse depth]) (reference PackNet01.py:138-174); here the
    concatenation is folded into the K loop of the split-bf16 forward and weight-gradient kernels.  Against F.conv2d on the
    concatenated tensor at the real layer shapes (autotuner on: every candidate tiling reads the three tensors)."""
    from test_kernels_emulated import _check_conv2d_cat
    _check_conv2d_cat(DEV, *case)
