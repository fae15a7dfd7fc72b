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


def _run_stack(steps, make_opt, x, tgt, check_arena=False):
    from packnet_sfm.hip import functional as HF
    torch.manual_seed(11)
    net = _BlockStack().to(DEV).train()
    opt = make_opt(net)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ((net(x) - tgt) ** 2).mean()
        loss.backward()
        if check_arena:
            inner = getattr(opt, '_opt', opt)
            ranges = _arena_range(inner)
            n_in = 0
            for name, p in net.named_parameters():
                if p.dim() == 4:            # Conv2d weights: leaves used once -> the kernel wrote them into the arena slice
                    view = inner.grad_view([g for g in inner.param_groups if id(p) in g['_offs']][0], p)
                    inside = any(lo <= p.grad.data_ptr() < hi for lo, hi in ranges)
                    if name.startswith('pack.conv.conv_base'):
                        continue            # its gradient flows through the composed kernel (non-leaf): gathered, not slotted
                    assert inside and p.grad.data_ptr() == view.data_ptr(), '%s: gradient not produced inside the arena' % name
                    n_in += 1
            assert n_in >= 6, n_in
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return net, losses


def _noise_only(net):
    """Names of conv biases feeding a GroupNorm with ONE channel per group: their gradient is mathematically zero, both runs see
    pure round-off, and Adam turns the sign of that noise into +-lr steps (they do not influence the output)."""
    from packnet_sfm.networks.layers.packnet.layers01 import Conv2D
    return {n + '.conv_base.bias' for n, m in net.named_modules() if isinstance(m, Conv2D) and m.normalize.num_channels == m.normalize.num_groups}


def _stack_inputs():
    g = torch.Generator().manual_seed(5)
    return torch.randn(2, 16, 32, 64, generator=g).to(DEV), torch.randn(2, 16, 32, 64, generator=g).to(DEV)


def test_flat_adam_arena_slots_on_real_blocks():
    """Bench default path: FlatAdam + register_grad_slots.  Three optimizer steps on a stack of real blocks equal
    torch.optim.Adam on plain (freshly allocated) gradients of an identical replica; the conv weight gradients live inside the
    gradient arena (pointer check), i.e. the wgrad kernel -> arena slice -> adam_flat_kernel path is what ran."""
    from packnet_sfm.rccl.flat_adam import FlatAdam
    x, tgt = _stack_inputs()
    ref, lref = _run_stack(3, lambda n: torch.optim.Adam(n.parameters(), lr=2e-3), x, tgt)
    net, lflat = _run_stack(3, lambda n: FlatAdam([{'params': list(n.parameters()), 'lr': 2e-3}]), x, tgt, check_arena=True)
    for a, b in zip(lflat, lref):
        assert abs(a - b) <= 1e-5 * abs(b), (lflat, lref)
    skip = _noise_only(net)
    assert skip == {'unpack.conv.conv_base.bias', 'head.conv_base.bias'}, skip
    for (name, p), q in zip(net.named_parameters(), ref.parameters()):
        if name in skip:
            continue
        P.check(p, q, 1e-5, 'parameter ' + name, floor=1e-2)


def test_flat_adam_slots_released_with_the_optimizer():
    """gradient slots must not outlive their optimizer -- a second model whose parameters reuse the ids of a
    collected one must not find (and write into) the dead optimizer's arena."""
    import gc
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.rccl.flat_adam import FlatAdam
    x, tgt = _stack_inputs()
    n_before = len(HF._GRAD_SLOTS)
    net, _ = _run_stack(1, lambda n: FlatAdam([{'params': list(n.parameters()), 'lr': 2e-3}]), x, tgt)
    del net
    gc.collect()
    assert len(HF._GRAD_SLOTS) <= n_before, 'gradient slots of a collected optimizer are still registered'
    ref, lref = _run_stack(2, lambda n: torch.optim.Adam(n.parameters(), lr=2e-3), x, tgt)
    for p in ref.parameters():
        assert p.grad is not None and HF._slot_of(p) is None


# ------------------------------------------------------------------------------- (c) the reducer path, 1-rank RCCL group
def test_forced_collectives_step_equals_plain_step():
    """hvd.DistributedOptimizer(FlatAdam, force_collectives=True) in a 1-rank RCCL group -- buckets = slices of the gradient
    arena, all-reduce (ReduceOp.AVG) on the side stream from post-accumulate hooks, join before the update -- gives the same
    parameters as the plain FlatAdam step (reference: horovod_trainer.py:46-48,92-93)."""
    import torch.distributed as dist
    from packnet_sfm.rccl import hvd
    from packnet_sfm.rccl.flat_adam import FlatAdam
    x, tgt = _stack_inputs()
    plain, lplain = _run_stack(3, lambda n: FlatAdam([{'params': list(n.parameters()), 'lr': 2e-3}]), x, tgt)
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

        def make(n):
            opt = hvd.DistributedOptimizer(FlatAdam([{'params': list(n.parameters()), 'lr': 2e-3}]),
                                           named_parameters=n.named_parameters(), compression=hvd.Compression.none,
                                           bucket_bytes=64 << 10, force_collectives=True)
            assert opt._reducer.force and len(opt._reducer.buckets) >= 3
            return opt
        ddp, lddp = _run_stack(3, make, x, tgt, check_arena=True)
        assert lddp == pytest.approx(lplain, rel=1e-5)
        skip = _noise_only(ddp)
        for (name, p), q in zip(ddp.named_parameters(), plain.parameters()):
            if name in skip:
                continue
            P.check(p, q, 1e-5, 'parameter ' + name, floor=1e-2)
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
    for scale in (1e-30, 1e-34, 1e-37):
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
