"""Check EVERY configuration the conv autotuner may pick for a layer shape against torch's convolution, several
repetitions each (catches races that only some configurations / timings expose).
usage: python tools/conv_config_sweep.py [shape ...]   shape = B,Cin,Cout,H,W,ks"""
import ctypes
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from packnet_sfm.hip import _lib, ops  # noqa: E402

lib = _lib.get()
SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [
    (1, 3, 64, 64, 96, 5), (1, 64, 64, 64, 96, 7), (1, 64, 64, 32, 48, 3), (1, 256, 64, 32, 48, 7), (1, 128, 128, 16, 24, 3),
    (1, 256, 256, 8, 12, 3), (1, 512, 512, 4, 6, 3), (1, 129, 64, 64, 96, 3), (1, 64, 32, 32, 48, 3), (1, 512, 512, 2, 3, 3),
    (1, 256, 256, 8, 12, 1), (4, 256, 256, 24, 80, 3), (4, 64, 64, 96, 320, 3), (4, 512, 512, 12, 40, 3)]


def key(*v):
    return (ctypes.c_int * 7)(*v)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


bad = 0
for (B, Cin, Cout, H, W, ks) in SHAPES:
    g = torch.Generator().manual_seed(B + Cin + Cout + H + W + ks)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5).cuda()
    dy = torch.randn(B, Cout, H, W, generator=g).cuda()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=ks // 2)
    yr.backward(dy.double())
    wf, wb = ops.conv2d_pack(w)
    worst = {}
    for variant in (0, 1, 2):
        for nt in (1, 2):
            for fmt in (0, 1):
                for split in (1, 2, 3, 4, 8, 16):
                    cfg = nt | (variant << 4) | (fmt << 8)
                    lib.pnsfm_tune_set(key(10, B, Cin, Cout, H, W, ks), cfg, split)
                    lib.pnsfm_tune_set(key(11, B, Cout, Cin, H, W, ks), cfg, split)
                    for rep in range(3):
                        try:
                            e1 = rel(ops.conv2d_forward(x, wf, None, Cout, ks).double(), yr)
                            e2 = rel(ops.conv2d_backward_data(dy, wb, Cin, ks).double(), xr.grad)
                        except Exception as ex:     # configuration does not fit (LDS): the tuner would skip it too
                            e1 = e2 = -1.0
                            break
                        worst[('fwd', variant)] = max(worst.get(('fwd', variant), 0), e1)
                        worst[('dgrad', variant)] = max(worst.get(('dgrad', variant), 0), e2)
                        if max(e1, e2) > 2e-5:
                            bad += 1
                            print('  BAD fwd/dgrad', (B, Cin, Cout, H, W, ks), 'variant', variant, 'NT', nt, 'fMT', fmt, 'split', split, 'rep', rep, '%.2e %.2e' % (e1, e2))
    Wk = 32 if ks == 1 else W
    for variant in (0, 1):
        for split in (1, 2, 3, 5, 8, 16, 40, 120):
            lib.pnsfm_tune_set(key(12, B, Cin, Cout, H * W, Wk, ks), split, variant)
            for rep in range(3):
                try:
                    dw, db = ops.conv2d_backward_weight(x, dy, ks)
                except Exception:
                    break
                e = rel(dw.double(), wr.grad)
                eb = rel(db.double(), dy.double().sum((0, 2, 3)))
                worst[('wgrad', variant)] = max(worst.get(('wgrad', variant), 0), e)
                if max(e, eb) > 5e-5:
                    bad += 1
                    print('  BAD wgrad', (B, Cin, Cout, H, W, ks), 'variant', variant, 'split', split, 'rep', rep, '%.2e %.2e' % (e, eb))
    print((B, Cin, Cout, H, W, ks), {('%s/v%d' % k): '%.1e' % v for k, v in sorted(worst.items())}, flush=True)
print('BAD configurations:', bad)
