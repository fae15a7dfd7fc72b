"""GPU lab for the split-bf16 conv kernels: pin (NT, variant, narrow-M, K-split) through pnsfm_tune_set, report the error
against MIOpen and the time per launch.  usage: bx3_lab.py [shape-set]"""
import ctypes, os, sys, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
import torch.nn.functional as F
from packnet_sfm.hip import _lib, ops, functional as HF

dev = torch.device('cuda:0')
lib = _lib.get()
SHAPES = [(4, 256, 64, 96, 320, 7), (4, 256, 64, 48, 160, 5), (4, 384, 256, 24, 80, 3), (4, 64, 64, 192, 640, 7)]
CFGS = [(2, 3, 0, 1), (2, 3, 0, 2), (2, 3, 0, 4), (2, 3, 1, 4), (1, 3, 0, 4), (2, 4, 0, 4), (2, 5, 0, 4)]
if len(sys.argv) > 1:
    CFGS = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for shape in SHAPES:
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5).to(dev)
    ref = F.conv2d(x, w, None, padding=ks // 2)
    gf = 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9
    for cfg in CFGS:
        NT, variant, narrow, split = cfg
        bx3 = variant >= 3
        HF.set_conv_math('bx3' if bx3 else 'f32')
        lib.pnsfm_set_conv_variant(0)
        key = (ctypes.c_int * 7)(10 + (100 if bx3 else 0), B, Cin, Cout, H, W, ks)
        lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8), split)
        wf, _ = ops.conv2d_pack(w, want_bwd=False)
        errs = []
        for rep in range(3):
            y = ops.conv2d_forward(x, wf, None, Cout, ks)
            errs.append(float((y - ref).abs().max() / ref.abs().max()))
        ms = timeit(lambda: ops.conv2d_forward(x, wf, None, Cout, ks))
        print(shape, 'NT %d var %d narrow %d split %d: err %s  %.3f ms  %.1f TF' % (NT, variant, narrow, split,
              ' '.join('%.1e' % e for e in errs), ms, gf / ms), flush=True)
