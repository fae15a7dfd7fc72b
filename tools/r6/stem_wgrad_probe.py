"""The stem's weight gradient (3 -> C, 5x5): conv2d_wgrad_stem5_kernel (split-bf16 arithmetic) against the generic f32 kernel, over
pixel splits; both checked against fp64.
usage: python tools/r6/stem_wgrad_probe.py"""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
import torch.nn.functional as F
from packnet_sfm.hip import _lib, ops
lib = _lib.get()
lib.pnsfm_set_autotune(0)


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best * 1e3


for (B, Cin, Cout, H, W, ks) in [(4, 3, 64, 192, 640, 5), (2, 3, 64, 384, 1280, 5), (4, 3, 32, 192, 640, 5)]:
    x = torch.randn(B, Cin, H, W, device='cuda'); dy = torch.randn(B, Cout, H, W, device='cuda')
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, ks, ks), dy.double(), padding=ks // 2)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), (Cout, Cin, ks, ks), dy.double().abs(), padding=ks // 2)
    key = (ctypes.c_int * 7)(12, B, Cin, Cout, H * W, W, ks)
    run = lambda: ops.conv2d_backward_weight(x, dy, ks)
    for math, name, splits in ((0, 'generic f32', (480, 960, 1920)), (1, 'stem kernel', (128, 192, 256, 320, 384, 512, 640, 768, 960, 1280, 1920))):
        lib.pnsfm_set_conv_math(math)
        for split in splits:
            assert lib.pnsfm_tune_set(key, split, 0) == 0
            dw, db = run()
            err = float(((dw.double() - ref).abs() / mag).max())
            eb = float(((db.double() - dy.double().sum((0, 2, 3))).abs() / dy.double().abs().sum((0, 2, 3))).max())
            out = (ctypes.c_int * 8)(); lib.pnsfm_conv2d_last_config(out)
            t = timeit(run)
            print('%s %-12s split %4d: %.1f us  err %.1e  dbias err %.1e  %s' % ((B, Cin, Cout, H, W, ks), name, split, t, err, eb, list(out)[:5]), flush=True)
    lib.pnsfm_set_conv_math(1)
    lib.pnsfm_set_wgrad_variant(-1)
