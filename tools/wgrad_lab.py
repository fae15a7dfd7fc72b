"""GPU lab for the weight-gradient kernels: one autotuned call per shape (PNSFM_TUNE_LOG lists every candidate kernel / split
with its time), error of the chosen configuration against MIOpen, time per launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops, functional as HF
if os.environ.get('PNSFM_LAB_LIB'):        # another build of the library (same-box A/B), its tuned_gfx950.db next to it
    _lib.LIB_PATH = os.path.abspath(os.environ['PNSFM_LAB_LIB'])

dev = torch.device('cuda:0')
SHAPES = [(4, 64, 64, 192, 640, 7), (4, 64, 256, 96, 320, 7), (4, 256, 64, 96, 320, 7), (4, 129, 64, 192, 640, 3),
          (4, 64, 64, 96, 320, 3), (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3), (4, 8192, 256, 12, 40, 3),
          (8, 2048, 64, 4, 320, 5), (4, 512, 128, 24, 80, 5), (4, 128, 128, 48, 160, 3), (4, 16384, 512, 6, 20, 3),
          (4, 512, 512, 6, 20, 3), (8, 2048, 64, 96, 4, 5)]
if len(sys.argv) > 1 and sys.argv[1] == 'step':
    # (shape, launches per step) of the pixel-split layers of the 192x640 batch-4 step
    SHAPES = [(4, 64, 64, 192, 640, 7), (4, 256, 64, 96, 320, 7), (4, 64, 256, 96, 320, 7), (4, 129, 64, 192, 640, 3), (4, 64, 64, 96, 320, 3),
              (4, 128, 128, 48, 160, 3), (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3), (4, 256, 64, 48, 160, 5), (4, 512, 128, 24, 80, 5),
              (4, 129, 64, 96, 320, 3), (4, 193, 128, 48, 160, 3), (4, 64, 64, 96, 320, 1), (4, 256, 256, 24, 80, 1)]
elif len(sys.argv) > 1:
    SHAPES = SHAPES[int(sys.argv[1]):]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mode in ('bx3',):
    HF.set_conv_math(mode)
    for shape in SHAPES:
        B, Cin, Cout, H, W, ks = shape
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Cin, H, W, generator=g).to(dev)
        dy = torch.randn(B, Cout, H, W, generator=g).to(dev)
        dw, db = ops.conv2d_backward_weight(x, dy, ks)
        if os.environ.get('PNSFM_LAB_NOREF'):
            err = float('nan')
        else:
            ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, ks, ks), dy, padding=ks // 2)
            err = float((dw - ref).abs().max() / ref.abs().max())
        errb = float((db - dy.sum((0, 2, 3))).abs().max() / dy.sum((0, 2, 3)).abs().max())
        ms = timeit(lambda: ops.conv2d_backward_weight(x, dy, ks), reps=20)
        gf = 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9
        print(os.environ.get('PNSFM_LAB_TAG', 'base' if os.environ.get('PNSFM_LAB_LIB') else 'new'), mode, shape, 'err dw %.1e db %.1e  %.3f ms  %.1f TF' % (err, errb, ms, gf / ms), flush=True)
