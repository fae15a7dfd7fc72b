"""GPU lab: time the forward launches of the step's body layers under the library's tuned configurations.  Same-box A/B of two builds:
    PNSFM_LAB_LIB=tools/micro/libpnsfm_base.so python tools/conv_lab.py ; python tools/conv_lab.py
(PNSFM_LAB_LIB: another build of libpnsfm_hip.so, e.g. the previous commit's, with its tuned_gfx950.db next to it; PNSFM_LAB_TAG
labels the output lines)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib
if os.environ.get('PNSFM_LAB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['PNSFM_LAB_LIB'])
from packnet_sfm.hip import ops

dev = torch.device('cuda:0')
SHAPES = [(4, 64, 64, 96, 320, 3), (4, 128, 128, 48, 160, 3), (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3),
          (4, 64, 129, 192, 640, 3), (4, 129, 64, 192, 640, 3), (4, 64, 64, 192, 640, 7), (4, 256, 64, 96, 320, 7),
          (4, 256, 64, 48, 160, 5), (4, 512, 128, 24, 80, 5), (4, 384, 256, 24, 80, 3), (4, 64, 64, 96, 320, 1)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


tag = os.environ.get('PNSFM_LAB_TAG', 'base' if os.environ.get('PNSFM_LAB_LIB') else 'new')
tot = 0.0
for shape in SHAPES:
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5).to(dev)
    wf, _ = ops.conv2d_pack(w, want_bwd=False)
    gf = 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9
    ms = timeit(lambda: ops.conv2d_forward(x, wf, None, Cout, ks))
    tot += ms
    print('lab %-5s %-28s %.4f ms  %.1f TF' % (tag, shape, ms, gf / ms), flush=True)
print('lab %-5s total %.4f ms' % (tag, tot), flush=True)
