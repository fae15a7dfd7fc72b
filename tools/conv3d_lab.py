"""GPU lab for the Conv3d stencil kernels: time per launch of forward / backward-data / backward-weight on the step's shapes.
usage: [PNSFM_STENCIL_XCD_MAP=0|1] python tools/conv3d_lab.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops
if os.environ.get('PNSFM_LAB_LIB'):        # another build of the library (same-box A/B)
    _lib.LIB_PATH = os.path.abspath(os.environ['PNSFM_LAB_LIB'])

dev = torch.device('cuda:0')
SHAPES = [(4, 32, 96, 320), (4, 32, 48, 160), (4, 64, 24, 80), (4, 128, 12, 40), (4, 256, 6, 20), (4, 1024, 12, 40), (4, 2048, 6, 20),
          (8, 256, 5, 320), (8, 256, 96, 5)]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tag = os.environ.get('PNSFM_LAB_TAG', os.environ.get('PNSFM_STENCIL_XCD_MAP', '1'))
tot = [0.0, 0.0, 0.0]
for B, D, H, W in SHAPES:
    g = torch.Generator().manual_seed(1)
    p = torch.randn(B, D, H, W, generator=g).to(dev)
    w3 = torch.randn(8, 1, 3, 3, 3, generator=g).to(dev)
    b3 = torch.randn(8, generator=g).to(dev)
    dout = torch.randn(B, 8 * D, H, W, generator=g).to(dev)
    t = (timeit(lambda: ops.conv3d_forward(p, w3, b3)), timeit(lambda: ops.conv3d_backward_data(dout, w3)),
         timeit(lambda: ops.conv3d_backward_weight(p, dout)))
    tot = [a + b for a, b in zip(tot, t)]
    mb = dout.numel() * 4 / 1e6
    print('xcd-map %s  %-20s dout %6.1f MB  fwd %7.1f us  dgrad %7.1f us (%.2f TB/s)  wgrad %7.1f us' % (
        tag, (B, D, H, W), mb, t[0], t[1], mb * 1.125 / t[1], t[2]), flush=True)
print('xcd-map %s  total fwd %.1f dgrad %.1f wgrad %.1f us' % (tag, *tot))
