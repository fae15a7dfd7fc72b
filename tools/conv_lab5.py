"""GPU lab (round 5): the forward / backward-data launches of the step under the library's tuned configuration against pinned
configurations of a candidate variant (default: 7, the ping-pong workgroup of conv2d_bx3pp.h).
    python tools/conv_lab5.py [--variant 7] [--shapes all|body|big] [--g G]
Prints per shape: tuned ms / TFLOP/s, the best candidate (NT, narrow-M, tile mode, split) ms / TFLOP/s, max |difference|."""
import argparse
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument('--variant', type=int, default=7)
ap.add_argument('--shapes', default='all')
ap.add_argument('--splits', default='1,2,4,8,16')
args = ap.parse_args()
dev = torch.device('cuda:0')
lib = _lib.get()
BODY = [(4, 64, 64, 96, 320, 3), (4, 128, 128, 48, 160, 3), (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3)]
BIG = [(4, 64, 64, 192, 640, 7), (4, 256, 64, 96, 320, 7), (4, 64, 256, 96, 320, 7), (4, 64, 129, 192, 640, 3), (4, 129, 64, 192, 640, 3),
       (4, 512, 16384, 6, 20, 3), (4, 16384, 512, 6, 20, 3), (4, 256, 8192, 12, 40, 3), (4, 8192, 256, 12, 40, 3),
       (8, 2048, 64, 4, 320, 5), (8, 64, 2048, 4, 320, 5)]
MID = [(4, 256, 64, 48, 160, 5), (4, 512, 128, 24, 80, 5), (4, 128, 512, 24, 80, 5), (4, 64, 256, 48, 160, 5), (4, 384, 256, 24, 80, 3),
       (4, 64, 129, 96, 320, 3), (4, 129, 64, 96, 320, 3), (4, 128, 193, 48, 160, 3), (4, 193, 128, 48, 160, 3), (4, 256, 384, 24, 80, 3),
       (4, 512, 768, 12, 40, 3), (4, 768, 512, 12, 40, 3), (4, 64, 64, 96, 320, 1), (4, 256, 256, 24, 80, 1)]
SHAPES = {'body': BODY, 'big': BIG, 'mid': MID, 'all': BODY + BIG + MID}[args.shapes]


def timeit(fn, reps=10):
    for _ in range(2):
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


tot_base = tot_best = 0.0
for shape in SHAPES:
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5).to(dev)
    wf, _ = ops.conv2d_pack(w, want_bwd=False)
    gf = 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9
    key = (ctypes.c_int * 7)(110, B, Cin, Cout, H, W, ks)
    lib.pnsfm_set_conv_variant(3)            # clears pins: the tuned (database / autotuned) configuration
    y0 = ops.conv2d_forward(x, wf, None, Cout, ks)
    base = timeit(lambda: ops.conv2d_forward(x, wf, None, Cout, ks))
    best = (1e9, None, 0.0)
    tms = (0, 1, 2) if W % 32 != 0 else ((0, 1) if ks >= 5 else (0,))
    for NT in (2, 1):
        for narrow in (0, 1):
            for tm in tms:
                last = None
                for split in [int(v) for v in args.splits.split(',')]:
                    if split > max(1, (Cin + 15) // 16):
                        break
                    lib.pnsfm_tune_set(key, NT | (args.variant << 4) | (narrow << 8) | (tm << 9), split)
                    try:
                        y = ops.conv2d_forward(x, wf, None, Cout, ks)
                    except Exception as e:
                        continue
                    ms = timeit(lambda: ops.conv2d_forward(x, wf, None, Cout, ks), reps=6)
                    if last is not None and abs(ms - last) < 1e-6:
                        continue
                    last = ms
                    if ms < best[0]:
                        best = (ms, (NT, narrow, tm, split), float((y - y0).abs().max() / y0.abs().max()))
    lib.pnsfm_set_conv_variant(3)
    tot_base += base
    tot_best += min(best[0], base)
    print('%-30s tuned %.4f ms %6.1f TF | variant %d best %s %.4f ms %6.1f TF (%+.1f %%) relerr %.1e' % (
        shape, base, gf / base, args.variant, best[1], best[0], gf / best[0], 100.0 * (base / best[0] - 1.0), best[2]), flush=True)
print('total tuned %.4f ms, with the candidate where it wins %.4f ms (%.1f %%)' % (tot_base, tot_best, 100.0 * (tot_base / tot_best - 1.0)))
