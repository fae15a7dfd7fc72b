"""The depth networks' 3 -> C 5x5 stem (forward, f32 MFMA): conv2d_stem5_kernel (variant 0 on these shapes) against the generic kernels
(variant 1: patch by LDS-DMA, 2: pipelined), NT = 1 / 2, checked against fp64 and bit-compared with variant 1.
usage: python tools/r6/stem_probe.py"""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
import torch.nn.functional as F
from packnet_sfm.hip import _lib, ops
lib = _lib.get()
lib.pnsfm_set_autotune(0)
for (B, Cin, Cout, H, W, ks) in [(4, 3, 64, 192, 640, 5), (2, 3, 64, 384, 1280, 5), (4, 3, 32, 192, 640, 5)]:
    x = torch.randn(B, Cin, H, W, device='cuda'); w = torch.randn(Cout, Cin, ks, ks, device='cuda') * 0.1; b = torch.randn(Cout, device='cuda')
    wf, wb = ops.conv2d_pack(w)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    key = (ctypes.c_int * 7)(10, B, Cin, Cout, H, W, ks)
    ys = {}
    for variant in (1, 0, 2):
        for NT in (1, 2):
            assert lib.pnsfm_tune_set(key, NT | (variant << 4), 1) == 0
            y = ops.conv2d_forward(x, wf, b, Cout, ks)
            ys[(variant, NT)] = y
            err = float((y.double() - ref).abs().max())
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv2d_forward(x, wf, b, Cout, ks)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            print('%s variant %d NT %d: %.1f us  max err %.2e  %s' % ((B, Cin, Cout, H, W, ks), variant, NT, best * 1e3, err,
                                                                    'bits == variant 1: %s' % torch.equal(y, ys[(1, NT)]) if variant != 1 else ''), flush=True)
