"""Where does a mid-resolution 3x3 layer lose its time?  Launch time of the split-bf16 forward kernel as a function of the K extent
(input channels) at a fixed tile grid: time = fixed cost per launch/workgroup + slope * K.  usage: conv_scaling_lab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops, functional as HF

dev = torch.device('cuda:0')
lib = _lib.get()
HF.set_conv_math('bx3')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (B, Cout, H, W, ks) in [(8, 256, 24, 80, 3), (8, 512, 12, 40, 3), (16, 512, 6, 20, 3)]:
    for cfg in [(1, 3, 0, 0), (2, 3, 0, 0), (1, 3, 0, 1), (2, 3, 0, 1), (2, 4, 0, 1), (1, 3, 1, 1), (1, 3, 0, 2), (2, 3, 0, 2), (1, 3, 1, 2)]:
        NT, variant, narrow, tm = cfg
        if W % 32 == 0 and tm:
            continue
        line = []
        for Cin in (32, 64, 128, 256, 512, 1024):
            x = torch.randn(B, Cin, H, W, device=dev)
            w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
            key = (ctypes.c_int * 7)(110, B, Cin, Cout, H, W, ks)
            lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8) | (tm << 9), 1)
            wf, _ = ops.conv2d_pack(w, want_bwd=False)
            ms = timeit(lambda: ops.conv2d_forward(x, wf, None, Cout, ks))
            gf = 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9
            line.append('%4d: %6.1f us %5.1f TF' % (Cin, ms * 1e3, gf / ms))
        print('B%d Cout %d %dx%d k%d  NT %d var %d narrowM %d tm %d | ' % (B, Cout, H, W, ks, NT, variant, narrow, tm) + ' | '.join(line), flush=True)
