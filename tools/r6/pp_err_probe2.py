"""Round-6 probe: fp64 error of the f32-MFMA kernels and the split-bf16 kernels at the SAME K split (the error follows the length of
the fp32 accumulation chain), and of the reference's own fp32 convolution on the CPU."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'packnet-sfm_amd'))
import torch, torch.nn.functional as F
from packnet_sfm.hip import _lib, ops, functional as HF
lib = _lib.get()
DEV = 'cuda'
def last():
    out = (ctypes.c_int * 8)(); lib.pnsfm_conv2d_last_config(out); return list(out)
for shape in [(1, 64, 64, 48, 160, 7), (1, 2048, 64, 24, 80, 5), (4, 512, 512, 6, 20, 3), (1, 64, 64, 96, 320, 3)]:
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    y64 = F.conv2d(x.double(), w.double(), padding=ks // 2)
    ymag = F.conv2d(x.double().abs(), w.double().abs(), padding=ks // 2)
    ycpu = F.conv2d(x, w, padding=ks // 2)
    print(shape, 'CPU fp32 conv err %.3e' % float(((ycpu.double() - y64).abs() / ymag).max()))
    for mode, base, variant in (('f32', 10, 0), ('bx3', 110, 3), ('bx3', 110, 7)):
        HF.set_conv_math(mode)
        wf, _ = ops.conv2d_pack(w.to(DEV), want_bwd=False)
        key = (ctypes.c_int * 7)(base, B, Cin, Cout, H, W, ks)
        for split in (1, 2, 4, 8, 16):
            lib.pnsfm_tune_set(key, 1 | (variant << 4), split)
            y = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks)
            c = last()
            e = (y.cpu().double() - y64).abs() / ymag
            print('  %s v%d split %d (ran v%d split %d): max %.3e  rms %.3e' % (mode, variant, split, c[0], c[4], float(e.max()), float((e ** 2).mean().sqrt())))
        lib.pnsfm_set_conv_variant(3)
    HF.set_conv_math('bx3')
