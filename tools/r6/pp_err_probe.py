"""Round-6 probe: error vs fp64 of the split-bf16 forward kernels by variant / tiling on one shape (is variant 7 different from 3?)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'packnet-sfm_amd'))
import torch, torch.nn.functional as F
from packnet_sfm.hip import _lib, ops, functional as HF
lib = _lib.get()
DEV = 'cuda'
def last():
    out = (ctypes.c_int * 8)(); lib.pnsfm_conv2d_last_config(out); return list(out)
for shape in [(1, 64, 64, 48, 160, 7), (1, 2048, 64, 24, 80, 5), (4, 512, 512, 6, 20, 3)]:
    B, Cin, Cout, H, W, ks = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
    y64 = F.conv2d(x.double(), w.double(), padding=ks // 2)
    ymag = F.conv2d(x.double().abs(), w.double().abs(), padding=ks // 2)
    res = {}
    HF.set_conv_math('f32')
    wf, _ = ops.conv2d_pack(w.to(DEV), want_bwd=False)
    y = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks)
    print(shape, 'f32 err %.3e' % float(((y.cpu().double() - y64).abs() / ymag).max()), last())
    HF.set_conv_math('bx3')
    wf, _ = ops.conv2d_pack(w.to(DEV), want_bwd=False)
    key = (ctypes.c_int * 7)(110, B, Cin, Cout, H, W, ks)
    for variant in (3, 4, 5, 7):
        for NT in (1, 2):
            for tm in (0, 1, 2):
                for split in (1, 4):
                    lib.pnsfm_tune_set(key, NT | (variant << 4) | (tm << 9), split)
                    y = ops.conv2d_forward(x.to(DEV), wf, None, Cout, ks)
                    c = last()
                    if c[0] != variant or c[1] != NT or c[5] != tm or c[4] != split:
                        continue
                    e = float(((y.cpu().double() - y64).abs() / ymag).max())
                    k = (NT, tm, split)
                    same = ''
                    if k in res:
                        same = 'equal to v%d: %s' % (res[k][0], torch.equal(res[k][1], y))
                    else:
                        res[k] = (variant, y)
                    print('  v%d NT%d tm%d split%d G%d err %.3e %s' % (variant, NT, tm, split, c[3], e, same))
    lib.pnsfm_set_conv_variant(3)
