"""Nine-taps weight gradient: two pixel shares per workgroup (cfg bit 12) against the shipped decision, per 3x3 layer of the tuning
database (single-source keys).  For each layer: the shipped decision's time, then SH = 2 over ci tiles per workgroup x pixel splits;
prints the best and the largest difference of the results.  Needs the kernel variant of tools/r6/wgrad4_two_shares.patch (NOT in the
tree: the measured gain was 1-8 % on a third of the layers, ~0.1 % of the step -- profiles/r06_wgrad4_two_shares.txt): git apply it and rebuild.
usage: python tools/r6/wgrad4_shares.py"""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib
lib = _lib.get()
vp = ctypes.c_void_p
db = [list(map(int, l.split())) for l in open(os.path.join(ROOT, 'packnet-sfm_amd', 'csrc', 'tuned_gfx950.db')) if l.strip()]
rows = [r for r in db if r[0] == 112 and r[6] == 3 and r[5] % 4 == 0 and r[2] >= 16 and r[3] >= 16]
lib.pnsfm_set_autotune(0)


def timeit(run):
    for _ in range(3):
        assert run() == 0
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


tot0 = tot1 = 0.0
for r in rows:
    _, B, Cin, Cout, HW, W, ks, d0, d1 = r
    H = HW // W
    x = torch.randn(B, Cin, H, W, device='cuda'); dy = torch.randn(B, Cout, H, W, device='cuda')
    dw = torch.empty(Cout, Cin, 3, 3, device='cuda'); dbias = torch.empty(Cout, device='cuda')
    run = lambda: lib.pnsfm_conv2d_backward_weight(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), vp(dbias.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
    key = (ctypes.c_int * 7)(*r[:7])
    assert lib.pnsfm_tune_set(key, d0, d1) == 0
    t0 = timeit(run)
    ref = dw.clone()
    best = (1e9, None)
    for WCI in (1, 2):
        base = -(-Cin // (16 * WCI)) * -(-Cout // (32 * (4 // WCI)))
        TGs = (4, 5) if W > 24 else (0,)
        for TG in TGs:
            tw = 8 * (TG or 3)
            tiles = B * -(-W // tw) * -(-H // 4)
            seen = set()
            for wgs in (128, 192, 256, 384, 512, 768):
                split = max(1, min(tiles // 2, round(wgs / base)))
                tps = -(-tiles // split); split = -(-tiles // tps)
                if split in seen or split < 2:
                    continue
                seen.add(split)
                cfg = WCI | (TG << 4) | (1 << 12)
                assert lib.pnsfm_tune_set(key, split, 3 | (cfg << 4)) == 0
                t = timeit(run)
                err = float((dw - ref).abs().max() / ref.abs().max())
                assert err < 1e-5, (r, WCI, TG, split, err)
                if t < best[0]:
                    best = (t, (WCI, TG, split, base * split))
    lib.pnsfm_set_wgrad_variant(-1)
    tot0 += t0; tot1 += min(t0, best[0])
    print('%-30s shipped (%d, %d; kernel %d) %.1f us | two shares best %.1f us %s%s' %
          ((B, Cin, Cout, H, W), d0, d1, d1 & 15, t0, best[0], best[1], '   <--' if best[0] < 0.97 * t0 else ''), flush=True)
print('sum over layers: shipped %.1f us, with two shares where faster %.1f us' % (tot0, tot1))
