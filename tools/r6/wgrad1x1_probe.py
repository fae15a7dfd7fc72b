"""The LDS-free 1x1 weight gradient (csrc/conv2d_wgrad1.hip, variant 4) against the shipped decision, per 1x1 backward-weight shape of the
tuning database: times wave tile x pixel split, checks every result against fp64.  Needs tools/r6/wgrad1x1_lds_free.patch (NOT in the
tree: no gain -- profiles/r06_wgrad1x1_probe.txt): git apply it and rebuild.
usage: python tools/r6/wgrad1x1_probe.py"""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops
lib = _lib.get()
db = [list(map(int, l.split())) for l in open(os.path.join(ROOT, 'packnet-sfm_amd', 'csrc', 'tuned_gfx950.db')) if l.strip()]
rows = [r for r in db if r[0] == 112 and r[6] == 1]
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


tot0 = tot1 = 0.0
for r in rows:
    _, B, Cin, Cout, HW, W, ks, d0, d1 = r
    if HW % 16:
        continue
    H = HW // 32
    x = torch.randn(B, Cin, H, 32, device='cuda'); dy = torch.randn(B, Cout, H, 32, device='cuda')
    ref = torch.einsum('bop,bip->oi', dy.double().flatten(2), x.double().flatten(2))
    mag = torch.einsum('bop,bip->oi', dy.double().abs().flatten(2), x.double().abs().flatten(2))
    key = (ctypes.c_int * 7)(*r[:7])
    run = lambda: ops.conv2d_backward_weight(x, dy, 1)
    assert lib.pnsfm_tune_set(key, d0, d1) == 0
    t0 = timeit(run)
    e0 = float(((run()[0].double().flatten(1) - ref).abs() / mag).max())
    best = (1e9, None)
    steps = B * HW // 16
    for T in (1, 2):
        base = -(-Cin // (32 * T)) * -(-Cout // (32 * T))
        seen = set()
        for wgs in (0, 64, 128, 256, 512, 1024):
            split = max(1, min(steps, round(wgs / base))) if wgs else 1
            if split in seen:
                continue
            seen.add(split)
            assert lib.pnsfm_tune_set(key, split, 4 | (T << 4)) == 0
            dw, db_ = run()
            err = float(((dw.double().flatten(1) - ref).abs() / mag).max())
            eb = float((db_.double() - dy.double().sum((0, 2, 3))).abs().max() / dy.double().abs().sum((0, 2, 3)).max())
            assert err < 2e-7 and eb < 2e-7, (r, T, split, err, eb)
            t = timeit(run)
            if t < best[0]:
                best = (t, (T, split, base * split), err)
    lib.pnsfm_set_wgrad_variant(-1)
    tot0 += t0; tot1 += min(t0, best[0])
    print('%-34s shipped (%d, %d) %.1f us err %.1e | no-LDS best %.1f us %s err %.1e%s' %
          (tuple(r[:7]), d0, d1, t0, e0, best[0], best[1], best[2], '   <--' if best[0] < 0.97 * t0 else ''), flush=True)
print('sum over shapes: shipped %.1f us, with the LDS-free kernel where faster %.1f us' % (tot0, tot1))
