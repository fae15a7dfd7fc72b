"""The LDS-free 1x1 kernel (tuner variant 8, csrc/conv2d_bx3_1x1.h) against the shipped decision, per 1x1 forward / backward-data shape
of the tuning database: times NT x narrow-M x K split, compares the bits with variant 3 at the same split.
usage: python tools/r6/conv1x1_probe.py"""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import _lib, ops
lib = _lib.get()
vp = ctypes.c_void_p
db = [list(map(int, l.split())) for l in open(os.path.join(ROOT, 'packnet-sfm_amd', 'csrc', 'tuned_gfx950.db')) if l.strip()]
rows = [r for r in db if r[0] in (110, 111) and r[6] == 1]
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
    kind, B, K, M, H, W, ks, d0, d1 = r
    x = torch.randn(B, K, H, W, device='cuda')
    w = torch.randn(M, K, 1, 1, device='cuda') * 0.1        # as a forward layer K -> M: the kernel is the same for both kinds
    bias = torch.randn(M, device='cuda')
    wf, _ = ops.conv2d_pack(w)
    keyf = (ctypes.c_int * 7)(110, B, K, M, H, W, ks)
    run = lambda: ops.conv2d_forward(x, wf, bias, M, 1)
    assert lib.pnsfm_tune_set(keyf, d0, d1) == 0
    t0 = timeit(run)
    best = (1e9, None)
    for NT in (1, 2):
        for narrow in (0, 1):
            for split in (1, 2, 4):
                if split > K // 16:
                    continue
                assert lib.pnsfm_tune_set(keyf, NT | (8 << 4) | (narrow << 8), split) == 0
                y8 = run()
                t = timeit(run)
                assert lib.pnsfm_tune_set(keyf, NT | (3 << 4) | (narrow << 8), split) == 0
                y3 = run()
                assert torch.equal(y8, y3), (r, NT, narrow, split, float((y8 - y3).abs().max()))
                if t < best[0]:
                    best = (t, (NT, narrow, split))
    lib.pnsfm_set_conv_variant(3)
    tot0 += t0; tot1 += min(t0, best[0])
    print('%-34s shipped (%d, %d) %.1f us | no-LDS best %.1f us %s%s' % (tuple(r[:7]), d0, d1, t0, best[0], best[1], '   <--' if best[0] < 0.97 * t0 else ''), flush=True)
print('sum over shapes: shipped %.1f us, with the LDS-free kernel where faster %.1f us' % (tot0, tot1))
