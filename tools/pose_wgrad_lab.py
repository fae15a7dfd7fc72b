"""GPU lab: the stride-2 weight gradients of PoseNet (f32 MFMA kernel conv2d_wgrad_kernel) at the 192x640 batch-4 step's shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import ops

dev = torch.device('cuda:0')
B, H, W = 4, 192, 640
chain = [(9, 16, 7), (16, 32, 5), (32, 64, 3), (64, 128, 3), (128, 256, 3), (256, 256, 3), (256, 256, 3)]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = 0.0
for cin, cout, ks in chain:
    P = ks // 2
    Ho, Wo = (H + 2 * P - ks) // 2 + 1, (W + 2 * P - ks) // 2 + 1
    x = torch.randn(B, cin, H, W, device=dev)
    dy = torch.randn(B, cout, Ho, Wo, device=dev)
    t = timeit(lambda: ops.conv2d_backward_weight_strided(x, dy, ks, 2, want_bias=True))
    gf = 2.0 * B * cin * cout * ks * ks * Ho * Wo / 1e9
    print('%3d -> %3d k%d s2  in %3dx%3d out %3dx%3d  %7.1f us  %6.2f GF  %5.1f TF' % (cin, cout, ks, H, W, Ho, Wo, t, gf, gf / t * 1e-3 * 1e3 / 1e3), flush=True)
    tot += t
    H, W = Ho, Wo
print('total %.1f us' % tot)
