"""Does the conv kernel's speed depend on the DATA (power-limited clock)?  Times one layer with zero / random tensors.
usage: python tools/conv_data_micro.py B Cin Cout H W ks"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch  # noqa: E402
from packnet_sfm.hip import ops  # noqa: E402

B, Cin, Cout, H, W, ks = (int(v) for v in sys.argv[1:7])
flops = 2.0 * B * Cin * Cout * ks * ks * H * W
for kind in ('random', 'zeros', 'random', 'zeros'):
    mk = (lambda *s: torch.randn(*s, device='cuda')) if kind == 'random' else (lambda *s: torch.zeros(*s, device='cuda'))
    x, w, dy = mk(B, Cin, H, W), mk(Cout, Cin, ks, ks) * 0.05, mk(B, Cout, H, W)
    wf, wb = ops.conv2d_pack(w)
    for name, fn in (('fwd', lambda: ops.conv2d_forward(x, wf, None, Cout, ks)), ('wgrad', lambda: ops.conv2d_backward_weight(x, dy, ks))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print('%-6s %-6s %8.3f ms %6.1f TFLOP/s' % (kind, name, dt * 1e3, flops / dt / 1e12), flush=True)
