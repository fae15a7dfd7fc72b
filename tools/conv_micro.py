"""Micro-benchmark of the conv entry points for one shape (for rocprofv3 --pmc runs and A/B timing on the GPU box).
usage: python tools/conv_micro.py B Cin Cout H W ks [iters] [what=fwd|dgrad|wgrad|all]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch  # noqa: E402
from packnet_sfm.hip import ops  # noqa: E402

B, Cin, Cout, H, W, ks = (int(v) for v in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
what = sys.argv[8] if len(sys.argv) > 8 else 'all'
dev = 'cuda'
x = torch.randn(B, Cin, H, W, device=dev)
w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
b = torch.randn(Cout, device=dev)
dy = torch.randn(B, Cout, H, W, device=dev)
wf, wb = ops.conv2d_pack(w)
flops = 2.0 * B * Cin * Cout * ks * ks * H * W


def run(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('%-6s %8.3f ms  %6.1f TFLOP/s' % (name, dt * 1e3, flops / dt / 1e12), flush=True)


if what in ('fwd', 'all'):
    run('fwd', lambda: ops.conv2d_forward(x, wf, b, Cout, ks))
if what in ('dgrad', 'all'):
    run('dgrad', lambda: ops.conv2d_backward_data(dy, wb, Cin, ks))
if what in ('wgrad', 'all'):
    run('wgrad', lambda: ops.conv2d_backward_weight(x, dy, ks))
