"""GPU lab: autotune the weight gradient of a few shapes from scratch and dump every candidate the tuner timed (PNSFM_TUNE_LOG):
    PNSFM_TUNE_DB= PNSFM_TUNE_LOG=gpurun_out/wgrad_tune.log python tools/wgrad_tune_dump.py [B,Cin,Cout,H,W,ks ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
import torch
from packnet_sfm.hip import ops

SHAPES = [(4, 64, 64, 96, 320, 3), (4, 129, 64, 192, 640, 3), (4, 128, 128, 48, 160, 3), (4, 256, 256, 24, 80, 3)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for B, Cin, Cout, H, W, ks in SHAPES:
    x = torch.randn(B, Cin, H, W, device='cuda')
    dy = torch.randn(B, Cout, H, W, device='cuda')
    ops.conv2d_backward_weight(x, dy, ks)        # first call: the autotuner times its candidates
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_backward_weight(x, dy, ks)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print((B, Cin, Cout, H, W, ks), '%.4f ms  %.1f TF' % (ms, 2.0 * B * Cin * Cout * H * W * ks * ks / 1e9 / ms), flush=True)
