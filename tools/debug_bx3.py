"""Debug aid: run training steps with every stride-1 conv forward / backward-data checked against MIOpen (torch) on the fly."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'packnet-sfm_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from packnet_sfm.hip import ops, functional as HF

bad = []
orig_f, orig_b = ops.conv2d_forward, ops.conv2d_backward_data
cur = {}

def fwd(x, wp, bias, Cout, ks):
    y = orig_f(x, wp, bias, Cout, ks)
    w = cur.get('w')
    if w is not None:
        ref = F.conv2d(x, w, bias, padding=ks // 2)
        err = float((y - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
        if not (err < 1e-4):
            print('FWD BAD', tuple(x.shape), Cout, ks, 'err', err, flush=True); bad.append(('f', tuple(x.shape), Cout, ks))
    return y

def bwd(dy, wp, Cin, ks):
    dx = orig_b(dy, wp, Cin, ks)
    w = cur.get('w')
    if w is not None:
        ref = F.conv_transpose2d(dy, w, padding=ks // 2)
        err = float((dx - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
        if not (err < 1e-4):
            print('DGRAD BAD', tuple(dy.shape), Cin, ks, 'err', err, flush=True); bad.append(('b', tuple(dy.shape), Cin, ks))
    return dx

ops.conv2d_forward, ops.conv2d_backward_data = fwd, bwd
of, ob = HF.Conv2dFn.forward, HF.Conv2dFn.backward

def F2(ctx, x, weight, bias, cache, recording=True):
    cur['w'] = weight.detach()
    ctx.w_dbg = weight.detach()
    try:
        return of(ctx, x, weight, bias, cache, recording)
    finally:
        cur['w'] = None

def B2(ctx, dy):
    cur['w'] = ctx.w_dbg
    try:
        return ob(ctx, dy)
    finally:
        cur['w'] = None

HF.Conv2dFn.forward = staticmethod(F2)
HF.Conv2dFn.backward = staticmethod(B2)

import bench
torch.manual_seed(0)
dev = torch.device('cuda:0')
model = bench.build_model(dev)
batch = bench.synthetic_batch(4, 192, 640, 1, dev)
opt = torch.optim.Adam(model.parameters(), lr=2e-4)
for step in range(3):
    opt.zero_grad()
    out = model(batch, progress=0.0)
    loss = out['loss']
    loss.backward()
    torch.cuda.synchronize()
    nonfinite = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print('step', step, 'loss', float(loss), 'non-finite grads:', nonfinite[:8], flush=True)
    opt.step()
print('bad calls:', len(bad))
