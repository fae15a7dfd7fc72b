"""Bisect hipGraph replay vs eager: which part of the step differs?  (debug aid, GPU only)"""
import os, sys, random
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_cases as P
from packnet_sfm.hip import functional as HF
from tests.test_gpu_parity import _selfsup, _step_batch  # noqa
fx = dict(P.golden('step')['step_flip0'])
batch = _step_batch(fx)
side = os.environ.get('SIDE', '1') == '1'
HF.set_wgrad_stream(side)
model, dn, pn = _selfsup('cuda', fx)
model.flip_lr_prob = 0.0

def fwd():
    out = model(batch, progress=0.0)
    return out

# eager reference
out = fwd(); out['loss'].backward()
torch.cuda.synchronize()
ref_loss = out['loss'].detach().clone()
ref_d = [d.detach().clone() for d in out['inv_depths']]
ref_m = {k: v.detach().clone() for k, v in out['metrics'].items()}
ref_g = {n: p.grad.detach().clone() for n, p in dn.named_parameters()}
model.zero_grad(set_to_none=True)

what = os.environ.get('WHAT', 'fwdbwd')
g = torch.cuda.CUDAGraph()
HF.bump_weight_epoch()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    o = fwd()
    if what == 'fwdbwd':
        o['loss'].backward()
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay', rep, 'loss graph %.7f eager %.7f' % (float(o['loss']), float(ref_loss)),
          'metrics', {k: float(v) for k, v in o['metrics'].items()}, 'eager', {k: float(v) for k, v in ref_m.items()})
    for i, (a, b) in enumerate(zip(o['inv_depths'], ref_d)):
        print('   inv_depth[%d] max abs diff %.3e' % (i, float((a - b).abs().max())))
    if what == 'fwdbwd':
        worst = max(((float((p.grad - ref_g[n]).abs().max()) / (float(ref_g[n].abs().max()) + 1e-12)), n) for n, p in dn.named_parameters())
        print('   worst grad rel diff %.3e %s' % worst)
