"""Gradient reproducibility: side stream on/off, repeated (debug aid, GPU only)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_cases as P
from packnet_sfm.hip import functional as HF
from tests.test_gpu_parity import _selfsup, _step_batch  # noqa
fx = dict(P.golden('step')['step_flip0'])
batch = _step_batch(fx)
model, dn, pn = _selfsup('cuda', fx)
def grads(side, n=1):
    HF.set_wgrad_stream(side)
    model.zero_grad(set_to_none=True)
    for _ in range(n):
        model(batch, progress=0.0)['loss'].backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in dn.named_parameters()}
ref = grads(False); ref = grads(False)
def cmp(a, b, tag):
    worst = sorted(((float((a[k] - b[k]).abs().max() / (b[k].abs().max() + 1e-20)), k) for k in a), reverse=True)[:4]
    print(tag, ' | '.join('%.2e %s' % w for w in worst))
cmp(grads(False), ref, 'off vs off      ')
cmp(grads(True), ref, 'on  vs off      ')
cmp(grads(True), ref, 'on  vs off again')
r2 = grads(False, 2)
cmp(grads(False, 2), r2, 'off x2 vs off x2')
cmp(grads(True, 2), r2, 'on x2 vs off x2 ')
cmp({k: v * 2 for k, v in ref.items()}, r2, '2*single vs off x2')
