"""Do back-to-back graph replays without host sync behave?  (debug aid)"""
import os, sys, random, copy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_cases as P
from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import graph as G
from tests.test_gpu_parity import _selfsup, _step_batch  # noqa
fx = dict(P.golden('step')['step_flip0'])
batch = _step_batch(fx)
model, dn, pn = _selfsup('cuda', fx)
model.flip_lr_prob = 0.5
opt = torch.optim.Adam([{'params': list(dn.parameters()), 'lr': 2e-4}, {'params': list(pn.parameters()), 'lr': 2e-4}], fused=True, capturable=True)
def eager():
    opt.zero_grad()
    out = model(batch, progress=0.0)
    out['loss'].backward()
    opt.step()
    return out['loss'].detach().clone().reshape(())
random.seed(7); eager(); torch.cuda.synchronize()
opt_tensors = [v for st in opt.state.values() for v in st.values() if torch.is_tensor(v)]
state = (copy.deepcopy(model.state_dict()), [t.clone() for t in opt_tensors])
def restore():
    model.load_state_dict(state[0])
    with torch.no_grad():
        for t, s in zip(opt_tensors, state[1]): t.copy_(s)
    HF.bump_weight_epoch(); random.seed(11)
restore(); le = torch.stack([eager() for _ in range(4)]).cpu()
gs = G.GraphedTrainStep(model, opt, batch, progress=0.0)
for mode in ('nosync', 'sync', 'nosync-noclone'):
    restore()
    out = []
    for _ in range(4):
        l = gs(batch)
        if mode == 'sync':
            torch.cuda.synchronize()
        out.append(l.detach().clone().reshape(()) if mode != 'nosync-noclone' else float(l))
    print(mode, [float(v) for v in out])
print('eager', le.tolist())
