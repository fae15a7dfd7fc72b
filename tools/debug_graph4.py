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
N = int(os.environ.get('N', '5'))
groups = [{'params': list(dn.parameters()), 'lr': 2e-4}, {'params': list(pn.parameters()), 'lr': 2e-4}]
opt = torch.optim.SGD(groups, lr=1e-3, foreach=True) if os.environ.get('OPT') == 'sgd' else torch.optim.Adam(groups, fused=True, capturable=True)
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
def fwd_only(flip):
    model._flip_override = flip
    with torch.no_grad():
        l = float(model(batch, progress=0.0)['loss'])
    model._flip_override = None
    return l
restore(); print('fwd-only at restored state: flip F %.8f  T %.8f' % (fwd_only(False), fwd_only(True)))
restore(); le = torch.stack([eager() for _ in range(N)]).cpu()
gs = G.GraphedTrainStep(model, opt, batch, progress=0.0)
restore(); print('fwd-only after capture+restore: flip F %.8f  T %.8f' % (fwd_only(False), fwd_only(True)))
restore()
lg = torch.stack([gs(batch).detach().clone().reshape(()) for _ in range(N)]).cpu()
print('eager', le.tolist()); print('graph', lg.tolist())
