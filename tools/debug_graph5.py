import os, sys, random, copy
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_cases as P
from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import graph as G
from packnet_sfm.models import SfmModel as SM
from tests.test_gpu_parity import _selfsup, _step_batch  # noqa
fx = dict(P.golden('step')['step_flip0'])
batch = _step_batch(fx)
model, dn, pn = _selfsup('cuda', fx)
model.flip_lr_prob = 0.5
log = []
orig = SM.SfmModel.depth_net_flipping
def spy(self, b, flip):
    log.append(bool(flip)); return orig(self, b, flip)
SM.SfmModel.depth_net_flipping = spy
groups = [{'params': list(dn.parameters()), 'lr': 2e-4}, {'params': list(pn.parameters()), 'lr': 2e-4}]
opt = torch.optim.Adam(groups, fused=True, capturable=True)
def eager(flip=None):
    opt.zero_grad()
    model._flip_override = flip
    out = model(batch, progress=0.0)
    model._flip_override = None
    out['loss'].backward()
    opt.step()
    return out['loss'].detach().clone().reshape(())
eager(False); torch.cuda.synchronize()
opt_tensors = [v for st in opt.state.values() for v in st.values() if torch.is_tensor(v)]
state = (copy.deepcopy(model.state_dict()), [t.clone() for t in opt_tensors])
def restore():
    model.load_state_dict(state[0])
    with torch.no_grad():
        for t, s in zip(opt_tensors, state[1]): t.copy_(s)
    HF.bump_weight_epoch()
def fwd_only(flip):
    model._flip_override = flip
    with torch.no_grad():
        l = float(model(batch, progress=0.0)['loss'])
    model._flip_override = None
    return l
restore(); print('fwd-only: F %.8f T %.8f' % (fwd_only(False), fwd_only(True)))
flips = [True, False, False, True, False]
del log[:]
restore(); le = [float(eager(f)) for f in flips]; print('eager flips seen by the model', log, le)
del log[:]
gs = G.GraphedTrainStep(model, opt, batch, progress=0.0); print('capture saw', log, 'graph keys', list(gs.graphs.keys()))
for f in (False, True):
    restore(); print('first replay of graph[%s]: %.8f' % (f, float(gs(batch, flip=f))))
restore(); print('graph', [float(gs(batch, flip=f)) for f in flips])
