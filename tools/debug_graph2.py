"""Graph replay vs eager from one saved state; knobs: FLIP (flip prob), POOL (share pool 0/1), SIDE (wgrad stream)."""
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
HF.set_wgrad_stream(os.environ.get('SIDE', '1') == '1')
flipp = float(os.environ.get('FLIP', '0.5'))
model, dn, pn = _selfsup('cuda', fx)
model.flip_lr_prob = flipp
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
restore(); le = [float(eager()) for _ in range(4)]
if os.environ.get('POOL', '1') == '0':
    orig = G.GraphedTrainStep._capture
    def cap(self, flip):
        self._pool = None
        orig(self, flip)
        self._pool = None
    G.GraphedTrainStep._capture = cap
gs = G.GraphedTrainStep(model, opt, batch, progress=0.0)
restore(); lg = [float(gs(batch)) for _ in range(4)]
random.seed(11); print('flips', [random.random() < flipp for _ in range(4)])
print('FLIP', flipp, 'POOL', os.environ.get('POOL', '1'), 'SIDE', os.environ.get('SIDE', '1'))
print(' eager', le); print(' graph', lg)
