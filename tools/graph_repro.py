"""Repeats the body of tests/test_gpu_parity.py::test_graphed_step_matches_eager and prints the per-step deviation of the replayed
loss sequence from the eager one (a flaky-capture hunter).  usage: python tools/graph_repro.py [iterations] [sgd|adam]"""
import copy, os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'tests'))
sys.path.insert(0, os.path.join(_ROOT, 'packnet-sfm_amd'))
sys.path.insert(0, _ROOT)
import torch
import parity_cases as P
import test_gpu_parity as T
from packnet_sfm.hip import functional as HF
from packnet_sfm.hip.graph import GraphedTrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
optimizer = sys.argv[2] if len(sys.argv) > 2 else 'sgd'
flips = [True, False, False, True, False, False, True, True]
worst = 0.0
for it in range(n):
    fx = dict(P.golden('step')['step_flip0'])
    batch = T._step_batch(fx)
    model, dn, pn = T._selfsup(T.DEV, fx)
    model.flip_lr_prob = 0.5
    groups = [{'params': list(dn.parameters()), 'lr': 2e-4}, {'params': list(pn.parameters()), 'lr': 2e-4}]
    opt = (torch.optim.Adam(groups, fused=True, capturable=True) if optimizer == 'adam'
           else torch.optim.SGD(groups, lr=1e-3, foreach=True))

    def eager(flip=False):
        opt.zero_grad()
        model._flip_override = flip
        out = model(batch, progress=0.0)
        model._flip_override = None
        out['loss'].backward()
        opt.step()
        return out['loss'].detach().clone().reshape(())

    eager()
    torch.cuda.synchronize()
    opt_tensors = [v for st in opt.state.values() for v in st.values() if torch.is_tensor(v)]
    state = (copy.deepcopy(model.state_dict()), [t.clone() for t in opt_tensors])

    def restore():
        model.load_state_dict(state[0])
        with torch.no_grad():
            for t, saved in zip(opt_tensors, state[1]):
                t.copy_(saved)
        HF.bump_weight_epoch()

    restore()
    le = torch.stack([eager(f) for f in flips]).cpu()
    restore()
    le2 = torch.stack([eager(f) for f in flips]).cpu()
    graphed = GraphedTrainStep(model, opt, batch, progress=0.0)
    restore()
    lg = torch.stack([graphed(batch, flip=f).detach().clone().reshape(()) for f in flips]).cpu()
    torch.cuda.synchronize()
    dev = ((lg - le).abs() / le).tolist()
    dev2 = ((le2 - le).abs() / le).tolist()
    worst = max(worst, max(dev))
    print(it, 'graph-vs-eager', ' '.join('%.1e' % d for d in dev), '| eager-vs-eager', '%.1e' % max(dev2), flush=True)
    del graphed, model, dn, pn, opt
print('worst', '%.2e' % worst)
