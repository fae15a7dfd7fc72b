"""FlatAdam: torch.optim.Adam semantics on ONE flat fp32 buffer per parameter group, updated by a single launch of the
gfx950 `adam_kernel` (csrc/elementwise.hip, 28 B/parameter of HBM traffic) instead of ~250 per-tensor updates.

The parameters of each group are re-homed as views into a flat buffer (their names/shapes -- the checkpoint contract --
do not change); gradients are expected as views into a matching flat buffer (GradBucketReducer already keeps them that
way), otherwise they are gathered first.  Matches `Adam(lr, betas, eps, weight_decay)` of the reference's
configure_optimizers (packnet_sfm/models/model_wrapper.py:128-149): two groups ('Depth', 'Pose'), StepLR-compatible
(`param_groups[i]['lr']` is read every step).
"""
import torch

from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops


class FlatAdam:
    def __init__(self, param_groups, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if isinstance(param_groups, (list, tuple)) and param_groups and not isinstance(param_groups[0], dict):
            param_groups = [{'params': list(param_groups)}]
        self.param_groups = []
        self.state = {}
        for g in param_groups:
            params = [p for p in g['params'] if p.requires_grad]
            group = {'lr': g.get('lr', lr), 'betas': g.get('betas', betas), 'eps': g.get('eps', eps),
                     'weight_decay': g.get('weight_decay', weight_decay), 'name': g.get('name', ''), 'params': params}
            n = sum(p.numel() for p in params)
            dev = params[0].device
            flat = torch.empty(n, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + k].view_as(p)          # parameter now lives inside the flat buffer
                off += k
            group['_flat'] = flat
            group['_grad'] = torch.zeros(n, dtype=torch.float32, device=dev)
            group['_m'] = torch.zeros(n, dtype=torch.float32, device=dev)
            group['_v'] = torch.zeros(n, dtype=torch.float32, device=dev)
            group['_step'] = 0
            self._bind_grads(group)
            self.param_groups.append(group)
        self.grad_scale = 1.0      # e.g. 1/world_size when gradients were sum-reduced

    @staticmethod
    def _bind_grads(group):
        off = 0
        for p in group['params']:
            k = p.numel()
            p.grad = group['_grad'][off:off + k].view_as(p)
            off += k

    def zero_grad(self, set_to_none=False):
        for g in self.param_groups:
            g['_grad'].zero_()
            self._bind_grads(g)

    def _gather_grads(self, group):
        """Make sure group['_grad'] holds the gradients (no copy when p.grad is already the flat view)."""
        off = 0
        for p in group['params']:
            k = p.numel()
            view = group['_grad'][off:off + k]
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad.reshape(-1))
            off += k

    @torch.no_grad()
    def step(self, closure=None):
        for g in self.param_groups:
            self._gather_grads(g)
            g['_step'] += 1
            b1, b2 = g['betas']
            ops.adam_step(g['_flat'], g['_grad'], g['_m'], g['_v'], g['lr'], b1, b2, g['eps'], g['weight_decay'],
                          self.grad_scale, g['_step'])
        HF.bump_weight_epoch()      # parameters changed through raw pointers: invalidate packed conv weights

    def state_dict(self):
        return {'groups': [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in g.items() if k != 'params'}
                           for g in self.param_groups]}

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd['groups']):
            for k in ('_m', '_v'):
                g[k].copy_(s[k])
            g['_step'] = s['_step']
            for k in ('lr', 'betas', 'eps', 'weight_decay'):
                g[k] = s[k]
