"""FlatAdam: torch.optim.Adam semantics on flat fp32 arenas -- parameters, gradients, exp_avg, exp_avg_sq of a parameter
group are each ONE buffer, updated by ONE launch of the gfx950 `adam_flat_kernel` (csrc/elementwise.hip: float4 accesses,
28 B/parameter of HBM traffic, step counter and hyper-parameters read from device memory: nothing the launch needs is a host
value that changes from step to step) instead of ~250 per-tensor updates.

SURVEY.md 8(f) N1: the optimizer CONSUMES THE ALL-REDUCE BUCKETS IN PLACE.  The gradient arena is laid out in reverse
registration order (the order backward produces gradients) and `grad_buckets()` cuts it into the contiguous <=32 MiB
slices that `GradBucketReducer` all-reduces, so reduce and update touch the same memory: no second flat gradient buffer,
no gather pass.  The conv weight-gradient kernels even write straight into their slice of the arena
(`hip.functional.register_grad_slots`), so for them not even the bucket gather copy exists; gradients produced elsewhere
(GroupNorm affine, Conv3d, PoseNet head) are gathered with one multi-tensor copy per group.

Matches `Adam(lr, betas, eps, weight_decay)` of the reference's configure_optimizers
(packnet_sfm/models/model_wrapper.py:128-166): two groups ('Depth', 'Pose') with their own lr / weight decay, StepLR-
compatible (`param_groups[i]['lr']` is read before every step; `default_config.py:70-74`), and `state_dict()` /
`load_state_dict()` speak torch.optim.Adam's layout ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups'}), which
is what the reference's checkpoints store (`model.optimizer.state_dict()`), so optimizer state moves between the two.

Round 5 -- the FUSED TAIL (on by default; PNSFM_ADAM_FUSED=0 / FlatAdam(..., fused=False) switch it off): once the conv layers own
packed weight copies, step() is two launches for the whole model -- `adam_pack_table_kernel` (csrc/conv2d_bx3.h) updates every
split-bf16 conv weight AND writes its forward / backward-data images from the values it holds (40 B per parameter instead of the 28 of
the flat update + the 20 of the re-pack), `adam_segments_kernel` updates the rest of the arenas; both use the inline update of
csrc/adam_math.h, like the flat kernel: bit-identical parameters and moments (tests/test_kernels_emulated.py).
(Round 5 also had the update bucket by bucket UNDERNEATH the backward pass -- post-accumulate hooks, an update stream: +0.2 % for 1.5 ms of
host work per step, profiles/r05_ab_adam_overlap.txt -- removed in round 6.)
"""
import os

import torch

from packnet_sfm.hip import functional as HF
from packnet_sfm.hip import ops

_ALIGN = 4          # floats: every parameter starts on a 16-byte boundary of the arena (float4 kernel, 16-byte DMA)


def _round_up(n, a):
    return (n + a - 1) // a * a


class FlatAdam:
    def __init__(self, param_groups, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_slots=True, fused=None):
        if isinstance(param_groups, (list, tuple)) and param_groups and not isinstance(param_groups[0], dict):
            param_groups = [{'params': list(param_groups)}]
        self.param_groups = []
        self.state = {}                 # torch-style handle; the tensors live in the arenas
        self.grad_scale = 1.0           # e.g. 1/world_size when gradients were sum-reduced
        slots = []
        for g in param_groups:
            params = [p for p in g['params'] if p.requires_grad]
            if not params:
                raise ValueError('FlatAdam: a parameter group without trainable parameters')
            group = {'lr': g.get('lr', lr), 'betas': tuple(g.get('betas', betas)), 'eps': g.get('eps', eps),
                     'weight_decay': g.get('weight_decay', weight_decay), 'name': g.get('name', ''), 'params': params,
                     'amsgrad': False, 'maximize': False}
            dev = params[0].device
            order = list(reversed(params))          # arena order = the order backward produces gradients
            offs, n = {}, 0
            for p in order:
                offs[id(p)] = n
                n += _round_up(p.numel(), _ALIGN)
            flat = torch.zeros(n, dtype=torch.float32, device=dev)
            grad = torch.zeros(n, dtype=torch.float32, device=dev)
            for p in order:
                o, k = offs[id(p)], p.numel()
                flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + k].view_as(p)   # the parameter now lives inside the arena (names / shapes unchanged)
                p.grad = None
            group.update(_order=order, _offs=offs, _flat=flat, _grad=grad,
                         _m=torch.zeros(n, dtype=torch.float32, device=dev), _v=torch.zeros(n, dtype=torch.float32, device=dev),
                         _hp=torch.zeros(12, dtype=torch.float32, device=dev), _hp_host=None)
            self._sync_group(group)
            self.param_groups.append(group)
            slots += [(p, self.grad_view(group, p)) for p in order]
        # ---- the fused optimizer tail (round 5): conv weights updated AND re-packed by one launch, everything else by a second one
        self._fused = (os.environ.get('PNSFM_ADAM_FUSED', '1') != '0') if fused is None else bool(fused)
        self._plan = None               # [signature, item table, n, blocks, (cache, parameter) pairs it covers, segment table, n, blocks]
        self._slots = HF.register_grad_slots(slots) if grad_slots else None
        if self._slots is not None:
            # the slots are views of THIS optimizer's gradient arena: they go when the optimizer goes (and, independently,
            # when a parameter is collected or re-registered by a newer optimizer -- hip/functional.py)
            import weakref
            weakref.finalize(self, self._slots.remove)
        HF.bump_weight_epoch()

    # ---- arena access -------------------------------------------------------------------------------------------------
    @staticmethod
    def grad_view(group, p):
        o = group['_offs'][id(p)]
        return group['_grad'][o:o + p.numel()].view_as(p)

    def grad_buckets(self, bucket_bytes=32 << 20):
        """[(flat slice of the gradient arena, [parameters], [their views])] -- contiguous, cut at parameter boundaries, in
        the order backward fills them (last group first).  What GradBucketReducer all-reduces in place."""
        out = []
        for g in reversed(self.param_groups):
            cur, start, end = [], None, None
            for p in g['_order']:
                o = g['_offs'][id(p)]
                e = o + _round_up(p.numel(), _ALIGN)
                if cur and (e - start) * 4 > bucket_bytes:
                    out.append((g['_grad'][start:end], cur, [self.grad_view(g, q) for q in cur]))
                    cur, start = [], None
                if start is None:
                    start = o
                cur.append(p)
                end = e
            if cur:
                out.append((g['_grad'][start:end], cur, [self.grad_view(g, q) for q in cur]))
        return out

    # ---- hyper-parameters live on the device (the update kernel reads them: replayable) --------------------------------
    def _sync_group(self, g):
        b1, b2 = float(g['betas'][0]), float(g['betas'][1])
        host = (float(g['lr']), b1, b2, float(g['eps']), float(g['weight_decay']), float(self.grad_scale), 1.0 - b1, 1.0 - b2)
        if g['_hp_host'] != host:
            g['_hp_host'] = host
            g['_hp'][1:9].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)

    def sync_hyperparams(self):
        """Push lr / betas / eps / weight decay / grad_scale to the device if they changed (`step()` does it itself)."""
        for g in self.param_groups:
            self._sync_group(g)

    # ---- fused tail: Adam + re-pack of the conv weights in one launch, the rest of the arenas in a second ---------------------------
    def _fused_plan(self):
        """Device tables of the two-launch tail, rebuilt when the set of (parameter, packed buffers) or the arithmetic mode changes.
        None while no conv weight has packed copies yet (first steps) or when fusing is off."""
        if not self._fused or not HF._pack_batch_on():
            return None
        mine = {}
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                mine[id(p)] = (g, p)
        pairs = []
        for dev, prs in HF._pack_pairs(only=set(mine)).items():
            pairs += [(c, w) for c, w in prs if w.dim() == 4 and w.device == self.param_groups[0]['_flat'].device]
        if not pairs:
            return None
        sig = (HF.get_conv_math(),) + tuple((w.data_ptr(), c.wp_fwd.data_ptr(), c.wp_bwd.data_ptr()) for c, w in pairs)
        if self._plan is not None and self._plan[0] == sig:
            return self._plan
        dev = self.param_groups[0]['_flat'].device

        def slices(g, p):
            o, k = g['_offs'][id(p)], p.numel()
            return [g[a][o:o + k] for a in ('_flat', '_grad', '_m', '_v')]

        items = []
        for c, w in pairs:
            g, p = mine[id(w)]
            fl, gr, m, v = slices(g, p)
            items.append((fl.view_as(p), gr.view_as(p), m.view_as(p), v.view_as(p), g['_hp'], c.wp_fwd, c.wp_bwd))
        table, n, blocks, covered = ops.adam_fused_plan(items, [], dev)
        taken = {id(pairs[i][1]) for i in covered}
        # segments: maximal runs of the arenas between the covered conv weights (arena order; parameters are 16-byte aligned)
        segs = []
        for g in self.param_groups:
            run = None
            for p in g['_order']:
                o = g['_offs'][id(p)]
                e = o + _round_up(p.numel(), _ALIGN)
                if id(p) in taken:
                    if run is not None:
                        segs.append((g, run[0], run[1]))
                        run = None
                else:
                    run = [o, e] if run is None else [run[0], e]
            if run is not None:
                segs.append((g, run[0], run[1]))
        seg_t, nseg, sblocks = ops.adam_segment_table([(g['_flat'][a:b], g['_grad'][a:b], g['_m'][a:b], g['_v'][a:b], g['_hp']) for g, a, b in segs], dev)
        self._plan = [sig, table, n, blocks, [pairs[i] for i in covered], seg_t, nseg, sblocks]
        return self._plan

    # ---- torch.optim.Optimizer surface --------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        """Gradients are dropped, not zero-filled: the next backward writes every element of the arena slices it uses (the
        conv kernels directly, the rest through the gather), and a parameter without a gradient is zeroed in the gather."""
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    def _gather_grads(self, g):
        """Make g['_grad'] hold the gradients: no copy for gradients that already live in their arena slice.  Returns the
        parameters that have NO gradient this step."""
        src, dst, missing = [], [], []
        # address of every parameter's arena slice, computed once per arena (246 slice + view_as + data_ptr per step took the host
        # ~1 ms: profiles/r06_host_profile.txt); a view is only built for the gradients that are NOT already in place
        ptrs = g.get('_gptr')
        base = g['_grad'].data_ptr()
        if ptrs is None or ptrs[0] != base:
            ptrs = g['_gptr'] = (base, {id(p): base + 4 * g['_offs'][id(p)] for p in g['_order']})
        at = ptrs[1]
        for p in g['_order']:
            gr = p.grad
            if gr is None:
                self.grad_view(g, p).zero_()
                missing.append(p)
            elif gr.data_ptr() != at[id(p)]:
                view = self.grad_view(g, p)
                src.append(gr.detach().reshape(view.shape))
                dst.append(view)
        if src:
            torch._foreach_copy_(dst, src)
        return missing

    @torch.no_grad()
    def step(self, closure=None):
        """One Adam update per group = one `adam_flat_kernel` launch over the whole arena.

        A parameter whose .grad is None is SKIPPED like torch.optim.Adam does (parameter, exp_avg and exp_avg_sq keep their
        values: they are saved around the flat launch and restored -- three small multi-tensor copies, only on steps that
        have such parameters; e.g. PackNetSAN01's sparse-depth branch on batches without input_depth).  One deviation
        remains and is deliberate: the step counter is per GROUP (one device-resident scalar per flat launch), so a parameter that skipped k steps uses the group's step in its bias corrections where torch
        would use its own, k smaller; `state_dict()` reports the group's step for every parameter.  Under
        hvd.DistributedOptimizer unused parameters arrive with ZERO gradients (the reducer fills their bucket slice, as
        horovod's synchronize() does for the reference) and are therefore updated, exactly like the reference's DDP path."""
        loss = closure() if closure is not None else None
        plan = self._fused_plan()
        if plan is not None:
            # the two-launch tail: every gradient present (a parameter without one is skipped like torch does: plain path this step)
            missing = [self._gather_grads(g) for g in self.param_groups]
            if not any(missing):
                for g in self.param_groups:
                    self._sync_group(g)
                    ops.adam_flat_update(g['_flat'][:0], g['_grad'][:0], g['_m'][:0], g['_v'][:0], g['_hp'], tick=True)    # step counter only
                ops.adam_pack_table_run(plan[1], plan[2], plan[3])
                ops.adam_segments_run(plan[5], plan[6], plan[7])
                HF.bump_weight_epoch()
                HF.stamp_packed(plan[4])
                HF.repack_all(exclude={id(w) for _, w in plan[4]})
                return loss
        for g in self.param_groups:
            # the whole arena in one launch (a parameter without a gradient keeps parameter and moments, like torch.optim.Adam)
            missing = self._gather_grads(g)
            self._sync_group(g)
            keep = None
            if missing:
                views = [g[k][g['_offs'][id(p)]:g['_offs'][id(p)] + p.numel()] for p in missing for k in ('_flat', '_m', '_v')]
                keep = (views, [v.clone() for v in views])
            ops.adam_flat_step(g['_flat'], g['_grad'], g['_m'], g['_v'], g['_hp'])
            if keep is not None:
                torch._foreach_copy_(keep[0], keep[1])
        HF.bump_weight_epoch()      # parameters changed through raw pointers: invalidate packed conv weights ...
        HF.repack_all()             # ... and re-pack them in one launch (the lazy path covers what is left)
        return loss

    # ---- checkpoints in torch.optim.Adam's layout ----------------------------------------------------------------------
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            step = g['_hp'][0:1].detach().clone().reshape(()).cpu()
            ids = []
            for p in g['params']:
                o, k = g['_offs'][id(p)], p.numel()
                state[idx] = {'step': step.clone(), 'exp_avg': g['_m'][o:o + k].view_as(p).clone(),
                              'exp_avg_sq': g['_v'][o:o + k].view_as(p).clone()}
                ids.append(idx)
                idx += 1
            groups.append({'lr': g['lr'], 'betas': g['betas'], 'eps': g['eps'], 'weight_decay': g['weight_decay'],
                           'amsgrad': False, 'maximize': False, 'name': g['name'], 'params': ids})
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        if 'param_groups' not in sd or len(sd['param_groups']) != len(self.param_groups):
            raise ValueError('FlatAdam.load_state_dict: expected torch.optim.Adam layout with %d groups' % len(self.param_groups))
        for g, sg in zip(self.param_groups, sd['param_groups']):
            if len(sg['params']) != len(g['params']):
                raise ValueError('FlatAdam.load_state_dict: group size mismatch')
            for k in ('lr', 'eps', 'weight_decay'):
                if k in sg:
                    g[k] = sg[k]
            if 'betas' in sg:
                g['betas'] = tuple(sg['betas'])
            step = 0.0
            for p, i in zip(g['params'], sg['params']):
                st = sd['state'].get(i)
                if st is None:
                    continue
                o, k = g['_offs'][id(p)], p.numel()
                g['_m'][o:o + k].copy_(st['exp_avg'].reshape(-1))
                g['_v'][o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
                step = max(step, float(st['step']))
            g['_hp'][0:1].fill_(step)
            g['_hp_host'] = None
            self._sync_group(g)
