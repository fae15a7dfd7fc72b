"""A `horovod.torch`-shaped facade over torch.distributed/RCCL.

The reference's trainer talks to horovod only through: init, rank, size, local_rank, allreduce(tensor, average, name),
DistributedOptimizer(optimizer, named_parameters, compression), Compression.none
(packnet_sfm/trainers/horovod_trainer.py:5-48, packnet_sfm/utils/horovod.py:1-48).  This module exports exactly those
names so trainer code written against horovod runs unchanged on an MI355X node, launched one process per GPU with
`python -m torch.distributed.run --nproc-per-node N ...` instead of mpirun.
"""
import torch
import torch.distributed as dist

from packnet_sfm.rccl.reducer import GradBucketReducer, init_process_group

_state = {'rank': 0, 'size': 1, 'local_rank': 0, 'init': False}


def init():
    r, w, lr = init_process_group()
    _state.update(rank=r, size=w, local_rank=lr, init=True)
    return True


def rank():
    return _state['rank']


def size():
    return _state['size']


def local_rank():
    return _state['local_rank']


def allreduce(tensor, average=True, name=None):
    """Out-of-place all-reduce (sum or mean), like hvd.allreduce."""
    if _state['size'] == 1:
        return tensor.clone()
    out = tensor.clone()
    on_cpu = not out.is_cuda and dist.get_backend() == 'nccl'
    if on_cpu:
        out = out.cuda()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    if average:
        out = out / _state['size']
    return out.cpu() if on_cpu else out


def broadcast_parameters(params, root_rank=0):
    """Not used by the reference (replicas agree by seeding); offered for explicit consistency."""
    if _state['size'] == 1:
        return
    items = params.items() if isinstance(params, dict) else params
    for _, p in items:
        dist.broadcast(p.data if hasattr(p, 'data') else p, src=root_rank)


class Compression:
    none = None


class DistributedOptimizer:
    """optimizer.zero_grad() / backward() / optimizer.step() with gradient averaging across ranks in between.
    Gradients are reduced bucket-by-bucket on a side stream while backward is still running."""

    def __init__(self, optimizer, named_parameters=None, compression=None, bucket_bytes=128 << 20,
                 force_collectives=False):
        self._opt = optimizer
        params = [p for g in optimizer.param_groups for p in g['params']]
        self._reducer = GradBucketReducer(params, bucket_bytes=bucket_bytes, average=True,
                                          force_collectives=force_collectives)

    def zero_grad(self, set_to_none=False):
        self._reducer.zero_grad()

    def synchronize(self):
        self._reducer.synchronize()

    def step(self, closure=None):
        self._reducer.synchronize()
        return self._opt.step(closure) if closure is not None else self._opt.step()

    def __getattr__(self, name):
        return getattr(self._opt, name)
