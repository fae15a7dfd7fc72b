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
    backend = dist.get_backend()
    if backend == 'nccl' and not out.is_cuda:          # RCCL reduces device memory only
        out = out.cuda()
    elif backend != 'nccl' and out.is_cuda:            # gloo (CPU tests / shared-device rehearsal): stage through the host
        out = out.cpu()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    if average:
        out = out / _state['size']
    return out.to(tensor.device)


def broadcast_parameters(params, root_rank=0):
    """Not used by the reference (replicas agree by seeding); offered for explicit consistency."""
    if _state['size'] == 1:
        return
    items = params.items() if isinstance(params, dict) else params
    for _, p in items:
        dist.broadcast(p.data if hasattr(p, 'data') else p, src=root_rank)
    # `.data` writes do not bump tensor._version: invalidate the packed conv-weight copies explicitly.  (Any other raw
    # `.data` update of a conv weight -- an EMA, p.data.copy_() -- must be followed by HF.bump_weight_epoch() as well.)
    from packnet_sfm.hip import functional as HF
    HF.bump_weight_epoch()


class Compression:
    none = None


class DistributedOptimizer:
    """optimizer.zero_grad() / backward() / optimizer.step() with gradient averaging across ranks in between.
    Gradients are reduced bucket-by-bucket on a side stream while backward is still running."""

    def __init__(self, optimizer, named_parameters=None, compression=None, bucket_bytes=96 << 20,
                 force_collectives=False, overlap=None, chunk_bytes=512 << 20):
        self._opt = optimizer
        params = [p for g in optimizer.param_groups for p in g['params']]
        # an optimizer that keeps its gradients in a flat arena (rccl/flat_adam.py) lends its buckets: reduce + update in place
        buckets = optimizer.grad_buckets(bucket_bytes) if hasattr(optimizer, 'grad_buckets') else None
        self._reducer = GradBucketReducer(params, bucket_bytes=bucket_bytes, average=True,
                                          force_collectives=force_collectives, buckets=buckets, overlap=overlap,
                                          chunk_bytes=chunk_bytes)

    def zero_grad(self, set_to_none=False):
        self._reducer.zero_grad()

    def synchronize(self):
        """Wait for the averaged gradients (idempotent within a step, so the horovod idiom
        `optimizer.synchronize(); clip_grad_norm_(...); optimizer.step()` reduces once)."""
        self._reducer.synchronize()

    class _SkipSync:
        def __init__(self, outer):
            self.outer = outer

        def __enter__(self):
            self.outer._skip_sync = True

        def __exit__(self, *exc):
            self.outer._skip_sync = False

    def skip_synchronize(self):
        """Context manager: `with optimizer.skip_synchronize(): optimizer.step()` after a manual synchronize() (horovod API)."""
        return DistributedOptimizer._SkipSync(self)

    def step(self, closure=None):
        if not self.__dict__.get('_skip_sync', False):
            self._reducer.synchronize()
        return self._opt.step(closure) if closure is not None else self._opt.step()

    def __getattr__(self, name):
        return getattr(self._opt, name)
