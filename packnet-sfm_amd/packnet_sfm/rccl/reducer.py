"""Bucketed gradient all-reduce, overlapped with backward on a side HIP stream.

Replaces what `hvd.DistributedOptimizer(optimizer, named_parameters, compression=none)` does for the reference
(packnet_sfm/trainers/horovod_trainer.py:46-48,92-93): every rank computes gradients on its shard of the batch and
the ranks average them before the optimizer step.  One process per GPU; `torch.distributed` backend "nccl" is RCCL on
ROCm (xGMI on an MI355X node), "gloo" is used by the CPU tests.

MI355X-first choices (not a translation of horovod's tensor-fusion queue):
  * parameters are laid into a few LARGE flat fp32 buckets once, in reverse registration order (~ the order backward
    produces them).  Autograd hands each parameter its freshly computed gradient tensor (no accumulate pass, no
    zero-fill); when the last gradient of a bucket exists, ONE multi-tensor copy (`torch._foreach_copy_`) gathers the
    bucket and every `param.grad` is re-pointed at its bucket view, so the optimizer reads the reduced values in place;
  * xGMI is point-to-point and a ring all-reduce is bound by one link (~153 GB/s), so a step wants big collectives --
    but the LAST bucket of a step completes when backward ends, so its all-reduce is fully exposed: with 128 MiB buckets
    that was 112 MB (pack4.conv3d ... pre_calc: ~1 ms on 8 GPUs).  The default is therefore 32 MiB (large enough to run at
    link speed: ~0.2 ms of transfer against ~30 us of latency), cut at parameter boundaries: 519.5 MB of PackNet01+PoseNet
    gradients -> 10 buckets, the 302 MB pack5 and 75 MB pack4 weights are buckets of their own (they complete in the
    middle of backward and hide behind the rest of it), <= 32 MB stays exposed at the end.  A bucket larger than
    `chunk_bytes` (64 MiB: a single parameter that big) is reduced as several back-to-back collectives over contiguous
    slices of its flat buffer, so the overlap with backward never hinges on one giant ring pass;
  * a bucket's all-reduce is enqueued on a dedicated side stream the moment its last gradient has been accumulated
    (post-accumulate-grad hooks), while the compute stream keeps running backward; `synchronize()` joins the side
    stream before the optimizer reads the gradients and applies the 1/world_size averaging.
"""
import os
import time as _time

import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun or the bench driver).
    Returns (rank, world_size, local_rank).  Single-process runs (no env) do not create a group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if (world > 1 or os.environ.get('PNSFM_FORCE_DDP') == '1') and not dist.is_initialized():
        if backend is None:
            # RCCL needs one device per rank ("duplicate GPU" otherwise).  More ranks than devices (a 1-GPU box rehearsing
            # the N>1 path) fall back to gloo with host-staged buckets: functional, not a performance path.
            ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            backend = 'nccl' if ndev >= int(os.environ.get('LOCAL_WORLD_SIZE', world)) and ndev > 0 else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            # keep dmabuf IPC (the only mode the host driver supports) for RCCL's intra-node transport
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _Bucket:
    def __init__(self, params, device, dtype, flat=None, views=None):
        self.params = params
        self.pending = len(params)
        self.launched = False
        self.work = None
        self.event = None
        if flat is not None:            # a slice of somebody else's arena (FlatAdam's gradient buffer): reduced in place
            self.flat, self.views = flat, list(views)
            return
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, device=device, dtype=dtype)
        off = 0
        self.views = []
        for p in params:
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(v)
            off += p.numel()


class GradBucketReducer:
    """
    Parameters
    ----------
    params : iterable of nn.Parameter (requires_grad ones are bucketed)
    bucket_bytes : int        flat-bucket capacity
    chunk_bytes : int         largest single collective: a bigger bucket (one huge parameter) is reduced in slices of this size
    process_group             torch.distributed group (default: WORLD)
    average : bool            divide by world size (horovod's `average=True` semantics)
    """

    def __init__(self, params, bucket_bytes=96 << 20, process_group=None, average=True, force_collectives=False, buckets=None,
                 overlap=None, chunk_bytes=512 << 20):
        self.group = process_group
        self.chunk_bytes = int(chunk_bytes)
        # overlap=False (or PNSFM_DDP_OVERLAP=0): no collective starts before synchronize() -- the A/B leg that shows what the
        # side-stream overlap with backward is worth (tests/rccl_two_ranks.py, bench.py)
        self.overlap = (os.environ.get('PNSFM_DDP_OVERLAP', '1') != '0') if overlap is None else bool(overlap)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # run the collectives even in a 1-rank group (exercises the RCCL / side-stream path on a single GPU)
        self.force = bool(force_collectives) and dist.is_initialized()
        self.average = average
        # RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG, so the CPU tests scale afterwards
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._fused_avg = bool(average) and backend == 'nccl'
        self._host_staged = False       # set below: gloo + device tensors -> stage each bucket through host memory
        # (a ONE-rank group -- the single-GPU rehearsal of this path, PNSFM_FORCE_DDP=1 -- sums: the mean over one rank is the sum, and RCCL
        # runs a real 32-workgroup kernel over the whole bucket for a one-rank AVG: 0.75 ms per step for 520 MB, rocprofv3,
        # profiles/r06_ddp_probe.txt -- the "11 % rehearsal cost" of round 5 was that kernel, not the reducer)
        self._reduce_op = dist.ReduceOp.AVG if (self._fused_avg and self.world > 1) else dist.ReduceOp.SUM
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('GradBucketReducer: no trainable parameters')
        self.device = params[0].device
        self.buckets = []
        self._bucket_of = {}
        if buckets is not None:
            # `buckets`: [(flat slice, [parameters], [views])] handed over by the optimizer (FlatAdam.grad_buckets): the
            # collective runs in place on the optimizer's gradient arena -- no second flat buffer, no gather for gradients
            # the kernels already wrote there
            covered = set()
            for flat, bparams, views in buckets:
                b = _Bucket(list(bparams), self.device, flat.dtype, flat=flat, views=views)
                for p in bparams:
                    self._bucket_of[p] = b
                    covered.add(id(p))
                self.buckets.append(b)
            if covered != {id(p) for p in params}:
                raise ValueError('GradBucketReducer: the optimizer buckets do not cover the trainable parameters')
        else:
            cur, cur_bytes = [], 0
            for p in reversed(params):                      # backward produces gradients roughly back-to-front
                nbytes = p.numel() * p.element_size()
                if cur and cur_bytes + nbytes > bucket_bytes:
                    self._close(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._close(cur)
        self._host_staged = backend == 'gloo' and self.device.type == 'cuda'
        self.side_stream = torch.cuda.Stream(device=self.device) if (self.device.type == 'cuda' and not self._host_staged) else None
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                p.grad = None
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))
        self._launched = 0
        self.collectives_issued = 0      # all-reduce calls since construction (a bucket > chunk_bytes counts once per slice)
        self._next = 0          # index of the next bucket to launch
        self._synced = False
        self._exposed = None    # [(event before the join, event after it)] while bench.py measures the exposed all-reduce time
        self._host_us = None    # host microseconds per collective call (enqueue + stream-ordered wait) while bench.py measures

    # ---- measurement: how long the compute stream waits for the communication stream at the end-of-backward join ---------
    def exposed_reset(self, on):
        self._exposed = [] if (on and self.side_stream is not None) else None
        self._host_us = [] if (on and self.side_stream is not None) else None

    def host_us_per_collective(self):
        """Mean host time of one collective call (ProcessGroup enqueue + the stream-ordered wait) over the recorded steps, or None."""
        if not self._host_us:
            return None
        return float(sum(self._host_us) / len(self._host_us))

    def exposed_ms(self):
        """Sum over the recorded steps of the time between reaching the join and passing it on the compute stream = the part
        of the gradient all-reduce that backward did NOT hide (None when not recording / no side stream)."""
        if self._exposed is None:
            return None
        torch.cuda.synchronize(self.device)
        return float(sum(a.elapsed_time(b) for a, b in self._exposed))

    def _close(self, params):
        b = _Bucket(params, params[0].device, params[0].dtype)
        for p in params:
            self._bucket_of[p] = b
        self.buckets.append(b)

    def _make_hook(self, bucket):
        def hook(param):
            bucket.pending -= 1
            if bucket.pending == 0 and self.overlap:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Launch complete buckets strictly in bucket-index order (a complete bucket waits for its predecessors): every
        rank issues the same sequence of collectives even when ranks see different sets of unused parameters."""
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            b = self.buckets[self._next]
            self._gather(b)
            self._launch(b)
            self._next += 1

    def _gather(self, bucket):
        """Copy the bucket's gradients into its flat buffer with one multi-tensor launch and alias p.grad to the views."""
        # Weight gradients may still be in flight on the side streams of hip/functional.py (weight-gradient stream, pose branch).
        # Round 5: with a communication stream, IT waits for them (and for the compute stream) and runs the gather copy itself --
        # the compute stream is never joined in the middle of backward.  (Joining it here, once per bucket, serialised backward-data
        # behind every outstanding weight gradient ten times per step: the 1-rank rehearsal ran 7.7 % under the plain step.)
        from packnet_sfm.hip import functional as HF
        comm = self.side_stream
        if comm is not None:
            comm.wait_stream(torch.cuda.current_stream(self.device))
            for st in HF.side_streams(self.device):
                if st != comm:
                    comm.wait_stream(st)
        else:
            HF.join_wgrad_stream(bucket.flat.device)
        with (torch.cuda.stream(comm) if comm is not None else HF._nullctx()):
            src, dst = [], []
            for p, v in zip(bucket.params, bucket.views):
                g = p.grad
                if g is None:
                    v.zero_()                               # unused parameter this step
                elif g.data_ptr() != v.data_ptr():
                    src.append(g.detach())
                    dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)
                if comm is not None:
                    for g in src:                           # read on the communication stream after p.grad lets go of them
                        g.record_stream(comm)
        for p, v in zip(bucket.params, bucket.views):
            p.grad = v

    def _launch(self, bucket):
        self._launched += 1
        bucket.launched = True
        if self.world == 1 and not self.force:
            return
        op = self._reduce_op
        # the same slices on every rank (a function of the bucket size only): the collective sequences pair up
        per = max(1, self.chunk_bytes // bucket.flat.element_size())
        n = bucket.flat.numel()
        chunks = [bucket.flat] if n <= per else [bucket.flat[o:min(o + per, n)] for o in range(0, n, per)]
        self.collectives_issued += len(chunks)
        if self._host_staged:
            for c in chunks:
                host = c.cpu()                          # synchronises: rehearsal path only
                dist.all_reduce(host, op=op, group=self.group)
                c.copy_(host)
            bucket.work = None
        elif self.side_stream is not None:
            # (the gather that precedes this call has already made the communication stream wait for the compute stream and the
            # side streams: nothing was enqueued in between)
            t0 = _time.perf_counter() if self._host_us is not None else 0.0
            with torch.cuda.stream(self.side_stream):
                works = [dist.all_reduce(c, op=op, group=self.group, async_op=True) for c in chunks]
                # Round 6: the COMMUNICATION stream waits for the collective right away (Work.wait() is stream-ordered: the host does not
                # block), so that synchronize() needs ONE join of the compute stream with it.  Before, the compute stream itself waited for
                # every Work at the end of backward: 15 cross-stream event waits in a row on the critical path, 1.5-1.9 ms per step on
                # one GPU although a one-rank collective launches no kernel (profiles/r05_ddp_probe.txt).
                for w in works:
                    w.wait()
            if self._host_us is not None:
                self._host_us.append(1e6 * (_time.perf_counter() - t0) / len(chunks))
            bucket.work = None
        else:
            bucket.work = [dist.all_reduce(c, op=op, group=self.group, async_op=True) for c in chunks]

    def zero_grad(self):
        """Drop the gradients (autograd will hand out fresh tensors; nothing is zero-filled) and re-arm the hooks."""
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            b.launched = False
            for p in b.params:
                p.grad = None
        self._launched = 0
        self._next = 0
        self._synced = False

    def synchronize(self):
        """Block the compute stream until every bucket is reduced; buckets whose hooks never all fired (unused
        parameters) are reduced now, in bucket-index order on every rank (ranks may see different sets of unused
        parameters; the collectives must still pair up).  Applies the averaging.  Idempotent until zero_grad()."""
        if self._synced:
            return
        self._synced = True
        for b in self.buckets[self._next:]:            # some parameters got no gradient: gather what exists, reduce now
            self._gather(b)
            self._launch(b)
        self._next = len(self.buckets)
        cur = torch.cuda.current_stream(self.device) if self.side_stream is not None else None
        if self._exposed is not None and cur is not None:
            before = torch.cuda.Event(enable_timing=True)
            before.record(cur)                 # backward's last kernel is behind this point of the compute stream
        for b in self.buckets:
            if b.work is not None:
                for w in b.work:
                    w.wait()                   # compute stream waits for the collective (stream-ordered, the host does not block)
                b.work = None
        if self.side_stream is not None:           # (always: the gather copies run there even when no collective does)
            cur.wait_stream(self.side_stream)
            if self._exposed is not None and (self.world > 1 or self.force):
                after = torch.cuda.Event(enable_timing=True)
                after.record(cur)
                self._exposed.append((before, after))
        if self.average and self.world > 1 and not self._fused_avg:
            scale = 1.0 / self.world
            for b in self.buckets:
                b.flat.mul_(scale)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def total_bytes(self):
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)
