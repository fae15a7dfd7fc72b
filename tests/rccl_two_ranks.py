"""Two data-parallel ranks of the gradient-averaging path on real blocks (launched by tests/test_gpu_round3.py under
torch.distributed.run; not a pytest file).

Every rank: the same seeded block stack (replicas agree by seeding, like the reference: model_wrapper.py:44), its HALF of a
4-image batch, FlatAdam whose gradient arena doubles as the all-reduce buckets, hvd.DistributedOptimizer on top
(trainers/horovod_trainer.py:46-48,92-93).  Checked on every rank against a single-process run of the FULL batch in the same
process: averaged gradients (after synchronize) and the parameters after two optimizer steps.  With one device per rank
(RCCL over xGMI) the step is also timed with the side-stream overlap on and off.  Ranks that have to share a device fall
back to gloo (host-staged buckets): the same code path, functional only.  Rank 0 prints one JSON line."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'packnet-sfm_amd'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from packnet_sfm.rccl import hvd
    from packnet_sfm.rccl.flat_adam import FlatAdam
    from test_gpu_round3 import _BlockStack
    hvd.init()
    rank, world = hvd.rank(), hvd.size()
    assert world == 2, world
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', hvd.local_rank() % ndev)
    torch.cuda.set_device(dev)
    backend = dist.get_backend()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 16, 32, 64, generator=g).to(dev)
    tgt = torch.randn(4, 16, 32, 64, generator=g).to(dev)

    def build():
        torch.manual_seed(11)
        return _BlockStack().to(dev).train()

    def loss_of(net, sl):
        return ((net(x[sl]) - tgt[sl]) ** 2).mean()

    # ---- reference: the gradient of the FULL batch, computed in this process
    ref = build()
    loss_of(ref, slice(0, 4)).backward()
    ref_grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}

    # ---- two ranks, half the batch each; `rep` mirrors the update with torch.optim.Adam on the averaged gradient VALUES (two
    # independently produced gradients differ in the last bits, and Adam's division by sqrt(v) turns that into 1e-4-level
    # parameter differences on the elements whose gradient is ~0: feeding both optimizers the same numbers removes that)
    net = build()
    rep = build()
    ropt = torch.optim.Adam(rep.parameters(), lr=2e-3)
    opt = hvd.DistributedOptimizer(FlatAdam([{'params': list(net.parameters()), 'lr': 2e-3}]), named_parameters=net.named_parameters(),
                                   compression=hvd.Compression.none, bucket_bytes=256 << 10)
    mine = slice(2 * rank, 2 * rank + 2)
    worst = pworst = 0.0
    gscale = max(float(v.abs().max()) for v in ref_grads.values())
    for step in range(2):
        opt.zero_grad()
        loss_of(net, mine).backward()
        opt.synchronize()
        torch.cuda.synchronize()
        for (n, p), q in zip(net.named_parameters(), rep.parameters()):
            if step == 0:       # same parameters as `ref`: the average over the ranks must be the full-batch gradient
                e = float((p.grad - ref_grads[n]).abs().max()) / max(float(ref_grads[n].abs().max()), 0.05 * gscale)
                worst = max(worst, e)
                assert e <= 2e-4, 'rank %d: averaged gradient of %s off by %.2e' % (rank, n, e)
            q.grad = p.grad.detach().clone()
        with opt.skip_synchronize():
            opt.step()
        ropt.step()
        for (n, p), q in zip(net.named_parameters(), rep.parameters()):
            e = float((p.detach() - q.detach()).abs().max()) / max(float(q.detach().abs().max()), 1e-2)
            pworst = max(pworst, e)
            assert e <= 1e-5, 'rank %d: parameter %s after step %d off by %.2e' % (rank, n, step, e)
    torch.cuda.synchronize()
    # replicas stay identical: every rank holds the same parameters bit for bit
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    other = flat.clone() if backend == 'nccl' else flat.cpu()
    dist.broadcast(other, src=0)
    assert torch.equal(other.to(flat.device), flat), 'rank %d: replicas diverged' % rank

    timing = None
    if backend == 'nccl':
        def timed(overlap, steps=20):
            n2 = build()
            o2 = hvd.DistributedOptimizer(FlatAdam([{'params': list(n2.parameters()), 'lr': 2e-3}]), named_parameters=n2.named_parameters(),
                                          compression=hvd.Compression.none, bucket_bytes=256 << 10, overlap=overlap)
            for i in range(3 + steps):
                if i == 3:
                    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
                o2.zero_grad()
                loss_of(n2, mine).backward()
                o2.step()
            dist.barrier(); torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / steps
        timing = {'overlap_on_ms_per_step': round(timed(True), 3), 'overlap_off_ms_per_step': round(timed(False), 3)}
    dist.barrier()
    if rank == 0:
        print(json.dumps({'ok': True, 'backend': backend, 'devices': ndev, 'worst_grad_err': worst, 'worst_param_err': pworst,
                          'timing': timing}), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
