"""Where does the N>1 path spend its time on ONE GPU?  A 1-rank RCCL group; the training step (a) plain FlatAdam, (b) under
hvd.DistributedOptimizer without collectives (hooks + bucket bookkeeping only), (c) with the forced 1-rank collectives, overlapped,
(d) the same with every collective at the end of backward.  GPU time between events on the compute stream + free-running ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
os.environ['PNSFM_FORCE_DDP'] = '1'
import bench, torch
from packnet_sfm.rccl import hvd
from packnet_sfm.rccl.flat_adam import FlatAdam
hvd.init()
dev = torch.device('cuda', 0)
batch = bench.synthetic_batch(4, 192, 640, 1234, dev)


def measure(tag, wrap):
    model = bench.build_model(dev)
    opt = FlatAdam([{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4}, {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4}])
    opt = wrap(opt, model)

    def step(ev=None):
        opt.zero_grad()
        if ev: ev[0].record()
        out = model(batch, progress=0.0)
        if ev: ev[1].record()
        out['loss'].backward()
        if ev: ev[2].record()
        opt.step()
        if ev: ev[3].record()
    for _ in range(6):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize(); free = 1e3 * (time.perf_counter() - t0) / 20
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    fw = bw = op = 0.0
    for _ in range(10):
        step(e); torch.cuda.synchronize()
        fw += e[0].elapsed_time(e[1]); bw += e[1].elapsed_time(e[2]); op += e[2].elapsed_time(e[3])
    ti = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ti.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print('%-44s free-running %.2f ms/step | GPU between events: forward %.2f backward %.2f optimizer(+join) %.2f | host enqueue %.2f ms' % (tag, free, fw / 10, bw / 10, op / 10, sorted(ti)[1]), flush=True)
    del model, opt
    torch.cuda.empty_cache()


W = lambda **kw: (lambda opt, model: hvd.DistributedOptimizer(opt, named_parameters=model.named_parameters(), compression=hvd.Compression.none, **kw))
for rep in range(2):
    measure('plain FlatAdam', lambda opt, model: opt)
    measure('reducer, no collectives (hooks only)', W(force_collectives=False))
    measure('reducer + 1-rank collectives, overlapped', W(force_collectives=True))
    measure('reducer + 1-rank collectives at the end', W(force_collectives=True, overlap=False))
