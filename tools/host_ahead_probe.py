"""How far does the host run ahead of the GPU?  Times the training step (a) as bench.py does (no fence between steps), (b) with a
device synchronisation after every step, (c) with one after every forward pass too.  If (a) ~ (b) - host enqueue time, the host is
normally a full step ahead and the GPU never starves; the differences show what each fence costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
from packnet_sfm.rccl.flat_adam import FlatAdam
opt = FlatAdam([{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4}, {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4}])
batch = bench.synthetic_batch(4, 192, 640, 1234, dev)


def run(n, sync_step, sync_fwd):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad(); out = model(batch, progress=0.0)
        if sync_fwd: torch.cuda.synchronize()
        out['loss'].backward(); opt.step()
        if sync_step: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(6):
    run(1, False, False)
for rep in range(2):
    print('free-running %.2f ms/step | sync every step %.2f | sync after forward and after step %.2f' % (run(20, False, False), run(20, True, False), run(20, True, True)))
# GPU time of the forward and backward halves (events), free-running
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
fw = bw = 0.0
for _ in range(10):
    opt.zero_grad(); e[0].record(); out = model(batch, progress=0.0); e[1].record(); out['loss'].backward(); e[2].record(); opt.step(); e[3].record()
    torch.cuda.synchronize()
    fw += e[0].elapsed_time(e[1]); bw += e[1].elapsed_time(e[2])
print('GPU time between events (main stream): forward %.2f ms, backward %.2f ms, optimizer %.2f ms' % (fw / 10, bw / 10, e[2].elapsed_time(e[3])))
