"""Throughput probe (results WRONG by construction: the next forward races with the update): what would the step gain if the optimizer
tail (adam_pack_table_kernel, 1.1 ms, HBM-bound) ran on a second stream underneath the NEXT step's forward pass?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from packnet_sfm.rccl.flat_adam import FlatAdam
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
opt = FlatAdam([{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4}, {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4}])
batch = bench.synthetic_batch(4, 192, 640, 1234, dev)
side = torch.cuda.Stream()


def step(overlap):
    opt.zero_grad()
    out = model(batch, progress=0.0)
    out['loss'].backward()
    if overlap:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            opt.step()
    else:
        opt.step()


for mode in (False, True, False, True):
    for _ in range(5):
        step(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step(mode)
    torch.cuda.synchronize()
    print('optimizer tail on a second stream underneath the next forward: %s -> %.3f ms/step' % (mode, 1e3 * (time.perf_counter() - t0) / 20), flush=True)
