"""Which aten ops (by input shape) the training step still launches around the HIP kernels: torch.profiler over 3 eager steps of
bench.py's workload.  usage: python tools/aten_shapes.py [H W B]"""
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, 'packnet-sfm_amd'))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 640, 4)
dev = torch.device('cuda', 0)
model = bench.build_model(dev, 'PackNet01')
from packnet_sfm.rccl.flat_adam import FlatAdam
groups = [{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0},
          {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0}]
opt = FlatAdam(groups)
batch = bench.synthetic_batch(B, H, W, 0, dev)


def step():
    opt.zero_grad()
    out = model(batch, progress=0.0)
    out['loss'].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0 and e.key.startswith('aten::'):
        rows.append((t / N, e.count / N, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('aten device time per step: %.1f us in %d (op, shape) groups' % (tot, len(rows)))
for t, c, k, s in rows[:70]:
    print('%8.1f us %5.1f x  %-28s %s' % (t, c, k, s))

# ---- by call site: the innermost frame inside this repository's package
import collections
sites = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = getattr(e, 'self_cuda_time_total', 0)
    if t <= 0 or not e.name.startswith('aten::'):
        continue
    where = '?'
    for fr in (e.stack or []):
        if 'packnet' in fr and 'site-packages' not in fr and 'dist-packages' not in fr:
            where = fr.split('packnet-sfm_amd/')[-1]
            break
    a = sites[(where, e.name)]
    a[0] += t / N
    a[1] += 1.0 / N
print()
print('by call site (innermost frame of this package):')
for (where, name), (t, c) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:60]:
    print('%8.1f us %5.1f x  %-34s %s' % (t, c, name, where[:110]))
