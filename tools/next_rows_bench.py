"""Throughput of the SURVEY 8(f) 'next' rows on one MI355X, one JSON object (profiles/rNN_next_rows.json):
  N2b  device input pipeline (resize 375x1242 -> 192x640 + colour jitter + to-tensor, batch of 4 triplets) vs the PIL oracle on the host;
  N3   PackNetSAN01 forward + backward (dense path, and with a 5 %-filled sparse depth map), batch 4 at 192x640;
  N4   GenericCamera.project (Neural Ray Surfaces), 384x384 fisheye frame (192x192 internal), forward + backward.
Inputs are synthetic and resident in HBM; times are hipEvent brackets around 10 repetitions after 2 warm-ups."""
import json, os, random, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

dev = torch.device('cuda:0')


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
# ---- N2b
from packnet_sfm.datasets.device_transforms import DeviceTrainTransform
rng = np.random.default_rng(0)
B, H, W = 4, 375, 1242
frames = [torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(dev) for _ in range(3)]
K = torch.eye(3, dtype=torch.float64, device=dev).repeat(B, 1, 1)
t = DeviceTrainTransform((192, 640), (0.2, 0.2, 0.2, 0.05), ())
random.seed(1)
ms = timed(lambda: t({'rgb': frames[0], 'rgb_context': frames[1:], 'intrinsics': K}))
out['N2b_device_train_transform'] = {'ms_per_batch_of_4_triplets': round(ms, 3), 'images_per_sec': round(B / ms * 1e3, 1)}
try:
    from PIL import Image
    from oracle import augment_oracle as AO
    fr = [f.cpu().numpy() for f in frames]
    random.seed(1)
    t0 = time.time()
    for b in range(B):
        AO.train_transforms({'rgb': Image.fromarray(fr[0][b]), 'rgb_context': [Image.fromarray(fr[1][b]), Image.fromarray(fr[2][b])],
                             'intrinsics': np.eye(3)}, (192, 640), (0.2, 0.2, 0.2, 0.05), ())
    out['N2b_device_train_transform']['pil_host_images_per_sec_1_core'] = round(B / (time.time() - t0), 1)
except Exception as e:      # the oracle is test infrastructure; the comparison is optional
    out['N2b_device_train_transform']['pil_host'] = 'unavailable: %s' % e

# ---- N3
from packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
torch.manual_seed(0)
net = PackNetSAN01(dropout=0.0, version='1A').to(dev)
rgb = torch.rand(4, 3, 192, 640, device=dev)
sparse = torch.rand(4, 1, 192, 640, device=dev) * 50 * (torch.rand(4, 1, 192, 640, device=dev) < 0.05)

def san(with_depth):
    net.zero_grad(set_to_none=True)
    o = net(rgb, input_depth=sparse if with_depth else None)
    inv = o['inv_depths'] if isinstance(o, dict) else o
    inv = inv if isinstance(inv, (list, tuple)) else [inv]
    loss = sum(i.mean() for i in inv)
    if isinstance(o, dict) and 'inv_depths_rgbd' in o:       # the completion outputs and the feature-consistency term take part in the
        loss = loss + sum(i.mean() for i in o['inv_depths_rgbd']) + o['depth_loss']     # loss, so the sparse branch's backward runs too
    loss.backward()
net.train()
try:
    ms0 = timed(lambda: san(False), reps=5)
    ms1 = timed(lambda: san(True), reps=5)
    out['N3_packnetsan01_fwd_bwd'] = {'dense_ms': round(ms0, 2), 'dense_images_per_sec': round(4 / ms0 * 1e3, 1),
                                     'with_sparse_depth_ms': round(ms1, 2), 'with_sparse_depth_images_per_sec': round(4 / ms1 * 1e3, 1),
                                     'note': 'training-mode call with input_depth = RGB pass + RGB-D pass (PackNetSAN01.py:226-229), loss over both '
                                             'outputs + depth_loss; the sparse branch runs on the active-site kernels of csrc/sparse.hip'}
except Exception as e:
    out['N3_packnetsan01_fwd_bwd'] = 'failed: %r' % (e,)

# ---- N4
from packnet_sfm.geometry.camera_generic import GenericCamera
torch.manual_seed(1)
Hn = Wn = 384
rays = torch.nn.functional.normalize(torch.randn(1, 3, Hn, Wn, device=dev), dim=1).requires_grad_(True)
X = (torch.randn(1, 3, Hn, Wn, device=dev) + torch.tensor([0., 0., 3.], device=dev).view(1, 3, 1, 1)).requires_grad_(True)
cam = GenericCamera(rays)

def nrs():
    rays.grad = None; X.grad = None
    g = cam.project(X, 10.0, downsample=True, frame='c')
    g.sum().backward()
ms = timed(nrs, reps=5)
out['N4_generic_camera_project_fwd_bwd'] = {'frame': '384x384 (192x192 internal, 1681 candidates per pixel)', 'ms': round(ms, 3),
                                            'note': 'the reference materialises a [3, 36864, 1681] patch tensor (743 MB) per call'}
print(json.dumps(out))
