"""rocprofv3 target: a few PackNetSAN01 training-mode steps with a 5 % sparse depth map (which kernels the sparse branch spends its time in)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd')); sys.path.insert(0, ROOT)
import torch
from packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = PackNetSAN01(dropout=0.0, version='1A').to(dev).train()
rgb = torch.rand(4, 3, 192, 640, device=dev)
sparse = torch.rand(4, 1, 192, 640, device=dev) * 50 * (torch.rand(4, 1, 192, 640, device=dev) < 0.05)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    net.zero_grad(set_to_none=True)
    o = net(rgb, input_depth=sparse)
    (sum(i.mean() for i in o['inv_depths']) + sum(i.mean() for i in o['inv_depths_rgbd']) + o['depth_loss']).backward()
torch.cuda.synchronize()
print('ok')
