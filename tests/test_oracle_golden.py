"""CPU: the oracle (oracle/packnet_oracle.py) reproduces the REFERENCE's outputs stored in tests/golden/*.pt.
(The fixtures were produced by running the reference's own modules: oracle/pin_against_reference.py.)"""
import torch

import parity_cases as P
from oracle import packnet_oracle as O


def _sd(fx, prefix='l.'):
    return {prefix + k: v for k, v in fx['sd'].items()}


def test_oracle_conv_blocks():
    L = P.golden('layers')
    for name in ('conv2d_k3', 'conv2d_k5', 'conv2d_k7'):
        fx = L[name]
        P.check(O.conv2d_gn_elu(fx['x'], _sd(fx), 'l', fx['k']), fx['y'], 2e-5, name)
    fx = L['residual_conv']
    P.check(O.residual_conv(fx['x'], _sd(fx), 'l'), fx['y'], 2e-5, 'residual_conv')
    fx = L['invdepth']
    P.check(O.inv_depth_head(fx['x'], _sd(fx), 'l'), fx['y'], 2e-5, 'invdepth')


def test_oracle_packing_blocks():
    L = P.golden('layers')
    P.check(O.packing(L['packing']['x']), L['packing']['y'], 0.0, 'packing')
    assert torch.equal(O.packing(L['packing']['x']), torch.nn.functional.pixel_unshuffle(L['packing']['x'], 2))
    for name in ('pack_k3', 'pack_k5'):
        fx = L[name]
        P.check(O.pack_layer_conv3d(fx['x'], _sd(fx), 'l', fx['k']), fx['y'], 2e-5, name)
    fx = L['unpack']
    P.check(O.unpack_layer_conv3d(fx['x'], _sd(fx), 'l', 3), fx['y'], 2e-5, 'unpack')


def test_oracle_d4_blocks_and_packnetslim01():
    """d = 4 packing / unpacking blocks and PackNetSlim01 (reference: PackNetSlim01.py, num_3d_feat = 4)."""
    S = P.golden('slim')
    for name in ('pack_d4_k3', 'pack_d4_k5'):
        fx = S[name]
        P.check(O.pack_layer_conv3d(fx['x'], _sd(fx), 'l', fx['k']), fx['y'], 2e-5, name)
    fx = S['unpack_d4']
    P.check(O.unpack_layer_conv3d(fx['x'], _sd(fx), 'l', 3), fx['y'], 2e-5, 'unpack_d4')
    fx = S['packnetslim01']
    sd = O.init_params(O.packnet01_param_shapes('1A', ni=32, n1=32, d=4), seed=fx['seed'], randomize_affine=True)
    disps = O.packnet01_forward(sd, fx['rgb'], '1A', True)
    for a, b in zip(disps, fx['disps']):
        P.check(a, b, 5e-5, 'packnetslim01 disp')
    fx = S['packnet01_1B']
    sd = O.init_params(O.packnet01_param_shapes('1B'), seed=fx['seed'], randomize_affine=True)
    for a, b in zip(O.packnet01_forward(sd, fx['rgb'], '1B', True), fx['disps']):
        P.check(a, b, 5e-5, 'packnet01 1B disp')


def test_oracle_supervised_loss():
    for method, fx in P.golden('slim')['supervised'].items():
        pred = [t.clone().requires_grad_(True) for t in fx['pred']]
        loss = O.supervised_loss(pred, fx['gt'], method, 2)
        P.check(loss, fx['loss'][0], 1e-6, 'supervised ' + method)
        loss.backward()
        for p, g in zip(pred, fx['dpred']):
            P.check(p.grad, g, 1e-5, 'supervised grad ' + method)


def test_oracle_loss_and_grads():
    for name, fx in (list(P.golden('loss').items()) + list(P.golden('slim')['loss_clip'].items()) +
                     list(P.golden('slim')['loss_padding'].items()) + list(P.golden('loss_l1').items())):
        inv = [t.clone().requires_grad_(True) for t in fx['inv_depths']]
        pv = fx['pose_vec'].clone().requires_grad_(True)
        mats = [O.pose_vec2mat44(pv[:, i]) for i in range(2)]
        loss, photo, smooth = O.multiview_photometric_loss(fx['image'], fx['context'], inv, fx['K'], fx['K'], mats, **fx['kwargs'])
        P.check(loss, fx['loss'], 1e-5, name)
        P.check(smooth, fx['smoothness_loss'], 1e-5, name + '.smooth')
        loss.sum().backward()
        for i in range(4):
            P.check(inv[i].grad, fx['d_inv_depths'][i], 2e-4, '%s.dinv%d' % (name, i))
        P.check(pv.grad, fx['d_pose_vec'], 2e-4, name + '.dpose')


def test_oracle_posenet_and_pose_algebra():
    fx = P.golden('network')['posenet']
    psd = O.init_params(O.posenet_param_shapes(2), seed=fx['seed'], randomize_affine=True)
    P.check(O.posenet_forward(psd, fx['image'], fx['context']), fx['pose_vec'], 1e-5, 'posenet')
    # rotation matrices are orthonormal, translation lands in the last column
    v = torch.tensor([[0.1, -0.2, 0.3, 0.05, -0.02, 0.01]])
    T = O.pose_vec2mat44(v)
    assert torch.allclose(T[:, :3, :3] @ T[:, :3, :3].transpose(1, 2), torch.eye(3).unsqueeze(0), atol=1e-6)
    assert torch.allclose(T[:, :3, 3], v[:, :3])


def test_oracle_packnet01_forward():
    fx = P.golden('network')['packnet01']
    sd = O.init_params(O.packnet01_param_shapes('1A'), seed=fx['seed'], randomize_affine=True)
    assert sum(v.numel() for v in sd.values()) == 128294020
    with torch.no_grad():
        disps = O.packnet01_forward(sd, fx['rgb'], '1A', True)
    for a, b in zip(disps, fx['disps']):
        P.check(a, b, 5e-5, 'packnet01 disp')
    assert [tuple(d.shape[-2:]) for d in disps] == [(32, 64), (16, 32), (8, 16), (4, 8)]


def test_oracle_properties():
    """Size-independent properties of the loss path: identity warp at T = I, SSIM(x, x) = 1, flip equivariance."""
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 3, 24, 40, generator=g)
    K = torch.tensor([[0.58 * 40, 0., 20.], [0., 1.92 * 24, 12.], [0., 0., 1.]]).repeat(2, 1, 1)
    inv = 0.1 + torch.rand(2, 1, 24, 40, generator=g)
    T = torch.eye(4).repeat(2, 1, 1)
    assert torch.allclose(O.view_synthesis(img, inv, K, K, T), img, atol=2e-5)
    assert torch.allclose(O.ssim(img, img), torch.ones_like(img), atol=1e-5)
    assert float(O.photometric_map(img, img).abs().max()) < 1e-5
    sx, sy = O.smoothness_terms(torch.ones(2, 1, 24, 40), img)
    assert float(sx) == 0.0 and float(sy) == 0.0
