"""CPU: the drop-in host-side helpers (batch plumbing, depth utilities, camera / pose algebra, evaluation metrics) give
the outputs the REFERENCE's own helpers gave on the same inputs (fixture 'host' of tests/golden/slim.pt, written by
oracle/pin_against_reference.py from /root/reference).  No kernels involved: this pins the plumbing around them."""
import types

import torch

import parity_cases as P


def _fx():
    return P.golden('slim')['host']


def test_flip_and_upsample_plumbing():
    from packnet_sfm.models import model_utils as MU
    fx = _fx()['model_utils']
    b = fx['batch']
    mine = MU.flip_batch_input({k: (list(v) if isinstance(v, list) else v.clone()) for k, v in b.items()})
    assert torch.equal(mine['rgb'], fx['flipped']['rgb'])
    for a, r in zip(mine['rgb_context'], fx['flipped']['rgb_context']):
        assert torch.equal(a, r)
    assert torch.equal(mine['intrinsics'], fx['flipped']['intrinsics'])
    out = MU.flip_output({'inv_depths': [t.clone() for t in fx['inv_depths']]})
    up = MU.upsample_output({'inv_depths': [t.clone() for t in fx['inv_depths']]}, mode='nearest', align_corners=None)
    for a, r in zip(out['inv_depths'], fx['flipped_output']):
        assert torch.equal(a, r)
    for a, r in zip(up['inv_depths'], fx['upsampled']):
        assert torch.equal(a, r)


def test_depth_and_image_helpers():
    from packnet_sfm.geometry.camera_utils import scale_intrinsics
    from packnet_sfm.losses.loss_base import ProgressiveScaling
    from packnet_sfm.utils import depth as D, image as I
    host = _fx()
    fx, inv = host['depth'], host['model_utils']['inv_depths']
    assert torch.equal(D.depth2inv(fx['depth'].clone()), fx['depth2inv'])
    assert torch.equal(D.inv2depth(inv[0]), fx['inv2depth'])
    for a, r in zip(D.inv_depths_normalize([t.clone() for t in inv]), fx['normalized']):
        P.check(a, r, 1e-6, 'inv_depths_normalize')
    rgb = host['model_utils']['batch']['rgb']
    for a, r in zip(I.match_scales(rgb, inv, 4), host['image']['match_bilinear']):
        P.check(a, r, 1e-6, 'match_scales bilinear')
    for a, r in zip(I.match_scales(inv[0], inv, 4, mode='nearest', align_corners=None), host['image']['match_nearest']):
        assert torch.equal(a, r)
    K = host['model_utils']['batch']['intrinsics']
    P.check(scale_intrinsics(K.clone(), 0.5, 0.25), host['image']['scaled_K'], 1e-7, 'scale_intrinsics')
    for ps, expected in host['progressive'].items():
        sched = ProgressiveScaling(ps, 4)
        assert [sched(p) for p in (0.0, 0.1, 0.26, 0.5, 0.76, 1.0)] == expected


def test_camera_and_pose_algebra():
    from packnet_sfm.geometry.camera import Camera
    from packnet_sfm.geometry.pose import Pose
    from packnet_sfm.geometry.pose_utils import invert_pose
    fx = _fx()['camera']
    T = Pose.from_vec(fx['vec'], 'euler')                  # float64: the torch path of from_vec
    P.check(T.mat, fx['T'], 1e-12, 'Pose.from_vec')
    P.check(invert_pose(T.mat), fx['Tinv'], 1e-12, 'invert_pose')
    P.check((T @ T).mat, fx['TT'], 1e-12, 'Pose @ Pose')
    cam = Camera(fx['K'].clone(), Tcw=T)
    P.check(cam.Kinv, fx['Kinv'], 1e-12, 'Kinv')
    P.check(cam.reconstruct(fx['depth'], 'c'), fx['Xc'], 1e-12, 'reconstruct c')
    Xw = cam.reconstruct(fx['depth'], 'w')
    P.check(Xw, fx['Xw'], 1e-12, 'reconstruct w')
    P.check(cam.project(fx['Xw'], 'w'), fx['uv_w'], 1e-10, 'project w')
    P.check(cam.project(fx['Xw'], 'c'), fx['uv_c'], 1e-10, 'project c')
    P.check(cam.scaled(0.5).K, fx['K_half'], 1e-12, 'scaled')
    assert cam.scaled(1.) is cam and len(cam) == fx['K'].shape[0]


def test_depth_metrics():
    from packnet_sfm.utils.depth import compute_depth_metrics
    fx = _fx()['metrics']
    for (crop, use_gt_scale), expected in fx['values'].items():
        cfg = types.SimpleNamespace(crop=crop, min_depth=0.0, max_depth=80.0, scale_output='resize')
        got = compute_depth_metrics(cfg, fx['gt'], fx['pred'], use_gt_scale)
        P.check(got, expected, 1e-5, 'depth metrics crop=%r gt_scale=%r' % (crop, use_gt_scale))
