"""Neural-Ray-Surface projection: GenericCamera.project on the fused HIP kernels against the REFERENCE's own
GenericCamera.project (tests/golden/nrs.pt, written by oracle/pin_against_reference.py from /root/reference: grid and autograd
gradients w.r.t. the 3-D points and the ray surface, at two stages of the temperature annealing)."""
import pytest
import torch

import parity_cases as P


def _check(device, name):
    from packnet_sfm.geometry.camera_generic import GenericCamera
    fx = P.golden('nrs')[name]
    R = fx['rays'].clone().to(device).requires_grad_(True)
    X = fx['X'].clone().to(device).requires_grad_(True)
    cam = GenericCamera(R)
    grid = cam.project(X, fx['progress'], downsample=True, frame='c')
    assert grid.shape == fx['grid'].shape
    # grid values are normalised image coordinates in [-1, 1]; one pixel of the 48x56 half-resolution map is ~0.04.  The softmax
    # is sharp: T = 1e-4 at progress 0 and 3e-6 at progress 35, i.e. logits up to 3e5 whose fp32 ulp is 0.03 -- the weights of the
    # reference itself carry percent-level rounding noise there, so the bound scales with 1/T: 2e-4 (0.005 px) / 1e-3 (0.025 px),
    # and the mean error must stay 20x below it.
    diff = (grid.detach().cpu() - fx['grid']).abs()
    err, mean = float(diff.max()), float(diff.mean())
    tol = 2e-4 if fx['progress'] < 10 else 1e-3
    print(name, 'max |grid - reference| = %.3e, mean %.3e' % (err, mean))
    assert err <= tol and mean <= tol / 20, (err, mean)
    (grid * fx['dy'].to(device)).sum().backward()
    P.check_robust(X.grad, fx['gX'], 2e-2, name + ' d/dX')
    P.check_robust(R.grad, fx['gR'], 2e-2, name + ' d/dR')


@pytest.mark.parametrize('name', ['nrs_start', 'nrs_late'])
def test_nrs_project_emulated(emulated_kernels, name):
    _check('cpu', name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['nrs_start', 'nrs_late'])
def test_nrs_project_gpu(name):
    _check('cuda', name)


def test_generic_camera_reconstruct_and_errors(emulated_kernels):
    from packnet_sfm.geometry.camera_generic import GenericCamera
    fx = P.golden('nrs')['nrs_start']
    cam = GenericCamera(fx['rays'])
    depth = torch.full((1, 1, 96, 112), 3.0)
    P.check(cam.reconstruct(depth, frame='c'), fx['rays'] * 3.0, 1e-6, 'reconstruct')
    with pytest.raises(ValueError):
        cam.reconstruct(depth, frame='x')
    with pytest.raises(NotImplementedError):
        cam.project(torch.randn(2, 3, 96, 112), 0.0)
