"""Drop-in boundary, executed through the REFERENCE's own resolver.

`PYTHONPATH=<repo>/packnet-sfm_amd:<reference checkout>` must give ONE `packnet_sfm` package: modules this repository
re-implements resolve here, everything else (utils/load.py, utils/config.py, ...) resolves in the reference, and the
reference's plug-in loader builds OUR PackNet01 and restores a checkpoint written from the REFERENCE's PackNet01.

Runs in this container only (it needs /root/reference; the GPU box has no reference checkout): CPU, no kernels launched.
Each scenario is a fresh interpreter so that sys.path order, not this pytest process's import state, decides.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'packnet-sfm_amd')
REF = '/root/reference'

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'packnet_sfm')), reason='no reference checkout here')


def _run(code, pythonpath, timeout=600):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath), PYTHONDONTWRITEBYTECODE='1')
    prelude = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)            # oracle/_refstubs.py: stubs for cv2 / torchvision / yacs / termcolor (absent in this image)
        from oracle import _refstubs
        _install = _refstubs.install
        def _stubs_only():
            path = list(sys.path)
            _install()
            sys.path[:] = path            # keep the PYTHONPATH order under test (install() would put the reference first)
        _stubs_only()
    ''' % ROOT)
    r = subprocess.run([sys.executable, '-c', prelude + textwrap.dedent(code)], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, 'child failed:\nSTDOUT:\n%s\nSTDERR:\n%s' % (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_merged_package_resolves_each_module_in_the_right_tree():
    out = _run('''
        import packnet_sfm, packnet_sfm.utils.load as L, packnet_sfm.utils.config, packnet_sfm.networks.depth.PackNet01 as PN
        import packnet_sfm.networks.layers.packnet.layers01 as LY, packnet_sfm.losses.multiview_photometric_loss as ML
        import packnet_sfm.models.SelfSupModel as SS, packnet_sfm.geometry.camera as CAM, packnet_sfm.trainers as TR
        print('load', L.__file__); print('cfg', packnet_sfm.utils.config.__file__)
        for m in (PN, LY, ML, SS, CAM): print('ours', m.__file__)
        print('trainer', TR.HorovodTrainer.__module__)
        # names of shadowed modules that the hot path does not re-implement are served from the reference's file
        from packnet_sfm.utils.depth import viz_inv_depth, post_process_inv_depth, inv2depth, compute_depth_metrics
        from packnet_sfm.utils.image import load_image, flip_lr
        from packnet_sfm.utils.types import is_cfg, is_list
        from packnet_sfm.losses.multiview_photometric_loss import SSIM, MultiViewPhotometricLoss
        from packnet_sfm.geometry.camera_utils import view_synthesis_generic, view_synthesis
        print('fallback', viz_inv_depth.__module__, SSIM.__module__, inv2depth.__module__, MultiViewPhotometricLoss.__module__)
    ''', [PKG, REF])
    lines = dict(l.split(' ', 1) for l in out.strip().splitlines() if ' ' in l and not l.startswith('ours'))
    assert lines['load'].startswith(REF) and lines['cfg'].startswith(REF)
    ours = [l.split(' ', 1)[1] for l in out.strip().splitlines() if l.startswith('ours')]
    assert len(ours) == 5 and all(p.startswith(PKG) for p in ours), ours
    assert lines['trainer'] == 'packnet_sfm.trainers.horovod_trainer'
    fb = lines['fallback'].split()
    assert fb[0].endswith('.__reference__') and fb[1].endswith('.__reference__')       # reference's implementations
    assert fb[2] == 'packnet_sfm.utils.depth' and fb[3] == 'packnet_sfm.losses.multiview_photometric_loss'   # ours


def test_reference_loader_builds_our_packnet01_and_restores_a_reference_checkpoint(tmp_path):
    ckpt = str(tmp_path / 'reference_packnet01.ckpt')
    # (1) only the reference on the path: ITS PackNet01, seeded, saved the way model_wrapper/ModelCheckpoint do
    #     ({'state_dict': {'model.depth_net.<name>': tensor}}, utils/load.py:139-151 strips everything up to 'depth_net.')
    _run('''
        import torch
        from packnet_sfm.networks.depth.PackNet01 import PackNet01
        import packnet_sfm.networks.depth.PackNet01 as M
        assert M.__file__.startswith(%r), M.__file__
        torch.manual_seed(7)
        net = PackNet01(dropout=0.0, version='1A')
        sd = net.state_dict()
        assert len(sd) == 216, len(sd)
        torch.save({'state_dict': {'model.depth_net.' + k: v for k, v in sd.items()}}, %r)
        print(float(sum(v.double().sum() for v in sd.values())))
    ''' % (REF, ckpt), [REF])
    # (2) merged path, ours first: the reference's resolver + checkpoint loader on our module
    out = _run('''
        import io, contextlib, torch
        from packnet_sfm.utils.load import load_class, load_class_args_create, load_network, filter_args
        import packnet_sfm.utils.load as L
        assert L.__file__.startswith(%r), L.__file__
        # config.model.depth_net carries more keys than the ctor takes (utils/load.py:35-56 filters them)
        args = {'name': 'PackNet01', 'checkpoint_path': '', 'version': '1A', 'dropout': 0.0}
        cls = load_class('PackNet01', paths=['packnet_sfm.networks.depth'])
        assert cls.__module__ == 'packnet_sfm.networks.depth.PackNet01'
        import packnet_sfm.networks.depth.PackNet01 as M
        assert M.__file__.startswith(%r), M.__file__
        assert set(filter_args(cls, args)) == {'version', 'dropout'}
        net = load_class_args_create('PackNet01', paths=['packnet_sfm.networks.depth'], args=args)
        assert type(net) is cls and len(net.state_dict()) == 216
        before = {k: v.clone() for k, v in net.state_dict().items()}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            net = load_network(net, %r, ['depth_net', 'disp_network'])
        assert '216/216' in buf.getvalue(), buf.getvalue()
        saved = torch.load(%r, map_location='cpu')['state_dict']
        changed = 0
        for k, v in net.state_dict().items():
            assert torch.equal(v, saved['model.depth_net.' + k]), k
            changed += int(not torch.equal(v, before[k]))
        assert changed >= 50, changed           # every conv weight differs from our own (differently seeded) init: really overwritten
        # the other plug-ins of the hot path resolve the same way (model_wrapper.py:410-471)
        pose = load_class_args_create('PoseNet', paths=['packnet_sfm.networks.pose'], args={'nb_ref_imgs': 2, 'rotation_mode': 'euler'})
        model = load_class('SelfSupModel', paths=['packnet_sfm.models'])(num_scales=4, ssim_loss_weight=0.85, automask_loss=True,
                                                                      photometric_reduce_op='min', flip_lr_prob=0.5)
        model.add_depth_net(net); model.add_pose_net(pose)
        assert type(model).__module__ == 'packnet_sfm.models.SelfSupModel' and 'depth_net' in model.network_requirements
        print('OK', float(sum(v.double().sum() for v in net.state_dict().values())))
    ''' % (REF, PKG, ckpt, ckpt), [PKG, REF])
    assert out.strip().splitlines()[-1].startswith('OK')


def test_reference_horovod_imports_resolve_to_the_rccl_facade():
    """`import horovod.torch as hvd` (reference trainers/horovod_trainer.py:5, utils/horovod.py:3-7) unchanged: with our tree
    on the path the import lands on the RCCL facade, and the REFERENCE's own utils/horovod.py -- executed from its file, not our
    shadowing module -- runs its helpers on it (single process: rank 0 of 1; reduce_value is the identity)."""
    out = _run('''
        import importlib.util, torch
        import horovod.torch as hvd
        import packnet_sfm.rccl.hvd as facade
        for name in ('init', 'rank', 'size', 'local_rank', 'allreduce', 'broadcast_parameters', 'DistributedOptimizer', 'Compression'):
            assert getattr(hvd, name) is getattr(facade, name), name
        assert hvd.Compression.none is None
        spec = importlib.util.spec_from_file_location('ref_utils_horovod', %r)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)                     # the reference's file: `try: import horovod.torch as hvd`
        assert ref.HAS_HOROVOD is True and ref.hvd is hvd
        assert ref.hvd_init() is True and ref.rank() == 0 and ref.world_size() == 1
        t = torch.arange(4.)
        assert torch.equal(ref.reduce_value(t, average=True, name='x'), t)
        # the reference's trainer module body imports cleanly against the shim (constructing it needs a GPU: set_device)
        src = open(%r).read()
        assert 'import horovod.torch as hvd' in src
        spec = importlib.util.spec_from_file_location('ref_horovod_trainer', %r)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.hvd is hvd and hasattr(mod, 'HorovodTrainer')
        print('OK')
    ''' % (os.path.join(REF, 'packnet_sfm', 'utils', 'horovod.py'), os.path.join(REF, 'packnet_sfm', 'trainers', 'horovod_trainer.py'),
           os.path.join(REF, 'packnet_sfm', 'trainers', 'horovod_trainer.py')), [PKG, REF])
    assert out.strip().splitlines()[-1] == 'OK'
