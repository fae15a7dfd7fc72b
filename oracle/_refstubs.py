"""TEST INFRASTRUCTURE (this container only): make the reference's hot-path modules importable from
/root/reference without copying them.  Stubs the third-party modules the image lacks (cv2, torchvision,
yacs, termcolor), patches matplotlib.cm.get_cmap (removed in matplotlib>=3.9, used at utils/depth.py:6) and
works around `ref_image.get_device()` == -1 on CPU (losses/multiview_photometric_loss.py:150,156-157).
Recipe from SURVEY.md Appendix C.  Nothing here runs on the GPU box (no /root/reference there)."""
import os
import sys
import types

REFERENCE = '/root/reference'


def install():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError('reference checkout not present at %s' % REFERENCE)
    import torch
    import matplotlib
    import matplotlib.cm as cm

    def _mod(n):
        m = types.ModuleType(n)
        sys.modules[n] = m
        return m

    if 'cv2' not in sys.modules:
        _mod('cv2')
    if 'torchvision' not in sys.modules:
        tv = _mod('torchvision')
        tv.transforms = _mod('torchvision.transforms')
    if 'yacs' not in sys.modules:
        y = _mod('yacs')
        y.config = _mod('yacs.config')
        y.config.CfgNode = type('CfgNode', (dict,), {})
    if 'MinkowskiEngine' not in sys.modules:
        # PackNetSAN01 imports MinkowskiEngine (third-party, un-versioned, not installed).  Parameter-free placeholders let the
        # reference's DENSE path (input_depth=None) be constructed and run; the sparse branch itself cannot be executed here.
        me = _mod('MinkowskiEngine')

        class _Placeholder(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

            def forward(self, *a, **k):
                raise RuntimeError('MinkowskiEngine is not available: the sparse branch of the reference cannot run here')
        for name in ('MinkowskiConvolution', 'MinkowskiBatchNorm', 'MinkowskiReLU', 'MinkowskiMaxPooling', 'MinkowskiSigmoid'):
            setattr(me, name, _Placeholder)
        me.SparseTensor = _Placeholder
        me.utils = _mod('MinkowskiEngine.utils')
    if 'termcolor' not in sys.modules:
        _mod('termcolor').colored = lambda s, *a, **k: s
    if not hasattr(cm, 'get_cmap'):
        cm.get_cmap = lambda n: matplotlib.colormaps[n]
    if not getattr(torch.Tensor.get_device, '_pnsfm_patched', False):
        _gd = torch.Tensor.get_device

        def get_device(self):
            d = _gd(self)
            return self.device if d < 0 else d
        get_device._pnsfm_patched = True
        torch.Tensor.get_device = get_device
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
