"""Merging this package with a checkout of the reference (TRI-ML/packnet-sfm) that sits further down sys.path.

The reference resolves its plug-ins by module path (`load_class('PackNet01', ['packnet_sfm.networks.depth'])`,
packnet_sfm/utils/load.py:79-111), so the drop-in must live under the SAME package name -- but it only provides the
hot-path modules.  Two mechanisms make `PYTHONPATH=<this>/packnet-sfm_amd:<reference checkout>` work as one package:

  * every package `__init__` here calls `extend(__path__, __name__)` (pkgutil.extend_path): sub-modules this tree does not
    have (utils/load.py, utils/config.py, models/model_wrapper.py, datasets/, loggers/, ...) are found in the reference's
    directories, while modules both trees have resolve to THIS tree (first on the path);
  * a module here that shadows a reference module but implements only the hot-path part of it installs
    `__getattr__ = reference_fallback(__name__, __file__)` (PEP 562): any other public name of the shadowed module
    (`utils.depth.viz_inv_depth`, `utils.image.load_image`, ...) is served from the reference's file, loaded on first use.

Without a reference checkout on the path both are no-ops: the package is then just the stand-alone hot path.
"""
import importlib.util
import os
import sys
from pkgutil import extend_path as extend  # noqa: F401  (re-exported for the package __init__ files)

_loaded = {}


def _reference_file(module_name, own_file):
    """Path of the same-named module in another `packnet_sfm` tree on the (extended) package path, or None."""
    pkg_name, _, leaf = module_name.rpartition('.')
    pkg = sys.modules.get(pkg_name)
    own = os.path.realpath(own_file)
    for d in list(getattr(pkg, '__path__', [])):
        cand = os.path.join(d, leaf + '.py')
        if os.path.isfile(cand) and os.path.realpath(cand) != own:
            return cand
    return None


def reference_fallback(module_name, own_file):
    """-> a module-level __getattr__ serving names this module does not define from the reference's module of the same
    dotted name (executed under the alias `<module_name>.__reference__`; it imports its own dependencies through the
    merged package, i.e. it sees the MI355X modules wherever they shadow the reference's)."""
    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        mod = _loaded.get(module_name)
        if mod is None:
            path = _reference_file(module_name, own_file)
            if path is None:
                raise AttributeError('module %r has no attribute %r (and no reference checkout of packnet_sfm is on '
                                     'sys.path to provide it)' % (module_name, name))
            spec = importlib.util.spec_from_file_location(module_name + '.__reference__', path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _loaded[module_name] = mod
        try:
            return getattr(mod, name)
        except AttributeError:
            raise AttributeError('module %r has no attribute %r' % (module_name, name)) from None
    return __getattr__
