"""Datasets package: only the device-side input pipeline lives here; the reference's dataset classes, host augmentations and
transforms resolve in a reference checkout further down sys.path (see packnet_sfm/_merge.py)."""
from packnet_sfm._merge import extend as _extend
__path__ = _extend(__path__, __name__)
