"""Trainers (API of the reference's packnet_sfm/trainers/__init__.py: `from packnet_sfm.trainers import HorovodTrainer`)."""
# one package with a reference checkout further down sys.path (see packnet_sfm/_merge.py)
from packnet_sfm._merge import extend as _extend
__path__ = _extend(__path__, __name__)

from packnet_sfm.trainers.horovod_trainer import HorovodTrainer  # noqa: E402

__all__ = ["HorovodTrainer"]
