"""Data-parallel gradient exchange over RCCL/xGMI (replaces the horovod/NCCL path of the reference's trainer)."""
from packnet_sfm.rccl.reducer import GradBucketReducer, init_process_group  # noqa: F401
# one package with a reference checkout further down sys.path (see packnet_sfm/_merge.py)
from packnet_sfm._merge import extend as _extend
__path__ = _extend(__path__, __name__)
