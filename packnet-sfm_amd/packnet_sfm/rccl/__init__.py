"""Data-parallel gradient exchange over RCCL/xGMI (replaces the horovod/NCCL path of the reference's trainer)."""
from packnet_sfm.rccl.reducer import GradBucketReducer, init_process_group  # noqa: F401
