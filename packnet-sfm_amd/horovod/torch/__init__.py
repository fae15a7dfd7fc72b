"""`import horovod.torch as hvd` -> the RCCL facade.  Exactly the names the reference's trainer and helpers use
(packnet_sfm/trainers/horovod_trainer.py:5-48,92-93; packnet_sfm/utils/horovod.py:1-48): init, rank, size, local_rank,
allreduce(tensor, average, name), broadcast_parameters, DistributedOptimizer(optimizer, named_parameters, compression),
Compression.none."""
from packnet_sfm.rccl.hvd import (Compression, DistributedOptimizer, allreduce, broadcast_parameters, init, local_rank,  # noqa: F401
                                   rank, size)

__all__ = ['Compression', 'DistributedOptimizer', 'allreduce', 'broadcast_parameters', 'init', 'local_rank', 'rank', 'size']
