"""Process-group helpers under the names of the reference's packnet_sfm/utils/horovod.py, served by the RCCL facade
(`packnet_sfm.rccl.hvd`, one process per GPU)."""
import functools

from packnet_sfm.rccl import hvd

HAS_HOROVOD = True      # the facade is part of this package, so the "horovod present?" switch is always on

rank = hvd.rank
world_size = hvd.size


def hvd_init():
    hvd.init()
    return HAS_HOROVOD


def on_rank_0(func):
    """Decorator: run `func` on rank 0 only (other ranks return None)."""
    @functools.wraps(func)
    def only_first_rank(*args, **kwargs):
        return func(*args, **kwargs) if hvd.rank() == 0 else None
    return only_first_rank


@on_rank_0
def print0(string='\n'):
    print(string)


def reduce_value(value, average, name):
    """All-reduce a tensor over the ranks (mean when `average`, else sum)."""
    return hvd.allreduce(value, average=average, name=name)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
