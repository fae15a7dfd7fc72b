"""rank / world-size helpers with the API of the reference's packnet_sfm/utils/horovod.py, on RCCL."""
from packnet_sfm.rccl import hvd

HAS_HOROVOD = True   # the RCCL facade is always available


def hvd_init():
    hvd.init()
    return True


def on_rank_0(func):
    def wrapper(*args, **kwargs):
        if rank() == 0:
            func(*args, **kwargs)
    return wrapper


def rank():
    return hvd.rank()


def world_size():
    return hvd.size()


@on_rank_0
def print0(string='\n'):
    print(string)


def reduce_value(value, average, name):
    """Mean (or sum) of a tensor over all ranks."""
    return hvd.allreduce(value, average=average, name=name)
