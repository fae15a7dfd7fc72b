"""Type predicates with the names the reference's packnet_sfm/utils/types.py exports (minus its yacs `is_cfg`).

Table-driven: one closure per python / numpy container kind; tensors are matched on the exact class (a Parameter or
another Tensor subclass is not "a tensor" for the callers that branch on this, e.g. batch stacking)."""
import numpy as np
import torch

_CONTAINER_KINDS = {
    'numpy': np.ndarray,
    'tuple': tuple,
    'list': list,
    'dict': dict,
    'str': str,
    'int': int,
}


def _predicate(kind):
    cls = _CONTAINER_KINDS[kind]

    def check(data):
        return isinstance(data, cls)
    check.__name__ = 'is_' + kind
    check.__doc__ = 'True when `data` is a %s.' % cls.__name__
    return check


is_numpy, is_tuple, is_list, is_dict, is_str, is_int = (_predicate(k) for k in ('numpy', 'tuple', 'list', 'dict', 'str', 'int'))


def is_tensor(data):
    return type(data) is torch.Tensor


def is_seq(data):
    """list or tuple"""
    return isinstance(data, (list, tuple))


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
