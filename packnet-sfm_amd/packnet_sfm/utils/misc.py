"""Small helpers (API subset of the reference's packnet_sfm/utils/misc.py)."""
from packnet_sfm.utils.types import is_list


def filter_dict(dictionary, keywords):
    """Keywords that are keys of `dictionary`, in the order given."""
    return [key for key in keywords if key in dictionary]


def make_list(var, n=None):
    var = var if is_list(var) else [var]
    if n is None:
        return var
    assert len(var) == 1 or len(var) == n, 'Wrong list length for make_list'
    return var * n if len(var) == 1 else var


def same_shape(shape1, shape2):
    return len(shape1) == len(shape2) and all(a == b for a, b in zip(shape1, shape2))


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
