"""Type predicates (API of the reference's packnet_sfm/utils/types.py, minus the yacs dependency)."""
import numpy as np
import torch


def is_numpy(data):
    return isinstance(data, np.ndarray)


def is_tensor(data):
    return type(data) == torch.Tensor


def is_tuple(data):
    return isinstance(data, tuple)


def is_list(data):
    return isinstance(data, list)


def is_dict(data):
    return isinstance(data, dict)


def is_str(data):
    return isinstance(data, str)


def is_int(data):
    return isinstance(data, int)


def is_seq(data):
    return is_tuple(data) or is_list(data)
