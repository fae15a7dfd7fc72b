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


def parse_crop_borders(borders, shape):
    """Resolve a crop specification of the reference's configs against an image of `shape` = (height, width) and return pixel
    borders (left, top, right, bottom) -- same rules as the reference's utils/misc.py:77-146:

      ()                       -> the whole image
      (y, height, x, width)    integers: y / x < 0 count from the bottom / right edge, height / width <= 0 mean "up to that far
                               from the far edge", otherwise they are extents from y / x;
                               floats y / x: centre of the crop as a fraction of the image, height / width centred on it
      (y, x)                   integers: crop |y| rows (|x| columns) from the top/left if positive, from the bottom/right if
                               negative;  floats: (fraction, size) -- a square of `size` centred at that fraction of both axes
    """
    H, W = int(shape[0]), int(shape[1])
    if len(borders) == 0:
        return 0, 0, W, H

    def axis(start, extent, size):
        if isinstance(start, int):
            lo = start + size if start < 0 else start
            hi = extent + size if extent <= 0 else extent + lo
            return lo, hi
        centre, half = start * size, extent / 2
        return int(centre - half), int(centre + half)

    if len(borders) == 4:
        y, h, x, w = borders
        left, right = axis(x, w, W)
        top, bottom = axis(y, h, H)
    elif len(borders) == 2:
        y, x = borders
        if isinstance(x, int):
            left, top, right, bottom = max(0, x), max(0, y), W + min(0, x), H + min(0, y)
        else:       # (fraction, size): the reference reads the pair as (size, fraction) after its swap -- kept as is
            frac, size = x, y
            left, right = int(frac * W - size / 2), int(frac * W + size / 2)
            top, bottom = int(frac * H - size / 2), int(frac * H + size / 2)
    else:
        raise NotImplementedError('Crop tuple must have 2 or 4 values.')
    if not (0 <= left < right <= W and 0 <= top < bottom <= H):
        raise AssertionError('Crop borders {} are invalid'.format((left, top, right, bottom)))
    return left, top, right, bottom


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
