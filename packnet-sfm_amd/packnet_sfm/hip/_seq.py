"""Loader of the block sequencer (csrc/seq/pnsfm_seq.cpp -> _pnsfm_seq.so next to this file): the bodies of the hot autograd nodes as
single C++ calls over the C ABI.  `get()` returns the module bound to the kernel library packnet_sfm.hip._lib currently holds (the
gfx950 build; tests/emu swap in the host-emulated build and get a re-bound module), or None when PNSFM_SEQ=0 asks for the pure-Python
bodies of hip/functional.py (same launches, same order: the A/B switch and the reference implementation of what the extension does)."""
import ctypes
import importlib.util
import os

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
SEQ_PATH = os.path.join(_HERE, "_pnsfm_seq.so")
_SYMBOLS = ("pnsfm_last_error", "pnsfm_conv2d_forward", "pnsfm_conv2d_forward_cat", "pnsfm_conv2d_backward_data",
            "pnsfm_conv2d_backward_data_add", "pnsfm_conv2d_backward_weight", "pnsfm_conv2d_backward_weight_cat",
            "pnsfm_groupnorm_act_forward", "pnsfm_groupnorm_act_backward", "pnsfm_stream_wait_stream", "pnsfm_region_ops")
_ON = os.environ.get("PNSFM_SEQ", "1") != "0"
_MOD = None
_BOUND = None      # (library handle, REQUIRE_CUDA) the module is bound to


def set_enabled(on):
    global _ON
    _ON = bool(on)


def enabled():
    return _ON


def _load():
    global _MOD
    if _MOD is None:
        if not os.path.exists(SEQ_PATH):
            raise ImportError("%s not found -- build it with `python __graft_entry__.py` (packnet-sfm_amd/csrc/build.py: build_seq); "
                              "PNSFM_SEQ=0 runs the pure-Python bodies instead" % SEQ_PATH)
        spec = importlib.util.spec_from_file_location("_pnsfm_seq", SEQ_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _MOD = mod
    return _MOD


def get():
    """The sequencer bound to the current kernel library, or None (switched off)."""
    global _BOUND
    if not _ON:
        return None
    lib = _lib.get()
    key = (lib, _lib.REQUIRE_CUDA)
    if _BOUND is None or _BOUND[0] is not lib or _BOUND[1] != _lib.REQUIRE_CUDA:
        mod = _load()
        mod.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _SYMBOLS}, bool(_lib.REQUIRE_CUDA))
        _BOUND = key
    return _MOD
