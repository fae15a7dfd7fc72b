"""TEST INFRASTRUCTURE: route packnet_sfm.hip.ops to the host-emulated build of the kernels (tests/emu).

Only tests call this.  It swaps the ctypes handle inside packnet_sfm.hip._lib for libpnsfm_emu.so and lifts the
"device tensors only" check, so the *same* Python wrappers + the *same* kernel sources run on CPU tensors.
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
PKG = os.path.join(ROOT, "packnet-sfm_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def use_emulated_kernels():
    from build_emu import build_emu
    from packnet_sfm.hip import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu()))
    assert lib.pnsfm_build_target() == b"emu"
    _lib._LIB = lib
    _lib.REQUIRE_CUDA = False
    return lib
