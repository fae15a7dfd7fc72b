"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pnsfm.h declares.
No compute calls here (no GPU in this tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def lib_path():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd', 'csrc'))
    import build as build_mod
    return build_mod.build_hip(force=False, verbose=False)


def test_header_symbols_exported(lib_path):
    from packnet_sfm.hip import _lib
    header = open(os.path.join(ROOT, 'include', 'pnsfm.h')).read()
    declared = set(re.findall(r'\b(pnsfm_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), 'symbol %s missing from %s' % (name, lib_path)
    _lib.bind(lib)
    assert lib.pnsfm_build_target() == b'gfx950'
    assert lib.pnsfm_version() >= 1


def test_packed_weight_sizes(lib_path):
    """Host-side geometry helpers (no kernel launch): padded packed-weight sizes."""
    from packnet_sfm.hip import _lib
    lib = _lib.bind(ctypes.CDLL(lib_path))
    # shapes the split-bf16 kernels take (>= 16 K-channels; 1x1 layers too since round 3) are packed as three bf16 pieces: 6 bytes per element
    assert lib.pnsfm_conv2d_packed_elems_fwd(64, 64, 3) == 9 * 64 * 64 * 3 // 2
    assert lib.pnsfm_conv2d_packed_elems_fwd(3, 64, 5) == 25 * 16 * 64         # K rows padded to whole 16-channel chunks
    assert lib.pnsfm_conv2d_packed_elems_bwd(3, 64, 5) == 25 * 64 * 32 * 3 // 2   # backward-data: K = 64 output channels
    assert lib.pnsfm_conv2d_packed_elems_fwd(129, 64, 3) == 9 * 144 * 64 * 3 // 2
    assert lib.pnsfm_conv2d_packed_elems_bwd(129, 64, 3) == 9 * 64 * 160 * 3 // 2   # M = 129 -> 5 tiles of 32
    assert lib.pnsfm_conv2d_packed_elems_fwd(256, 1, 3) == 9 * 256 * 32 * 3 // 2
    assert lib.pnsfm_conv2d_packed_elems_fwd(256, 64, 1) == 256 * 64 * 3 // 2   # 1x1: split-bf16 since round 3
    assert lib.pnsfm_conv2d_packed_elems_fwd(8, 64, 1) == 16 * 64              # < 16 K-channels: f32 kernels


def test_product_loader_refuses_cpu_tensors(lib_path):
    """No CPU fallback: the product wrappers reject host tensors."""
    import torch
    from packnet_sfm.hip import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.space_to_depth(torch.zeros(1, 1, 2, 2))


def test_product_never_imports_oracle_or_emulator():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'packnet-sfm_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')) and not f.endswith('.emu.o'):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+(oracle|emu_loader|build_emu)\b', src, re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
