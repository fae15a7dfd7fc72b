import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'packnet-sfm_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def emulated_kernels():
    """Run packnet_sfm.hip ops on the host-emulated build of the kernel sources (tests/emu); restores the
    product loader state afterwards.  Test infrastructure only."""
    from packnet_sfm.hip import _lib
    import emu_loader
    saved = (_lib._LIB, _lib.REQUIRE_CUDA)
    lib = emu_loader.use_emulated_kernels()
    lib.pnsfm_set_conv_math(1)          # every test starts from the library defaults (tests that switch them need not restore)
    lib.pnsfm_set_conv_variant(0)
    lib.pnsfm_set_conv_variant(3)
    lib.pnsfm_set_wgrad_variant(-1)
    yield
    _lib._LIB, _lib.REQUIRE_CUDA = saved
