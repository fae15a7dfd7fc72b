import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'packnet-sfm_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: 437 tests, most of them on the host-emulated kernels) takes 10.5 min in one process and 3 min
    on six: when pytest-xdist is there and the caller did not choose (`-n ...`), spread it over the host's cores.  GPU runs (`-m gpu`,
    or no marker expression) stay in one process -- one device, and some of those tests time kernels.  PNSFM_TEST_WORKERS=<n>
    overrides (0: one process)."""
    try:
        if hasattr(config, 'workerinput') or 'not gpu' not in (config.option.markexpr or ''):
            return
        if getattr(config.option, 'numprocesses', 0) is not None or config.getoption('usepdb', False) or config.getoption('collectonly', False):
            return                  # no xdist (attribute missing -> 0), or the caller passed -n
        import xdist  # noqa: F401
        n = int(os.environ.get('PNSFM_TEST_WORKERS', min(6, max(1, (os.cpu_count() or 1) - 2))))
        if n > 1:
            config.option.numprocesses = n
            config.option.dist = 'load'
            config.option.tx = ['popen'] * n
    except Exception:
        pass


# Collection order (one noise-bound whole-step test in the middle of the suite hid 58 kernel tests from a `-x` run):
# kernel-vs-oracle and block-level golden tests first, then whole networks, then optimizer / trainer / multi-step tests, and the
# full-size (BASELINE.json shapes) and multi-process tests last.  Within a tier the file order is kept.
_TIERS = (
    (3, ('full_size', 'two_rank', 'gradient_accumulation', 'trainer_fit', 'deterministic_step')),
    (2, ('flat_adam', 'trainer_loop', 'forced_collectives', 'follow_the_optimizer', 'adam_vs_torch', 'training_step_golden',
         'semisup', 'packnet_san')),
    (1, ('packnet01', 'packnetslim01', 'posenet_golden', 'dropout_and_eval')),
)


def _tier(item):
    name = item.name
    for tier, keys in _TIERS:
        if any(k in name for k in keys):
            return tier
    return 0


def pytest_collection_modifyitems(config, items):
    items.sort(key=_tier)        # stable


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
    lib.pnsfm_set_gn_fused(1)
    yield
    _lib._LIB, _lib.REQUIRE_CUDA = saved
