import os
import sys

import pytest

# the oracle's torch-CPU steps run fastest on a few dozen threads, not on all 128 of the GPU box's (bench.py's thread sweep:
# 16 threads 180 img/s, 128 threads 23) - before torch is imported, and inherited by the tests' subprocesses
os.environ.setdefault('OMP_NUM_THREADS', '32')
os.environ.setdefault('MMDGAN_WINO_MIN_TILES', '32')     # let the small parity cases reach the Winograd kernels
os.environ.setdefault('MMDGAN_WINO2', '2')               # ... and the F(2x2,2x2) stride-2 kernels in both directions
os.environ.setdefault('MMDGAN_WINO43', '2')              # ... and F(4x4,3x3) wherever H and W are multiples of 4
os.environ.setdefault('MMDGAN_WINO43_WGRAD_MIN_TILES', '64')   # ... its weight gradient from 64 tiles on (below: F(2x2,3x3) keeps its cases)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'mmd-gan_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # (pytest.ini at the repo root: `timeout = 600` - with pytest-timeout installed a hung GPU test fails after ten minutes
    # instead of holding the box until the caller's limit)


def _gpu_usable():
    """a gfx950 device this process can launch on (the library is the judge: mmdgan_device_ok)"""
    try:
        import torch
        if not torch.cuda.is_available():
            return False
        from mmdgan_hip import _lib
        return _lib.load().mmdgan_device_ok() == 1
    except Exception:                                    # no library / no torch: the gpu tests cannot run
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without an MI355X: the gpu-marked tests are SKIPPED (with the reason), not run
    into require_device()'s error - a CPU-only CI then shows real regressions only.  On a GPU box nothing changes."""
    if not any('gpu' in it.keywords for it in items) or _gpu_usable():
        return
    skip = pytest.mark.skip(reason='needs a gfx950 (MI355X) device: run with -m gpu on the GPU box')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """a fresh checkout has no libmmdgan_hip.so yet (built artefacts are not in git): build it once, the way
    __graft_entry__.build() does (hipcc cross-compiles for gfx950 without a GPU).  The product code itself never
    builds on demand - a missing library is an error there."""
    import importlib.util
    lib = os.path.join(PKG, 'lib', 'libmmdgan_hip.so')
    if not os.path.exists(lib):
        spec = importlib.util.spec_from_file_location('_mmdgan_build_ext', os.path.join(PKG, 'build_ext.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)


def pytest_terminal_summary(terminalreporter):
    """how often a step test fell back to the comparison under the engine's own sign decisions (helpers.note_knife_edge_retry):
    printed in every gpu session's log, zero included, so the frequency of that path is on record"""
    try:
        import helpers
    except ImportError:
        return
    n = len(helpers.KNIFE_EDGE_RETRIES)
    terminalreporter.write_line('knife-edge retries (fallback to the engine\'s sign decisions): %d' % n)
    for what in helpers.KNIFE_EDGE_RETRIES:
        terminalreporter.write_line('  knife-edge retry: ' + what)
    # ... and how often a gradient needed the escape clause of helpers.assert_grads_within_fp32_floor (above the plain 1e-4 L2 /
    # 1e-3 entry-wise bars, inside twice the fp32 oracle's own loss)
    terminalreporter.write_line('steps whose gradients were held against the audited fp64 evaluation: %d' % len(helpers.AUDITED_STEPS))
    for what in helpers.AUDITED_STEPS:
        terminalreporter.write_line('  audited step: ' + what)
    terminalreporter.write_line('gradient tensors that took the fp32-floor clause: %d' % len(helpers.FLOOR_CLAUSE_USES))
    for what in helpers.FLOOR_CLAUSE_USES:
        terminalreporter.write_line('  fp32-floor clause: ' + what)
