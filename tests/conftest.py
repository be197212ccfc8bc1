import os
import sys

import pytest

os.environ.setdefault('MMDGAN_WINO_MIN_TILES', '32')
os.environ.setdefault('MMDGAN_WINO2', '2')               # ... and the F(2x2,2x2) stride-2 kernels in both directions     # let the small parity cases reach the Winograd kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'mmd-gan_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
