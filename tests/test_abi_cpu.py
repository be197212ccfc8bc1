"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mmdgan_hip.h declares, and the product path refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT
from mmdgan_hip import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mmdgan_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mmdgan_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, 'loader signature table lacks %s' % n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mmdgan_version() >= 100


def test_no_cpu_fallback():
    from mmdgan_hip import ops
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.HipLibraryError):
        ops.mmd_loss(torch.zeros(4, 16), torch.zeros(4, 16))
    with pytest.raises(_lib.HipLibraryError):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 3), torch.zeros(3, 3, 3, 8), 1)


def test_argument_errors_do_not_need_a_device():
    lib = _lib.load()
    # validation happens before any launch: bad batch size / weights return MMDGAN_E_ARG
    assert lib.mmdgan_mmd_loss(1, 1, 1, 16, 0, 0.0, -1.0, 0.25, 4.0, 1, None, None, None, 1, None) == -1
    assert b'batch_size' in lib.mmdgan_last_error()
    assert lib.mmdgan_mmd_loss(1, 1, 8, 16, 0, 0.5, -1.0, 0.25, 4.0, 1, None, None, None, 1, None) == -1
    assert b'w[0]-w[1] must be 1' in lib.mmdgan_last_error()       # math_func.py:1340
    assert lib.mmdgan_mmd_loss(1, 1, 8, 16, 7, 0.0, -1.0, 0.25, 4.0, 1, None, None, None, 1, None) == -1
