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


def test_kernel_selection_switches_are_listed_in_one_place(monkeypatch):
    """the library reports its kernel-selection switches (csrc/tuning.h) and which of them are off their default; the host
    side's switches live in mmdgan_hip/settings.py.  tests/conftest.py sets two of the library's for this process."""
    from mmdgan_hip import ops, settings
    t = ops.tuning()
    assert set(t) == {'force_direct', 'thin_valu', 'wino', 'wino_min_tiles', 'wino_ksplit_below', 'wino_wgrad', 'wino_wgrad_slab',
                      'wino2', 'wino2_ksplit', 'wino2_ksplit_below', 'wino2_wgrad', 'wino2_wgrad_min_tiles', 'wgrad_cus', 'gemm_skinny',
                      'gemm_panel', 'mmd_d16', 'wino43', 'wino43_min_tiles', 'wino43_ksplit_below', 'wino43_wgrad', 'wino43_wgrad_min_tiles', 'wino43_wgrad_cus'}
    assert t['wino_min_tiles'] == (32, False) and t['wino2'] == (2, False) and t['wino43'] == (2, False)   # conftest's thresholds
    assert t['wgrad_cus'] == (224, True) and t['wino'] == (1, True) and t['wino43_wgrad'] == (1, True) and t['wino43_wgrad_min_tiles'] == (64, False)
    assert settings.describe() == {} or all(k.startswith('MMDGAN_') for k in settings.describe())
    monkeypatch.setenv('MMDGAN_SN_STREAMS', '1')
    monkeypatch.setenv('MMDGAN_NO_SUCH_SWITCH', '1')
    assert settings.describe().get('MMDGAN_SN_STREAMS') == '1' and settings.get('MMDGAN_SN_STREAMS') == '1'
    assert 'MMDGAN_NO_SUCH_SWITCH' in settings.unknown() and 'MMDGAN_WINO2' not in settings.unknown()
    # no other place reads the environment for a switch: the sources hold no getenv / os.environ beside these two files
    import glob
    root = os.path.join(os.path.dirname(__file__), '..', 'mmd-gan_amd')
    for path in glob.glob(os.path.join(root, 'csrc', '*')):
        if not path.endswith('tuning.h'):
            assert 'getenv(' not in open(path).read(), path
    for path in glob.glob(os.path.join(root, 'mmdgan_hip', '*.py')) + glob.glob(os.path.join(root, '*', '*.py')):
        if not path.endswith(('settings.py', 'dist.py', '_lib.py')):
            assert "environ.get('MMDGAN_" not in open(path).read(), path


def test_handles_and_plan_bookkeeping_without_a_device():
    """handles own what used to be process-global state; a plan's segment bookkeeping is host logic"""
    import ctypes
    lib = _lib.load()
    h1, h2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.mmdgan_create(ctypes.byref(h1)) == 0 and lib.mmdgan_create(ctypes.byref(h2)) == 0 and h1.value != h2.value
    assert lib.mmdgan_make_current(h1) == 0
    assert lib.mmdgan_plan_end(ctypes.byref(ctypes.c_int())) == -1 and b'not recording' in lib.mmdgan_last_error()
    assert lib.mmdgan_plan_begin() == 0
    assert lib.mmdgan_plan_begin() == -1 and b'already recording' in lib.mmdgan_last_error()
    assert lib.mmdgan_set_outputs_prezeroed(1) == 0          # a mode switch inside a recording is a node
    assert lib.mmdgan_plan_mark() == 1
    assert lib.mmdgan_set_outputs_prezeroed(0) == 0
    pid = ctypes.c_int(-1)
    assert lib.mmdgan_plan_end(ctypes.byref(pid)) == 0 and pid.value == 0
    assert lib.mmdgan_plan_segments(0) == 2 and lib.mmdgan_plan_nodes(0) == 2
    assert lib.mmdgan_plan_describe(0, None, 0) == 1        # no kernel launch in it: an empty listing (terminator only)
    # the other handle knows nothing of it
    assert lib.mmdgan_make_current(h2) == 0
    assert lib.mmdgan_plan_segments(0) < 0
    assert lib.mmdgan_plan_replay(0, -1) == -1 and b'no plan 0' in lib.mmdgan_last_error()
    assert lib.mmdgan_make_current(h1) == 0
    assert lib.mmdgan_plan_replay(0, 2) == -1 and b'has 2 segments' in lib.mmdgan_last_error()
    assert lib.mmdgan_plan_destroy(0) == 0 and lib.mmdgan_plan_segments(0) < 0
    assert lib.mmdgan_make_current(None) == 0
    assert lib.mmdgan_destroy(h1) == 0 and lib.mmdgan_destroy(h2) == 0 and lib.mmdgan_destroy(None) == 0
