"""ctypes loader for oracle/mmd_oracle.c (plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libmmd_oracle.so')
LOSS = {'rep': 0, 'rmb': 1}


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    return ctypes.CDLL(_SO)


def mmd(s_gen, s_x, loss_type='rep', rep_weights=(0.0, -1.0), lb=0.25, ub=4.0, dtype=np.float32):
    """returns dict(loss_gen, loss_dis, stats[5], dist[3,B,B], masks[3,B,B], grads[4,B,d])."""
    lib = _lib()
    x = np.ascontiguousarray(s_gen, dtype=dtype)
    y = np.ascontiguousarray(s_x, dtype=dtype)
    B, d = x.shape
    fn = lib.mmd_oracle_f32 if dtype == np.float32 else lib.mmd_oracle_f64
    real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    ptr = ctypes.POINTER(real)
    fn.restype = ctypes.c_int
    fn.argtypes = [ptr, ptr, ctypes.c_int, ctypes.c_int, ctypes.c_int, real, real, real, real,
                   ptr, ptr, ptr, ctypes.POINTER(ctypes.c_ubyte), ptr]
    losses, stats = np.zeros(2, dtype), np.zeros(5, dtype)
    dist, masks = np.zeros((3, B, B), dtype), np.zeros((3, B, B), np.uint8)
    grads = np.zeros((4, B, d), dtype)
    p = lambda a: a.ctypes.data_as(ptr)
    rc = fn(p(x), p(y), B, d, LOSS[loss_type], rep_weights[0], rep_weights[1], lb, ub,
            p(losses), p(stats), p(dist), masks.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), p(grads))
    if rc != 0:
        raise ValueError('mmd_oracle: bad arguments (rc=%d)' % rc)
    return {'loss_gen': losses[0], 'loss_dis': losses[1], 'stats': stats, 'dist': dist,
            'masks': masks.astype(bool), 'grads': grads}
