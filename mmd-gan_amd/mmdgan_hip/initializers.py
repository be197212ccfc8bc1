"""Variable initialisers of the reference (layer_func.py:14-80, 709-747), as NumPy draws.

`FLAGS.WEIGHT_INITIALIZER` (misc_fun.py) selects the kernel initialiser for every d / c / tc kernel:
  'default'   tf.variance_scaling_initializer by activation (layer_func.py:27-52):
              relu    -> truncated normal, stddev sqrt(2 * init_w_scale / fan_in)            (:34-36)
              lrelu   -> truncated normal, stddev sqrt(2 / 1.01 * init_w_scale / fan_in)     (:42-44)
              sigmoid -> uniform +- sqrt(3 * 16 * init_w_scale / fan_avg)                    (:45-47)
              other   -> uniform +- sqrt(3 * init_w_scale / fan_avg)  (Xavier)               (:50-52)
              init_w_scale == 0 -> zeros                                                     (:28-29)
  'sn_paper'  truncated normal, stddev 0.02, whatever the activation                        (:55-58)
  'pg_paper'  truncated normal, stddev 1                                                     (:59-62)
  anything else: NotImplementedError('The initializer ... is not implemented.')              (:63-64)
TF-1.8 semantics underneath: a truncated normal re-draws samples beyond two standard deviations and `stddev` is the
parameter of the un-truncated normal (no 0.8796 correction); variance scaling takes fan_in = shape[-2] * receptive
field and fan_out = shape[-1] * receptive field for EVERY kernel, so a transposed-conv kernel [k, k, Cout, Cin]
gets its fan_in from Cout - kept.
Biases: truncated normal, stddev 1e-5 (layer_func.py:747 -> :78); BN gamma 1 / beta 0; SN start vectors truncated
normal(0, 1), NOT normalised (math_func.py:565-567).

The random STREAM is NumPy's, not TF's: the distributions are the contract (parity tests inject weights explicitly).
"""
import math

import numpy as np

WEIGHT_INITIALIZERS = ('default', 'sn_paper', 'pg_paper')


def check_mode(mode):
    if mode not in WEIGHT_INITIALIZERS:
        raise NotImplementedError('The initializer {} is not implemented.'.format(mode))      # layer_func.py:64
    return mode


def trunc_normal(rng, shape, stddev):
    out = rng.randn(*shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.randn(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def fans(shape):
    """TF's fan rule: (fan_in, fan_out) = (shape[-2], shape[-1]) x receptive field size"""
    receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return shape[-2] * receptive, shape[-1] * receptive


def weight_initializer(rng, shape, act, mode='default', init_w_scale=1.0):
    """one kernel of `shape` for a layer whose activation is `act` (layer_func.py:14-66)"""
    check_mode(mode)
    shape = [int(v) for v in shape]
    if mode == 'sn_paper':
        return trunc_normal(rng, shape, 0.02)
    if mode == 'pg_paper':
        return trunc_normal(rng, shape, 1.0)
    if init_w_scale == 0.0:
        return np.zeros(shape, np.float32)
    fan_in, fan_out = fans(shape)
    if act == 'relu':
        return trunc_normal(rng, shape, math.sqrt(2.0 * init_w_scale / max(1.0, fan_in)))
    if act == 'lrelu':
        return trunc_normal(rng, shape, math.sqrt(2.0 / 1.01 * init_w_scale / max(1.0, fan_in)))
    scale = 16.0 * init_w_scale if act == 'sigmoid' else 1.0 * init_w_scale
    lim = math.sqrt(3.0 * scale / max(1.0, (fan_in + fan_out) / 2.0))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def bias_initializer(rng, shape, init_b_scale=1e-5):
    """layer_func.py:69-80; every bias of the hot path is created with init_b_scale = 1e-5 (:747)"""
    if init_b_scale == 0.0:
        return np.zeros(shape, np.float32)
    return trunc_normal(rng, [int(v) for v in shape], init_b_scale)
