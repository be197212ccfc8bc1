"""shared helpers for the parity tests (oracle = checker, never the thing measured)."""
import ast
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# Tolerances (BASELINE.json north_star): fp32 MMD loss and conv activations within 1e-4
# relative; index masks bit-exact.  "relative" for a tensor = max|a-b| / max|b|.
RTOL = 1e-4


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.max(np.abs(b))
    if scale == 0.0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)


def elementwise_err(a, b, floor_frac=1e-2):
    """the ELEMENT-WISE companion of rel_err: max over elements of |a-b| / (|b| + floor_frac * max|b|).  rel_err is
    norm-wise (max|a-b| / max|b|) and leaves small-magnitude entries unconstrained; here every entry above floor_frac
    (1 %) of the tensor's scale is held to the relative bar ON ITS OWN, and the entries below it to an absolute
    floor_frac * bar * scale (1e-6 of the scale at the 1e-4 bar).  The floor is what fp32 allows: a sum of hundreds of
    products of either sign carries an absolute error of ~1e-7 of the SCALE of its terms whatever the size of the
    result (measured on the Winograd kernels: 1.2e-7 of max|y| at the zero crossings), so an entry at 1e-4 of the scale
    has three significant digits in ANY fp32 implementation, the reference's included."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor_frac * np.max(np.abs(b)))))


def sign_flips(got_nhwc, ref_nchw):
    """how many activation outputs have a different SIGN in an fp32 evaluation (engine buffer, NHWC or [N,F]) than in
    the fp64 oracle's (NCHW or [N,F]): the lrelu / relu derivative masks that differ between the two.  One such flip
    moves ONE element of the back-propagated gradient by 0.9 (lrelu) or 1.0 (relu) of itself - against a tensor of N
    elements that is ~1/sqrt(N) of its L2 norm times the element's relative size, i.e. 2e-3 ... 1e-2 for the 1e5 ... 1e6
    elements of a D layer (tools/flip_probe.py: with NO flip above a layer its gradients agree to 3e-6, with one flip
    to 6e-3).  Pre-activations land within fp32 resolution (~1e-6 of their scale) of zero about once per million."""
    got, ref = np.asarray(got_nhwc), np.asarray(ref_nchw)
    if ref.ndim == 4:
        ref = ref.transpose(0, 2, 3, 1)
    return int(((got.reshape(ref.shape) > 0) != (ref > 0)).sum())


def golden(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def designs_of(fx):
    return ast.literal_eval(str(fx['designs_repr']))
