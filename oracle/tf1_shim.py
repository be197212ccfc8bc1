"""Eager TensorFlow-1.x API subset backed by torch-CPU.  TEST INFRASTRUCTURE ONLY.

Purpose: TensorFlow is not installed (and cannot be installed) in the build
container, so the reference's own modules (`/root/reference/GeneralTools/
{misc_fun,math_func,layer_func}.py`) cannot be imported as they are.  This
module registers itself as `sys.modules['tensorflow']` so that those modules
import and run UNMODIFIED, in this container only, for one purpose: generating
the golden vectors under `tests/golden/` (see `oracle/make_golden.py`).

What this pins and what it does not (said plainly, see DESIGN.md "Oracle"):
  * pins: the reference's algebra, control flow, shapes, defaults, variable
    names and quirks (they are executed from the reference's own source);
  * does not pin: the rounding of TF-1.8's Eigen/cuDNN kernels - the arithmetic
    primitives underneath are torch-CPU (oneDNN / MKL) fp32 or fp64.

Nothing in the product path (`mmd-gan_amd/`) imports this file.  It never
travels to the GPU box in a form that matters: `/root/reference` does not exist
there, so this shim has nothing to import.

Only the tf symbols the hot path touches are provided (SURVEY.md section 8(c)
lists them); anything else raises AttributeError loudly.
"""
import builtins
import contextlib
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# global eager state
# ---------------------------------------------------------------------------
DTYPE = torch.float32          # switch to torch.float64 for the fp64 goldens


class _State:
    def __init__(self):
        self.reset()

    def reset(self):
        self.variables = {}        # full name -> torch tensor (leaf)
        self.trainable = []        # names in creation order
        self.scope = []            # variable scope stack
        self.update_ops = []       # list of (variable name, new value tensor)
        self.rng = np.random.RandomState(0)


STATE = _State()


def set_dtype(dtype):
    global DTYPE
    DTYPE = dtype


# ---------------------------------------------------------------------------
# tensor helpers
# ---------------------------------------------------------------------------
class _Shape(list):
    def as_list(self):
        return list(self)


def _get_shape(self):
    return _Shape(self.shape)


torch.Tensor.get_shape = _get_shape          # tf.Tensor.get_shape().as_list()


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x), dtype=DTYPE)


# ---------------------------------------------------------------------------
# scopes / variables / collections
# ---------------------------------------------------------------------------
@contextlib.contextmanager
def name_scope(name=None, *a, **k):
    yield


class _VarScope:
    def reuse_variables(self):
        pass

    @property
    def name(self):
        return '/'.join(STATE.scope)


@contextlib.contextmanager
def variable_scope(name_or_scope=None, reuse=None, *a, **k):
    if isinstance(name_or_scope, _VarScope) or name_or_scope is None:
        yield _VarScope()
        return
    # absolute scopes (name ends with '/') are not used by the hot path
    parts = [p for p in str(name_or_scope).split('/') if p]
    STATE.scope.extend(parts)
    try:
        yield _VarScope()
    finally:
        del STATE.scope[len(STATE.scope) - len(parts):]


def get_variable_scope():
    return _VarScope()


AUTO_REUSE = 'AUTO_REUSE'


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **k):
    full = '/'.join(STATE.scope + [name])
    if full not in STATE.variables:
        if isinstance(shape, int):
            shape = [shape]
        if callable(initializer) and not isinstance(initializer, torch.Tensor):
            try:
                value = initializer(list(shape))
            except TypeError:                     # initializer class, not instance
                value = initializer()(list(shape))
            if callable(value) and not isinstance(value, torch.Tensor):
                value = initializer()(list(shape))  # the factory itself was passed (tf.zeros_initializer, math_func.py:1995)
        else:
            value = _t(initializer)
        value = value.to(DTYPE).clone().detach()
        value.requires_grad_(builtins.bool(trainable))
        value.tf_name = full
        STATE.variables[full] = value
        if trainable:
            STATE.trainable.append(full)
    return STATE.variables[full]


class GraphKeys:
    UPDATE_OPS = 'update_ops'
    TRAINABLE_VARIABLES = 'trainable_variables'


class _Assign:
    def __init__(self, var, value):
        self.var, self.value = var, value


def assign(var, value, *a, **k):
    return _Assign(var, value.detach())


def add_to_collection(key, value):
    assert key == GraphKeys.UPDATE_OPS
    STATE.update_ops.append(value)


def get_collection(key, scope=None):
    if key == GraphKeys.TRAINABLE_VARIABLES:
        return [STATE.variables[n] for n in STATE.trainable
                if scope is None or n.startswith(scope)]
    if key == GraphKeys.UPDATE_OPS:
        return list(STATE.update_ops)
    raise KeyError(key)


def run_update_ops():
    """apply (and clear) the registered UPDATE_OPS: all reads precede all writes."""
    for op in STATE.update_ops:
        with torch.no_grad():
            op.var.copy_(op.value)
    STATE.update_ops = []


@contextlib.contextmanager
def control_dependencies(ops):
    yield


# ---------------------------------------------------------------------------
# initializers (numpy RandomState; goldens inject explicit weights anyway)
# ---------------------------------------------------------------------------
def _trunc_normal(shape, stddev=1.0, mean=0.0):
    out = STATE.rng.randn(*shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = STATE.rng.randn(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return torch.as_tensor(out * stddev + mean, dtype=DTYPE)


def truncated_normal_initializer(mean=0.0, stddev=1.0, **k):
    return lambda shape: _trunc_normal(shape, stddev, mean)


def zeros_initializer(*a, **k):
    return lambda shape: torch.zeros(shape, dtype=DTYPE)


def ones_initializer(*a, **k):
    return lambda shape: torch.ones(shape, dtype=DTYPE)


def variance_scaling_initializer(scale=1.0, mode='fan_in', distribution='normal', **k):
    def init(shape):
        receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in = shape[-2] * receptive if len(shape) > 1 else shape[0]
        fan_out = shape[-1] * receptive
        n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.0}[mode]
        s = scale / max(1.0, n)
        if distribution == 'normal':
            # TF 1.8: truncated normal with stddev sqrt(s) (no .8796 correction)
            return _trunc_normal(shape, stddev=np.sqrt(s))
        limit = np.sqrt(3.0 * s)
        return torch.as_tensor(STATE.rng.uniform(-limit, limit, size=shape), dtype=DTYPE)
    return init


# ---------------------------------------------------------------------------
# math
# ---------------------------------------------------------------------------
def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return torch.matmul(a, b)


def identity(x, name=None):
    return x


def reshape(x, shape, name=None):
    return x.reshape([int(s) for s in shape])


def norm(x, ord='euclidean', axis=None, keepdims=False, name=None):
    assert ord in ('euclidean', 2)
    if axis is None:
        return torch.sqrt(torch.sum(x * x))
    return torch.sqrt(torch.sum(x * x, dim=axis, keepdim=keepdims))


def constant(value, dtype=None, shape=None, name=None):
    if dtype in (int32, int64):
        return torch.as_tensor(value, dtype=dtype)
    return torch.as_tensor(value, dtype=DTYPE)


def diag_part(x, name=None):
    return torch.diagonal(x)


matrix_diag_part = diag_part


def expand_dims(x, axis, name=None):
    return x.unsqueeze(axis)


def maximum(a, b, name=None):
    return torch.maximum(_t(a), _t(b).to(_t(a).dtype))


def minimum(a, b, name=None):
    return torch.minimum(_t(a), _t(b).to(_t(a).dtype))


def multiply(a, b, name=None):
    return a * b


def add(a, b, name=None):
    return a + b


def cos(x, name=None):
    return torch.cos(_t(x))


def sin(x, name=None):
    return torch.sin(_t(x))


def eye(num_rows, num_columns=None, dtype=None, name=None):
    return torch.eye(int(num_rows), int(num_columns if num_columns is not None else num_rows), dtype=DTYPE)


def exp(x, name=None):
    return torch.exp(x)


def sqrt(x, name=None):
    return torch.sqrt(x)


def square(x, name=None):
    return x * x


def reduce_sum(x, axis=None, keepdims=False, name=None):
    if axis is None:
        return torch.sum(x)
    return torch.sum(x, dim=axis, keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False, name=None):
    if axis is None:
        return torch.mean(x)
    return torch.mean(x, dim=axis, keepdim=keepdims)


def transpose(x, perm=None, name=None):
    return x.permute(*perm) if perm is not None else x.t()


def concat(values, axis, name=None):
    return torch.cat(list(values), dim=axis)


def split(value, num_or_size_splits, axis=0, name=None):
    return list(torch.chunk(value, num_or_size_splits, dim=axis))


def while_loop(cond, body, loop_vars, **k):
    v = tuple(loop_vars)
    while builtins.bool(cond(*v)):
        v = tuple(body(*v))
    return v


def zeros(shape, dtype=None, name=None):
    return torch.zeros(shape, dtype=DTYPE)


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, name=None):
    return torch.as_tensor(STATE.rng.randn(*shape) * stddev + mean, dtype=DTYPE)


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, name=None):
    """tf.random_uniform; the draw is kept in STATE.last_uniform so that a fixture can record it as an INPUT (the
    build takes `uni` as an argument instead of reproducing TF's random stream, math_func.py:2079)"""
    u = STATE.rng.uniform(minval, maxval, size=list(shape)).astype(np.float32)      # tf.float32 draws in every precision
    STATE.last_uniform = u
    return torch.as_tensor(u, dtype=DTYPE)


def greater(a, b, name=None):
    return _t(a) > _t(b)


def logical_not(x, name=None):
    return torch.logical_not(x)


def boolean_mask(tensor, mask, axis=0, name=None):
    idx = torch.nonzero(mask, as_tuple=False).reshape(-1)
    return torch.index_select(tensor, axis, idx)


def gather(params, indices, axis=0, name=None):
    return torch.index_select(params, axis, indices.long())


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return torch.clamp(_t(t), min=clip_value_min, max=clip_value_max)


float32 = torch.float32
float64 = torch.float64
int32 = torch.int32
int64 = torch.int64
bool = torch.bool            # noqa: A001  (tf.bool)
string = 'string'
Tensor = torch.Tensor


# ---------------------------------------------------------------------------
# tf.nn
# ---------------------------------------------------------------------------
def _stride_hw(strides, data_format):
    assert data_format in ('NCHW', 'channels_first'), 'shim implements the reference default NCHW only'
    return int(strides[2]), int(strides[3])


def _same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _dilation_hw(dilations, data_format):
    if dilations is None:
        return 1, 1
    assert data_format in ('NCHW', 'channels_first')
    return int(dilations[2]), int(dilations[3])


def _conv2d(x, kernel, strides, padding, use_cudnn_on_gpu=True, data_format='NHWC', dilations=None, name=None):
    """tf.nn.conv2d, NCHW: 'SAME' (pad_before = total // 2 on the dilated extent) or 'VALID'; dilations [1,1,d,d]"""
    sh, sw = _stride_hw(strides, data_format)
    dh, dw = _dilation_hw(dilations, data_format)
    assert padding in ('SAME', 'VALID'), padding
    kh, kw = (kernel.shape[0] - 1) * dh + 1, (kernel.shape[1] - 1) * dw + 1       # dilated extents
    if padding == 'SAME':
        pt, pb = _same_pad(x.shape[2], kh, sh)
        pl, pr = _same_pad(x.shape[3], kw, sw)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, kernel.permute(3, 2, 0, 1), stride=(sh, sw), dilation=(dh, dw))


def _conv2d_transpose(value, kernel, output_shape, strides, padding='SAME', data_format='NHWC', name=None):
    """gradient of _conv2d w.r.t. its input; kernel is [k,k,out_channels,in_channels]."""
    sh, sw = _stride_hw(strides, data_format)
    assert padding in ('SAME', 'VALID'), padding
    k = kernel.shape[0]
    oh, ow = int(output_shape[2]), int(output_shape[3])
    pt, pb = _same_pad(oh, k, sh) if padding == 'SAME' else (0, 0)
    pl, pr = _same_pad(ow, k, sw) if padding == 'SAME' else (0, 0)
    full = F.conv_transpose2d(value, kernel.permute(3, 2, 0, 1), stride=(sh, sw))
    # full spatial size is (in-1)*s+k; the conv's padded input was oh+pt+pb (VALID: rows the conv never read get zeros)
    fh, fw = full.shape[2], full.shape[3]
    if fh < oh + pt + pb or fw < ow + pl + pr:
        full = F.pad(full, (0, max(ow + pl + pr - fw, 0), 0, max(oh + pt + pb - fh, 0)))
    return full[:, :, pt:pt + oh, pl:pl + ow]


def _bias_add(x, bias, data_format=None, name=None):
    if data_format in ('NCHW', 'channels_first'):
        return x + bias.reshape(1, -1, 1, 1)
    return x + bias


def _avg_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
    """tf.nn.avg_pool, NCHW, window == stride, 'SAME' on sizes the window divides (what ImageScaling 'avg' issues,
    layer_func.py:1155-1159)"""
    kh, kw = _stride_hw(ksize, data_format)
    assert (kh, kw) == _stride_hw(strides, data_format) and value.shape[2] % kh == 0 and value.shape[3] % kw == 0
    return F.avg_pool2d(value, (kh, kw))


def depth_to_space(x, block_size, data_format='NHWC', name=None):
    """tf.depth_to_space, NCHW: out[n, c, h*r + i, w*r + j] = in[n, (i*r + j)*C + c, h, w]"""
    assert data_format == 'NCHW'
    r = int(block_size)
    n, cr2, h, w = x.shape
    c = cr2 // (r * r)
    return x.reshape(n, r, r, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, c, h * r, w * r)


def space_to_depth(x, block_size, data_format='NHWC', name=None):
    """tf.space_to_depth, NCHW: out[n, (i*r + j)*C + c, h, w] = in[n, c, h*r + i, w*r + j]"""
    assert data_format == 'NCHW'
    r = int(block_size)
    n, c, hr, wr = x.shape
    h, w = hr // r, wr // r
    return x.reshape(n, c, h, r, w, r).permute(0, 3, 5, 1, 2, 4).reshape(n, r * r * c, h, w)


def _max_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
    """tf.nn.max_pool, NCHW, window == stride on sizes the window divides (ImageScaling 'max', layer_func.py:1149-1153);
    the gradient goes to the first maximum of a window in row-major order, as TF's kernels route it"""
    kh, kw = _stride_hw(ksize, data_format)
    assert (kh, kw) == _stride_hw(strides, data_format) and value.shape[2] % kh == 0 and value.shape[3] % kw == 0
    return F.max_pool2d(value, (kh, kw))


nn = types.SimpleNamespace(
    avg_pool=_avg_pool,
    max_pool=_max_pool,
    conv2d=_conv2d,
    conv2d_transpose=_conv2d_transpose,
    bias_add=_bias_add,
    leaky_relu=lambda x, alpha=0.2, name=None: torch.where(x > 0, x, x * alpha),
    relu=lambda x, name=None: torch.relu(x),
    tanh=lambda x, name=None: torch.tanh(x),
    sigmoid=lambda x, name=None: torch.sigmoid(x),
    softplus=lambda x, name=None: F.softplus(x),
)


# ---------------------------------------------------------------------------
# tf.layers.batch_normalization (TF defaults momentum=.99, epsilon=1e-3, fused)
# ---------------------------------------------------------------------------
def _batch_normalization(x, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True,
                         beta_initializer=None, gamma_initializer=None, gamma_constraint=None,
                         training=False, renorm=False, fused=None, name='batch_normalization', **k):
    c = x.shape[axis]
    with variable_scope(name):
        gamma = get_variable('gamma', [c], initializer=gamma_initializer or ones_initializer())
        beta = get_variable('beta', [c], initializer=beta_initializer or zeros_initializer())
        mm = get_variable('moving_mean', [c], initializer=zeros_initializer(), trainable=False)
        mv = get_variable('moving_variance', [c], initializer=ones_initializer(), trainable=False)
    dims = [d for d in range(x.dim()) if d != (axis % x.dim())]
    bshape = [1] * x.dim()
    bshape[axis % x.dim()] = c
    if training:
        mean = x.mean(dim=dims)
        var = ((x - mean.reshape(bshape)) ** 2).mean(dim=dims)          # biased
        n = x.numel() // c
        var_unbiased = var * (n / max(n - 1.0, 1.0)) if x.dim() == 4 else var   # fused kernel only for 4-D
        add_to_collection(GraphKeys.UPDATE_OPS,
                          assign(mm, mm * momentum + mean.detach() * (1 - momentum)))
        add_to_collection(GraphKeys.UPDATE_OPS,
                          assign(mv, mv * momentum + var_unbiased.detach() * (1 - momentum)))
    else:
        mean, var = mm, mv
    y = (x - mean.reshape(bshape)) / torch.sqrt(var.reshape(bshape) + epsilon)
    return y * gamma.reshape(bshape) + beta.reshape(bshape)


def _resize_bilinear(images, size, align_corners=False, name=None):
    """tf.image.resize_bilinear on NHWC with align_corners=True (what ImageScaling 'bil' issues, layer_func.py:1128-1137):
    source coordinate = y * (in - 1) / (out - 1), linear interpolation between the two neighbours"""
    assert align_corners
    x = images.permute(0, 3, 1, 2)
    y = F.interpolate(x, size=(int(size[0]), int(size[1])), mode='bilinear', align_corners=True)
    return y.permute(0, 2, 3, 1)


def _bicubic_matrix(out_n, in_n):
    """dense [out_n, in_n] interpolation matrix of tf.image.resize_bicubic(align_corners=True), TF 1.x legacy kernel
    (resize_bicubic_op.cc): per output position the four clamped taps around floor(o * (in-1)/(out-1)), weights from the
    1024-entry Keys-cubic table (A = -0.75); clamped taps that coincide add up.  Built entry by entry (independent of
    oracle/restatement.py's gather form)."""
    import math
    import numpy as np
    table = np.zeros((1025, 2), np.float32)
    for i in range(1025):
        x = float(np.float32(i / 1024.0))
        table[i, 0] = np.float32((((-0.75 + 2) * x - (-0.75 + 3)) * x * x + 1))
        x = float(np.float32(np.float32(x) + np.float32(1.0)))
        table[i, 1] = np.float32((((-0.75 * x - 5 * -0.75) * x + 8 * -0.75) * x - 4 * -0.75))
    scale = np.float32(in_n - 1) / np.float32(out_n - 1) if out_n > 1 else np.float32(in_n) / np.float32(out_n)
    m = np.zeros((out_n, in_n), np.float64)
    for o in range(out_n):
        loc = np.float32(scale * np.float32(o))
        base = int(loc)
        delta = np.float32(loc - np.float32(base))
        f = float(delta * np.float32(1024))
        off = int(round(f)) if abs(f - math.floor(f) - 0.5) > 0 else int(2 * round(f / 2))        # lrintf: half to even
        w4 = (table[off, 1], table[off, 0], table[1024 - off, 0], table[1024 - off, 1])
        for t in range(4):
            m[o, min(in_n - 1, max(0, base - 1 + t))] += float(w4[t])
    return m


def _resize_bicubic(images, size, align_corners=False, name=None):
    """tf.image.resize_bicubic on NHWC with align_corners=True (ImageScaling 'bic', layer_func.py:1138-1147)"""
    assert align_corners
    n, h, w, c = images.shape
    my = torch.tensor(_bicubic_matrix(int(size[0]), int(h)), dtype=images.dtype)
    mx = torch.tensor(_bicubic_matrix(int(size[1]), int(w)), dtype=images.dtype)
    return torch.einsum('ph,nhwc,qw->npqc', my, images, mx)


image = types.SimpleNamespace(resize_bilinear=_resize_bilinear, resize_bicubic=_resize_bicubic)
layers = types.SimpleNamespace(batch_normalization=_batch_normalization)
summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None,
                                image=lambda *a, **k: None)


# ---------------------------------------------------------------------------
def install():
    """register this module as `tensorflow` and patch numpy aliases the reference needs."""
    if not hasattr(np, 'int'):
        np.int = int                      # math_func.py:190 uses np.int (removed in numpy>=1.24)
    sys.modules['tensorflow'] = sys.modules[__name__]


def __getattr__(name):                    # anything not provided fails loudly
    raise AttributeError('tf1_shim: tf.%s is not provided (not on the hot path)' % name)
