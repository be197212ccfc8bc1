"""CPU restatement of the reference's SN-DCGAN + repulsive-MMD training step.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product path
(`mmd-gan_amd/`) never does and fails loudly without its HIP library.

This is a from-scratch restatement (torch-CPU tensors as the array library,
fp32 or fp64 selectable) of exactly the slice of richardwth/MMD-GAN that
SURVEY.md section 8 puts on the hot path, and of the section 8(f) rows built after
it (the other in-kernel losses, flattened-kernel spectral norm, residual blocks
and scaling ops, FID / sprite helpers).  Every function cites the reference
file:line it follows (paths relative to /root/reference).  It is pinned against
the reference's own code, executed through `oracle/tf1_shim.py`, by the golden
vectors in `tests/golden/` (`oracle/make_golden.py` writes them,
`tests/test_oracle_golden.py` checks them).  The reference itself ships no
tests or golden vectors, so those fixtures are the pin.

Layouts here are the reference's: activations NCHW (misc_fun.py:50-51), conv
kernels HWIO `[k,k,Cin,Cout]`, transposed-conv kernels `[k,k,Cout,Cin]`
(layer_func.py:584,595), dense kernels `[in,out]`; variable names as TF scopes
give them (`dis/l2_ds/kernel/kernel`, `.../kernel/SN/in_rand`, `.../bias/bias`,
`gen/l2_up/BN/BN/gamma` ...).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

EPSI = 1e-10            # misc_fun.py:29 FLAGS.EPSI
BN_EPS = 1e-3           # tf.layers.batch_normalization default epsilon (layer_func.py:960)
BN_MOMENTUM = 0.99      # tf.layers.batch_normalization default momentum
LRELU_ALPHA = 0.1       # layer_func.py:112


# ---------------------------------------------------------------------------
# architecture dict -> layer specs (layer_func.py:1189-1275, 2118-2151, 2221-2391)
# ---------------------------------------------------------------------------
_TEMPLATE = {'name': None, 'type': 'default', 'op': 'c', 'out': None, 'bias': 'b',
             'act': 'linear', 'act_nm': None, 'act_k': False, 'w_nm': None, 'w_p': None,
             'kernel': 3, 'strides': 1, 'dilation': 1, 'padding': 'SAME', 'scale': None,
             'in_reshape': None, 'out_reshape': None, 'aux': None}


def layer_defaults(design):
    """update_layer_design, layer_func.py:1230-1247 (defaults; BN drops the bias)."""
    d = dict(_TEMPLATE)
    d.update(design)
    if d['act_nm'] in ('bn', 'BN') and d['bias'] in ('b', 'bias'):
        d['bias'] = None
    if d['op'] == 'tc':
        d['scale'] = None
    if d['op'] not in ('d', 'c', 'tc', 'i'):
        raise AttributeError('layer op {} not supported.'.format(d['op']))
    if d['type'] not in ('default',) + RES_TYPES:
        raise NotImplementedError('{} is not implemented.'.format(d['type']))
    if d['type'] in RES_TYPES and d['op'] not in ('c', 'tc'):
        raise NotImplementedError('residual blocks are restated for op "c" and "tc"')
    if d['scale'] is not None:
        assert isinstance(d['scale'], (list, tuple)), 'Value for key "scale" must be list or tuple.'   # :1250-1252
    return d


RES_TYPES = ('res', 'res_i', 'res_v1')                       # layer_func.py:2062


def _pick(value, index):
    """Layer._update_design_ (layer_func.py:1380-1395): list-valued design entries are per kernel"""
    return value[index] if isinstance(value, (list, tuple)) else value


def _scaled_shape(shape, scale):
    """ImageScaling._get_shape_ (layer_func.py:1076-1109) for the methods restated here"""
    method, factor = scale
    c, h, w = shape
    if method in ('avg', 'max'):
        if factor > 0:
            raise AttributeError('{} can only be used for downsampling'.format(method))
    elif method == 'unpool':
        if factor < 0:
            raise AttributeError('unpool can only be used for upsampling')
        if factor != 2:
            raise AttributeError('unpool can only deal with factor = 2')
    elif method in ('bil', 'bic'):                            # any integer factor, either direction
        pass
    elif method == 'ps':                                      # periodic shuffling moves pixels into / out of channels
        nh, nw = (int(h * factor), int(w * factor)) if factor > 0 else (int(-h / factor), int(-w / factor))
        return [int(c * h * w / nh / nw), nh, nw]                # layer_func.py:1105-1107
    else:
        raise NotImplementedError('Method {} not implemented.'.format(method))
    return [c, int(h * factor), int(w * factor)] if factor > 0 else [c, int(-h / factor), int(-w / factor)]


def _rescale(x, scale):
    """ImageScaling.__call__ (layer_func.py:1125-1163): 'avg' = avg_pool with window = stride = -factor;
    'unpool' = four channel copies through depth_to_space = every pixel repeated 2 x 2; 'ps' = periodic shuffling"""
    method, factor = scale
    if method == 'avg':
        return F.avg_pool2d(x, -factor)
    if method == 'max':                                       # layer_func.py:1149-1153
        return F.max_pool2d(x, -factor)
    if method == 'bil':                                       # layer_func.py:1128-1137: tf.image.resize_bilinear,
        n, c, h, w = x.shape                                  # align_corners=True
        size = (int(h * factor), int(w * factor)) if factor > 0 else (int(-h / factor), int(-w / factor))
        return bilinear_resize(x, size)
    if method == 'bic':                                       # layer_func.py:1138-1147: tf.image.resize_bicubic,
        n, c, h, w = x.shape                                  # align_corners=True
        size = (int(h * factor), int(w * factor)) if factor > 0 else (int(-h / factor), int(-w / factor))
        return bicubic_resize(x, size)
    if method == 'ps':                                        # layer_func.py:197-244, 1125-1127: tf.depth_to_space /
        r = abs(int(factor))                                  # tf.space_to_depth on NCHW, block-major channel order
        n, c, h, w = x.shape
        if factor > 0:
            co = c // (r * r)
            return x.reshape(n, r, r, co, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, co, h * r, w * r)
        return x.reshape(n, c, h // r, r, w // r, r).permute(0, 3, 5, 1, 2, 4).reshape(n, r * r * c, h // r, w // r)
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def bilinear_resize(x, size):
    """tf.image.resize_bilinear(align_corners=True) written out (resize_bilinear_op.cc semantics): for an output row y
    the source coordinate is y * (in_h - 1) / (out_h - 1); rows floor(.) and min(floor(.) + 1, in_h - 1) are blended
    with the fractional part, columns likewise.  NCHW in, NCHW out."""
    n, c, h, w = x.shape
    oh, ow = size

    def axis(out_n, in_n):
        scale = (in_n - 1) / (out_n - 1) if out_n > 1 else 0.0
        src = torch.arange(out_n, dtype=x.dtype) * scale
        lo = torch.floor(src).long().clamp(max=in_n - 1)
        hi = torch.clamp(lo + 1, max=in_n - 1)
        return lo, hi, (src - lo.to(x.dtype))
    y0, y1, wy = axis(oh, h)
    x0, x1, wx = axis(ow, w)
    top = x[:, :, y0][:, :, :, x0] * (1 - wx) + x[:, :, y0][:, :, :, x1] * wx
    bot = x[:, :, y1][:, :, :, x0] * (1 - wx) + x[:, :, y1][:, :, :, x1] * wx
    return top * (1 - wy)[:, None] + bot * wy[:, None]


def bicubic_taps(out_n, in_n):
    """indices [out_n,4] and weights [out_n,4] (fp32 values, as TF's coefficient table holds them) of
    tf.image.resize_bicubic(align_corners=True), TF 1.x legacy sampling (resize_bicubic_op.cc GetWeightsAndIndices):
    source = o * (in - 1) / (out - 1) in fp32, floor, the fraction rounded half-to-even onto a 1/1024 grid, Keys cubic with
    A = -0.75 evaluated in double and stored as fp32, taps floor-1 .. floor+2 clamped to the image."""
    scale = np.float32(in_n - 1) / np.float32(out_n - 1) if out_n > 1 else np.float32(in_n) / np.float32(out_n)
    loc = (np.float32(scale) * np.arange(out_n, dtype=np.float32)).astype(np.float32)
    base = loc.astype(np.int64)
    off = np.rint((loc - base.astype(np.float32)).astype(np.float32) * np.float32(1024)).astype(np.int64)
    A = -0.75

    def near(i):
        x = (i.astype(np.float32) / np.float32(1024)).astype(np.float64)
        return (((A + 2) * x - (A + 3)) * x * x + 1).astype(np.float32)

    def far(i):
        x = (i.astype(np.float32) / np.float32(1024) + np.float32(1)).astype(np.float64)
        return (((A * x - 5 * A) * x + 8 * A) * x - 4 * A).astype(np.float32)
    wgt = np.stack([far(off), near(off), near(1024 - off), far(1024 - off)], 1)
    idx = np.clip(base[:, None] + np.arange(-1, 3)[None, :], 0, in_n - 1)
    return idx, wgt


def bicubic_resize(x, size):
    """tf.image.resize_bicubic(align_corners=True): rows interpolated along x, then along y.  NCHW in, NCHW out."""
    n, c, h, w = x.shape
    oh, ow = size
    iy, wy = bicubic_taps(oh, h)
    ix, wx = bicubic_taps(ow, w)
    wy_t, wx_t = torch.tensor(wy, dtype=x.dtype), torch.tensor(wx, dtype=x.dtype)
    cols = sum(x[:, :, :, torch.as_tensor(ix[:, b])] * wx_t[:, b] for b in range(4))              # [n,c,h,ow]
    return sum(cols[:, :, torch.as_tensor(iy[:, a])] * wy_t[:, a][:, None] for a in range(4))     # [n,c,oh,ow]


def _kernel_spec(layer_scope, op_name, d, index, in_shape, sn_mode, op='c'):
    """one ParametricOperation of a block (Layer._add_kernel_, layer_func.py:1415-1452): a conv kernel - or, op='tc', a
    transposed-conv kernel [k, k, out, in] (layer_func.py:590-600) - with its own spectral norm; variables live under
    <layer>/<op_name>/"""
    sub = {'op': op, 'type': 'default', 'in_reshape': None, 'out_reshape': None}
    for key in ('out', 'act', 'act_k', 'w_nm', 'kernel', 'strides', 'dilation', 'padding'):
        sub[key] = _pick(d[key], index)
    c, h, w = in_shape
    if op == 'tc':
        if sub['dilation'] != 1 or sub['padding'] != 'SAME':
            raise NotImplementedError('{}/{}: dilation / VALID on a transposed conv are not restated'.format(layer_scope, op_name))
        ks = {'design': sub, 'scope': '{}/{}'.format(layer_scope, op_name), 'in_shape': list(in_shape),
              'kernel_shape': [sub['kernel'], sub['kernel'], sub['out'], c],
              'op_out_shape': [sub['out'], h * sub['strides'], w * sub['strides']]}
    else:
        ks = {'design': sub, 'scope': '{}/{}'.format(layer_scope, op_name), 'in_shape': list(in_shape),
              'kernel_shape': [sub['kernel'], sub['kernel'], c, sub['out']],
              'op_out_shape': [sub['out'], _same_out(h, sub['strides']), _same_out(w, sub['strides'])]}
    if sub['w_nm'] == 's':
        if sn_mode in ('sn_paper', 'PIM', 'pim'):
            # layer_func.py:811-814: the kernel flattened to [k*k*shape[2], shape[3]] (a 'tc' kernel's last axis is the layer's INPUT)
            num_in, num_out = int(np.prod(ks['kernel_shape'][:3])), ks['kernel_shape'][3]
            ks['use_u'] = num_in <= num_out
            ks['sn_x_shape'] = [1, num_in] if ks['use_u'] else [1, num_out]
            ks['pim'] = True
        else:
            ks['use_u'] = int(np.prod(in_shape)) <= int(np.prod(ks['op_out_shape']))
            if op == 'tc':                                   # math_func.py:512-528: the conv the layer is the transpose of
                ks['sn_x_shape'] = [1] + (list(ks['op_out_shape']) if ks['use_u'] else list(in_shape))
            else:
                ks['sn_x_shape'] = [1] + (list(in_shape) if ks['use_u'] else list(ks['op_out_shape']))
        if not isinstance(sub['act_k'], (float, int)) or sub['act_k'] is False:
            raise ValueError('{}: w_nm="s" needs a numeric act_k'.format(ks['scope']))
    return ks


def _build_res(s, d, shape, sn_mode):
    """Layer._add_layer_res_ (layer_func.py:1687-1771)"""
    sc, bn = s['scope'], d['act_nm'] in ('bn', 'BN')
    up = d['scale'] is not None and d['scale'][1] > 0
    down = d['scale'] is not None and d['scale'][1] < 0
    res = {'bn0': bn and d['type'] != 'res_v1', 'bn1': bn, 'up': up, 'down': down,
           'bias': d['bias'] is not None, 'bias_sc': True}        # :1745 "'bias' in self.design" holds for every design
    cur = list(shape)
    if up:
        cur = _scaled_shape(cur, d['scale'])
    op = d['op']                                             # 'tc' (:1725-1727): kernel_0 and kernel_sc transposed, kernel_1 a conv
    res['k0'] = _kernel_spec(sc, 'kernel_0', d, 0, cur, sn_mode, op)
    cur = res['k0']['op_out_shape']
    res['k1'] = _kernel_spec(sc, 'kernel_1', d, 1, cur, sn_mode)
    cur = res['k1']['op_out_shape']
    if down:
        cur = _scaled_shape(cur, d['scale'])
    sc_shape = list(shape)
    if d['type'] == 'res':
        if up:
            sc_shape = _scaled_shape(sc_shape, d['scale'])
        res['ksc'] = _kernel_spec(sc, 'kernel_sc', d, 2, sc_shape, sn_mode, op)
        sc_shape = res['ksc']['op_out_shape']
        if down:
            sc_shape = _scaled_shape(sc_shape, d['scale'])
    elif d['type'] == 'res_v1':
        if d['scale'] is not None:
            if not down:
                raise AttributeError('{}: res_v1 is only used with downsampling.'.format(sc))
            sc_shape = _scaled_shape(sc_shape, d['scale'])
        res['ksc'] = _kernel_spec(sc, 'kernel_sc', d, 2, sc_shape, sn_mode, op)
        sc_shape = res['ksc']['op_out_shape']
    assert sc_shape == cur, '{}: Resnet shape {} and shortcut shape {} do not match.'.format(sc, cur, sc_shape)
    s['res'] = res
    return list(cur)



def _same_out(size, stride):
    return -(-size // stride)          # math_func.py:172-193 ('SAME')


def _conv_geometry(d):
    """(stride, dilation, padding) of a conv design as the reference applies them: dilation is ignored when the stride is
    above 1 (layer_func.py:546-549)"""
    stride, dil = d['strides'], d.get('dilation', 1)
    if stride > 1 and dil > 1:
        dil = 1
    return stride, dil, d.get('padding', 'SAME')


def _conv_out(size, kernel, stride, dil, padding):
    """spatial_shape_after_conv, math_func.py:172-193"""
    if padding in ('same', 'SAME'):
        return -(-size // stride)
    return -(-(size - (kernel - 1) * dil) // stride)


def build_net(designs, input_shape, net_name, sn_mode='default'):
    """Net + Routine.add_input_layers/seq_links shape inference (layer_func.py:2118,2221,2349).

    input_shape excludes the batch dimension: [z] or [C,H,W].  Returns a list of spec dicts."""
    specs, shape = [], list(input_shape)
    for design in designs:
        d = layer_defaults(design)
        if d['in_reshape'] is not None:
            shape = list(d['in_reshape'])
        s = {'design': d, 'scope': '{}/{}'.format(net_name, d['name']), 'in_shape': list(shape)}
        if d['type'] in RES_TYPES or d['op'] == 'i':
            if d['type'] in RES_TYPES:
                out = _build_res(s, d, shape, sn_mode)
            else:                                            # identity kernel: BN / activation only (:1275, 1646-1685)
                if d['scale'] is not None:
                    raise NotImplementedError('scaling on an identity layer is not restated')
                out = list(shape)
            s['op_out_shape'] = list(out)
            if d['out_reshape'] is not None:
                out = list(d['out_reshape'])
            s['out_shape'] = list(out)
            specs.append(s)
            shape = list(out)
            continue
        if d['scale'] is not None:                           # layer_func.py:1627-1629: up-sampling precedes the kernel
            if d['op'] != 'c':
                raise NotImplementedError('{}: scaling is restated for conv layers'.format(s['scope']))
            if d['scale'][1] > 0:
                shape = _scaled_shape(shape, d['scale'])
                s['in_shape_scaled'] = list(shape)
        if d['op'] == 'd':                                   # layer_func.py:576-578
            assert len(shape) == 1, '{}: dense layer needs a flat input'.format(s['scope'])
            s['kernel_shape'] = [shape[0], d['out']]
            out = [d['out']]
        elif d['op'] == 'c':                                 # layer_func.py:579-589
            c, h, w = shape
            s['kernel_shape'] = [d['kernel'], d['kernel'], c, d['out']]
            st_, dil_, pad_ = _conv_geometry(d)
            out = [d['out'], _conv_out(h, d['kernel'], st_, dil_, pad_), _conv_out(w, d['kernel'], st_, dil_, pad_)]
        else:                                                # 'tc', layer_func.py:590-600
            if d.get('dilation', 1) != 1 or d.get('padding', 'SAME') != 'SAME':
                raise NotImplementedError('{}: dilation / VALID on a transposed conv are not restated'.format(s['scope']))
            c, h, w = shape
            s['kernel_shape'] = [d['kernel'], d['kernel'], d['out'], c]
            out = [d['out'], h * d['strides'], w * d['strides']]
        s['op_out_shape'] = list(out)
        if d['w_nm'] == 's':                                 # math_func.py:481-486, 512-528
            if d['op'] == 'd':
                use_u = shape[0] <= d['out']
                s['sn_x_shape'] = [1, shape[0]] if use_u else [1, d['out']]
            elif d['op'] in ('c', 'tc') and sn_mode in ('sn_paper', 'PIM', 'pim'):
                # layer_func.py:801, 811-814: the kernel flattened to [k*k*shape[2], shape[3]] goes through the dense routine
                # (a 'tc' kernel is [k, k, out, in]: its LAST axis is the layer's input channels)
                num_in, num_out = int(np.prod(s['kernel_shape'][:3])), s['kernel_shape'][3]
                use_u = num_in <= num_out
                s['sn_x_shape'] = [1, num_in] if use_u else [1, num_out]
                s['pim'] = True
            else:
                use_u = int(np.prod(shape)) <= int(np.prod(out))
                if d['op'] == 'c':
                    s['sn_x_shape'] = [1] + (list(shape) if use_u else list(out))
                else:
                    s['sn_x_shape'] = [1] + (list(out) if use_u else list(shape))
            s['use_u'] = use_u
            if not isinstance(d['act_k'], (float, int)) or d['act_k'] is False:
                # layer_func.py:835: act_k=False passes isinstance(int) and zeroes the kernel;
                # every shipped config sets act_k, and the build rejects the quirk
                raise ValueError('{}: w_nm="s" needs a numeric act_k'.format(s['scope']))
        if d['scale'] is not None and d['scale'][1] < 0:     # layer_func.py:1640-1642: down-sampling is the last op
            out = _scaled_shape(out, d['scale'])
        if d['out_reshape'] is not None:
            out = list(d['out_reshape'])
        s['out_shape'] = list(out)
        specs.append(s)
        shape = list(out)
    return specs


# ---------------------------------------------------------------------------
# initialisers (layer_func.py:14-80; TF variance_scaling fan rules)
# ---------------------------------------------------------------------------
def _trunc_normal(rng, shape, stddev):
    out = rng.randn(*shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.randn(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * stddev


def init_kernel(rng, shape, act):
    """weight_initializer 'default' mode (layer_func.py:27-52).  TF computes fan_in from
    shape[-2] and fan_out from shape[-1] times the receptive field, for every op (tc quirk)."""
    receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
    if act == 'relu':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / fan_in))
    if act == 'lrelu':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / 1.01 / fan_in))
    limit = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
    return rng.uniform(-limit, limit, size=shape)


def init_params(specs, rng, dtype=torch.float32):
    """all variables of one net, in TF creation order, reference names and layouts."""
    p = OrderedDict()
    def bn_vars(prefix, c):
        p[prefix + '/BN/gamma'] = torch.ones(c, dtype=dtype)
        p[prefix + '/BN/beta'] = torch.zeros(c, dtype=dtype)
        p[prefix + '/BN/moving_mean'] = torch.zeros(c, dtype=dtype)
        p[prefix + '/BN/moving_variance'] = torch.ones(c, dtype=dtype)

    def kernel_vars(ks, bias_name):
        kd = ks['design']
        p[ks['scope'] + '/kernel'] = torch.as_tensor(init_kernel(rng, ks['kernel_shape'], kd['act']), dtype=dtype)
        if kd['w_nm'] == 's':
            p[ks['scope'] + '/SN/in_rand'] = torch.as_tensor(_trunc_normal(rng, ks['sn_x_shape'], 1.0), dtype=dtype)
        if bias_name is not None:
            p[bias_name] = torch.as_tensor(_trunc_normal(rng, [kd['out']], 1e-5), dtype=dtype)

    for s in specs:
        d, sc = s['design'], s['scope']
        if 'res' in s:                                                # creation order of layer_func.py:1687-1771
            r = s['res']
            if r['bn0']:
                bn_vars(sc + '/BN_0', s['in_shape'][0])
            kernel_vars(r['k0'], sc + '/bias_0/bias' if r['bias'] else None)
            if r['bn1']:
                bn_vars(sc + '/BN_1', r['k0']['design']['out'])
            kernel_vars(r['k1'], sc + '/bias_1/bias' if r['bias'] else None)
            if 'ksc' in r:
                kernel_vars(r['ksc'], sc + '/bias_sc/bias')
            continue
        if d['op'] == 'i':
            if d['act_nm'] in ('bn', 'BN'):
                bn_vars(sc + '/BN', s['in_shape'][0])
            continue
        p[sc + '/kernel/kernel'] = torch.as_tensor(init_kernel(rng, s['kernel_shape'], d['act']), dtype=dtype)
        if d['w_nm'] == 's':                                          # math_func.py:565-567
            p[sc + '/kernel/SN/in_rand'] = torch.as_tensor(_trunc_normal(rng, s['sn_x_shape'], 1.0), dtype=dtype)
        if d['bias'] is not None:                                     # layer_func.py:745-747
            p[sc + '/bias/bias'] = torch.as_tensor(_trunc_normal(rng, [s['op_out_shape'][0]], 1e-5), dtype=dtype)
        if d['act_nm'] in ('bn', 'BN'):                               # layer_func.py:953-966
            c = s['op_out_shape'][0]
            p[sc + '/BN/BN/gamma'] = torch.ones(c, dtype=dtype)
            p[sc + '/BN/BN/beta'] = torch.zeros(c, dtype=dtype)
            p[sc + '/BN/BN/moving_mean'] = torch.zeros(c, dtype=dtype)
            p[sc + '/BN/BN/moving_variance'] = torch.ones(c, dtype=dtype)
    return p


def trainable_names(params):
    return [n for n in params if not (n.endswith('in_rand') or '/moving_' in n)]


# ---------------------------------------------------------------------------
# linear operators (layer_func.py:909-928; SURVEY A.4 semantics)
# ---------------------------------------------------------------------------
def conv2d_same(x, w_hwio, stride, dilation=1, padding='SAME'):
    """tf.nn.conv2d(x, W[k,k,Cin,Cout], padding, NCHW, dilations): cross-correlation; 'SAME': pad_before = total//2 on the
    dilated extent (k - 1) * dilation + 1; 'VALID': no padding (layer_func.py:912-916)."""
    k = (w_hwio.shape[0] - 1) * dilation + 1
    if padding in ('same', 'SAME'):
        def pads(size):
            total = max((_same_out(size, stride) - 1) * stride + k - size, 0)
            return total // 2, total - total // 2
        (pt, pb), (pl, pr) = pads(x.shape[2]), pads(x.shape[3])
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=stride, dilation=dilation)


def conv2d_transpose_same(v, w, out_hw, stride, padding='SAME'):
    """tf.nn.conv2d_transpose(v, W[k,k,Cout,Cin], padding): the input-gradient of conv2d_same (dilation 1)."""
    k = w.shape[0]
    oh, ow = out_hw

    def pads(size):
        if padding not in ('same', 'SAME'):
            return 0, 0
        total = max((_same_out(size, stride) - 1) * stride + k - size, 0)
        return total // 2, total - total // 2
    (pt, pb), (pl, pr) = pads(oh), pads(ow)
    full = F.conv_transpose2d(v, w.permute(3, 2, 0, 1), stride=stride)
    fh, fw = full.shape[2], full.shape[3]
    if fh < oh + pt + pb or fw < ow + pl + pr:
        full = F.pad(full, (0, max(ow + pl + pr - fw, 0), 0, max(oh + pt + pb - fh, 0)))
    return full[:, :, pt:pt + oh, pl:pl + ow]


# ---------------------------------------------------------------------------
# spectral normalisation (math_func.py:470-569, 661-672, 674-746)
# ---------------------------------------------------------------------------
def _l2(x):
    return torch.sqrt(torch.sum(x * x))                      # tf.norm(axis=None), math_func.py:651


def sn_power_iteration(w, x, spec):
    """one power-iteration step.  Returns (sigma, x_update).  sigma = ||F(x)|| from the
    PRE-update x (math_func.py:668-670); x is a constant (non-trainable variable)."""
    d = spec['design']
    if d['op'] == 'd' or spec.get('pim'):
        w = w.reshape(-1, w.shape[-1])                       # no-op for a dense kernel
        if 1 in w.shape:                                     # math_func.py:702-704
            return _l2(w), x
        if spec['use_u']:
            fwd, bwd = (lambda t: t @ w), (lambda t: t @ w.t())          # math_func.py:583-602
        else:
            fwd, bwd = (lambda t: t @ w.t()), (lambda t: t @ w)
    else:
        stride, dil, padding = _conv_geometry(d)
        if dil > 1:
            # math_func.py:613-616, 630-634: tf.nn.atrous_conv2d(_transpose) - NHWC-only ops fed the NCHW tensors of the
            # reference's default data format: not a defined computation, not restated
            raise NotImplementedError('{}: spectral norm on a dilated kernel'.format(spec['scope']))
        in_hw = spec.get('in_shape_scaled', spec['in_shape'])[1:] if d['op'] == 'c' else spec['op_out_shape'][1:]
        conv = lambda t: conv2d_same(t, w, stride, 1, padding)                    # math_func.py:604-619
        conv_t = lambda t: conv2d_transpose_same(t, w, in_hw, stride, padding)    # math_func.py:621-637
        fwd, bwd = (conv, conv_t) if spec['use_u'] else (conv_t, conv)   # math_func.py:527-528
    u = fwd(x)
    sigma = _l2(u)
    y = u / (sigma + EPSI)                                   # math_func.py:659
    xb = bwd(y)
    x_update = xb / (_l2(xb) + EPSI)
    return sigma, x_update.detach()


# ---------------------------------------------------------------------------
# forward pass of one net (layer_func.py:870-928, 946-966, 1646-1685, 2078-2100)
# ---------------------------------------------------------------------------
def _act(x, name, mask=None, audit=None):
    """mask (tests only, see net_forward): the SIGN DECISIONS of another evaluation of the same layer (bool tensor, True where that
    evaluation's output was positive).  relu / lrelu then take their slope from it instead of from x - the two differ only
    where a pre-activation lies within rounding of zero.  An fp32 run of this oracle forced to an fp32 kernel's masks is the
    floor the kernel's gradients are measured against: what fp32 arithmetic loses against fp64 under the SAME decisions.
    audit (a list): receives, per forced activation, (how many decisions differ from this evaluation's own, the largest
    |pre-activation| among those as a fraction of the layer's largest, the activation's element count) - the caller's proof
    that the forced decisions differ from the natural ones at knife edges only."""
    if mask is not None and name in ('relu', 'lrelu'):
        if audit is not None:
            differ = (x > 0) != mask
            n = int(differ.sum())
            audit.append((n, float(x.detach().abs()[differ].max() / x.detach().abs().max()) if n else 0.0, x.numel()))
        lo = 0.0 if name == 'relu' else LRELU_ALPHA
        return x * torch.where(mask, torch.ones((), dtype=x.dtype), torch.full((), lo, dtype=x.dtype))
    if name == 'linear':
        return x
    if name == 'relu':
        return torch.relu(x)
    if name == 'lrelu':
        return torch.where(x > 0, x, x * LRELU_ALPHA)        # layer_func.py:104-112
    if name == 'tanh':
        return torch.tanh(x)
    raise NotImplementedError('Function {} is not implemented.'.format(name))


def net_forward(specs, params, x, is_training=True, collect=None, masks=None):
    """returns (output, updates) - updates maps variable name -> new value (the UPDATE_OPS of
    graph_func.py:848: SN in_rand assignments and BN moving statistics)."""
    updates = OrderedDict()
    # masks (tests only): {'gen': [...], 'dis': [...]} - for every relu / lrelu this net evaluates, in evaluation order,
    # the sign decisions another evaluation took there (_act)
    mask_iter = iter(masks[specs[0]['scope'].split('/')[0]]) if masks is not None else None

    audit = masks.get('audit') if masks is not None else None

    def act(t, name):
        if collect is not None and name in ('relu', 'lrelu'):
            # graph-attached PRE-activations in evaluation order (oracle/make_golden.py measures and repairs the fixtures'
            # activation margin on them: a relu's exact-zero outputs hide how close to zero their inputs were)
            collect.setdefault('pre_acts', []).append(t)
        return _act(t, name, next(mask_iter) if (mask_iter is not None and name in ('relu', 'lrelu')) else None, audit)

    def batch_norm(t, prefix):                                # layer_func.py:953-966, SURVEY A.4
        axis_shape, dims = [1, -1, 1, 1], [0, 2, 3]
        if is_training:
            mean = t.mean(dim=dims)
            var = ((t - mean.reshape(axis_shape)) ** 2).mean(dim=dims)
            cnt = t.numel() // t.shape[1]
            var_u = var * (cnt / max(cnt - 1.0, 1.0))
            mm, mv = params[prefix + '/BN/moving_mean'], params[prefix + '/BN/moving_variance']
            updates[prefix + '/BN/moving_mean'] = mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            updates[prefix + '/BN/moving_variance'] = mv * BN_MOMENTUM + var_u.detach() * (1 - BN_MOMENTUM)
        else:
            mean, var = params[prefix + '/BN/moving_mean'], params[prefix + '/BN/moving_variance']
        t = (t - mean.reshape(axis_shape)) / torch.sqrt(var.reshape(axis_shape) + BN_EPS)
        return t * params[prefix + '/BN/gamma'].reshape(axis_shape) + params[prefix + '/BN/beta'].reshape(axis_shape)

    def conv_op(t, ks, bias_name):                            # ParametricOperation.apply, layer_func.py:870-950
        kd = ks['design']
        w = params[ks['scope'] + '/kernel']
        if kd['w_nm'] == 's':
            sigma, x_new = sn_power_iteration(w, params[ks['scope'] + '/SN/in_rand'], ks)
            updates[ks['scope'] + '/SN/in_rand'] = x_new
            w = w * (kd['act_k'] / sigma)
            if collect is not None:
                collect[ks['scope'] + '/sigma'] = sigma.detach()
        if kd['op'] == 'tc':                                  # layer_func.py:918-928
            t = conv2d_transpose_same(t, w, ks['op_out_shape'][1:], kd['strides'])
        else:
            t = conv2d_same(t, w, kd['strides'])
        if bias_name is not None:
            t = t + params[bias_name].reshape(1, -1, 1, 1)
        return t

    for s in specs:
        d, sc = s['design'], s['scope']
        n = x.shape[0]
        if d['in_reshape'] is not None:
            x = x.reshape([n] + list(d['in_reshape']))
        assert list(x.shape[1:]) == s['in_shape'], \
            '{}: the input shape {} does not match existed shape {}.'.format(sc, list(x.shape[1:]), s['in_shape'])
        if 'res' in s or d['op'] == 'i':
            if d['op'] == 'i':                                # identity kernel, then BN and activation (:1646-1685)
                if d['act_nm'] in ('bn', 'BN'):
                    x = batch_norm(x, sc + '/BN')
                x = act(x, d['act'])
            else:                                             # Layer._apply_layer_res_, layer_func.py:1773-1842
                r = s['res']
                res = x
                if d['type'] != 'res_v1':
                    if r['bn0']:
                        res = batch_norm(res, sc + '/BN_0')
                    res = act(res, d['act'])
                if r['up']:
                    res = _rescale(res, d['scale'])
                res = conv_op(res, r['k0'], sc + '/bias_0/bias' if r['bias'] else None)
                if r['bn1']:
                    res = batch_norm(res, sc + '/BN_1')
                res = act(res, d['act'])
                res = conv_op(res, r['k1'], sc + '/bias_1/bias' if r['bias'] else None)
                if r['down']:
                    res = _rescale(res, d['scale'])
                short = x
                if d['type'] == 'res':
                    if r['up']:
                        short = _rescale(short, d['scale'])
                    short = conv_op(short, r['ksc'], sc + '/bias_sc/bias')
                    if r['down']:
                        short = _rescale(short, d['scale'])
                elif d['type'] == 'res_v1':
                    if r['down']:
                        short = _rescale(short, d['scale'])
                    short = conv_op(short, r['ksc'], sc + '/bias_sc/bias')
                x = res + short
            if collect is not None:
                collect[sc + '/out'] = x.detach()
                collect[sc + '/out_live'] = x
            if d['out_reshape'] is not None:
                x = x.reshape([n] + list(d['out_reshape']))
            continue
        if d['scale'] is not None and d['scale'][1] > 0:     # layer_func.py:1654-1656
            x = _rescale(x, d['scale'])
        w = params[sc + '/kernel/kernel']
        if d['w_nm'] == 's':                                  # layer_func.py:884-887, 913
            sigma, x_new = sn_power_iteration(w, params[sc + '/kernel/SN/in_rand'], s)
            updates[sc + '/kernel/SN/in_rand'] = x_new
            w = w * (d['act_k'] / sigma)
            if collect is not None:
                collect[sc + '/sigma'] = sigma.detach()
        if d['op'] == 'd':
            x = x @ w
        elif d['op'] == 'c':
            x = conv2d_same(x, w, *_conv_geometry(d))
        else:
            x = conv2d_transpose_same(x, w, s['op_out_shape'][1:], d['strides'])
        if d['bias'] is not None:                             # layer_func.py:946-950
            b = params[sc + '/bias/bias']
            x = x + (b.reshape(1, -1, 1, 1) if x.dim() == 4 else b)
        if d['act_nm'] in ('bn', 'BN'):                       # layer_func.py:953-966, SURVEY A.4
            axis_shape = [1, -1, 1, 1] if x.dim() == 4 else [1, -1]
            dims = [0, 2, 3] if x.dim() == 4 else [0]
            if is_training:
                mean = x.mean(dim=dims)
                var = ((x - mean.reshape(axis_shape)) ** 2).mean(dim=dims)
                cnt = x.numel() // x.shape[1]
                var_u = var * (cnt / max(cnt - 1.0, 1.0)) if x.dim() == 4 else var
                mm, mv = params[sc + '/BN/BN/moving_mean'], params[sc + '/BN/BN/moving_variance']
                updates[sc + '/BN/BN/moving_mean'] = (mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
                updates[sc + '/BN/BN/moving_variance'] = (mv * BN_MOMENTUM + var_u.detach() * (1 - BN_MOMENTUM))
            else:
                mean, var = params[sc + '/BN/BN/moving_mean'], params[sc + '/BN/BN/moving_variance']
            x = (x - mean.reshape(axis_shape)) / torch.sqrt(var.reshape(axis_shape) + BN_EPS)
            x = x * params[sc + '/BN/BN/gamma'].reshape(axis_shape) + params[sc + '/BN/BN/beta'].reshape(axis_shape)
        x = act(x, d['act'])
        if d['scale'] is not None and d['scale'][1] < 0:     # layer_func.py:1682-1684
            x = _rescale(x, d['scale'])
        if collect is not None:
            collect[sc + '/out'] = x.detach()
            collect[sc + '/out_live'] = x              # graph-attached (tests differentiate w.r.t. it)
        if d['out_reshape'] is not None:                      # C,H,W order flatten (my_test_cifar.py:36)
            x = x.reshape([n] + list(d['out_reshape']))
    return x, updates


# ---------------------------------------------------------------------------
# pairwise distances and the repulsive / bounded MMD losses
# ---------------------------------------------------------------------------
def get_squared_dist(x, y):
    """math_func.py:799-840, mode 'xxxyyy': Gram form, diag from the Gram matrix, clamp at 0."""
    xxt, xyt, yyt = x @ x.t(), x @ y.t(), y @ y.t()
    dx, dy = torch.diagonal(xxt), torch.diagonal(yyt)
    zero = torch.zeros((), dtype=x.dtype)
    dist_xx = torch.maximum(dx[:, None] - 2.0 * xxt + dx[None, :], zero)
    dist_xy = torch.maximum(dx[:, None] - 2.0 * xyt + dy[None, :], zero)
    dist_yy = torch.maximum(dy[:, None] - 2.0 * yyt + dy[None, :], zero)
    return dist_xx, dist_xy, dist_yy


def matrix_mean_wo_diagonal(m, num_row):
    """math_func.py:1064: (sum - trace)/(m(m-1)), also for the cross block."""
    return (torch.sum(m) - torch.sum(torch.diagonal(m))) / (num_row * (num_row - 1.0))


def mmd_g(dist_xx, dist_xy, dist_yy, batch_size, sigma=1.0, custom_weights=None):
    """math_func.py:1312-1343 (no bounds)."""
    k_xx = torch.exp(-dist_xx / (2.0 * sigma ** 2))
    k_yy = torch.exp(-dist_yy / (2.0 * sigma ** 2))
    k_xy = torch.exp(-dist_xy / (2.0 * sigma ** 2))
    m = float(batch_size)
    e_kxx, e_kxy, e_kyy = (matrix_mean_wo_diagonal(k, m) for k in (k_xx, k_xy, k_yy))
    stats = {'kxx': e_kxx, 'kxy': e_kxy, 'kyy': e_kyy}
    mmd1 = e_kxx + e_kyy - 2.0 * e_kxy
    if custom_weights is None:
        return mmd1, None, stats
    assert custom_weights[0] - custom_weights[1] == 1.0, 'w[0]-w[1] must be 1'
    mmd2 = custom_weights[0] * e_kxy - e_kxx - custom_weights[1] * e_kyy
    return mmd1, mmd2, stats


def mmd_g_bounded(dist_xx, dist_xy, dist_yy, batch_size, sigma=1.0, lower_bound=0.25, upper_bound=4.0,
                  custom_weights=(0.0, -1.0)):
    """math_func.py:1380-1422."""
    s2 = 2.0 * sigma ** 2
    lb = torch.tensor(lower_bound, dtype=dist_xx.dtype)
    ub = torch.tensor(upper_bound, dtype=dist_xx.dtype)
    k_xx, k_yy, k_xy = torch.exp(-dist_xx / s2), torch.exp(-dist_yy / s2), torch.exp(-dist_xy / s2)
    k_xx_b = torch.exp(-torch.maximum(dist_xx, lb) / s2)
    k_xy_b = torch.exp(-torch.minimum(dist_xy, ub) / s2) if custom_weights[0] > 0 else k_xy
    if custom_weights[1] > 0:
        k_yy_b = torch.exp(-torch.maximum(dist_yy, lb) / s2)
    else:
        k_yy_b = torch.exp(-torch.minimum(dist_yy, ub) / s2)
    m = float(batch_size)
    e_kxx, e_kxy, e_kyy = (matrix_mean_wo_diagonal(k, m) for k in (k_xx, k_xy, k_yy))
    e_kxx_b, e_kyy_b = matrix_mean_wo_diagonal(k_xx_b, m), matrix_mean_wo_diagonal(k_yy_b, m)
    e_kxy_b = matrix_mean_wo_diagonal(k_xy_b, m) if custom_weights[0] < 0 else e_kxy
    assert custom_weights[0] - custom_weights[1] == 1.0, 'w[0]-w[1] must be 1'
    mmd1 = e_kxx + e_kyy - 2.0 * e_kxy
    mmd2 = custom_weights[0] * e_kxy_b - e_kxx_b - custom_weights[1] * e_kyy_b
    stats = {'kxx': e_kxx, 'kxy': e_kxy, 'kyy': e_kyy, 'kxx_b': e_kxx_b, 'kyy_b': e_kyy_b}
    return mmd1, mmd2, stats


def gan_loss(score_gen, score_data, loss_type, batch_size, rep_weights=(0.0, -1.0)):
    """GANLoss.__call__ for the hot-path loss names (math_func.py:2556-2658, 2505-2550).
    x = generated scores, y = real scores (my_sngan.py:284-286).  Returns
    (loss_gen, loss_dis, stats)."""
    if loss_type in ('rep', 'rep_mmd_g'):
        d = get_squared_dist(score_gen, score_data)
        return mmd_g(*d, batch_size, sigma=1.0, custom_weights=list(rep_weights))
    if loss_type in ('rmb', 'rep_b', 'rep_mmd_b'):
        d = get_squared_dist(score_gen, score_data)
        return mmd_g_bounded(*d, batch_size, sigma=1.0, lower_bound=0.25, upper_bound=4.0,
                             custom_weights=list(rep_weights))
    # ---- the other in-kernel losses
    if loss_type in ('mmd_g', 'fixed_g'):                 # math_func.py:2160-2173 + mixture_mmd_g :1435-1462
        d = get_squared_dist(score_gen, score_data)
        total, stats = 0.0, {'kxx': 0.0, 'kxy': 0.0, 'kyy': 0.0}
        for sigma in MIXTURE_SIGMA:
            mmd_i, _, st = mmd_g(*d, batch_size, sigma=sigma)
            total = total + mmd_i
            stats = {k: stats[k] + st[k] for k in stats}
        return total, -total, stats
    if loss_type == 'mgb':                                # math_func.py:2175-2193
        d = get_squared_dist(score_gen, score_data)
        loss_gen, _, stats = mmd_g(*d, batch_size, sigma=1.0)
        mmd_b, st_b = mmd_g_clamped(*d, batch_size, sigma=1.0, upper_bound=4.0, lower_bound=0.25)
        stats.update(st_b)
        return loss_gen, -mmd_b, stats
    if loss_type == 'hinge':                              # math_func.py:2137-2143
        loss_dis = torch.relu(1.0 + score_gen).mean() + torch.relu(1.0 - score_data).mean()
        return (-score_gen).mean(), loss_dis, {}
    if loss_type in ('logistic', ''):                     # math_func.py:2128-2135 (non-saturating)
        sp = torch.nn.functional.softplus
        return sp(-score_gen).mean(), (sp(score_gen) + sp(-score_data)).mean(), {}
    raise NotImplementedError('Not implemented.')


MIXTURE_SIGMA = [1.0, math.sqrt(2.0), 2.0, math.sqrt(8.0), 4.0]      # math_func.py:2108

MIX_LOSSES = {'mmd_g_mix': 1.0, 'fixed_g_mix': 1.0, 'sgm': 0.2}       # default mix_threshold, math_func.py:2195, 2230


def mix_groups(mix_indices):
    """slice_pairwise_distance with indices (math_func.py:2052-2053): over the 2B rows [gen ; data], group 1 takes
    gen_i where the coin says 'original' and data_i where it does not, group 2 the complement."""
    return torch.cat([mix_indices, ~mix_indices]), torch.cat([~mix_indices, mix_indices])


def gan_loss_mix(score_gen, score_data, loss_type, batch_size, uni, state, mix_threshold=None,
                 loss_average_update=0.01, mix_prob_update=0.01):
    """GANLoss._mmd_g_mix_ / _single_mmd_g_mix_ (math_func.py:2195-2263) with get_mix_coin (:2061-2085):
      pair_dist over concat(score_gen, score_data) (:2203), loss_gen = the un-mixed (mixture) MMD (:2208-2210),
      mix_indices = uni > mix_prob (:2079-2080; `uni` is tf.random_uniform([B]) - an INPUT here),
      loss_dis = -MMD between the two mixed groups, boolean_mask row order (:2215-2220, mat_slice :377-378).
    state = (loss_average, mix_prob), the two non-trainable variables 'coin/gen_average' and 'coin/prob'; their
    UPDATE_OPS (:1999-2011, 2031-2033) give new_state - every read sees the pre-update values:
      loss_average' = (1 - rho) loss_average + rho loss_gen;  mix_prob' = clip(mix_prob + rho (loss_average - thr), 0, 0.5)
    Returns (loss_gen, loss_dis, info)."""
    if mix_threshold is None:
        mix_threshold = MIX_LOSSES[loss_type]
    sigmas = [1.0] if loss_type == 'sgm' else MIXTURE_SIGMA
    both = torch.cat([score_gen, score_data], 0)
    gram = both @ both.t()                                            # get_squared_dist mode 'xx', :801-805
    diag = torch.diagonal(gram)
    pair = torch.maximum(diag[:, None] - 2.0 * gram + diag[None, :], torch.zeros((), dtype=both.dtype))
    b = int(batch_size)

    def mmd_of(dxx, dxy, dyy):
        total = 0.0
        for sg in sigmas:
            total = total + mmd_g(dxx, dxy, dyy, batch_size, sigma=sg)[0]
        return total
    loss_gen = mmd_of(pair[:b, :b], pair[:b, b:], pair[b:, b:])       # slice_pairwise_distance without indices, :2047-2050
    loss_average, mix_prob = state
    dt = both.dtype
    mix_indices = torch.as_tensor(np.asarray(uni), dtype=dt) > torch.as_tensor(np.asarray(mix_prob), dtype=dt)
    g1, g2 = mix_groups(mix_indices)
    i1, i2 = torch.nonzero(g1).reshape(-1), torch.nonzero(g2).reshape(-1)
    loss_mix = mmd_of(pair[i1][:, i1], pair[i1][:, i2], pair[i2][:, i2])
    la, mp = float(loss_average), float(mix_prob)
    new_state = ((1.0 - loss_average_update) * la + loss_average_update * float(loss_gen.detach()),
                 min(max(mp + mix_prob_update * (la - mix_threshold), 0.0), 0.5))
    info = {'mix_indices': mix_indices, 'mix_group_1': g1, 'mix_group_2': g2, 'new_state': new_state}
    return loss_gen, -loss_mix, info


def mmd_g_clamped(dist_xx, dist_xy, dist_yy, batch_size, sigma=1.0, upper_bound=None, lower_bound=None):
    """mmd_g with its bounds arguments (math_func.py:1312-1322, 1324-1335): k_xx, k_yy from max(dist, lower_bound),
    k_xy from min(dist, upper_bound)."""
    s2 = 2.0 * sigma ** 2
    dxx = dist_xx if lower_bound is None else torch.clamp(dist_xx, min=lower_bound)
    dyy = dist_yy if lower_bound is None else torch.clamp(dist_yy, min=lower_bound)
    dxy = dist_xy if upper_bound is None else torch.clamp(dist_xy, max=upper_bound)
    m = float(batch_size)
    e_kxx, e_kxy, e_kyy = (matrix_mean_wo_diagonal(torch.exp(-t / s2), m) for t in (dxx, dxy, dyy))
    return e_kxx + e_kyy - 2.0 * e_kxy, {'kxx_b': e_kxx, 'kxy_b': e_kxy, 'kyy_b': e_kyy}


def mmd_masks(score_gen, score_data, lower_bound=0.25, upper_bound=4.0):
    """the bit-exact index masks of SURVEY A.3: clamp-active sets of the rmb loss."""
    dxx, dxy, dyy = get_squared_dist(score_gen, score_data)
    return {'xx_lt_lb': dxx < lower_bound, 'yy_gt_ub': dyy > upper_bound, 'xy_gt_ub': dxy > upper_bound}


# ---------------------------------------------------------------------------
# TF-Adam (graph_func.py:518-527; tf.train.AdamOptimizer semantics)
# ---------------------------------------------------------------------------
class AdamTF:
    def __init__(self, names, params, lr, beta1=0.5, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = {n: torch.zeros_like(params[n]) for n in names}
        self.v = {n: torch.zeros_like(params[n]) for n in names}
        self.t = 0

    def apply(self, params, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for n, g in grads.items():
            self.m[n] = self.b1 * self.m[n] + (1.0 - self.b1) * g
            self.v[n] = self.b2 * self.v[n] + (1.0 - self.b2) * g * g
            params[n] = params[n] - lr_t * self.m[n] / (torch.sqrt(self.v[n]) + self.eps)


# ---------------------------------------------------------------------------
# one training step (my_sngan.py:271-323, 424-425; graph_func.py:851-854)
# ---------------------------------------------------------------------------
class OracleGan:
    """G + D + losses + two TF-Adam optimisers; one `step` = one sess.run of graph_func.py:853."""

    def __init__(self, architecture, loss_type='rep', lr_list=(5e-4, 2e-4), rep_weights=(0.0, -1.0),
                 seed=0, dtype=torch.float32, params=None, sn_mode='default', mix_threshold=None):
        self.arch, self.loss_type, self.rep_weights, self.dtype = architecture, loss_type, rep_weights, dtype
        self.code_size = architecture['code'][0][0]
        self.gen_specs = build_net(architecture['generator'], [self.code_size], 'gen', sn_mode)
        self.dis_specs = build_net(architecture['discriminator'], list(architecture['input'][0]), 'dis', sn_mode)
        assert self.gen_specs[-1]['out_shape'] == list(architecture['input'][0])
        rng = np.random.RandomState(seed)
        self.params = OrderedDict()
        self.params.update(init_params(self.gen_specs, rng, dtype))
        self.params.update(init_params(self.dis_specs, rng, dtype))
        if params is not None:
            for k, v in params.items():
                assert k in self.params and list(self.params[k].shape) == list(v.shape), k
                self.params[k] = torch.as_tensor(np.asarray(v), dtype=dtype).clone()
        names = trainable_names(self.params)
        self.dis_names = [n for n in names if n.startswith('dis')]      # my_sngan.py:301
        self.gen_names = [n for n in names if n.startswith('gen')]      # my_sngan.py:303
        self.opt_d = AdamTF(self.dis_names, self.params, lr_list[0])
        self.opt_g = AdamTF(self.gen_names, self.params, lr_list[1])
        self.global_step = 0
        self.mix_state, self.mix_threshold = (0.0, 0.0), mix_threshold     # 'coin/gen_average', 'coin/prob' (zeros_initializer)

    def set_adam_state(self, m, v, t):
        """resume both optimisers mid-run: first / second moments by variable name and the steps taken so far"""
        for opt in (self.opt_d, self.opt_g):
            for n in opt.m:
                opt.m[n] = torch.as_tensor(np.asarray(m[n]), dtype=self.dtype).clone()
                opt.v[n] = torch.as_tensor(np.asarray(v[n]), dtype=self.dtype).clone()
            opt.t = int(t)
        self.global_step = int(t)

    def forward_losses(self, z, real, collect=None, uni=None, masks=None):
        p = self.params
        gen, up_g = net_forward(self.gen_specs, p, z, True, collect, masks)
        dis_out, up_d = net_forward(self.dis_specs, p, torch.cat([real, gen], 0), True, collect, masks)   # my_sngan.py:278
        b = z.shape[0]
        s_x, s_gen = dis_out[:b], dis_out[b:]                                                    # my_sngan.py:279
        if self.loss_type in MIX_LOSSES:          # the coin's state is two more UPDATE_OPS variables
            assert uni is not None, 'the *_mix losses need this step\'s uniform draw'
            loss_gen, loss_dis, stats = gan_loss_mix(s_gen, s_x, self.loss_type, b, uni, self.mix_state, self.mix_threshold)
            self._pending_mix_state = stats['new_state']
        else:
            loss_gen, loss_dis, stats = gan_loss(s_gen, s_x, self.loss_type, b, self.rep_weights)
        updates = OrderedDict(up_g)
        updates.update(up_d)
        return loss_gen, loss_dis, stats, updates, (gen, s_x, s_gen)

    def grads(self, z, real, collect=None, uni=None, masks=None):
        leaves = {n: self.params[n].detach().clone().requires_grad_(True)
                  for n in self.dis_names + self.gen_names}
        saved = dict(self.params)
        self.params.update(leaves)
        try:
            loss_gen, loss_dis, stats, updates, aux = self.forward_losses(z, real, collect, uni, masks)
            gd = torch.autograd.grad(loss_dis, [leaves[n] for n in self.dis_names], retain_graph=True)
            gg = torch.autograd.grad(loss_gen, [leaves[n] for n in self.gen_names])
        finally:
            self.params.update(saved)
        return (loss_gen.detach(), loss_dis.detach(), stats, updates,
                dict(zip(self.dis_names, gd)), dict(zip(self.gen_names, gg)), aux)

    def step(self, z, real, uni=None, masks=None):
        """losses are pre-update values; D and G update simultaneously from one forward
        (SURVEY 3.1); all reads precede all writes for the UPDATE_OPS."""
        loss_gen, loss_dis, stats, updates, gd, gg, _ = self.grads(z, real, uni=uni, masks=masks)
        self.opt_d.apply(self.params, gd)
        self.opt_g.apply(self.params, gg)
        for n, v in updates.items():
            self.params[n] = v
        if self.loss_type in MIX_LOSSES:
            self.mix_state = self._pending_mix_state
        self.global_step += 1                                  # my_sngan.py:424
        return float(loss_gen), float(loss_dis)


# ---------------------------------------------------------------------------
# eval helpers - NumPy in the reference too
# ---------------------------------------------------------------------------
def mean_cov(x):
    """math_func.py:56-67: column means and the unbiased covariance of a 2-D array"""
    x = np.asarray(x)
    mu = x.mean(axis=0)
    xc = x - mu
    return mu, xc.T @ xc / (x.shape[0] - 1.0)


def sqrt_sym_mat(mat, eps=EPSI):
    """math_func.py:2671-2683: U diag(sqrt(s)) V^T from an SVD, singular values below eps cut to 0"""
    u, s, vh = np.linalg.svd(mat)
    return (u * np.where(s < eps, 0.0, np.sqrt(s))) @ vh


def trace_sqrt_product(cov1, cov2):
    """math_func.py:2686-2699: trace sqrt(cov1 cov2) = trace sqrt(sqrt(cov1) cov2 sqrt(cov1))"""
    r = sqrt_sym_mat(cov1)
    return np.trace(sqrt_sym_mat(r @ cov2 @ r))


def fid_from_pool3(x, y):
    """graph_func.py:1733-1745 (my_fid_from_pool3): either argument may be features [N,D] or a [mean, cov] pair"""
    mx, cx = x if isinstance(x, (list, tuple)) else mean_cov(x)
    my, cy = y if isinstance(y, (list, tuple)) else mean_cov(y)
    return np.sum((mx - my) ** 2) + np.trace(cx) + np.trace(cy) - 2.0 * trace_sqrt_product(cx, cy)


def sprite_grid(images, mesh_num=None, if_invert=False):
    """graph_func.py:222-263 (write_sprite up to the file write): per-image min/max scaling to [0,1], optional
    inversion, zero padding to a square mesh when none is given, row-major tiling, uint8 truncation of x*255."""
    images = np.asarray(images)
    if images.ndim == 3:
        images = np.tile(images[..., None], (1, 1, 1, 3))
    if images.shape[3] == 1:
        images = np.tile(images, (1, 1, 1, 3))
    images = images.astype(np.float32)
    n = images.shape[0]
    images = images - images.reshape(n, -1).min(axis=1)[:, None, None, None]
    images = images / images.reshape(n, -1).max(axis=1)[:, None, None, None]
    if if_invert:
        images = 1 - images
    if mesh_num is None:
        side = int(np.ceil(np.sqrt(n)))
        mesh_num = (side, side)
        images = np.pad(images, ((0, side * side - n), (0, 0), (0, 0), (0, 0)), mode='constant', constant_values=0)
    rows, cols = tuple(mesh_num)
    h, w, c = images.shape[1:]
    grid = images.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)
    return (grid * 255).astype(np.uint8)
