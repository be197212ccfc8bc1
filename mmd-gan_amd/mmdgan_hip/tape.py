"""Engine for architectures with residual blocks (reference layer_func.py:1687-1842).

The DCGAN path (engine.py) is a hand-scheduled chain of fused launches; residual nets are not a chain - a block's
input feeds two branches, its pre-activation cannot ride on the producer's epilogue, scaling ops sit between the
convolutions.  Here every net is lowered to a short list of primitive ops over named values (dense, conv, bn+act,
act, resample up / down, periodic shuffle, bilinear resize, add, reshape), run forward and then backward in reverse order, each primitive calling the
same C-ABI kernels as the DCGAN path (MFMA / Winograd convs, GEMM, BN, spectral-norm helpers, fused MMD loss,
multi-tensor Adam) plus the block-specific elementwise kernels of csrc/resample.hip.  Same semantics as
GanEngine.step: one spectral-norm power iteration per SN kernel per step (every kernel of a block has its own,
layer_func.py:1415-1452), D sees [real ; fake], both gradients are taken before either update.

Layout: NHWC on the device; variables are exchanged in the reference's layouts (get_variables / set_variables).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import initializers, ops, settings
from .engine import _Arena, _chw_perm, _trunc_normal

_TEMPLATE = {'name': None, 'type': 'default', 'op': 'c', 'out': None, 'bias': 'b', 'act': 'linear', 'act_nm': None,
             'act_k': False, 'w_nm': None, 'w_p': None, 'kernel': 3, 'strides': 1, 'dilation': 1, 'padding': 'SAME',
             'scale': None, 'in_reshape': None, 'out_reshape': None, 'aux': None}
RES_TYPES = ('res', 'res_i', 'res_v1')                                   # layer_func.py:2062
_PIM = ('sn_paper', 'PIM', 'pim')


def needs_primitive_ops(designs):
    """a net the hand-scheduled chain does not express: residual blocks, identity layers, scaling ops, an input reshape,
    'VALID' padding, dilation"""
    def plain(v, default):
        return all(x == default for x in v) if isinstance(v, (list, tuple)) else v == default
    return any(d.get('type', 'default') in RES_TYPES or d.get('op') == 'i' or d.get('scale') is not None
               or d.get('in_reshape') is not None or not plain(d.get('dilation', 1), 1) or not plain(d.get('padding', 'SAME'), 'SAME')
               for d in designs)


def has_residual_blocks(architecture):
    return any(needs_primitive_ops(architecture[net]) for net in ('generator', 'discriminator'))


def needs_tape_engine(architecture):
    """architectures the hand-scheduled GanEngine does not cover: residual blocks / scaling / identity layers, and
    batch norm anywhere in the discriminator (GanEngine's D backward pass has no BN path and refuses those)"""
    return has_residual_blocks(architecture) or any(d.get('act_nm') in ('bn', 'BN') for d in architecture['discriminator'])


def _pick(value, index):                                                 # Layer._update_design_, layer_func.py:1380-1395
    return value[index] if isinstance(value, (list, tuple)) else value


class _Kernel:
    """one ParametricOperation (layer_func.py:480-1038): a dense or conv kernel, optional bias, optional SN"""

    def __init__(self, scope, op, kernel_shape, in_ref, out_ref, act, w_nm, act_k, bias_name, stride, sn_mode, out=None):
        self.scope, self.op, self.kernel_shape = scope, op, list(kernel_shape)
        self.in_ref, self.out_ref = list(in_ref), list(out_ref)          # reference shapes without batch
        self.act_init, self.bias_name, self.stride = act, bias_name, stride
        self.R = kernel_shape[0] if op in ('c', 'tc') else 1
        self.out = kernel_shape[-1] if out is None else out           # a tc kernel is [R, R, out, in]
        self.sn, self.act_k, self.pim = w_nm == 's', act_k, False
        self.fold = None                                                 # 'unpool' / 'avg': scaling folded into this conv
        self.dil, self.valid = 1, False                                  # the layer's 'dilation' / 'padding' keys (_conv)
        self.row_perm = self.col_perm = None                             # dense kernels at an NCHW <-> NHWC seam
        if w_nm not in (None, 's'):
            raise NotImplementedError('{}: {} method not implemented'.format(scope, w_nm))       # layer_func.py:824
        if self.sn:
            if act_k is False or not isinstance(act_k, (float, int)):
                raise ValueError('{}: w_nm="s" needs a numeric act_k'.format(scope))
            if op == 'd' or sn_mode in _PIM:                             # math_func.py:481-486, layer_func.py:801, 811-814
                num_in, num_out = int(np.prod(kernel_shape[:-1])), kernel_shape[-1]     # (a tc kernel is [R, R, out, in])
                self.pim = op in ('c', 'tc')
                self.use_u = num_in <= num_out
                self.sn_x_ref = [1, num_in] if self.use_u else [1, num_out]
            else:                                                        # math_func.py:512-528
                # 'tc': the iteration runs on the conv whose transpose the layer is - its input is the layer's output
                self.use_u = int(np.prod(in_ref)) <= int(np.prod(out_ref))
                cin, cout = (in_ref, out_ref) if op == 'c' else (out_ref, in_ref)
                self.sn_x_ref = [1] + (list(cin) if self.use_u else list(cout))

    @property
    def w_name(self):
        return self.scope + '/kernel'

    def sn_native(self):
        r = self.sn_x_ref
        return [1, r[2], r[3], r[1]] if len(r) == 4 else list(r)

    def sn_u_shape(self):
        if self.op == 'd' or self.pim:
            return [1, self.kernel_shape[-1]] if self.use_u else [1, int(np.prod(self.kernel_shape[:-1]))]
        cin, cout = (self.in_ref, self.out_ref) if self.op == 'c' else (self.out_ref, self.in_ref)
        src = cout if self.use_u else cin
        return [1, src[1], src[2], src[0]]


class _Net:
    """one net lowered to primitives; owns its variables (flat arenas) and state"""

    def __init__(self, designs, in_ref, name, device, rng, sn_mode, weight_init='default'):
        self.name, self.device, self.sn_mode = name, device, sn_mode
        self.weight_init = initializers.check_mode(weight_init)          # FLAGS.WEIGHT_INITIALIZER
        # fold a block's 'unpool' x2 / 'avg' /2 into the 3x3 conv next to it (one 4x4 stride-2 launch: 4 taps per pixel
        # instead of 9, no up-sampled tensor in HBM; ResNet-SN config 7.32 -> 6.57 ms per step); MMDGAN_TAPE_COMPOSE=0: two ops
        self.compose = settings.on('MMDGAN_TAPE_COMPOSE')
        self.prims, self.kernels, self.bns = [], [], []
        self._bn_perm = {}                                               # BN prefix -> feature permutation (see _allocate)
        self._nval = 1                                                   # value 0 is the net input
        self.shapes = {0: list(in_ref)}                                  # value id -> reference shape (no batch)
        cur = 0
        for design in designs:
            cur = self._lower(design, cur)
        self.out_val = cur
        self._allocate(rng)

    # ---- lowering ---------------------------------------------------------------------------------------------
    def _new(self, ref_shape):
        v = self._nval
        self._nval += 1
        self.shapes[v] = list(ref_shape)
        return v

    def _emit(self, kind, ins, out_shape, **attrs):
        out = self._new(out_shape)
        self.prims.append(dict(kind=kind, ins=list(ins), out=out, **attrs))
        return out

    def _conv(self, scope, opname, d, index, x, bias_name, fold=None, ref_hw=None):
        """fold = 'unpool': x is the block's value BEFORE its x2 up-sampling, fold = 'avg': the result is wanted AFTER
        its /2 pooling - the kernel keeps the reference's geometry (spectral norm, names) and the scaling op is folded
        into it (one 4x4 stride-2 launch, csrc/resample.hip:compose_kernel)"""
        c, h, w = self.shapes[x]
        run_hw = (h, w)
        if fold == 'unpool':
            h, w = 2 * h, 2 * w
        if ref_hw is not None:                           # a 1x1 conv moved across its block's scaling op: the kernel keeps
            h, w = ref_hw                                # the reference's geometry (spectral norm), the launch runs on x
        R, stride, out = _pick(d['kernel'], index), _pick(d['strides'], index), _pick(d['out'], index)
        dil, padding = _pick(d['dilation'], index), _pick(d['padding'], index)
        if stride > 1 and dil > 1:                                       # layer_func.py:546-549
            dil = 1
        if padding not in ('SAME', 'same', 'VALID', 'valid'):
            raise NotImplementedError('{}: padding {} is not known'.format(scope, padding))
        valid = padding in ('VALID', 'valid')
        if dil != 1 or valid:
            # the 'dilation' / 'padding' keys (layer_func.py:541-556, 912-916; shapes math_func.py:172-193): compositions around the
            # 'SAME' kernels (TapeEngine._gconv_*; include/mmdgan_hip.h: mmdgan_strided_slice, mmdgan_space_batch)
            if fold is not None or ref_hw is not None:
                raise NotImplementedError('{}: dilation / VALID inside a block with scaling are not built'.format(scope))
            if dil > 1 and R % 2 == 0:
                raise NotImplementedError('{}: dilation on an even kernel is not built'.format(scope))
            ext = (R - 1) * dil
            if valid and (h <= ext or w <= ext):
                raise AssertionError('{}: a {}x{} input is smaller than the dilated {}x{} kernel'.format(scope, h, w, ext + 1, ext + 1))
            out_ref = [out, -(-(h - ext) // stride), -(-(w - ext) // stride)] if valid else [out, -(-h // stride), -(-w // stride)]
            k = _Kernel('{}/{}'.format(scope, opname), 'c', [R, R, c, out], [c, h, w], out_ref, _pick(d['act'], index),
                        _pick(d['w_nm'], index), _pick(d['act_k'], index), bias_name, stride, self.sn_mode)
            k.dil, k.valid = dil, valid
            if k.sn and not k.pim and dil > 1:
                # math_func.py:613-616, 630-634: tf.nn.atrous_conv2d(_transpose), NHWC-only ops on the NCHW tensors of the reference's
                # default data format - not a defined computation
                raise NotImplementedError('{}: spectral norm (default mode) on a dilated kernel'.format(scope))
            self.kernels.append(k)
            return self._emit('gconv', [x], out_ref, k=k)
        out_ref = [out, -(-h // stride), -(-w // stride)]
        k = _Kernel('{}/{}'.format(scope, opname), 'c', [R, R, c, out], [c, h, w], out_ref, _pick(d['act'], index),
                    _pick(d['w_nm'], index), _pick(d['act_k'], index), bias_name, stride, self.sn_mode)
        self.kernels.append(k)
        if ref_hw is not None:
            assert R == 1 and stride == 1 and fold is None
            return self._emit('conv', [x], [out, run_hw[0], run_hw[1]], k=k)
        if fold is None:
            return self._emit('conv', [x], out_ref, k=k)
        assert R == 3 and stride == 1
        k.fold = fold
        if fold == 'unpool':
            return self._emit('upconv', [x], out_ref, k=k)
        return self._emit('convdown', [x], [out, out_ref[1] // 2, out_ref[2] // 2], k=k)

    def _tconv(self, scope, opname, d, index, x, bias_name):
        """a transposed-conv kernel [R, R, out, in] of a layer or block (layer_func.py:590-600, 918-928)"""
        c, h, w = self.shapes[x]
        R, stride, out = _pick(d['kernel'], index), _pick(d['strides'], index), _pick(d['out'], index)
        if _pick(d['dilation'], index) != 1 or _pick(d['padding'], index) != 'SAME':
            raise NotImplementedError('{}: dilation / VALID are not built'.format(scope))
        out_ref = [out, h * stride, w * stride]
        k = _Kernel('{}/{}'.format(scope, opname), 'tc', [R, R, out, c], [c, h, w], out_ref, _pick(d['act'], index),
                    _pick(d['w_nm'], index), _pick(d['act_k'], index), bias_name, stride, self.sn_mode, out=out)
        self.kernels.append(k)
        return self._emit('tconv', [x], out_ref, k=k)

    def _can_fold(self, d, index, x, method):
        c, h, w = self.shapes[x]
        return (self.compose and d['scale'] is not None and d['scale'][0] == method and abs(d['scale'][1]) == 2
                and _pick(d['kernel'], index) == 3 and _pick(d['strides'], index) == 1
                and (method == 'unpool' or (h % 2 == 0 and w % 2 == 0)))

    def _bn_act(self, prefix, x, act):
        self.bns.append((prefix, self.shapes[x][0]))
        return self._emit('bn', [x], self.shapes[x], prefix=prefix, act=act)

    def _scale(self, x, scale):                                          # ImageScaling, layer_func.py:1041-1176
        method, factor = scale
        c, h, w = self.shapes[x]
        if method == 'avg':
            if factor > 0:
                raise AttributeError('avg can only be used for downsampling')             # :1098-1099
            f = -factor
            if h % f or w % f:
                raise NotImplementedError('avg pooling of a size its window does not divide is not built')
            return self._emit('down', [x], [c, h // f, w // f], f=f)
        if method == 'unpool':
            if factor < 0:
                raise AttributeError('unpool can only be used for upsampling')            # :1100-1101
            if factor != 2:
                raise AttributeError('unpool can only deal with factor = 2')              # :1102-1103
            return self._emit('up', [x], [c, h * 2, w * 2], f=2)
        if method == 'max':                                              # tf.nn.max_pool, :1149-1153
            if factor > 0:
                raise AttributeError('max can only be used for downsampling')             # :1098-1099
            f = -factor
            if h % f or w % f:
                raise NotImplementedError('max pooling of a size its window does not divide is not built')
            return self._emit('maxpool', [x], [c, h // f, w // f], f=f)
        if method in ('bil', 'bic'):                                     # tf.image.resize_bilinear / _bicubic, :1128-1147
            nh, nw = (int(h * factor), int(w * factor)) if factor > 0 else (int(-h / factor), int(-w / factor))
            return self._emit('bilinear' if method == 'bil' else 'bicubic', [x], [c, nh, nw])
        if method == 'ps':                                               # periodic shuffling, :1125-1127 / :197-244
            f = abs(int(factor))
            if factor > 0:
                if c % (f * f):
                    raise AssertionError('periodic shuffling needs a channel count divisible by {}'.format(f * f))
                return self._emit('shuffle', [x], [c // (f * f), h * f, w * f], f=f, to_big=True)
            if h % f or w % f:
                raise AssertionError('periodic shuffling needs a size divisible by {}'.format(f))
            return self._emit('shuffle', [x], [c * f * f, h // f, w // f], f=f, to_big=False)
        raise NotImplementedError('Method {} not implemented.'.format(method))            # :1165-1167

    def _lower(self, design, x):
        d = dict(_TEMPLATE)
        d.update(design)
        if d['act_nm'] in ('bn', 'BN') and d['bias'] in ('b', 'bias'):                    # layer_func.py:1241-1242
            d['bias'] = None
        if d['op'] == 'tc':
            d['scale'] = None                                                             # :1245-1247
        scope = '{}/{}'.format(self.name, d['name'])
        if d['act_nm'] not in (None, 'bn', 'BN'):
            raise NotImplementedError('{}: {} not implemented'.format(scope, d['act_nm']))
        if d['act'] not in ('linear', 'relu', 'lrelu', 'tanh'):
            raise NotImplementedError('Function {} is not implemented.'.format(d['act']))  # :149
        bn = d['act_nm'] in ('bn', 'BN')
        if d['in_reshape'] is not None:
            x = self._reshape(x, d['in_reshape'])
        if d['type'] in RES_TYPES:
            if d['op'] not in ('c', 'tc'):
                raise NotImplementedError('{}: residual blocks are built for op "c" and "tc"'.format(scope))
            y = self._lower_res(d, scope, x, bn)
        elif d['type'] != 'default':
            raise NotImplementedError('{}: {} is not implemented.'.format(scope, d['type']))   # :2067
        elif d['op'] == 'i':                                             # identity kernel, then BN / activation
            y = self._bn_act(scope + '/BN', x, d['act']) if bn else self._emit('act', [x], self.shapes[x], act=d['act'])
        elif d['op'] == 'tc':                                            # transposed conv (layer_func.py:590-600, 918-928)
            y = self._tconv(scope, 'kernel', d, None, x, scope + '/bias/bias' if d['bias'] is not None else None)
            if bn:
                y = self._bn_act(scope + '/BN', y, d['act'])
            elif d['act'] != 'linear':
                y = self._emit('act', [y], self.shapes[y], act=d['act'])
        elif d['op'] in ('d', 'c'):
            if d['scale'] is not None and d['scale'][1] > 0:             # :1627-1629
                x = self._scale(x, d['scale'])
            bias_name = scope + '/bias/bias' if d['bias'] is not None else None
            if d['op'] == 'd':
                assert len(self.shapes[x]) == 1, '{}: the input shape {} does not match a dense layer'.format(scope, self.shapes[x])
                k = _Kernel(scope + '/kernel', 'd', [self.shapes[x][0], d['out']], self.shapes[x], [d['out']], d['act'],
                            d['w_nm'], d['act_k'], bias_name, 1, self.sn_mode)
                self.kernels.append(k)
                y = self._emit('dense', [x], [d['out']], k=k)
            else:
                y = self._conv(scope, 'kernel', d, None, x, bias_name)
            if bn:
                y = self._bn_act(scope + '/BN', y, d['act'])
            elif d['act'] != 'linear':
                y = self._emit('act', [y], self.shapes[y], act=d['act'])
            if d['scale'] is not None and d['scale'][1] < 0:             # :1640-1642
                y = self._scale(y, d['scale'])
        else:
            raise AttributeError('layer op {} not supported.'.format(d['op']))                # layer_func.py:1275
        if d['out_reshape'] is not None:
            y = self._reshape(y, d['out_reshape'])
        return y

    def _reshape(self, x, ref_shape):
        ref_shape = list(ref_shape)
        assert int(np.prod(ref_shape)) == int(np.prod(self.shapes[x])), \
            'the output shape {} does not match existed shape {}.'.format(self.shapes[x], ref_shape)
        return self._emit('reshape', [x], ref_shape, src_shape=list(self.shapes[x]))

    def _lower_res(self, d, scope, x, bn):                               # layer_func.py:1687-1842
        act, typ = d['act'], d['type']
        up = d['scale'] is not None and d['scale'][1] > 0
        down = d['scale'] is not None and d['scale'][1] < 0
        bias = d['bias'] is not None
        r = x
        if typ != 'res_v1':
            if bn:
                r = self._bn_act(scope + '/BN_0', r, act)
            elif act != 'linear':
                r = self._emit('act', [r], self.shapes[r], act=act)
        # op 'tc' (:1725-1727): kernel_0 and kernel_sc are transposed convs - they ARE the up-sampling, 'scale' was dropped in
        # _lower (:1245-1247) - and kernel_1 stays a conv
        tc = d['op'] == 'tc'
        if tc:
            r = self._tconv(scope, 'kernel_0', d, 0, r, scope + '/bias_0/bias' if bias else None)
        elif up and self._can_fold(d, 0, r, 'unpool'):
            r = self._conv(scope, 'kernel_0', d, 0, r, scope + '/bias_0/bias' if bias else None, fold='unpool')
        else:
            if up:
                r = self._scale(r, d['scale'])
            r = self._conv(scope, 'kernel_0', d, 0, r, scope + '/bias_0/bias' if bias else None)
        if bn:
            r = self._bn_act(scope + '/BN_1', r, act)
        elif act != 'linear':
            r = self._emit('act', [r], self.shapes[r], act=act)
        if down and self._can_fold(d, 1, r, 'avg'):
            r = self._conv(scope, 'kernel_1', d, 1, r, scope + '/bias_1/bias' if bias else None, fold='avg')
        else:
            r = self._conv(scope, 'kernel_1', d, 1, r, scope + '/bias_1/bias' if bias else None)
            if down:
                r = self._scale(r, d['scale'])
        s = x
        if tc and typ in ('res', 'res_v1'):                              # (no scaling op on either side: 'scale' is None)
            s = self._tconv(scope, 'kernel_sc', d, 2, s, scope + '/bias_sc/bias')
        elif typ == 'res':
            # a 1x1 conv commutes with nearest-neighbour up-sampling (exactly) and with average pooling (to rounding):
            # it runs on the small side of the scaling op, a quarter of the work
            commute = (self.compose and _pick(d['kernel'], 2) == 1 and _pick(d['strides'], 2) == 1
                       and d['scale'] is not None and abs(d['scale'][1]) == 2)
            c0, h0, w0 = self.shapes[s]
            if up and commute and d['scale'][0] == 'unpool':
                s = self._conv(scope, 'kernel_sc', d, 2, s, scope + '/bias_sc/bias', ref_hw=(2 * h0, 2 * w0))
                s = self._scale(s, d['scale'])
            elif down and commute and d['scale'][0] == 'avg' and h0 % 2 == 0 and w0 % 2 == 0:
                s = self._scale(s, d['scale'])
                s = self._conv(scope, 'kernel_sc', d, 2, s, scope + '/bias_sc/bias', ref_hw=(h0, w0))
            else:
                if up:
                    s = self._scale(s, d['scale'])
                s = self._conv(scope, 'kernel_sc', d, 2, s, scope + '/bias_sc/bias')      # :1745 keeps the bias
                if down:
                    s = self._scale(s, d['scale'])
        elif typ == 'res_v1':
            if d['scale'] is not None:
                if not down:
                    raise AttributeError('{}: res_v1 is only used with downsampling.'.format(scope))
                s = self._scale(s, d['scale'])
            s = self._conv(scope, 'kernel_sc', d, 2, s, scope + '/bias_sc/bias')
        assert self.shapes[s] == self.shapes[r], \
            '{}: Resnet shape {} and shortcut shape {} do not match.'.format(scope, self.shapes[r], self.shapes[s])
        return self._emit('add', [r, s], self.shapes[r])

    # ---- variables --------------------------------------------------------------------------------------------
    def _allocate(self, rng):
        # dense kernels next to a [C,H,W] <-> flat reshape carry the NCHW <-> NHWC permutation (as engine.py does)
        produced_by = {p['out']: p for p in self.prims}
        for p in self.prims:
            if p['kind'] == 'dense':
                src = produced_by.get(p['ins'][0])
                if src is not None and src['kind'] == 'reshape' and len(src['src_shape']) == 3:
                    p['k'].row_perm = _chw_perm(*src['src_shape'])
            if p['kind'] == 'reshape' and len(self.shapes[p['out']]) == 3:
                # the dense layer's columns are stored in NHWC order of the image they become; a per-feature batch
                # norm / activation in between (my_test_stl.py:12: dense -> BN -> relu -> [512,6,6]) works on the
                # permuted features unchanged - its gamma / beta / moving statistics carry the same permutation
                perm = _chw_perm(*self.shapes[p['out']])
                src = produced_by.get(p['ins'][0])
                while src is not None and src['kind'] in ('bn', 'act'):
                    if src['kind'] == 'bn':
                        self._bn_perm[src['prefix']] = perm
                    src = produced_by.get(src['ins'][0])
                if src is not None and src['kind'] == 'dense':
                    src['k'].col_perm = perm
                else:
                    raise NotImplementedError('a reshape to an image must follow a dense layer')
        entries = []
        for item in self._creation_order():
            if isinstance(item, _Kernel):
                entries.append((item.w_name, item.kernel_shape))
                if item.bias_name is not None:
                    entries.append((item.bias_name, [item.out]))
            else:
                prefix, c = item
                entries += [(prefix + '/BN/gamma', [c]), (prefix + '/BN/beta', [c])]
        self.arena = _Arena(entries, self.device)
        self.params = self.arena.flat
        self.grads, self.adam_m, self.adam_v = self.arena.like(), self.arena.like(), self.arena.like()
        self.state, self.sn = OrderedDict(), {}
        for item in self._creation_order():
            if isinstance(item, _Kernel):
                if item.sn:
                    self.state[item.scope + '/SN/in_rand'] = torch.zeros(item.sn_native(), device=self.device)
                    z = lambda *shape: torch.zeros(*shape, device=self.device)
                    self.sn[item.scope] = dict(sigma=z(1), scale=z(1), dot=z(1), xbn=z(1), dsigma=z(item.kernel_shape),
                                               u=z(item.sn_u_shape()), un=z(item.sn_u_shape()), xb=z(item.sn_native()))
            else:
                prefix, c = item
                self.state[prefix + '/BN/moving_mean'] = torch.zeros(c, device=self.device)
                self.state[prefix + '/BN/moving_variance'] = torch.ones(c, device=self.device)
        self.build_optimizer()
        self._kernel_by_name = {}
        for k in self.kernels:
            self._kernel_by_name[k.w_name] = k
            if k.bias_name is not None:
                self._kernel_by_name[k.bias_name] = k
            if k.sn:
                self._kernel_by_name[k.scope + '/SN/in_rand'] = k
        self.init_variables(rng)

    def build_optimizer(self):
        """TF-Adam over the arena, one segment per variable; a spectrally normalised kernel's segment carries the fix-up of
        its gradient  dL/dW = scale * G - (scale / sigma) * <G, W> * dsigma/dW  (SURVEY A.2): the arena holds the RAW G and
        Adam applies the fix-up as it reads it (engine.Network).  Called again by TapeEngine once the <G, W> scalars have
        moved into its zero-each-step scratch."""
        t = self.opt.step_counter.clone() if hasattr(self, 'opt') else None
        sn_of = {k.w_name: self.sn[k.scope] for k in self.kernels if k.sn}
        segments = [(off, size, ({q: sn_of[name][q] for q in ('dsigma', 'dot', 'sigma', 'scale')} if name in sn_of else None))
                    for name, (off, size, _) in self.arena.offsets.items()]
        self.opt = ops.AdamArena(self.params, self.grads, self.adam_m, self.adam_v, segments)
        if t is not None:
            self.opt.step_counter.copy_(t)

    def effective_grad(self, name):
        """the gradient of variable `name` as the optimiser uses it (native layout, a new tensor): the fix-up applied to
        the raw gradient of a spectrally normalised kernel when Adam folds it in"""
        g = self.g(name)
        k = self._kernel_by_name.get(name)
        if not (k is not None and k.sn and name == k.w_name and self.opt.fold_fixup):
            return g.clone()
        st = self.sn[k.scope]
        return st['scale'] * g - (st['scale'] / st['sigma']) * st['dot'] * st['dsigma'].view(g.shape)

    def effective_grads_flat(self):
        """the whole gradient arena with every fix-up applied (tests / inspection)"""
        out = self.grads.clone()
        for name in self.arena.offsets:
            self.arena.view(name, out).copy_(self.effective_grad(name))
        return out

    def _creation_order(self):
        """kernels and BN ops in the order the primitives use them (= the reference's variable creation order)"""
        for p in self.prims:
            if p['kind'] in ('dense', 'conv', 'gconv', 'tconv', 'upconv', 'convdown'):
                yield p['k']
            elif p['kind'] == 'bn':
                yield (p['prefix'], self.shapes[p['out']][0])

    def p(self, name):
        return self.arena.view(name)

    def g(self, name):
        return self.arena.view(name, self.grads)

    def variable_names(self, trainable_only=False):
        names = list(self.arena.offsets)
        return names if trainable_only else names + list(self.state)

    def _dense_vec_perm(self, name):
        """a bias / BN vector that lives on the output of a dense layer feeding an image reshape"""
        k = self._kernel_by_name.get(name)
        if k is not None and k.op == 'd' and name == k.bias_name:
            return k.col_perm
        if '/BN/' in name:
            return self._bn_perm.get(name[:name.rindex('/BN/')])
        return None

    def to_native(self, name, ref):
        ref = np.asarray(ref, dtype=np.float32)
        k = self._kernel_by_name.get(name)
        if name.endswith('/SN/in_rand'):
            if ref.ndim == 4:
                return np.ascontiguousarray(ref.transpose(0, 2, 3, 1))
            perm = k.row_perm if (k.op == 'd' and k.use_u) else (k.col_perm if k.op == 'd' else None)
            return ref[:, perm] if perm is not None else ref
        if k is not None and k.op == 'd' and name == k.w_name:
            if k.row_perm is not None:
                ref = ref[k.row_perm, :]
            if k.col_perm is not None:
                ref = ref[:, k.col_perm]
            return np.ascontiguousarray(ref)
        perm = self._dense_vec_perm(name)
        return np.ascontiguousarray(ref[perm]) if perm is not None else ref

    def to_ref(self, name, nat):
        nat = np.asarray(nat, dtype=np.float32)
        k = self._kernel_by_name.get(name)

        def unperm(a, perm, axis):
            out = np.empty_like(a)
            if axis == 0:
                out[perm] = a
            else:
                out[:, perm] = a
            return out
        if name.endswith('/SN/in_rand'):
            if nat.ndim == 4:
                return np.ascontiguousarray(nat.transpose(0, 3, 1, 2))
            perm = k.row_perm if (k.op == 'd' and k.use_u) else (k.col_perm if k.op == 'd' else None)
            return unperm(nat, perm, 1) if perm is not None else nat
        if k is not None and k.op == 'd' and name == k.w_name:
            if k.col_perm is not None:
                nat = unperm(nat, k.col_perm, 1)
            if k.row_perm is not None:
                nat = unperm(nat, k.row_perm, 0)
            return nat
        perm = self._dense_vec_perm(name)
        return unperm(nat, perm, 0) if perm is not None else nat

    def tensor(self, name, grad=False):
        if name in self.arena.offsets:
            return self.effective_grad(name) if grad else self.p(name)
        return self.state[name]

    def set_variable(self, name, value):
        t = self.tensor(name)
        t.copy_(torch.as_tensor(self.to_native(name, value)).reshape(t.shape))

    def get_variable(self, name, grad=False):
        return self.to_ref(name, self.tensor(name, grad).detach().cpu().numpy())

    def init_variables(self, rng):
        """the reference's initialisers (layer_func.py:27-52, 745-747; TF fan rule), reference layouts"""
        for item in self._creation_order():
            if not isinstance(item, _Kernel):
                self.set_variable(item[0] + '/BN/gamma', np.ones(item[1], np.float32))
                continue
            w = initializers.weight_initializer(rng, item.kernel_shape, item.act_init, self.weight_init)
            self.set_variable(item.w_name, w)
            if item.sn:
                self.set_variable(item.scope + '/SN/in_rand', _trunc_normal(rng, item.sn_x_ref, 1.0))
            if item.bias_name is not None:
                self.set_variable(item.bias_name, _trunc_normal(rng, [item.out], 1e-5))


class NetForward:
    """ONE lowered net run forward eagerly: Routine.__call__ for architectures with residual blocks, scaling ops, identity
    layers or an input reshape (layer_func.py:2043-2067 dispatch -> :1687-1842).  Spectral norms from one power-iteration
    step on the stored vectors (updated in training mode only: UPDATE_OPS), batch norm from batch / moving statistics."""

    def __init__(self, designs, in_ref, name, device, sn_mode='default', weight_init='default', rng=None):
        ops.require_device()
        self.device = torch.device(device)
        self.net = _Net(designs, list(in_ref), name, self.device, rng if rng is not None else np.random.RandomState(0),
                        sn_mode, weight_init)
        if ops._workspace is None:
            ops.set_workspace(device=self.device)
        # the forward executor is TapeEngine's, on an instance that holds nothing but buffers
        run = TapeEngine.__new__(TapeEngine)
        run.device, run._bufs, run._wino, run._in_step, run._sn_zeroed = self.device, {}, {}, False, False
        run._folded = {k.scope: [None, None] for k in self.net.kernels}
        run._graphs, run._fusions, run._out_buffer, run._ready_wait = {}, {}, None, None
        run._act_fus, run._fuse_act = {}, settings.on('MMDGAN_TAPE_FUSE_ACT')
        self._run = run

    def __call__(self, x_nhwc, is_training=False):
        for k in self.net.kernels:
            if k.sn:
                self._run._sn_step(self.net, k, update=bool(is_training))
        vals = self._run._forward(self.net, x_nhwc.contiguous(), bool(is_training), 'fwd%d' % x_nhwc.shape[0])
        return vals[self.net.out_val]


_EV_GEN_READY, _EV_DIS_READY, _EV_GEN_SN_READY = 0, 1, 2                # named events of the engine's handle


def _native(ref_shape, n):
    return [n, ref_shape[1], ref_shape[2], ref_shape[0]] if len(ref_shape) == 3 else [n, ref_shape[0]]


class TapeEngine:
    """G + D + loss + two TF-Adam optimisers for nets with residual blocks; same interface as GanEngine."""

    def __init__(self, architecture, loss_type='rep', lr_list=(5e-4, 2e-4), rep_weights=(0.0, -1.0), batch_size=64,
                 seed=0, device=None, dist_group=None, use_graph=False, sn_mode='default', weight_init='default',
                 mix_threshold=None, launch_mode=None, dp_backend=None):
        ops.require_device()
        initializers.check_mode(weight_init)
        if loss_type not in ops.LOSS:
            raise NotImplementedError('Not implemented.')                                # math_func.py:2651
        assert rep_weights[0] - rep_weights[1] == 1.0, 'w[0]-w[1] must be 1'              # math_func.py:1340
        if sn_mode not in ('default', 'PICO', 'pico') + _PIM:
            raise NotImplementedError('spectral norm mode {} is not implemented.'.format(sn_mode))
        self.device = torch.device(device if device is not None else 'cuda')
        self.arch, self.loss_type, self.rep_weights = architecture, loss_type, tuple(rep_weights)
        self.lr_d, self.lr_g = float(lr_list[0]), float(lr_list[1])
        self.B, self.sn_mode = int(batch_size), sn_mode
        self.code_size = architecture['code'][0][0]
        self.in_shape_ref = list(architecture['input'][0])
        rng = np.random.RandomState(seed)
        self.gen = _Net(architecture['generator'], [self.code_size], 'gen', self.device, rng, sn_mode, weight_init)
        self.dis = _Net(architecture['discriminator'], self.in_shape_ref, 'dis', self.device, rng, sn_mode, weight_init)
        assert self.gen.shapes[self.gen.out_val] == self.in_shape_ref, \
            'generator output {} does not match the input shape {}'.format(self.gen.shapes[self.gen.out_val], self.in_shape_ref)
        assert len(self.dis.shapes[self.dis.out_val]) == 1, 'the discriminator must end in a score vector'
        self.score_size = self.dis.shapes[self.dis.out_val][0]
        self.global_step = 0
        self.dist_group, self.world, self.rank = dist_group, 1, 0
        if dist_group is not None:
            import torch.distributed as tdist
            self.world = tdist.get_world_size(dist_group)
            self.rank = tdist.get_rank(dist_group)
        settings.warn_unknown()                                          # (a switch of an earlier round would be ignored in silence)
        self._dp_force = settings.on('MMDGAN_DP_FORCE')
        # who carries the gradient exchange (dist.choose_dp_backend, as GanEngine): the library's own RCCL communicator under
        # an nccl group - its collectives are plan nodes, so a data-parallel step replays from one C call - else torch.distributed
        from . import dist as mdist
        self._dp_backend = mdist.choose_dp_backend(dist_group, self.device, dp_backend)
        self.use_graph = False                                           # eager issue only
        if ops._workspace is None:
            ops.set_workspace(device=self.device)
        self.losses = torch.zeros(8, device=self.device)
        self._static_z = torch.zeros(self.B, self.code_size, device=self.device)
        self._z_gen = torch.Generator(device=self.device)                # per-replica code sampler, see GanEngine
        self._z_gen.manual_seed((int(seed) * 1000003 + 7919 * self.rank + 12345) % (2 ** 63 - 1))
        self._dis_in = torch.zeros(_native(self.in_shape_ref, 2 * self.B), device=self.device)
        self._static_real = self._dis_in[:self.B]                        # the batch buffer IS the real half of D's input
        # this engine's own library state (include/mmdgan_hip.h "Handles"): workspace, prezeroed mode, launch plan, events
        self._handle = ops.Handle(device=self.device)
        # how a step reaches the GPU: 'eager' = library calls from Python, 'plan' = the library records one eager step and
        # re-issues it from ONE C call (engine.py); MMDGAN_LAUNCH_MODE overrides
        self.launch_mode = launch_mode or settings.get('MMDGAN_LAUNCH_MODE') or 'eager'
        if self.launch_mode == 'graph':
            self.launch_mode = 'plan'                                    # (no hipGraph capture here; the plan is its equal)
        assert self.launch_mode in ('eager', 'plan'), self.launch_mode
        self._plan, self._plan_stream, self._baked_lr = None, None, (self.lr_d, self.lr_g)
        self._out_buffer = None
        self._bufs = {}
        self._in_step = False                                            # transformed weights are valid inside step() only
        self._sn_zeroed = False                                          # inside step(): the power iteration's targets are zeroed
        self._exchange_pending = False
        self._fuse_fanin = settings.on('MMDGAN_TAPE_FUSE_ADD')
        self._ready_wait = None
        self._early_d_adam = settings.on('MMDGAN_EARLY_D_ADAM')
        self._wgrad_defer = settings.on('MMDGAN_WGRAD_DEFER')
        self._wg_after = []
        self._fuse_act = settings.on('MMDGAN_TAPE_FUSE_ACT')
        self._act_fus = {}
        self._bn_resign = settings.on('MMDGAN_BN_RESIGN')
        self._d_has_bn = bool(self.dis.bns)
        lib = ops.require_device()
        # side streams on hardware queues of their own (streams.py): the power iterations run under G's forward pass,
        # weight / bias gradients beside the input-gradient chain; MMDGAN_TAPE_STREAMS=0 keeps everything on one stream
        self._side = settings.on('MMDGAN_TAPE_STREAMS')
        if self._side:
            from .streams import distinct_queue_streams
            self._wg_stream, self._sn_stream = distinct_queue_streams(2, self.device)
            self._wg_raw, self._sn_raw = self._wg_stream.cuda_stream, self._sn_stream.cuda_stream
        self._graphs, self._fusions = {}, {}
        # D without batch norm: its rows are independent, so loss_dis (2B rows) and loss_gen (the fake half again, B rows) go
        # back through it TOGETHER as 3B rows (one launch per primitive instead of two passes) - _backward(extra_rows=B)
        self._d_joint = (not self._d_has_bn and all(p['kind'] in self._ROW_WISE for p in self.dis.prims)
                         and settings.on('MMDGAN_TAPE_JOINT'))
        # Winograd-eligible convolutions get their weights transformed once per step, off the critical path, instead
        # of inside every call (forward, and up to two input-gradient passes) - which also keeps the library's
        # shared workspace out of every launch of the main stream.  kernel scope -> {(dgrad, batch): tensor}
        self._wino, self._wino_jobs = {}, {}
        self._folded = {}                              # kernel scope -> (4x4 kernel, its gradient buffer)
        rows_d = (2 * self.B,) if self._d_has_bn else (2 * self.B, 3 * self.B, self.B)
        for net, batches in ((self.gen, (self.B,)), (self.dis, rows_d)):
            for k in net.kernels:
                if k.fold is not None:
                    cin, cout = k.kernel_shape[2], k.out
                    shape = (4, 4, cin, cout) if k.fold == 'avg' else (4, 4, cout, cin)
                    self._folded[k.scope] = [torch.zeros(shape, device=self.device), None]
                if k.op not in ('c', 'tc') or not self._side or k.dil != 1 or k.valid:
                    continue
                R, stride = (4, 2) if k.fold is not None else (k.R, k.stride)
                as_conv = k.op == 'c' and k.fold != 'unpool'
                if as_conv:                            # (a folded 'avg' conv: same input, 4x4 stride 2)
                    c, h, w, kout = k.in_ref[0], k.in_ref[1], k.in_ref[2], k.out
                else:                                  # the conv whose input-gradient the tc layer is: its input is
                    c, h, w, kout = k.out, k.out_ref[1], k.out_ref[2], k.in_ref[0]     # the layer's OUTPUT
                table = {}
                for dgrad in (False, True):
                    by_algo = {}                       # one transformed tensor per (direction, algorithm): F(4x4,3x3) and
                    for n in batches:                  # F(2x2,3x3) lay the weights out differently (ops.wino_algo)
                        # the forward pass of a conv (= the backward pass of a tc layer) runs at one batch size only
                        single = (not dgrad) if as_conv else dgrad
                        if single and n != batches[0]:
                            continue
                        algo = ops.wino_algo(n, h, w, c, kout, R, stride, dgrad)
                        if not algo:
                            continue
                        if algo not in by_algo:
                            by_algo[algo] = ops.wino_alloc(algo, c, kout, dgrad, self.device)
                        table[(dgrad, n)] = by_algo[algo]
                if table:
                    self._wino[k.scope] = (net, k, table)
        # everything the backward pass accumulates into with atomics and that is not a gradient arena: one flat
        # scratch, zeroed once per step (the backward pass runs with mmdgan_set_outputs_prezeroed(1))
        sizes = []
        for net in (self.gen, self.dis):
            for k in net.kernels:                          # what the power iteration and the fix-up accumulate into
                if k.sn:
                    st = net.sn[k.scope]
                    sizes.append((st, 'dot', 4))
                    sizes += [(st, q, st[q].numel()) for q in ('dsigma', 'u', 'xb')]
            for i, p in enumerate(net.prims):
                if p['kind'] == 'bn':
                    c = net.shapes[p['out']][0]
                    sizes.append((p, '_ws_bwd', max(lib.mmdgan_bn_workspace_bytes(c) // 4, 4)))
                    sizes.append((p, '_ws_fwd', max(lib.mmdgan_bn_workspace_bytes(c) // 4, 4)))   # (one memset per BN layer otherwise)
                    if net is self.dis:                # the loss_gen pass through a D with batch norm: totals of its own
                        sizes += [(p, '_ws_bwd2', max(lib.mmdgan_bn_workspace_bytes(c) // 4, 4)), (p, '_gg2', c), (p, '_gb2', c)]
            for k in net.kernels:                          # weight gradients of the folded 4x4 kernels (atomics)
                if k.fold is not None:
                    sizes.append((self._folded[k.scope], 1, self._folded[k.scope][0].numel()))
        total = sum((n + 3) // 4 * 4 for _, _, n in sizes)
        self._zero_scratch = torch.zeros(max(total, 4), device=self.device)
        off = 0
        for holder, key, n in sizes:
            old = holder.get(key) if isinstance(holder, dict) else None
            if torch.is_tensor(old):                       # (a tensor the net allocated for itself: same shape, new home)
                holder[key] = self._zero_scratch[off:off + old.numel()].view(old.shape)
            else:
                holder[key] = self._zero_scratch[off:off + n]
            off += (n + 3) // 4 * 4
        for entry in self._folded.values():
            entry[1] = entry[1].view(entry[0].shape)
        # the power iterations of a whole net as a few launches (csrc/sn_chain.hip: every stage of all chains at once, 8 kernels
        # per group) instead of five per kernel; MMDGAN_SN_FUSED=0 for the per-kernel chains
        self._sn_chains = {}
        if settings.on('MMDGAN_SN_FUSED'):
            for net in (self.gen, self.dis):
                layers = [(k, self._sn_chain_layer(net, k)) for k in net.kernels if k.sn]
                self._sn_chains[id(net)] = (ops.SnChains([L for _, L in layers if L is not None], self.device),
                                            [k for k, L in layers if L is None])
        self._grad_buckets = {id(net): self._make_buckets(net) for net in (self.gen, self.dis)}
        for net in (self.gen, self.dis):               # (the <G, W> scalars have moved: segments again)
            net.build_optimizer()
            # a single replica folds the spectral-norm fix-up into Adam's read of the gradient; data-parallel replicas apply
            # it before their all-reduce (engine.py: replicas must stay bit-identical)
            net.opt.fold_fixup = not (self.dist_group is not None and (self.world > 1 or self._dp_force))
        self._loss = ops.GanLossLauncher(loss_type, self.rep_weights, self.B, self.score_size, self.device, mix_threshold)

    # ---- buffers ----------------------------------------------------------------------------------------------
    def _buf(self, key, shape, zero=False):
        t = self._bufs.get(key)
        if t is None or list(t.shape) != list(shape):
            t = torch.zeros(list(shape), device=self.device)
            self._bufs[key] = t
        if zero:
            ops.memset_zero(t)                           # (a library call - also on first use: a launch plan records THIS step)
        return t

    def _sn_chain_layer(self, net, k):
        """the power iteration of kernel `k` as ops.SnChains describes it (the tensors of _sn_step), or None for a kernel with a
        unit dimension (math_func.py:702-704)"""
        w, st = net.p(k.w_name), net.sn[k.scope]
        if (k.op == 'd' or k.pim) and 1 in (int(np.prod(w.shape[:-1])), w.shape[-1]):
            return None
        if not (k.op == 'd' or k.pim) and (k.dil != 1 or k.valid):      # a composition: its own chain of launches (_sn_step)
            return None
        L = dict(w=w, x=net.state[k.scope + '/SN/in_rand'], sigma=st['sigma'], scale=st['scale'], dsigma=st['dsigma'], u=st['u'],
                 un=st['un'], xb=st['xb'], xb_norm=st['xbn'], act_k=k.act_k)
        if k.op == 'd' or k.pim:
            L.update(form=2 if k.use_u else 3, C=int(np.prod(w.shape[:-1])), K=w.shape[-1])
        else:
            h, wd = (k.in_ref if k.op == 'c' else k.out_ref)[1:]         # input of the conv ('tc': the layer's OUTPUT)
            L.update(form=0 if k.use_u else 1, H=h, W=wd, C=w.shape[2], K=w.shape[3], R=k.R, stride=k.stride)
        return L

    def _sn_all(self, net):
        """one power-iteration step of every spectrally normalised kernel of `net` (a training step's UPDATE_OPS)"""
        fused = self._sn_chains.get(id(net))
        if fused is None:
            for k in net.kernels:
                if k.sn:
                    self._sn_step(net, k)
            return
        fused[0].run(update=True)
        for k in fused[1]:
            self._sn_step(net, k)

    # ---- conv layers with 'padding': 'VALID' and / or 'dilation' > 1 (layer_func.py:541-556, 912-916) ------------------
    # compositions around the 'SAME' kernels: dilation d (stride 1) = the 'SAME' conv of each of the d*d phase images (space <->
    # batch, zero rows where the size is no multiple of d); 'VALID' = the stride-1 'SAME' result sampled at off + p * stride
    # with off = d * ((R - 1) // 2); the gradients take the adjoint ops.  key: buffer key prefix of the caller.
    def _gconv_slice(self, k):
        return k.dil * ((k.R - 1) // 2), k.stride, (k.out_ref[1], k.out_ref[2])

    def _gconv_fwd(self, k, a, w, key, bias=None, scale=None, out=None):
        n, h, wd = a.shape[0], a.shape[1], a.shape[2]
        d, K = k.dil, w.shape[3]
        s_eff = 1 if k.valid else k.stride
        xin = ops.space_batch(a, d, out=self._buf(key + ('xb',), [n * d * d, -(-h // d), -(-wd // d), a.shape[3]])) if d > 1 else a
        hf, wf = -(-xin.shape[1] // s_eff), -(-xin.shape[2] // s_eff)
        last = d == 1 and not k.valid
        full = out if last else self._buf(key + ('full',), [xin.shape[0], hf, wf, K])
        if n == 1 and self._sn_zeroed:                   # a batch-1 launch may split into an output it expects zeroed
            ops.memset_zero(full)
        ops.conv2d_fwd(xin, w, s_eff, bias=bias, scale=scale, out=full)
        if d > 1:
            full = ops.space_batch(full, d, hw=(h, wd), out=out if not k.valid else self._buf(key + ('unb',), [n, h, wd, K]))
        if k.valid:
            off, step, (P, Q) = self._gconv_slice(k)
            full = ops.strided_slice(full, off, step, (P, Q), out=out)
        return full

    def _gconv_dy(self, k, dy, in_hw, key):
        """the output gradient as the kernel launches see it: un-sliced ('VALID'), split into phase images (dilation)"""
        h, wd = in_hw
        if k.valid:
            off, step, _ = self._gconv_slice(k)
            dy = ops.strided_slice(dy, off, step, None, adjoint_hw=(h, wd), out=self._buf(key + ('dyf',), [dy.shape[0], h, wd, dy.shape[3]]))
        if k.dil > 1:
            d = k.dil
            dy = ops.space_batch(dy, d, out=self._buf(key + ('dyb',), [dy.shape[0] * d * d, -(-h // d), -(-wd // d), dy.shape[3]]))
        return dy

    def _gconv_dgrad(self, k, dy, w, in_hw, key, scale=None, out=None, dy_ready=None):
        h, wd = in_hw
        n, d, C = dy.shape[0], k.dil, w.shape[2]
        s_eff = 1 if k.valid else k.stride
        dyk = dy_ready if dy_ready is not None else self._gconv_dy(k, dy, in_hw, key)
        hd, wdd = -(-h // d), -(-wd // d)
        dxb = out if d == 1 else self._buf(key + ('dxb',), [n * d * d, hd, wdd, C])
        if n == 1 and self._sn_zeroed:
            ops.memset_zero(dxb)
        ops.conv2d_dgrad(dyk, w, (hd, wdd), s_eff, scale=scale, out=dxb)
        return ops.space_batch(dxb, d, hw=(h, wd), out=out) if d > 1 else dxb

    def _gconv_wgrad(self, k, a, dy, key, out, dbias=None, w=None, dot=None, dy_ready=None):
        h, wd, d = a.shape[1], a.shape[2], k.dil
        s_eff = 1 if k.valid else k.stride
        dyk = dy_ready if dy_ready is not None else self._gconv_dy(k, dy, (h, wd), key)
        xin = ops.space_batch(a, d, out=self._buf(key + ('xbw',), [a.shape[0] * d * d, -(-h // d), -(-wd // d), a.shape[3]])) if d > 1 else a
        ops.conv2d_wgrad(xin, dyk, k.R, s_eff, out=out, dbias=dbias, w=w, dot=dot)

    # ---- spectral norm (math_func.py:661-672), as engine.py:_sn_step -------------------------------------------
    def _sn_step(self, net, k, update=True):
        """update=False: sigma / scale from the stored vector only (inference: no UPDATE_OPS).  Inside step() the chain's
        accumulation targets (u, xb, dsigma) are zero on entry - the step's one scratch memset - and the launches may split"""
        st = net.sn[k.scope]
        oz = self._sn_zeroed
        w, x = net.p(k.w_name), net.state[k.scope + '/SN/in_rand']
        sigma, scale, dsig, u, un, xb, xbn = st['sigma'], st['scale'], st['dsigma'], st['u'], st['un'], st['xb'], st['xbn']
        if k.op == 'd' or k.pim:
            if k.pim:
                w, dsig = w.view(-1, w.shape[-1]), dsig.view(-1, w.shape[-1])
            if 1 in w.shape:                                             # math_func.py:702-704
                ops.sn_norm_scale(w.reshape(-1), k.act_k, sigma, scale, dsig.view(-1))
            elif k.use_u:
                ops.gemm(x, w, out=u, out_zeroed=oz)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    ops.gemm(x, un, trans_a=True, out=dsig, out_zeroed=oz)
                    ops.gemm(un, w, trans_b=True, out=xb, out_zeroed=oz)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
            else:
                ops.gemm(x, w, trans_b=True, out=u, out_zeroed=oz)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    ops.gemm(un, x, trans_a=True, out=dsig, out_zeroed=oz)
                    ops.gemm(un, w, out=xb, out_zeroed=oz)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
        elif k.dil != 1 or k.valid:                                      # a 'VALID' conv (dilated ones are refused at lowering)
            h, wd = k.in_ref[1:]
            key = ('sn', k.scope)
            if k.use_u:
                self._gconv_fwd(k, x, w, key + ('f',), out=u)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    self._gconv_wgrad(k, x, un, key + ('w',), out=dsig)
                    self._gconv_dgrad(k, un, w, (h, wd), key + ('b',), out=xb)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
            else:
                self._gconv_dgrad(k, x, w, (h, wd), key + ('f',), out=u)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    self._gconv_wgrad(k, un, x, key + ('w',), out=dsig)
                    self._gconv_fwd(k, un, w, key + ('b',), out=xb)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
        else:
            h, wd = (k.in_ref if k.op == 'c' else k.out_ref)[1:]         # input of the conv ('tc': the layer's OUTPUT)
            if k.use_u:
                ops.conv2d_fwd(x, w, k.stride, out=u)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    ops.conv2d_wgrad(x, un, k.R, k.stride, out=dsig)
                    ops.conv2d_dgrad(un, w, (h, wd), k.stride, out=xb)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
            else:
                ops.conv2d_dgrad(x, w, (h, wd), k.stride, out=u)
                ops.sn_norm_scale(u.view(-1), k.act_k, sigma, scale, un.view(-1))
                if update:
                    ops.conv2d_wgrad(un, x, k.R, k.stride, out=dsig)
                    ops.conv2d_fwd(un, w, k.stride, out=xb)
                    ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
        return scale

    # ---- forward ----------------------------------------------------------------------------------------------
    def _buf_of(self, key, shape):
        ob = self._out_buffer
        if ob is not None and key[:2] == ob[0] and key[2] == ob[1]:      # the net's last primitive writes where its reader reads
            assert list(ob[2].shape) == list(shape)
            return ob[2]
        return self._buf(key, shape)

    def _forward(self, net, x, training, tag, out_buffer=None):
        """runs the net's primitives; returns the value table (kept for the backward pass when training).
        out_buffer: where the net's output goes (instead of a buffer of the engine's own)"""
        n = x.shape[0]
        vals = {0: x}
        lib = ops.require_device()
        last = max(i for i, p in enumerate(net.prims) if p['out'] == net.out_val)
        self._out_buffer = ((tag, net.name), last, out_buffer) if (out_buffer is not None and net.prims[last]['kind'] != 'reshape') else None
        fused_addend, fused_add = self._add_fusions(net)
        act_after = self._act_fusions(net)               # value id -> the activation that is its only reader
        acted = set()                                    # values that were written ACTIVATED by their producer's epilogue
        for i, p in enumerate(net.prims):
            kind, a = p['kind'], vals[p['ins'][0]]
            key = (tag, net.name, i)
            out_shape = _native(net.shapes[p['out']], n)
            addend = vals[fused_addend[i]] if i in fused_addend else None     # a branch sum riding on this launch
            # an activation that is the only reader of this convolution / dense product rides on its epilogue (no pass of its own;
            # relu / lrelu / tanh are all differentiated from their OUTPUT, which is what the value table then holds)
            fact, okey = 'linear', key
            if kind in ('dense', 'conv', 'upconv', 'convdown', 'tconv') and addend is None and p['out'] in act_after and self._fuse_act:
                fact, j = act_after[p['out']]
                okey = (tag, net.name, j)                # the activation's buffer (the net's output buffer, if it is the last primitive)
                acted.add(p['out'])
            if self._ready_wait is not None and kind in ('conv', 'gconv', 'upconv', 'convdown', 'tconv'):
                ops.event_wait(self._ready_wait, ops._stream())          # this step's composed / transformed weights (_step_body)
                self._ready_wait = None
            if kind == 'reshape':
                y = a.reshape(out_shape)
            elif kind in ('dense', 'conv'):
                k = p['k']
                scale = net.sn[k.scope]['scale'] if k.sn else None
                bias = net.p(k.bias_name) if k.bias_name is not None else None
                y = self._buf_of(okey, out_shape)
                if kind == 'dense':
                    ops.gemm(a.reshape(n, -1), net.p(k.w_name), bias=bias, scale=scale, act=fact, out=y)
                else:
                    ops.conv2d_fwd(a, net.p(k.w_name), k.stride, bias=bias, scale=scale, act=fact, out=y, addend=addend,
                                   wino=self._wino_of(k, False, n) if (training and self._in_step) else None)
            elif kind == 'gconv':                                        # 'VALID' padding and / or dilation: a composition
                k = p['k']
                scale = net.sn[k.scope]['scale'] if k.sn else None
                bias = net.p(k.bias_name) if k.bias_name is not None else None
                y = self._gconv_fwd(k, a, net.p(k.w_name), key, bias=bias, scale=scale, out=self._buf_of(key, out_shape))
            elif kind in ('upconv', 'convdown'):                         # a conv with its block's scaling op folded in
                k = p['k']
                scale = net.sn[k.scope]['scale'] if k.sn else None
                bias = net.p(k.bias_name) if k.bias_name is not None else None
                w4 = self._folded[k.scope][0]
                if not (training and self._in_step):                     # outside step(): compose on the spot
                    w4 = ops.compose_scaled_conv(net.p(k.w_name), k.fold)
                y = self._buf_of(okey, out_shape)
                if kind == 'convdown':
                    ops.conv2d_fwd(a, w4, 2, bias=bias, scale=scale, act=fact, out=y, addend=addend,
                                   wino=self._wino_of(k, False, n) if (training and self._in_step) else None)
                else:
                    ops.conv2d_dgrad(a, w4, (out_shape[1], out_shape[2]), 2, bias=bias, scale=scale, act=fact, out=y, addend=addend,
                                     wino=self._wino_of(k, True, n) if (training and self._in_step) else None)
            elif kind == 'tconv':                                        # y = the input-gradient of a conv with kernel w
                k = p['k']
                bias = net.p(k.bias_name) if k.bias_name is not None else None
                scale = net.sn[k.scope]['scale'] if k.sn else None
                y = self._buf_of(okey, out_shape)
                ops.conv2d_dgrad(a, net.p(k.w_name), (out_shape[1], out_shape[2]), k.stride, bias=bias, scale=scale, act=fact, out=y,
                                 addend=addend, wino=self._wino_of(k, True, n) if (training and self._in_step) else None)
            elif kind == 'bn':
                y = self._buf_of(key, out_shape)
                c = out_shape[-1]
                pre = p['prefix']
                gamma, beta = net.p(pre + '/BN/gamma'), net.p(pre + '/BN/beta')
                mm, mv = net.state[pre + '/BN/moving_mean'], net.state[pre + '/BN/moving_variance']
                x2, y2 = a.reshape(-1, c), y.view(-1, c)
                if training:
                    mean, invstd = self._buf(key + ('mean',), [c]), self._buf(key + ('invstd',), [c])
                    p['_saved'] = (mean, invstd)
                    # inside a step the totals lie in the scratch the step zeroed with its first launch (the entry's own
                    # memset was 5 us on the main stream in front of every batch norm: 9 of them in the ResNet-SN generator)
                    zeroed = self._in_step and torch.is_tensor(p.get('_ws_fwd'))
                    ws = p['_ws_fwd'] if zeroed else self._buf(key + ('ws',), [max(lib.mmdgan_bn_workspace_bytes(c) // 4, 4)])
                    if zeroed:
                        lib.mmdgan_set_outputs_prezeroed(1)
                    try:
                        ops.bn_fwd_train(x2, gamma, beta, mm, mv, act=p['act'], unbiased=a.dim() == 4, new_moving_mean=mm,
                                         new_moving_var=mv, out=y2, save_mean=mean, save_invstd=invstd, workspace=ws)
                    finally:
                        if zeroed:
                            lib.mmdgan_set_outputs_prezeroed(0)
                else:
                    ops.bn_fwd_infer(x2, gamma, beta, mm, mv, act=p['act'], out=y2)
            elif kind == 'act':
                if p['ins'][0] in acted:                                 # written activated by its producer's epilogue
                    y = a
                else:
                    y = ops.act_fwd(a, p['act'], out=self._buf_of(key, out_shape))
            elif kind == 'down':
                y = ops.resample_down(a, p['f'], out=self._buf_of(key, out_shape))
            elif kind == 'up':
                if addend is not None:                                   # += into the other term of the sum that follows
                    y = ops.resample_up(a, p['f'], out=addend, accumulate=True)
                else:
                    y = ops.resample_up(a, p['f'], out=self._buf_of(key, out_shape))
            elif kind == 'shuffle':
                y = ops.periodic_shuffle(a, p['f'], p['to_big'], out=self._buf_of(key, out_shape))
            elif kind in ('bilinear', 'bicubic'):
                y = (ops.bilinear_resize if kind == 'bilinear' else ops.bicubic_resize)(a, out_shape[1:3], out=self._buf_of(key, out_shape))
            elif kind == 'maxpool':
                y = ops.max_pool(a, p['f'], out=self._buf_of(key, out_shape))
            elif kind == 'add':
                if i in fused_add:                                       # the sum was formed by the launch of its second term
                    y = vals[fused_add[i]]
                else:
                    y = ops.axpby(a, vals[p['ins'][1]], out=self._buf_of(key, out_shape))
            else:
                raise AssertionError(kind)
            vals[p['out']] = y
        if out_buffer is not None and self._out_buffer is None:
            ops.copy(out_buffer, vals[net.out_val].reshape(out_buffer.shape))
        self._out_buffer = None
        return vals

    # ---- backward ---------------------------------------------------------------------------------------------
    _ROW_WISE = ('reshape', 'dense', 'conv', 'gconv', 'upconv', 'convdown', 'tconv', 'act', 'down', 'up', 'shuffle', 'add')

    def _act_fusions(self, net):
        """{value id: activation name} for the values whose ONLY reader is an activation primitive (not the net's output)"""
        f = self._act_fus.get(id(net))
        if f is None:
            producer, uses = self._graph_of(net)
            f = {}
            for q in net.prims:
                if q['kind'] == 'act' and uses.get(q['ins'][0], 0) == 1 and q['ins'][0] != 0 and q['ins'][0] != net.out_val:
                    f[q['ins'][0]] = (q['act'], net.prims.index(q))
            self._act_fus[id(net)] = f
        return f

    def _graph_of(self, net):
        """value id -> the primitive that produces it, value id -> number of consumers (the net's output counts as one)"""
        g = self._graphs.get(id(net))
        if g is None:
            producer = {p['out']: p for p in net.prims}
            uses = {}
            for p in net.prims:
                for v in p['ins']:
                    uses[v] = uses.get(v, 0) + 1
            uses[net.out_val] = uses.get(net.out_val, 0) + 1
            g = self._graphs[id(net)] = (producer, uses)
        return g

    def _add_fusions(self, net):
        """which branch sums (layer_func.py:1842) ride on the launch that produces their second term instead of an axpby pass:
        ({index of a conv-like / 'up' primitive: the value it adds in its epilogue}, {index of an 'add' primitive: the value
        whose buffer then already holds the sum}).  A term qualifies when nothing else reads it and the other term exists
        before its producer runs: the 1x1 shortcut conv of a down-sampling block (its epilogue adds the branch), the
        'unpool' of an up-sampling block's shortcut (accumulated into the branch's buffer), the last conv of a block with
        an identity shortcut (adds the block input).  MMDGAN_TAPE_FUSE_ADD=0: every sum as its own pass."""
        f = self._fusions.get(id(net))
        if f is None:
            producer, uses = self._graph_of(net)
            where = {p['out']: i for i, p in enumerate(net.prims)}
            addend, alias = {}, {}
            if settings.on('MMDGAN_TAPE_FUSE_ADD'):
                for i, p in enumerate(net.prims):
                    if p['kind'] != 'add' or p['out'] == net.out_val:
                        continue
                    a, b = p['ins']
                    for term, other in ((b, a), (a, b)):
                        q = producer.get(term)
                        if q is None or uses.get(term, 0) != 1 or where[term] in addend:
                            continue
                        if where.get(other, -1) > where[term]:
                            continue                     # the other term does not exist yet when `term` is produced
                        if q['kind'] in ('conv', 'convdown', 'upconv', 'tconv'):
                            addend[where[term]], alias[i] = other, term
                            break
                        # (in place into the OTHER term's buffer: only where the backward pass never reads that buffer as its
                        # producer's output - a convolution's; an activation's, a max pool's or a batch norm's output is read back)
                        qo = producer.get(other)
                        if q['kind'] == 'up' and uses.get(other, 0) == 1 and other != 0 and qo is not None and \
                                qo['kind'] in ('conv', 'convdown', 'upconv', 'tconv'):
                            addend[where[term]], alias[i] = other, other      # accumulated INTO the other term's buffer
                            break
            f = self._fusions[id(net)] = (addend, alias)
        return f

    def _sn_grad_tail(self, net, k, gw, w, scale, dot_done):
        """the spectral-norm fix-up of a kernel's gradient: <G, W> next to the RAW gradient; the fix-up itself
        is applied by Adam as it reads the gradient (net.opt.fold_fixup) - or here, before a data-parallel exchange"""
        st = net.sn[k.scope]
        if not dot_done:
            ops.dot(gw.view(-1), w.view(-1), out=st['dot'])
        if not net.opt.fold_fixup:
            ops.sn_wgrad_fixup(gw.view(-1), st['dsigma'].view(-1), st['dot'], st['sigma'], scale)

    def _on_wg_stream(self, fn):
        # every weight gradient goes to the side stream (the thin first / last layers' partial sums and the Winograd-domain
        # slabs use the library workspace; workspace_acquire keeps the two streams' halves apart)
        if self._side:
            ops.stream_wait(self._wg_raw, ops._stream())
            with torch.cuda.stream(self._wg_stream):
                fn()
        else:
            fn()

    def _backward(self, net, vals, dout, tag, rows=None, param_grads=True, need_input_grad=False, extra_rows=0):
        """gradients of one scalar through the net.
        rows = (lo, hi): only those batch rows carry gradient (the loss_gen pass through D touches the fake half only);
        parameter gradients are then not formed.
        extra_rows = B: `dout` has B rows MORE than the values - [the rows of the values ; the gradient of a second scalar
        w.r.t. the LAST B rows of the values] (D: [loss_dis rows (2B) ; loss_gen rows of the fake half (B)]), back-
        propagated together: one launch per primitive for both scalars, parameter gradients from the first rows only,
        the input gradient (need_input_grad) for the extra rows only.  Row-wise primitives only (_ROW_WISE)."""
        lib = ops.require_device()
        nv = vals[0].shape[0]
        lo, hi = rows if rows is not None else (0, nv)
        n = hi - lo + extra_rows
        sl = (lambda t: t[lo:hi]) if rows is not None else (lambda t: t)
        producer, uses = self._graph_of(net)
        grads = {net.out_val: dout}

        def give(v, g):
            # fan-in: a value with two consumers sums their gradients.  The sum goes to a buffer of its own - the
            # first gradient may be shared with another value (both inputs of an add receive the same tensor)
            if v == 0 and extra_rows and g.shape[0] == n:    # the net's input: the gradient of the extra rows only
                g = g[nv:]
            if v in grads:
                acc = self._buf((tag, net.name, 'acc', v), list(g.shape))
                ops.axpby(grads[v], g, out=acc)
                grads[v] = acc
            else:
                grads[v] = g

        def first_rows(t):                                   # the rows parameter gradients are formed from
            return t[:nv] if extra_rows else t
        for i in range(len(net.prims) - 1, -1, -1):
            p = net.prims[i]
            if p['out'] not in grads:
                continue
            kind, dy = p['kind'], grads.pop(p['out'])
            vin = p['ins'][0]
            if vin == 0 and not need_input_grad and kind not in ('dense', 'conv', 'gconv', 'tconv', 'upconv', 'convdown', 'bn'):
                continue
            key = (tag, net.name, i, 'd')
            a = sl(vals[vin])
            if kind in ('dense', 'conv', 'gconv', 'upconv', 'convdown', 'tconv'):
                k = p['k']
                w = net.p(k.w_name)
                scale = net.sn[k.scope]['scale'] if k.sn else None
                if param_grads:
                    self._on_wg_stream(lambda: self._param_grads(net, p, a, first_rows(dy), w, scale))
                    self._exchange(net, p)
                # the input-gradient.  An activation in front of this primitive that nothing else reads: its derivative rides
                # on the epilogue (dact_of = the activation's output), the gradient goes straight to the activation's input
                q = producer.get(vin)
                fuse = q is not None and q['kind'] == 'act' and uses.get(vin, 0) == 1 and vin not in grads and kind != 'gconv'
                tgt = q['ins'][0] if fuse else vin
                if tgt == 0 and not need_input_grad:
                    continue
                act, dact, dact_batch, dyx = 'linear', None, 0, dy
                if tgt == 0 and extra_rows:                  # the net's input: the extra rows only
                    dyx = dy[nv:]
                    dact = vals[vin][nv - extra_rows:] if fuse else None
                elif fuse:
                    dact = sl(vals[vin])
                    dact_batch = nv if extra_rows else 0     # rows beyond the values take the LAST rows' activations
                if fuse:
                    act = q['act']
                nx = dyx.shape[0]
                in_shape = [nx] + list(a.shape[1:])
                dx = self._buf(key, in_shape)
                # fan-in: the value this gradient goes to already holds one from another consumer (a block's input: shortcut
                # and branch) - the sum rides on this launch's epilogue (give() would spend an axpby pass on it)
                fan = None
                if (self._fuse_fanin and tgt != 0 and kind in ('conv', 'convdown', 'upconv', 'tconv') and tgt in grads
                        and list(grads[tgt].shape) == in_shape and grads[tgt].is_contiguous()):
                    fan = grads.pop(tgt)
                if kind == 'dense':
                    ops.gemm(dyx.reshape(nx, -1), w, trans_b=True, scale=scale, act=act,
                             dact_of=dact.reshape(dact.shape[0], -1) if dact is not None else None, dact_rows=dact_batch,
                             out=dx.view(nx, -1))
                elif kind == 'conv':
                    ops.conv2d_dgrad(dyx, w, (in_shape[1], in_shape[2]), k.stride, scale=scale, act=act, dact_of=dact,
                                     dact_batch=dact_batch, out=dx, addend=fan, wino=self._wino_of(k, True, nx))
                elif kind == 'gconv':                        # 'VALID' / dilated: the adjoint composition
                    self._gconv_dgrad(k, dyx, w, (in_shape[1], in_shape[2]), key + ('x',), scale=scale, out=dx)
                elif kind == 'convdown':
                    ops.conv2d_dgrad(dyx, self._folded[k.scope][0], (in_shape[1], in_shape[2]), 2, scale=scale, act=act,
                                     dact_of=dact, dact_batch=dact_batch, out=dx, addend=fan, wino=self._wino_of(k, True, nx))
                elif kind == 'upconv':
                    ops.conv2d_fwd(dyx, self._folded[k.scope][0], 2, scale=scale, act=act, dact_of=dact,
                                   dact_batch=dact_batch, out=dx, addend=fan, wino=self._wino_of(k, False, nx))
                else:                                        # tconv: d/dx of dgrad(x, W) = conv(dy, W)
                    ops.conv2d_fwd(dyx, w, k.stride, scale=scale, act=act, dact_of=dact, dact_batch=dact_batch, out=dx, addend=fan,
                                   wino=self._wino_of(k, False, nx))
                give(tgt, dx)
                continue
            in_shape = [dy.shape[0]] + list(a.shape[1:])
            if kind == 'reshape':
                give(vin, dy.reshape(in_shape))
            elif kind == 'bn':
                if rows is not None or extra_rows:
                    raise NotImplementedError('a row-restricted backward pass cannot cross batch norm')
                c = in_shape[-1]
                pre = p['prefix']
                mean, invstd = p['_saved']
                dx = self._buf(key, in_shape)
                y = vals[p['out']]
                if param_grads:                                          # zeroed with the step's arenas / scratch
                    ws, gg, gb = p['_ws_bwd'], net.g(pre + '/BN/gamma'), net.g(pre + '/BN/beta')
                else:                                                    # a second pass through the same op
                    ws, gg, gb = p['_ws_bwd2'], p['_gg2'], p['_gb2']
                resign = self._bn_resign and p['act'] in ('linear', 'relu', 'lrelu')      # (GanEngine._backward_gen)
                ops.bn_bwd(a.reshape(-1, c), None if resign else y.reshape(-1, c), dy.reshape(-1, c), net.p(pre + '/BN/gamma'), mean,
                           invstd, act=p['act'], dgamma=gg, dbeta=gb, out=dx.view(-1, c), workspace=ws,
                           beta=net.p(pre + '/BN/beta') if resign else None)
                if param_grads:
                    self._exchange(net, p)
                if vin != 0 or need_input_grad:
                    give(vin, dx)
            elif kind == 'act':
                y = sl(vals[p['out']])
                dx = self._buf(key, in_shape)
                if extra_rows:                                           # the extra rows: the last rows' activations
                    ops.act_bwd(dy[:nv], y, p['act'], out=dx[:nv])
                    ops.act_bwd(dy[nv:], y[nv - extra_rows:], p['act'], out=dx[nv:])
                else:
                    ops.act_bwd(dy, y, p['act'], out=dx)
                give(vin, dx)
            elif kind == 'down':                                         # d avg-pool: spread over the window
                f = p['f']
                give(vin, ops.resample_up(dy, f, scale=1.0 / (f * f), out=self._buf(key, in_shape)))
            elif kind == 'up':                                           # d unpool: sum over the window
                give(vin, ops.resample_down(dy, p['f'], scale=1.0, out=self._buf(key, in_shape)))
            elif kind == 'maxpool':                                      # dy to each window's first maximum
                give(vin, ops.max_pool(a.contiguous(), p['f'], dy=dy.contiguous(), out=self._buf(key, in_shape)))
            elif kind in ('bilinear', 'bicubic'):                        # scatter with the forward weights (atomics)
                grad_of = ops.bilinear_resize_grad if kind == 'bilinear' else ops.bicubic_resize_grad
                give(vin, grad_of(dy.contiguous(), in_shape[1:3], out=self._buf(key, in_shape, zero=True)))
            elif kind == 'shuffle':                                      # a permutation: its gradient is the inverse one
                give(vin, ops.periodic_shuffle(dy.contiguous(), p['f'], not p['to_big'], out=self._buf(key, in_shape)))
            elif kind == 'add':
                give(vin, dy)
                give(p['ins'][1], dy)
            else:
                raise AssertionError(kind)
        return grads.get(0)

    # ---- deferred slab reductions (mmdgan_wgrad_defer): a Winograd-domain weight gradient leaves its slabs to the prologue of
    # the NEXT weight-gradient launch of its stream.  What has to read the finished gradient (the adjoint of a folded scaling op,
    # a spectral-norm fix-up under data parallelism) is queued here and issued behind that next launch - or behind a flush.
    def _wgrad(self, *args, after=None, **kw):
        ops.conv2d_wgrad(*args, **kw)                    # (sums the previous call's slabs first, or they were flushed before it)
        self._wgrad_run_after()
        if after is not None:
            self._wg_after.append(after)

    def _wgrad_run_after(self):
        todo, self._wg_after = self._wg_after, []
        for fn in todo:
            fn()

    def _wgrad_flush(self):
        """every weight gradient issued so far is complete in stream order behind this call"""
        if self._wg_after or self._wgrad_defer:
            if self._side and self._wg_after:            # the queued readers belong on the weight-gradient stream
                with torch.cuda.stream(self._wg_stream):
                    ops.wgrad_flush()
                    self._wgrad_run_after()
            else:
                ops.wgrad_flush()
                self._wgrad_run_after()

    def _param_grads(self, net, p, a, dy, w, scale):
        """weight / bias gradient of one dense / conv-like primitive from its input `a` and output gradient `dy`"""
        kind, k = p['kind'], p['k']
        n = dy.shape[0]
        gw = net.g(k.w_name)
        gb = net.g(k.bias_name) if k.bias_name is not None else None
        dot = net.sn[k.scope]['dot'] if k.sn else None
        dot_done = False
        tail = (lambda: self._sn_grad_tail(net, k, gw, w, scale, True)) if (k.sn and not net.opt.fold_fixup) else None
        if kind == 'dense':
            if gb is not None:
                ops.colsum(dy.reshape(n, -1), out=gb)
            ops.gemm(a.reshape(n, -1), dy.reshape(n, -1), trans_a=True, out=gw)
        elif kind == 'conv':                                             # (<G, W> rides on the weight-gradient launch)
            self._wgrad(a, dy, k.R, k.stride, out=gw, dbias=gb, w=w if k.sn else None, dot=dot, after=tail)
            return
        elif kind == 'gconv':                                            # 'VALID' / dilated
            self._gconv_wgrad(k, a, dy, ('pg', net.name, k.scope), out=gw, dbias=gb, w=w if k.sn else None, dot=dot)
            self._wgrad_run_after()                                      # (the launch above carried or followed the previous reduction)
            if tail is not None:
                ops.wgrad_flush()
            dot_done = True
        elif kind == 'tconv':                                            # W[R,R,out,in]: roles swapped
            if gb is not None:
                ops.colsum(dy.reshape(-1, dy.shape[-1]), out=gb)
            self._wgrad(dy, a, k.R, k.stride, out=gw, w=w if k.sn else None, dot=dot, after=tail)
            return
        else:                                                            # a 4x4 kernel with the block's scaling op folded in
            g4 = self._folded[k.scope][1]

            def unfold():                                                # the adjoint of the fold reads the FINISHED 4x4 gradient
                ops.compose_scaled_conv_grad(g4, k.fold, out=gw)
                if k.sn:
                    self._sn_grad_tail(net, k, gw, w, scale, False)
            if kind == 'convdown':
                self._wgrad(a, dy, 4, 2, out=g4, dbias=gb, after=unfold)
            else:                                                        # transposed form: roles swapped
                if gb is not None:
                    ops.colsum(dy.reshape(-1, dy.shape[-1]), out=gb)
                self._wgrad(dy, a, 4, 2, out=g4, after=unfold)
            return
        if k.sn:
            self._sn_grad_tail(net, k, gw, w, scale, dot_done)

    def _wino_of(self, k, dgrad, n):
        entry = self._wino.get(k.scope)
        return entry[2].get((dgrad, n)) if entry is not None else None

    def _compose_weights(self, net):
        for k in net.kernels:                              # 4x4 kernels of the folded scaling ops
            if k.fold is not None:
                ops.compose_scaled_conv(net.p(k.w_name), k.fold, out=self._folded[k.scope][0])

    def _transform_weights(self, net):
        self._compose_weights(net)
        if id(net) not in self._wino_jobs:               # every transform of the net as ONE launch (ops.WinoTransforms)
            jobs = []
            for scope, (owner, k, table) in self._wino.items():
                if owner is not net:
                    continue
                done = set()
                src = self._folded[k.scope][0] if k.fold is not None else net.p(k.w_name)
                for (dgrad, _), u in table.items():
                    if u.data_ptr() not in done:
                        jobs.append((src, u, dgrad))
                        done.add(u.data_ptr())
            self._wino_jobs[id(net)] = ops.WinoTransforms(jobs)
        self._wino_jobs[id(net)].run()

    # ---- one training step -----------------------------------------------------------------------------------
    def generate(self, z, is_training=False):
        for k in self.gen.kernels:                       # inference: spectral norms from the stored vectors, not updated
            if k.sn:
                self._sn_step(self.gen, k, update=False)
        vals = self._forward(self.gen, z, is_training, 'gen%d' % z.shape[0])
        return vals[self.gen.out_val]

    def discriminate(self, x_nhwc):
        """D(x) in INFERENCE mode (my_sngan.py:558-560, `self.Dis(..., is_training=False)`): spectral norms from the stored
        power-iteration vectors without updating them, batch norm from its moving statistics; [n, d] scores"""
        for k in self.dis.kernels:
            if k.sn:
                self._sn_step(self.dis, k, update=False)
        vals = self._forward(self.dis, x_nhwc.contiguous(), False, 'score%d' % x_nhwc.shape[0])
        return vals[self.dis.out_val].clone()

    def _make_buckets(self, net):
        """the gradient arena of `net` cut into exchange buckets in BACKWARD order: [(lowest item index, start, end)]
        (dist.layer_buckets; an item = one kernel with its bias, or one batch norm's gamma / beta, in creation order = arena
        order = the reverse of the order the backward pass completes them in).  Every primitive that owns parameters
        learns its item index (p['_item'])."""
        from .dist import layer_buckets
        target = int(float(settings.get('MMDGAN_DP_BUCKET_MB')) * (1 << 20)) // 4
        ranges = []
        for p in net.prims:
            if p['kind'] in ('dense', 'conv', 'gconv', 'tconv', 'upconv', 'convdown'):
                names = [p['k'].w_name] + ([p['k'].bias_name] if p['k'].bias_name is not None else [])
            elif p['kind'] == 'bn':
                names = [p['prefix'] + '/BN/gamma', p['prefix'] + '/BN/beta']
            else:
                continue
            lo = min(net.arena.offsets[n][0] for n in names)
            hi = max(net.arena.offsets[n][0] + (net.arena.offsets[n][1] + 3) // 4 * 4 for n in names)
            p['_item'] = len(ranges)
            ranges.append((lo, hi))
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[0][0] == 0, 'arena entries are not in creation order'
        ranges[-1] = (ranges[-1][0], net.arena.size)
        return layer_buckets(ranges, target)

    def _exchange(self, net, p):
        """called right after the parameter gradients of primitive `p` have been issued: if that completes an exchange
        bucket, its SUM all-reduce starts now on the power-iteration stream (idle from D's forward pass to the next step, on
        a hardware queue of its own - engine.py:_exchange) and travels underneath the backward kernels still to come; only
        the last bucket of G is exposed.  Averaging is Adam's grad_scale = 1 / world."""
        if not self._dp_active():
            return
        bucket = next((b for b in self._grad_buckets[id(net)] if b[0] == p['_item']), None)
        if bucket is None:
            return
        _, lo, hi = bucket
        self._wgrad_flush()                              # (a gradient of the bucket may still be waiting for its reduction)
        from . import dist as mdist
        capi = self._dp_backend == 'capi'

        def collective(stream_raw):
            if capi:                                     # a library call: recorded by a launch plan like any kernel
                ops.check(ops.require_device().mmdgan_allreduce_bucket(net.grads.data_ptr() + 4 * lo, hi - lo, stream_raw),
                          'allreduce_bucket')
            else:
                mdist.allreduce_sum_(net.grads[lo:hi], self.dist_group)
        if not self._side:
            collective(ops._stream())
            return
        # kernels' gradients are issued on the weight-gradient stream, batch-norm gradients on the main stream
        ops.stream_wait(self._sn_raw, self._wg_raw)
        ops.stream_wait(self._sn_raw, ops._stream())
        with torch.cuda.stream(self._sn_stream):
            collective(self._sn_raw)
        self._exchange_pending = True
        self.exchanged_buckets = getattr(self, 'exchanged_buckets', 0) + 1

    def _dp_active(self):
        return self.dist_group is not None and (self.world > 1 or self._dp_force)

    def _step_body(self):
        """one training step as library calls only (streams and dependencies included): what a launch plan records"""
        B = self.B
        lib = ops.require_device()
        main = ops._stream()
        # everything the step accumulates into that is not a gradient arena (power-iteration targets, <G, W> scalars,
        # batch-norm totals of the backward pass, folded-kernel gradients): one small launch, first thing
        ops.memset_zero_multi([self._zero_scratch] + [t for c in self._sn_chains.values() for t in c[0].zero_each_step])
        lib.mmdgan_set_outputs_prezeroed(1)
        self._sn_zeroed = True
        try:
            if self._side:
                # after the previous step's Adam: G's transformed weights first (its forward pass waits for them), then
                # the zeroing of the gradient arenas, then D's
                ops.stream_wait(self._wg_raw, main)
                with torch.cuda.stream(self._wg_stream):
                    self._transform_weights(self.gen)
                    ops.event_record(_EV_GEN_READY, self._wg_raw)
                    ops.memset_zero_multi([self.gen.grads, self.dis.grads])
                    self._transform_weights(self.dis)
                    ops.event_record(_EV_DIS_READY, self._wg_raw)
                    # (the step counts and bias-corrected learning rates of both updates need no gradient: here, not at the tail)
                    ops.adam_prepare_multi([(self.dis.opt, self.lr_d), (self.gen.opt, self.lr_g)])
                ops.stream_wait(self._sn_raw, main)
                with torch.cuda.stream(self._sn_stream):                 # depend on the weights only; G's first
                    if any(k.sn for k in self.gen.kernels):
                        self._sn_all(self.gen)
                        ops.event_record(_EV_GEN_SN_READY, self._sn_raw)
                        ops.event_wait(_EV_GEN_SN_READY, main)
                    self._sn_all(self.dis)
                # (G's composed / transformed weights are waited for where its first convolution starts - _forward: a dense
                # first layer and its batch norm run beside the transforms, which themselves wait for the previous step's Adam)
                self._ready_wait = _EV_GEN_READY
            else:
                ops.memset_zero_multi([self.gen.grads, self.dis.grads])
                self._compose_weights(self.gen)
                self._compose_weights(self.dis)
                for net in (self.gen, self.dis):
                    self._sn_all(net)
        finally:
            # (the forward pass runs without it: its launches with few tiles split their reductions into outputs the
            # library zeroes itself)
            lib.mmdgan_set_outputs_prezeroed(0)
            self._sn_zeroed = False
        self._in_step = True
        try:
            # G's output goes straight into the fake half of D's input (my_sngan.py:278: D sees [real ; fake]); the real half
            # IS the batch buffer
            gvals = self._forward(self.gen, self._static_z, True, 'g', out_buffer=self._dis_in[B:])
            self._ready_wait = None                                      # (no convolution in G: _EV_DIS_READY below is later on the same stream)
            if self._side:
                ops.stream_wait(main, self._sn_raw)
                ops.event_wait(_EV_DIS_READY, main)
            dvals = self._forward(self.dis, self._dis_in, True, 'd')
            self._last_vals = (gvals, dvals)                             # (the parity tests read activations from here)
            scores = dvals[self.dis.out_val]                             # [2B, d]: s_x = [:B], s_gen = [B:]
            self._loss.launch(scores, self.losses)
            ds = self._loss.grads.view(4 * B, -1)   # [dLd/ds_x ; dLd/ds_gen ; dLg/ds_gen ; dLg/ds_x]
            lib.mmdgan_set_outputs_prezeroed(1)     # gradient arenas and the scratch were zeroed at step start
            ops.wgrad_defer(self._wgrad_defer)      # the weight gradients of the backward pass as a chain (engine.py:_step_body)
            if self._d_joint:
                # loss_dis (2B rows) and loss_gen (the fake half again) through D together, 3B rows per launch
                d_in = self._backward(self.dis, dvals, ds[:3 * B], 'bd', param_grads=True, need_input_grad=True, extra_rows=B)
            else:
                self._backward(self.dis, dvals, ds[:2 * B], 'bd', param_grads=True)
                if self._d_has_bn:
                    # batch statistics couple the rows: the REAL scores depend on the fake images too (through the batch
                    # mean / variance), so loss_gen reaches G along dLg/ds_x as well - a full 2B-row pass with both halves
                    # of the loss_gen score gradient, [dLg/ds_x ; dLg/ds_gen] in D's [real ; fake] row order
                    dg = self._buf('dg_full', [2 * B, self.score_size])
                    ops.copy(dg[:B], ds[3 * B:4 * B])
                    ops.copy(dg[B:], ds[2 * B:3 * B])
                    d_in = self._backward(self.dis, dvals, dg, 'bg', param_grads=False, need_input_grad=True)[B:]
                else:
                    d_in = self._backward(self.dis, dvals, ds[2 * B:3 * B], 'bg', rows=(B, 2 * B), param_grads=False,
                                          need_input_grad=True)
            d_early = self._side and self._early_d_adam and not self._dp_active()
            self._wgrad_flush()                                          # D's last weight gradient: its reduction, and what waits for it
            if d_early:
                # D's gradients are complete once the weight-gradient stream has drained what it holds now and the main stream
                # has reached this point; nothing in G's backward pass reads D's weights: D's Adam runs beside it, not at the tail
                ops.stream_wait(self._wg_raw, main)
                with torch.cuda.stream(self._wg_stream):
                    self.dis.opt.step(self.lr_d, grad_scale=1.0)
            self._backward(self.gen, gvals, d_in, 'gb', param_grads=True)
            self._wgrad_flush()
            if self._side:
                ops.stream_wait(main, self._wg_raw)
        finally:
            self._wg_after = []
            lib.mmdgan_wgrad_defer(0)
            lib.mmdgan_set_outputs_prezeroed(0)
            self._in_step = False
        if self._exchange_pending:
            ops.stream_wait(main, self._sn_raw)
            self._exchange_pending = False
        gs = 1.0 / self.world
        if not d_early:
            self.dis.opt.step(self.lr_d, grad_scale=gs)
        self.gen.opt.step(self.lr_g, grad_scale=gs)

    def step(self, real_nhwc=None, z=None, uni=None):
        if z is None:
            self._static_z.normal_(generator=self._z_gen)                # my_sngan.py:123-124
        else:
            self._static_z.copy_(z)
        self._loss.draw(self._z_gen, uni)                                # the *_mix coin (math_func.py:2079)
        if real_nhwc is not None:
            self._static_real.copy_(real_nhwc)                           # (= the real half of D's input)
        mode = self.launch_mode
        if mode == 'plan' and self._dp_active() and self._dp_backend != 'capi':
            mode = 'eager'                                               # torch.distributed collectives are not plan nodes
        if (self.lr_d, self.lr_g) != self._baked_lr:                     # a recorded plan holds the learning rates by value
            self._baked_lr = (self.lr_d, self.lr_g)
            self._drop_plan()
        with self._handle:                                               # this engine's workspace / prezeroed mode / plan / events
            if mode == 'eager':
                self._step_body()
            else:
                self._plan_step()
        self.global_step += 1

    def plan_kernels(self):
        """the kernel launches of the recorded step in issue order (GanEngine.plan_kernels)"""
        if self._plan is None:
            raise RuntimeError('no recorded plan: launch_mode must be "plan" and one step must have run')
        with self._handle:
            return ops.plan_kernels(self._plan)

    def _drop_plan(self):
        if self._plan is not None:
            with self._handle:
                ops.require_device().mmdgan_plan_destroy(self._plan)
            self._plan = None

    def _plan_step(self):
        """record the step once (an ordinary eager step that the library notes down), replay it from one C call afterwards"""
        import ctypes
        lib = ops.require_device()
        main = ops._stream()
        if self._plan is not None and self._plan_stream != main:
            self._drop_plan()
        if self._plan is None:
            self._handle.forget_workspace_users()        # (GanEngine._plan_step)
            ops.check(lib.mmdgan_plan_begin(), 'plan_begin')
            try:
                self._step_body()
            except Exception:
                lib.mmdgan_plan_abort()
                raise
            pid = ctypes.c_int(-1)
            ops.check(lib.mmdgan_plan_end(ctypes.byref(pid)), 'plan_end')
            self._plan, self._plan_stream = pid.value, main
            return
        ops.check(lib.mmdgan_plan_replay(self._plan, 0), 'plan_replay')

    # ---- variables / checkpoints (reference names and layouts) -------------------------------------------------
    def _net_of(self, name):
        return self.gen if name.startswith('gen/') else self.dis

    def variable_names(self, trainable_only=False):
        return self.gen.variable_names(trainable_only) + self.dis.variable_names(trainable_only)

    def set_variables(self, values):
        for k, v in values.items():
            self._net_of(k).set_variable(k, v)

    def get_variables(self, names=None, grad=False):
        names = names if names is not None else self.variable_names(trainable_only=grad)
        return OrderedDict((k, self._net_of(k).get_variable(k, grad=grad)) for k in names)

    def set_adam_state(self, m, v, t):
        """Adam moments (reference names and layouts) and the step count of both optimisers, see GanEngine"""
        for k in m:
            net = self._net_of(k)
            for flat, src in ((net.adam_m, m[k]), (net.adam_v, v[k])):
                dst = net.arena.view(k, flat)
                dst.copy_(torch.as_tensor(net.to_native(k, src), device=self.device).reshape(dst.shape))
        for net in (self.gen, self.dis):
            net.opt.step_counter.fill_(int(t))

    def sigmas(self):
        """spectral norms of the last step, keyed like the reference's op scopes (<layer> for a plain layer's
        kernel, <layer>/kernel_0 ... inside a block)"""
        out = OrderedDict()
        for net in (self.dis, self.gen):
            for k in net.kernels:
                if k.sn:
                    scope = k.scope[:-len('/kernel')] if k.scope.endswith('/kernel') else k.scope
                    out[scope] = float(net.sn[k.scope]['sigma'].item())
        return out

    def get_adam_state(self):
        """(m, v, t): Adam moments by variable name in the reference's layouts and the step count (what set_adam_state takes)"""
        m, v = OrderedDict(), OrderedDict()
        for net in (self.gen, self.dis):
            for k in net.variable_names(trainable_only=True):
                m[k] = net.to_ref(k, net.arena.view(k, net.adam_m).detach().cpu().numpy())
                v[k] = net.to_ref(k, net.arena.view(k, net.adam_v).detach().cpu().numpy())
        return m, v, int(self.dis.opt.step_counter.item())

    def state_dict(self):
        """format 2 (GanEngine.state_dict): Adam moments by variable name, independent of the arena layout"""
        m, v, t = self.get_adam_state()
        return {'format': 2, 'global_step': self.global_step, 'variables': self.get_variables(),
                'loss_state': self._loss.state_dict(), 'adam_m': m, 'adam_v': v, 'adam_t': t}

    def load_state_dict(self, sd):
        self.set_variables(sd['variables'])
        self.global_step = int(sd['global_step'])
        self._loss.load_state_dict(sd.get('loss_state', {}))
        if sd.get('format', 1) >= 2:
            self.set_adam_state(sd['adam_m'], sd['adam_v'], sd['adam_t'])
        else:                                            # format 1: the flat arenas of the layout that wrote them
            for tag, net in (('gen', self.gen), ('dis', self.dis)):
                if sd[tag + '/adam_m'].numel() != net.adam_m.numel():
                    raise ValueError('checkpoint format 1 holds the Adam moments of %s as a flat arena of another layout' % tag)
                net.adam_m.copy_(sd[tag + '/adam_m'])
                net.adam_v.copy_(sd[tag + '/adam_v'])
                net.opt.step_counter.fill_(int(sd[tag + '/adam_t']))
        self._drop_plan()
