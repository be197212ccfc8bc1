"""The G+D training step as a fixed schedule of HIP kernel launches.

This is the build's restatement of the reference's graph (SURVEY.md sections 3.1, 8(a) rows A1-A11):
  architecture dict -> layer specs       layer_func.py:1189-1275, 2118-2151, 2221-2391
  G forward, D forward on [real ; fake]  my_sngan.py:271-279
  SN power iteration per D layer         math_func.py:661-672, 739-744
  rep / rmb loss                         math_func.py:2505-2550
  two gradient passes + TF-Adam          my_sngan.py:301-305, 424-425, graph_func.py:518-527
  UPDATE_OPS (SN vectors, BN stats)      graph_func.py:848-853

There is no autograd and no tracing: the step is static, so the host issues the same kernel
sequence every time (and can capture it into one hipGraph).  torch supplies device memory, the
stream and the z sampler only.

MI355X-first choices
  * NHWC activations; every weight lives in ONE flat arena per network (params / grads / Adam m /
    Adam v are four flat buffers) so Adam is one launch and the data-parallel all-reduce is a
    handful of large buckets over xGMI instead of 29 small ones.
  * D sees real and fake as the two halves of one 2B batch buffer; G's last layer writes its
    output straight into the second half (the reference's concat/split copies are gone).
  * loss_gen is back-propagated through D on the fake half only (the reference's TF graph also
    pushes zeros through the real half - same result, 1/7 less work).
  * the reference's NCHW layout shows up only at the API seam: dense weights next to a
    [C,H,W] reshape are stored with their rows/columns permuted so that the NHWC view is free;
    get_variables()/set_variables() convert to and from the reference's names and layouts.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import initializers, ops, settings

# named event slots (mmdgan_event_record / _wait)
_EV_WINO_GEN = 0                      # G's Winograd weights are ready
_EV_WINO_DIS = 1                      # D's
_EV_AHEAD_KEPT = 2                    # the tail of the step has saved scale / dsigma of D as this step used them (_ahead_tail)
_EV_SN0 = 8                           # + i: the power iteration of D layer i has produced its scale (i < 24)
_EV_SN_GEN0 = 32                      # + i: ... of G layer i

_TEMPLATE = {'name': None, 'type': 'default', 'op': 'c', 'out': None, 'bias': 'b',
             'act': 'linear', 'act_nm': None, 'act_k': False, 'w_nm': None, 'w_p': None,
             'kernel': 3, 'strides': 1, 'dilation': 1, 'padding': 'SAME', 'scale': None,
             'in_reshape': None, 'out_reshape': None, 'aux': None}
_ACTS = ('linear', 'relu', 'lrelu', 'tanh')


_trunc_normal = initializers.trunc_normal


def _chw_perm(c, h, w):
    """P with native[(h,w,c) flat] = ref[(c,h,w) flat][P]"""
    return np.arange(c * h * w).reshape(c, h, w).transpose(1, 2, 0).reshape(-1)


class LayerSpec:
    """one layer of a Net after update_layer_design defaults and shape inference."""

    def __init__(self, design, net_name, in_shape_ref, sn_mode='default'):
        d = dict(_TEMPLATE)
        d.update(design)
        if d['act_nm'] in ('bn', 'BN') and d['bias'] in ('b', 'bias'):     # layer_func.py:1241-1242
            d['bias'] = None
        self.scope = '{}/{}'.format(net_name, d['name'])
        if d['type'] != 'default':
            raise NotImplementedError('{}: {} is not implemented.'.format(self.scope, d['type']))   # :2067
        if d['op'] not in ('d', 'c', 'tc'):
            raise AttributeError('layer op {} not supported.'.format(d['op']))                      # :1275
        if d['act'] not in _ACTS:
            raise NotImplementedError('Function {} is not implemented.'.format(d['act']))           # :149
        if d['act_nm'] not in (None, 'bn', 'BN'):
            raise NotImplementedError('{}: {} not implemented'.format(self.scope, d['act_nm']))     # :1561
        if d['w_nm'] not in (None, 's'):
            raise NotImplementedError('{}: {} method not implemented'.format(self.scope, d['w_nm']))  # :824
        if d['in_reshape'] is not None or d['scale'] is not None or d['dilation'] != 1 or d['padding'] != 'SAME':
            raise NotImplementedError('{}: in_reshape / scale / dilation / VALID are not on this engine\'s schedule; use '
                                      'mmdgan_hip.tape.TapeEngine (SNGan.init_net does)'.format(self.scope))
        self.op, self.act, self.name = d['op'], d['act'], d['name']
        self.bn = d['act_nm'] in ('bn', 'BN')
        self.has_bias = d['bias'] is not None
        self.sn = d['w_nm'] == 's'
        self.act_k = d['act_k']
        self.init_w_scale = d.get('init_w_scale')                          # layer_func.py:517, 719-720
        if self.sn and (self.act_k is False or not isinstance(self.act_k, (float, int))):
            # layer_func.py:835 would silently multiply the kernel by False (= 0); documented deviation
            raise ValueError('{}: w_nm="s" needs a numeric act_k'.format(self.scope))
        self.R, self.stride, self.out = d['kernel'], d['strides'], d['out']
        self.in_shape_ref = list(in_shape_ref)          # [F] or [C,H,W]
        if self.op == 'd':
            assert len(in_shape_ref) == 1, '{}: the input shape {} does not match a dense layer'.format(self.scope, in_shape_ref)
            self.kernel_shape = [in_shape_ref[0], self.out]
            out = [self.out]
        elif self.op == 'c':
            c, h, w = in_shape_ref
            self.kernel_shape = [self.R, self.R, c, self.out]
            out = [self.out, -(-h // self.stride), -(-w // self.stride)]
        else:
            c, h, w = in_shape_ref
            self.kernel_shape = [self.R, self.R, self.out, c]
            out = [self.out, h * self.stride, w * self.stride]
        self.op_out_ref = out
        self.channels = out[0]
        self.pim = False
        if self.sn:                                                        # math_func.py:481-486, 512-528
            if self.op == 'd':
                self.use_u = in_shape_ref[0] <= self.out
                self.sn_x_ref = [1, in_shape_ref[0]] if self.use_u else [1, self.out]
            elif sn_mode in ('sn_paper', 'PIM', 'pim'):
                # layer_func.py:801, 811-814: power iteration on the kernel flattened to [k*k*shape[2], shape[3]] (the stored
                # kernel is already that matrix, row-major; a 'tc' kernel is [k,k,out,in]) through the dense routine
                num_in, num_out = int(np.prod(self.kernel_shape[:3])), self.kernel_shape[3]
                self.pim = True
                self.use_u = num_in <= num_out
                self.sn_x_ref = [1, num_in] if self.use_u else [1, num_out]
            else:
                # math_func.py:512-528.  'tc': the iteration runs on the CONV whose transpose the layer is (same kernel, same
                # spectral norm): its input is the layer's output - x has the layer's output shape when use_u
                self.use_u = int(np.prod(in_shape_ref)) <= int(np.prod(out))
                if self.op == 'c':
                    self.sn_x_ref = [1] + (list(in_shape_ref) if self.use_u else list(out))
                else:
                    self.sn_x_ref = [1] + (list(out) if self.use_u else list(in_shape_ref))
        self.out_reshape = d['out_reshape']
        self.out_shape_ref = list(self.out_reshape) if self.out_reshape is not None else list(out)
        assert int(np.prod(self.out_shape_ref)) == int(np.prod(out)), \
            '{}: the output shape {} does not match existed shape {}.'.format(self.scope, out, self.out_shape_ref)
        # permutations at the NCHW<->NHWC seam (None = identity)
        self.row_perm = None      # dense input features were a flattened [C,H,W] tensor
        self.col_perm = None      # dense output features become a [C,H,W] tensor
        if self.op == 'd' and self.out_reshape is not None and len(self.out_reshape) == 3:
            self.col_perm = _chw_perm(*self.out_reshape)


def build_specs(designs, input_shape_ref, net_name, sn_mode='default'):
    specs, shape, prev = [], list(input_shape_ref), None
    for design in designs:
        s = LayerSpec(design, net_name, shape, sn_mode)
        if s.op == 'd' and prev is not None and len(prev.op_out_ref) == 3:
            s.row_perm = _chw_perm(*prev.op_out_ref)                       # e.g. D l7 -> l8 (my_test_cifar.py:36)
        specs.append(s)
        shape, prev = s.out_shape_ref, s
    return specs


class _Arena:
    """flat fp32 arena with named views."""

    def __init__(self, entries, device):
        self.offsets, n = OrderedDict(), 0
        for name, shape in entries:
            size = int(np.prod(shape))
            self.offsets[name] = (n, size, tuple(shape))
            n += (size + 3) // 4 * 4                   # keep every view 16-byte aligned
        self.size = n
        self.flat = torch.zeros(max(n, 4), dtype=torch.float32, device=device)

    def view(self, name, flat=None):
        o, size, shape = self.offsets[name]
        return (self.flat if flat is None else flat)[o:o + size].view(shape)

    def like(self):
        return torch.zeros_like(self.flat)


class Network:
    """parameters, state and per-layer buffers of one net (G or D)."""

    def __init__(self, specs, device, rng, weight_init='default'):
        self.specs, self.device = specs, device
        self.weight_init = initializers.check_mode(weight_init)            # FLAGS.WEIGHT_INITIALIZER, layer_func.py:27-64
        entries = []
        self.name = specs[0].scope.split('/')[0]
        sn_specs = [s for s in specs if s.sn]
        # <G, W> of every spectrally normalised kernel (the scalar of its gradient's fix-up, below) lives at the HEAD of the
        # gradient arena: it is zeroed with the arena and summed with it by a data-parallel exchange (the fix-up is linear)
        self._dot_name = self.name + '/#sn_dot'
        if sn_specs:
            entries.append((self._dot_name, [len(sn_specs)]))
        for s in specs:
            entries.append((s.scope + '/kernel/kernel', s.kernel_shape))
            if s.has_bias:
                entries.append((s.scope + '/bias/bias', [s.channels]))
            if s.bn:
                nfeat = s.out if s.op == 'd' else s.channels
                entries.append((s.scope + '/BN/BN/gamma', [nfeat]))
                entries.append((s.scope + '/BN/BN/beta', [nfeat]))
        self.arena = _Arena(entries, device)
        self.params = self.arena.flat
        self.grads, self.adam_m, self.adam_v = self.arena.like(), self.arena.like(), self.arena.like()
        self.state = OrderedDict()                      # non-trainable: SN vectors, BN moving stats
        # everything the SN chain accumulates into with atomics (split reductions, dot) lives in ONE
        # scratch arena so that the step zeroes it with a single memset
        sn_entries = []
        for s in specs:
            if s.sn:
                sn_entries += [(s.scope + '#u', self._sn_u_shape(s)), (s.scope + '#xb', self._sn_native_shape(s)),
                               (s.scope + '#dsigma', s.kernel_shape)]
        self.sn_scratch = _Arena(sn_entries, device)
        # the power-iteration vectors and the spectral norms - what a step's UPDATE_OPS replace (math_func.py:661-672, 739-744) -
        # lie in ONE flat buffer, with a shadow of the same layout beside it: an engine that runs the iteration of step t+1 at
        # the tail of step t (GanEngine, "ahead") lets it write the shadow and commits it with one copy when step t+1 starts
        self.sn_live = _Arena([e for s in specs if s.sn for e in ((s.scope + '/kernel/SN/in_rand', self._sn_native_shape(s)),
                                                                  (s.scope + '#sigma', [1]))], device)
        self.sn_shadow = self.sn_live.like()
        # act_k / sigma of every normalised kernel, one flat buffer as well.  `readout` (set by an engine that runs the next step's
        # iteration at the tail of this one): {state key: tensor} - copies of scale and dsigma/dW AS THE LAST STEP USED THEM, for
        # whoever inspects that step's gradients afterwards (effective_grad)
        self.sn_scales = _Arena([(s.scope + '#scale', [1]) for s in specs if s.sn], device)
        self.readout = None
        for s in specs:
            if s.sn:
                self.state[s.scope + '#scale'] = self.sn_scales.view(s.scope + '#scale')
                self.state[s.scope + '/kernel/SN/in_rand'] = self.sn_live.view(s.scope + '/kernel/SN/in_rand')
                self.state[s.scope + '#sigma'] = self.sn_live.view(s.scope + '#sigma')
                for k in ('dsigma', 'u', 'xb'):
                    self.state[s.scope + '#' + k] = self.sn_scratch.view(s.scope + '#' + k)
                i = sn_specs.index(s)
                self.state[s.scope + '#dot'] = self.arena.view(self._dot_name, self.grads)[i:i + 1]
            if s.bn:
                nfeat = s.out if s.op == 'd' else s.channels
                self.state[s.scope + '/BN/BN/moving_mean'] = torch.zeros(nfeat, device=device)
                self.state[s.scope + '/BN/BN/moving_variance'] = torch.ones(nfeat, device=device)
        # TF-Adam over the arena, one segment per variable; a spectrally normalised kernel's segment carries the fix-up of its
        # gradient  dL/dW = scale * G - (scale / sigma) * <G, W> * dsigma/dW  (SURVEY A.2): the arena holds the RAW G
        segments = []
        for name, (off, size, _) in self.arena.offsets.items():
            if name == self._dot_name:
                continue
            s = self._spec_of(name)
            sn = None
            if s.sn and name.endswith('/kernel/kernel'):
                sn = {k: self.state[s.scope + '#' + k] for k in ('dsigma', 'dot', 'sigma', 'scale')}
            segments.append((off, size, sn))
        self.opt = ops.AdamArena(self.params, self.grads, self.adam_m, self.adam_v, segments)
        self.init_variables(rng)

    def effective_grad(self, name):
        """the gradient of variable `name` as the optimiser uses it: for a spectrally normalised kernel the fix-up applied to
        the raw gradient the arena holds (native layout, a new tensor)"""
        g = self.arena.view(name, self.grads)
        s = self._spec_of(name)
        if not (s.sn and name.endswith('/kernel/kernel') and self.opt.fold_fixup):
            return g.clone()
        st = self.state if self.readout is None else self.readout
        sc, sigma = st[s.scope + '#scale'], self.state[s.scope + '#sigma']
        return sc * g - (sc / sigma) * self.state[s.scope + '#dot'] * st[s.scope + '#dsigma'].view(g.shape)

    def effective_grads_flat(self):
        """the whole gradient arena with every fix-up applied (tests / inspection)"""
        out = self.grads.clone()
        for name in self.arena.offsets:
            if name != self._dot_name:
                self.arena.view(name, out).copy_(self.effective_grad(name))
            else:
                self.arena.view(name, out).zero_()       # the <G, W> scalars are no variable's gradient
        return out

    # ---- names / layouts -------------------------------------------------------------------
    @staticmethod
    def _sn_native_shape(s):
        r = s.sn_x_ref
        return [1, r[2], r[3], r[1]] if len(r) == 4 else list(r)

    @staticmethod
    def _sn_u_shape(s):
        if s.op == 'd' or s.pim:
            return [1, s.kernel_shape[-1]] if s.use_u else [1, int(np.prod(s.kernel_shape[:-1]))]
        # F(x): the conv's output when use_u, its input otherwise; the conv of a 'tc' layer maps the layer's output to its input
        cin, cout = (s.in_shape_ref, s.op_out_ref) if s.op == 'c' else (s.op_out_ref, s.in_shape_ref)
        src = cout if s.use_u else cin
        return [1, src[1], src[2], src[0]]

    def p(self, name):
        return self.arena.view(name)

    def g(self, name):
        return self.arena.view(name, self.grads)

    def variable_names(self, trainable_only=False):
        names = [n for n in self.arena.offsets if '#' not in n]
        if not trainable_only:
            names += [k for k in self.state if '#' not in k]
        return names

    def _spec_of(self, name):
        for s in self.specs:
            if name.startswith(s.scope + '/'):
                return s
        raise KeyError(name)

    def _to_native(self, name, ref):
        """reference layout (numpy) -> native layout (numpy)"""
        s = self._spec_of(name)
        ref = np.asarray(ref, dtype=np.float32)
        if name.endswith('in_rand'):
            if ref.ndim == 4:
                return np.ascontiguousarray(ref.transpose(0, 2, 3, 1))
            perm = s.row_perm if (s.op == 'd' and s.use_u) else s.col_perm
            return ref[:, perm] if perm is not None else ref
        if name.endswith('kernel/kernel') and s.op == 'd':
            if s.row_perm is not None:
                ref = ref[s.row_perm, :]
            if s.col_perm is not None:
                ref = ref[:, s.col_perm]
            return np.ascontiguousarray(ref)
        if s.op == 'd' and s.col_perm is not None and ref.ndim == 1:      # bias / BN vectors of a reshaped dense
            return np.ascontiguousarray(ref[s.col_perm])
        return ref

    def _to_ref(self, name, nat):
        s = self._spec_of(name)
        nat = np.asarray(nat, dtype=np.float32)
        if name.endswith('in_rand'):
            if nat.ndim == 4:
                return np.ascontiguousarray(nat.transpose(0, 3, 1, 2))
            perm = s.row_perm if (s.op == 'd' and s.use_u) else s.col_perm
            if perm is not None:
                out = np.empty_like(nat)
                out[:, perm] = nat
                return out
            return nat
        if name.endswith('kernel/kernel') and s.op == 'd':
            out = nat
            if s.col_perm is not None:
                t = np.empty_like(out)
                t[:, s.col_perm] = out
                out = t
            if s.row_perm is not None:
                t = np.empty_like(out)
                t[s.row_perm, :] = out
                out = t
            return out
        if s.op == 'd' and s.col_perm is not None and nat.ndim == 1:
            out = np.empty_like(nat)
            out[s.col_perm] = nat
            return out
        return nat

    def set_variable(self, name, ref_value):
        nat = torch.as_tensor(self._to_native(name, ref_value), device=self.device)
        dst = self.arena.view(name) if name in self.arena.offsets else self.state[name]
        assert tuple(dst.shape) == tuple(nat.shape), (name, tuple(dst.shape), tuple(nat.shape))
        dst.copy_(nat)

    def get_variable(self, name, grad=False):
        if name in self.arena.offsets:
            t = self.effective_grad(name) if grad else self.arena.view(name)
        else:
            t = self.state[name]
        return self._to_ref(name, t.detach().cpu().numpy())

    # ---- initialisers: weight_initializer in the mode FLAGS.WEIGHT_INITIALIZER names (layer_func.py:14-66,
    #      mmdgan_hip/initializers.py), bias 1e-5 (:747) --
    def init_variables(self, rng):
        for s in self.specs:
            w = initializers.weight_initializer(rng, s.kernel_shape, s.act, self.weight_init,
                                                1.0 if s.init_w_scale is None else s.init_w_scale)
            self.set_variable(s.scope + '/kernel/kernel', w)
            if s.sn:                                                        # math_func.py:565-567: NOT normalised
                self.set_variable(s.scope + '/kernel/SN/in_rand', _trunc_normal(rng, s.sn_x_ref, 1.0))
            if s.has_bias:
                n = s.out if s.op == 'd' else s.channels
                self.set_variable(s.scope + '/bias/bias', _trunc_normal(rng, [n], 1e-5))
            if s.bn:
                n = s.out if s.op == 'd' else s.channels
                self.set_variable(s.scope + '/BN/BN/gamma', np.ones(n, np.float32))
                self.set_variable(s.scope + '/BN/BN/beta', np.zeros(n, np.float32))


def _native_shape(shape_ref, batch):
    return [batch, shape_ref[1], shape_ref[2], shape_ref[0]] if len(shape_ref) == 3 else [batch, shape_ref[0]]


def sn_scratch_buffers(net, s, device):
    """the scratch sn_power_iteration needs for layer `s` when no engine owns it (Routine's eager path)"""
    u = net.state[s.scope + '#u']                         # the network's own scratch arena holds u and xb
    return {s.scope + '#u': u, s.scope + '#un': torch.zeros_like(u), s.scope + '#xb': net.state[s.scope + '#xb'],
            s.scope + '#xbnorm': torch.zeros(1, device=device)}


def sn_power_iteration(net, s, b, update=True, out_zeroed=True):
    """one power-iteration step of layer `s` of `net` (math_func.py:661-672, 739-744):
    sigma = ||F(x)|| from the pre-update x, dsigma/dW, then (update=True, the UPDATE_OPS of a training step)
    x <- normalised F^T(y) IN PLACE: every reader of the old x (F(x) and the dsigma outer product / weight gradient)
    is issued before the write on the same stream, which is the UPDATE_OPS ordering (reads precede writes).
    update=False evaluates sigma / scale only (inference: the reference's sessions do not run UPDATE_OPS there).
    `b` holds the scratch (`#u`, `#un`, `#xb`, `#xbnorm`); out_zeroed: `#u` / `#xb` / the state's `#dsigma` are zero on
    entry (the engine's once-per-step memset), so the split reductions may accumulate into them.
    Returns the device scalar act_k / sigma."""
    w = net.p(s.scope + '/kernel/kernel')
    x = net.state[s.scope + '/kernel/SN/in_rand']
    sigma, scale = net.state[s.scope + '#sigma'], net.state[s.scope + '#scale']
    dsig = net.state[s.scope + '#dsigma']
    u, un, xb, xbn = b[s.scope + '#u'], b[s.scope + '#un'], b[s.scope + '#xb'], b[s.scope + '#xbnorm']
    oz = bool(out_zeroed)
    if s.op == 'd' or s.pim:
        if s.pim:                                                        # layer_func.py:811-814
            w, dsig = w.view(-1, w.shape[-1]), dsig.view(-1, w.shape[-1])
        if 1 in w.shape:                                                 # math_func.py:702-704
            ops.sn_norm_scale(w.view(-1), s.act_k, sigma, scale, dsig.view(-1))
        elif s.use_u:
            ops.gemm(x, w, out=u, out_zeroed=oz)                         # u = x W          [1,out]
            ops.sn_norm_scale(u.view(-1), s.act_k, sigma, scale, un.view(-1))
            if update:
                ops.gemm(x, un, trans_a=True, out=dsig, out_zeroed=oz)   # dsigma/dW = x^T y
                ops.gemm(un, w, trans_b=True, out=xb, out_zeroed=oz)     # y W^T            [1,in]
                ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
        else:
            ops.gemm(x, w, trans_b=True, out=u, out_zeroed=oz)           # u = x W^T        [1,in]
            ops.sn_norm_scale(u.view(-1), s.act_k, sigma, scale, un.view(-1))
            if update:
                ops.gemm(un, x, trans_a=True, out=dsig, out_zeroed=oz)   # dsigma/dW = y^T x
                ops.gemm(un, w, out=xb, out_zeroed=oz)                   # y W              [1,out]
                ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
    else:
        # the conv with kernel w [R,R,C,K]: for a 'tc' layer ([R,R,out,in]) the conv whose transpose the layer is - its input is
        # the layer's output (math_func.py:520-528: forward / backward swap roles, the code below is the same)
        c, h, wd = s.in_shape_ref if s.op == 'c' else s.op_out_ref
        if s.use_u:
            ops.conv2d_fwd(x, w, s.stride, out=u)
            ops.sn_norm_scale(u.view(-1), s.act_k, sigma, scale, un.view(-1))
            if update:
                ops.conv2d_wgrad(x, un, s.R, s.stride, out=dsig)
                ops.conv2d_dgrad(un, w, (h, wd), s.stride, out=xb)
                ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
        else:
            ops.conv2d_dgrad(x, w, (h, wd), s.stride, out=u)
            ops.sn_norm_scale(u.view(-1), s.act_k, sigma, scale, un.view(-1))
            if update:
                ops.conv2d_wgrad(un, x, s.R, s.stride, out=dsig)
                ops.conv2d_fwd(un, w, s.stride, out=xb)
                ops.sn_norm(xb.view(-1), True, out_norm=xbn, out_v=x.view(-1))
    return scale


def sn_chain_layer(net, s, b, shadow=False):
    """the power iteration of layer `s` as ops.SnChains describes it (same tensors as sn_power_iteration), or None for a
    kernel with a unit dimension (math_func.py:702-704: no iteration, sigma = ||w||).  shadow: the new vector and the
    spectral norm go to the net's shadow buffer (Network.sn_shadow) instead of replacing the live ones"""
    w = net.p(s.scope + '/kernel/kernel')
    if (s.op == 'd' or s.pim) and 1 in (int(np.prod(w.shape[:-1])), w.shape[-1]):
        return None
    L = dict(w=w, x=net.state[s.scope + '/kernel/SN/in_rand'], sigma=net.state[s.scope + '#sigma'],
             scale=net.state[s.scope + '#scale'], dsigma=net.state[s.scope + '#dsigma'], u=b[s.scope + '#u'],
             un=b[s.scope + '#un'], xb=b[s.scope + '#xb'], xb_norm=b[s.scope + '#xbnorm'], act_k=s.act_k)
    if shadow:
        L['x_out'] = net.sn_live.view(s.scope + '/kernel/SN/in_rand', net.sn_shadow)
        L['sigma'] = net.sn_live.view(s.scope + '#sigma', net.sn_shadow)
    if s.op == 'd' or s.pim:                             # ('sn_paper': the conv kernel as its [R*R*C, K] matrix)
        L.update(form=2 if s.use_u else 3, C=int(np.prod(w.shape[:-1])), K=w.shape[-1])
    else:
        c, h, wd = s.in_shape_ref if s.op == 'c' else s.op_out_ref
        L.update(form=0 if s.use_u else 1, H=h, W=wd, C=w.shape[2], K=w.shape[3], R=s.R, stride=s.stride)
    return L


class GanEngine:
    """G + D + loss + two TF-Adam optimisers; `step()` = one sess.run of graph_func.py:853."""

    def __init__(self, architecture, loss_type='rep', lr_list=(5e-4, 2e-4), rep_weights=(0.0, -1.0),
                 batch_size=64, seed=0, device=None, dist_group=None, use_graph=False, sn_mode='default',
                 weight_init='default', mix_threshold=None, launch_mode=None, dp_backend=None):
        ops.require_device()
        initializers.check_mode(weight_init)
        if loss_type not in ops.LOSS:
            raise NotImplementedError('Not implemented.')                   # math_func.py:2651
        assert rep_weights[0] - rep_weights[1] == 1.0, 'w[0]-w[1] must be 1'   # math_func.py:1340
        self.device = torch.device(device if device is not None else 'cuda')
        self.arch, self.loss_type, self.rep_weights = architecture, loss_type, tuple(rep_weights)
        self.lr_d, self.lr_g = float(lr_list[0]), float(lr_list[1])
        self.B = int(batch_size)
        self.code_size = architecture['code'][0][0]
        self.in_shape_ref = list(architecture['input'][0])
        rng = np.random.RandomState(seed)
        if sn_mode not in ('default', 'PICO', 'pico', 'sn_paper', 'PIM', 'pim'):                      # layer_func.py:802-814
            raise NotImplementedError('spectral norm mode {} is not implemented.'.format(sn_mode))
        self.sn_mode = sn_mode
        self.gen = Network(build_specs(architecture['generator'], [self.code_size], 'gen', sn_mode), self.device, rng,
                           weight_init)
        self.dis = Network(build_specs(architecture['discriminator'], self.in_shape_ref, 'dis', sn_mode), self.device, rng,
                           weight_init)
        # this engine's hand-written schedule covers the DCGAN shape of the reference's drivers: batch norm in G only
        # (my_test_*.py); spectral norm anywhere, transposed-conv kernels included.  Anything else must not train silently
        # wrong: mmdgan_hip.tape.TapeEngine takes batch norm in D (SNGan.init_net routes there)
        for s in self.dis.specs:
            if s.bn:
                raise NotImplementedError('{}: batch norm in the discriminator is not on this engine\'s schedule; use '
                                          'mmdgan_hip.tape.TapeEngine (SNGan.init_net does)'.format(s.scope))
        if self.gen.specs[-1].out_shape_ref != self.in_shape_ref:
            raise AssertionError('gen: the output shape {} does not match existed shape {}.'.format(
                self.gen.specs[-1].out_shape_ref, self.in_shape_ref))
        self.score_size = self.dis.specs[-1].out
        self.global_step = 0
        self.dist_group = dist_group
        settings.warn_unknown()                                          # (a switch of an earlier round would be ignored in silence)
        self._dp_force = settings.on('MMDGAN_DP_FORCE')
        self.world, self.rank = 1, 0
        if dist_group is not None:
            import torch.distributed as tdist
            self.world, self.rank = tdist.get_world_size(dist_group), tdist.get_rank(dist_group)
        # the code sampler of this replica (my_sngan.py:123-124): a generator of its own, seeded per rank - torch's
        # default CUDA generator starts from the same constant in every process, which would hand all replicas the
        # same fake half of the batch
        self._z_gen = torch.Generator(device=self.device)
        self._z_gen.manual_seed((int(seed) * 1000003 + 7919 * self.rank + 12345) % (2 ** 63 - 1))
        self._recording, self._d_updated_early = False, False
        # set to {} by a caller (bench.py under data parallelism): eager steps then leave three events in it - start / end of
        # G's last exchange bucket on the exchange stream and the point where the main stream starts waiting for it
        self.exchange_probe = None
        self.bucket_probe = None             # a list (tools/scale_predict.py): eager steps append (name, bytes, event) per bucket
        # who carries the gradient exchange: 'torch' = torch.distributed on `dist_group` (RCCL through ProcessGroupNCCL; gloo in
        # the tests); 'capi' = the library's own RCCL communicator (mmdgan_comm_init / mmdgan_allreduce_bucket), whose
        # collectives are plan nodes like any launch - a data-parallel step then replays from one C call
        # Default: 'capi' under an nccl (= RCCL) group - it depends on nothing ProcessGroupNCCL does with streams and the
        # whole data-parallel step replays from one C call; 'torch' for any other backend (gloo in the tests)
        from . import dist as mdist
        self._dp_backend = mdist.choose_dp_backend(dist_group, self.device, dp_backend)      # verified, same on every rank
        # the power iterations of different layers are independent of each other too: two chains.
        # weight / bias gradients of a layer depend only on dz of that layer, not on the dgrad chain that
        # continues below it: they go to another stream so their blocks fill the tail of the dgrad
        # launches (each launch alone leaves CUs idle while its last wave of tiles drains).
        # All of them must sit on hardware queues of their own (streams.py: 2.43 vs 2.55 / 2.89 ms per step)
        from .streams import distinct_queue_streams
        n_sn = int(settings.get('MMDGAN_SN_STREAMS'))
        side = distinct_queue_streams(n_sn + 1, self.device)
        self._wg_stream, self._sn_streams = side[0], side[1:]
        self._comm_stream = side[1] if len(side) > 1 else side[0]      # gradient exchange, see _exchange
        self._wg_raw, self._comm_raw = self._wg_stream.cuda_stream, self._comm_stream.cuda_stream
        self._sn_raw = [st.cuda_stream for st in self._sn_streams]
        self._gen_tail_on_main = int(settings.get('MMDGAN_GEN_TAIL_MAIN')) if settings.on('MMDGAN_SIDE_WGRAD') else 0
        self._early_d_adam = settings.on('MMDGAN_EARLY_D_ADAM')
        self._wgrad_defer = settings.on('MMDGAN_WGRAD_DEFER')
        self._wg_after = []
        self._side_wgrad = settings.on('MMDGAN_SIDE_WGRAD')
        # round 4: dependencies that put a marker / barrier packet into the MAIN queue (~6 us of idle queue each) moved off it
        # where another ordering already covers them (MMDGAN_QUEUE_OPT=0: the round-3 placement)
        self._queue_opt = settings.on('MMDGAN_QUEUE_OPT') and self._side_wgrad
        self._bn_resign = settings.on('MMDGAN_BN_RESIGN')
        if ops._workspace is None:
            ops.set_workspace(device=self.device)                        # the default handle's (eval paths, stand-alone ops)
        # this engine's own library state (include/mmdgan_hip.h "Handles"): workspace, prezeroed mode, launch plan
        self._handle = ops.Handle(device=self.device)
        self._alloc(self.B)
        self.losses = torch.zeros(8, device=self.device)       # filled by the loss kernel each step
        self._loss = ops.GanLossLauncher(loss_type, self.rep_weights, self.B, self.score_size, self.device, mix_threshold)
        self.buf['mmd_grads'] = self._loss.grads
        # how a step reaches the GPU.  'eager': ~200 library calls from Python per step; 'graph': one captured hipGraph
        # (few host microseconds, but its branches overlap less: 2.69 vs 2.35 ms per CIFAR step); 'plan': the library
        # records one eager step and re-issues it from ONE C call (mmdgan_plan_replay) - the eager step's streams and
        # overlap with the graph's host cost.  MMDGAN_LAUNCH_MODE overrides; use_graph=True is the old spelling of 'graph'
        self.launch_mode = launch_mode or settings.get('MMDGAN_LAUNCH_MODE') or ('graph' if use_graph else 'eager')
        assert self.launch_mode in ('eager', 'graph', 'plan'), self.launch_mode
        self._graph, self._plan, self._plan_stream, self._plan_collectives = None, None, None, []
        self._baked_lr = (self.lr_d, self.lr_g)
        self._in_step = False                                  # True while step() runs (buffers on the zero list ARE zero)
        self._grad_buckets = {id(net): self._make_buckets(net) for net in (self.gen, self.dis)}
        # a single replica folds the spectral-norm fix-up of its gradients into Adam's read (Network.opt); data-parallel
        # replicas apply it BEFORE their all-reduce - sigma and dsigma/dW carry each replica's own atomics order in their last
        # bits, and replicas must stay bit-identical
        for net in (self.gen, self.dis):
            # data-parallel replicas fix a normalised kernel's gradient up BEFORE the exchange, each with its own dsigma/dW (equal
            # across replicas only up to the order of the power iteration's atomics): the all-reduce then hands every replica the
            # same bits and the replicas stay bit-identical.  (Folding the fix-up into Adam behind the exchange - the raw sums and
            # <G, W> are both in the arena - is the same mathematics, but lets those last bits of dsigma into the weights.)
            net.opt.fold_fixup = not self._dp_active()
        self._static_z = torch.zeros(self.B, self.code_size, device=self.device)
        # the real half of D's input buffer IS the batch buffer graph / plan replays read: a caller's batch is copied
        # once, straight to where the first conv reads it (my_sngan.py:278: D sees [real ; fake])
        self._static_real = self.buf['dis_in'][:self.B]

    @property
    def use_graph(self):
        return self.launch_mode == 'graph'

    @use_graph.setter
    def use_graph(self, on):
        self.launch_mode = 'graph' if on else ('eager' if self.launch_mode == 'graph' else self.launch_mode)

    def _make_buckets(self, net):
        """the gradient arena of `net` cut into exchange buckets, in BACKWARD order: [(lowest layer index, start, end)].
        A bucket is exchanged as soon as the parameter gradients of its lowest layer have been issued, so all but the
        last one travel underneath the rest of the backward pass.  Layers are contiguous in the arena (forward order);
        a bucket closes once it holds MMDGAN_DP_BUCKET_MB (default 8) - xGMI rings are latency-bound below a few MB."""
        from .dist import layer_buckets
        target = int(float(settings.get('MMDGAN_DP_BUCKET_MB')) * (1 << 20)) // 4
        first = {}                                   # layer index -> (start, end) float offsets of its entries
        for name, (o, size, _) in net.arena.offsets.items():
            # (the <G, W> scalars at the head of the arena are complete with layer 0's gradients: they travel with the last bucket)
            li = next((i for i, sp in enumerate(net.specs) if name.startswith(sp.scope + '/')), 0)
            lo, hi = first.get(li, (o, o))
            first[li] = (min(lo, o), max(hi, o + (size + 3) // 4 * 4))
        ranges = [first[li] for li in range(len(net.specs))]
        ranges[-1] = (ranges[-1][0], min(ranges[-1][1], net.arena.size))
        return layer_buckets(ranges, target)

    # ---------------------------------------------------------------------------------------
    def _alloc(self, B):
        """all per-step buffers, allocated once (graph capture needs static addresses)."""
        dev = self.device
        self.buf = {}
        c, h, w = self.in_shape_ref
        self.buf['dis_in'] = torch.zeros(2 * B, h, w, c, device=dev)          # [real ; fake]
        for net, batch in ((self.gen, B), (self.dis, 2 * B)):
            for s in net.specs:
                shp = _native_shape(s.op_out_ref, batch)
                if s.bn:
                    self.buf[s.scope + '#raw'] = torch.zeros(shp, device=dev)
                    self.buf[s.scope + '#dy'] = torch.zeros(shp, device=dev)      # gradient w.r.t. the BN+act output
                    nfeat = shp[-1]
                    self.buf[s.scope + '#mean'] = torch.zeros(nfeat, device=dev)
                    self.buf[s.scope + '#invstd'] = torch.zeros(nfeat, device=dev)
                is_gen_out = net is self.gen and s is self.gen.specs[-1]
                self.buf[s.scope + '#y'] = self.buf['dis_in'][B:] if is_gen_out else torch.zeros(shp, device=dev)
                # gradient w.r.t. the layer's pre-activation output (after act').  D back-propagates
                # [loss_dis rows (2B) ; loss_gen rows of the fake half (B)] together: 3B rows per layer
                self.buf[s.scope + '#dz'] = torch.zeros(_native_shape(s.op_out_ref, 3 * B) if net is self.dis else shp,
                                                        device=dev)
        self.buf['d_fake'] = torch.zeros(B, h, w, c, device=dev)
        # SN scratch per spectrally normalised layer (u / xb live in the network's zero-once-per-step arena)
        for net in (self.gen, self.dis):
            for s in net.specs:
                if s.sn:
                    self.buf[s.scope + '#u'] = net.state[s.scope + '#u']
                    self.buf[s.scope + '#un'] = torch.zeros_like(net.state[s.scope + '#u'])
                    self.buf[s.scope + '#xb'] = net.state[s.scope + '#xb']
                    self.buf[s.scope + '#xbnorm'] = torch.zeros(1, device=dev)
        # buffers some kernel accumulates into with atomics: zeroed once at the start of every step
        # (4 memset nodes instead of one per kernel, see mmdgan_set_outputs_prezeroed)
        self._zero_each_step = [self.gen.grads, self.dis.grads, self.dis.sn_scratch.flat]
        if any(s.sn for s in self.gen.specs):
            self._zero_each_step.append(self.gen.sn_scratch.flat)
        # batch-norm statistics are accumulated with fp64 atomics into per-layer totals ([forward | backward] x 2C),
        # which must be zero when the layer runs: one flat buffer, one memset per step
        bn = [(s.scope, s.out if s.op == 'd' else s.channels) for net in (self.gen, self.dis) for s in net.specs if s.bn]
        lib = ops.require_device()
        n64 = {c: lib.mmdgan_bn_workspace_bytes(c) // 8 for _, c in bn}      # doubles per call (the totals, several slots of them)
        self._bn_flat = torch.zeros(max(1, sum(2 * n64[c] for _, c in bn)), dtype=torch.float64, device=dev)
        self._bn_totals, off = {}, 0
        for scope, c in bn:
            self._bn_totals[scope] = (self._bn_flat[off:off + n64[c]], self._bn_flat[off + n64[c]:off + 2 * n64[c]])
            off += 2 * n64[c]
        if bn:
            self._zero_each_step.append(self._bn_flat)
        for net in (self.gen, self.dis):
            for s in net.specs:          # dense outputs whose K is long enough for the split-K gemm path
                if s.op == 'd' and s.kernel_shape[0] >= 512:
                    self._zero_each_step.append(self.buf[s.scope + ('#raw' if s.bn else '#y')])
        # 3x3 / stride-1 D layers that the library runs through Winograd: their weights change once per step, so
        # G w G^T is computed once per step on the parameter-gradient stream (idle during the forward pass)
        # instead of inside every conv call.  scope -> [forward tensor or None, input-gradient tensor or None]
        self._wino, self._wino_ok = {}, {}
        for net, nf, nb in ((self.dis, 2 * B, 3 * B), (self.gen, B, B)):
            for s in net.specs:
                if s.op == 'c':                              # conv geometry: input = layer input
                    c, h, w = s.in_shape_ref
                    k = s.out
                elif s.op == 'tc':                           # the conv whose input-gradient the tc layer is
                    k = s.in_shape_ref[0]
                    c, h, w = s.out, s.in_shape_ref[1] * s.stride, s.in_shape_ref[2] * s.stride
                else:
                    continue
                fw = ops.wino_algo(nf, h, w, c, k, s.R, s.stride, False)     # F(2x2,3x3) / F(4x4,3x3) / F(2x2,2x2) on 4 parity
                bw = ops.wino_algo(nb, h, w, c, k, s.R, s.stride, True)      # segments: each with its own weight layout
                if fw or bw:
                    self._wino[s.scope] = [ops.wino_alloc(fw, c, k, False, dev) if fw else None,
                                           ops.wino_alloc(bw, c, k, True, dev) if bw else None, net]
        self._wino_jobs = [ops.WinoTransforms([(net.p(scope + '/kernel/kernel'), u, dg) for scope, (uf, ub, net) in self._wino.items()
                                               if net is which for u, dg in ((uf, False), (ub, True)) if u is not None])
                           for which in (self.gen, self.dis)]
        # The batch-1 convolutions of the power iteration run on two concurrent chains and take the library's own route: a
        # batch-1 launch never goes through the handle's shared workspace (no in-call Winograd transform, no slab / partial-sum
        # weight gradient - csrc: "d.N > 1"), and any other cross-stream workspace user is ordered by workspace_acquire.
        gs = self.gen.specs
        for below, above in zip(gs, gs[1:]):     # input-gradient of a tc layer into a linear layer: split-K target
            if above.op == 'tc' and (below.bn or below.act == 'linear'):
                self._zero_each_step.append(self.buf[below.scope + ('#dy' if below.bn else '#dz')])
        self._zeroed_ptrs = {t.data_ptr() for t in self._zero_each_step}
        # the power iterations of a whole net as six launches (csrc/sn_chain.hip) instead of five per kernel;
        # MMDGAN_SN_FUSED=0: one chain of launches per layer, dealt to the power-iteration streams
        self._sn_fused = settings.on('MMDGAN_SN_FUSED') and len(self._sn_streams) > 0
        self._sn_chains = {}
        # "ahead": D's power iterations and D's Winograd weight transform of step t+1 run at the tail of step t, behind D's early
        # Adam and beside G's backward pass, instead of at the head of step t+1 beside G's small forward kernels (_step_body)
        # Measured (interleaved A/B of the bench step, plan replay, profiles/r05_ab_step_ahead.txt): CelebA 64x64 B=128 13.40 against
        # 13.52 ms with the tail, CIFAR 32x32 inside the spread (1.847 / 1.848), STL 48x48 3.705 against 3.677 - the side work moves
        # from beside G's forward pass to beside G's backward pass, and which of the two it hurts less depends on the net:
        # 'auto' (the default) pipelines from 64 x 64 images on, '1' / '0' force it.
        c_, h_, w_ = self.in_shape_ref
        want_ahead = settings.get('MMDGAN_STEP_AHEAD')
        want_ahead = (h_ * w_ >= 64 * 64) if want_ahead == 'auto' else want_ahead not in (None, '0', '')
        self._ahead = (self._sn_fused and want_ahead and self._side_wgrad and self._early_d_adam and
                       self._queue_opt and any(s.sn for s in self.dis.specs) and
                       all(sn_chain_layer(self.dis, s, self.buf) is not None for s in self.dis.specs if s.sn))
        self._ahead_valid = False
        self._tail_issued = False
        if self._ahead:
            self._sn_prev = (torch.zeros_like(self.dis.sn_scales.flat), torch.zeros_like(self.dis.sn_scratch.flat))
            self._sn_readout = {}
            for s in self.dis.specs:
                if s.sn:
                    self._sn_readout[s.scope + '#scale'] = self.dis.sn_scales.view(s.scope + '#scale', self._sn_prev[0])
                    self._sn_readout[s.scope + '#dsigma'] = self.dis.sn_scratch.view(s.scope + '#dsigma', self._sn_prev[1])
        if self._sn_fused:
            for net in (self.gen, self.dis):
                layers = [(s, sn_chain_layer(net, s, self.buf, shadow=self._ahead and net is self.dis)) for s in net.specs if s.sn]
                self._sn_chains[id(net)] = (ops.SnChains([L for _, L in layers if L is not None], dev),
                                            [s for s, L in layers if L is None])
                # the patch matrices that hold split products: zeroed with the step's scratch
                self._zero_each_step += self._sn_chains[id(net)][0].zero_each_step
            self._zeroed_ptrs = {t.data_ptr() for t in self._zero_each_step}

    # ---------------------------------------------------------------------------------------
    # spectral norm: one power-iteration step per D layer (math_func.py:661-672)
    # ---------------------------------------------------------------------------------------
    def _sn_step(self, net, s):
        return sn_power_iteration(net, s, self.buf)

    # ---------------------------------------------------------------------------------------
    def _layer_forward(self, net, s, x, is_training, scale):
        b = self.buf
        w = net.p(s.scope + '/kernel/kernel')
        bias = net.p(s.scope + '/bias/bias') if s.has_bias else None
        y = b[s.scope + '#y']
        n = x.shape[0]
        y = y[:n] if y.shape[0] != n else y
        fused_act = 'linear' if s.bn else s.act
        tgt = (b[s.scope + '#raw'][:n] if s.bn else y)
        if s.op == 'd':
            # a long-K dense output is on the step's zero list (_alloc): only then may the launch split K
            zeroed = self._in_step and tgt.data_ptr() in self._zeroed_ptrs
            ops.gemm(x.reshape(n, -1), w, bias=bias, scale=scale, act=fused_act, out=tgt.view(n, -1), out_zeroed=zeroed)
        # a batch norm behind a convolution: its statistics ride on the launch that writes the tensor (mmdgan_conv2d_*_stats)
        totals = self._bn_totals[s.scope][0] if (s.bn and is_training and self._in_step and s.op != 'd') else None
        if s.op == 'd':
            pass
        elif s.op == 'c':
            # transformed weights only for a batch the library runs Winograd at (its thresholds count tiles)
            wino = self._wino.get(s.scope, (None, None, None))[0] if is_training and self._wino_fwd_ok(net, s, n) else None
            ops.conv2d_fwd(x, w, s.stride, bias=bias, scale=scale, act=fused_act, out=tgt, wino=wino, bn_totals=totals)
        else:
            wino = self._wino.get(s.scope, (None, None, None))[1] if n == self.B and is_training else None
            ops.conv2d_dgrad(x, w, (tgt.shape[1], tgt.shape[2]), s.stride, bias=bias, scale=scale, act=fused_act, out=tgt,
                             wino=wino, bn_totals=totals)
        if s.bn:
            gamma, beta = net.p(s.scope + '/BN/BN/gamma'), net.p(s.scope + '/BN/BN/beta')
            mm, mv = net.state[s.scope + '/BN/BN/moving_mean'], net.state[s.scope + '/BN/BN/moving_variance']
            raw2d, y2d = tgt.view(-1, tgt.shape[-1]), y.view(-1, tgt.shape[-1])
            if is_training:                                  # moving statistics updated in place (UPDATE_OPS)
                ops.bn_fwd_train(raw2d, gamma, beta, mm, mv, act=s.act, unbiased=tgt.dim() == 4, new_moving_mean=mm,
                                 new_moving_var=mv, out=y2d, save_mean=b[s.scope + '#mean'], save_invstd=b[s.scope + '#invstd'],
                                 workspace=self._bn_totals[s.scope][0], have_totals=totals is not None)
            else:
                ops.bn_fwd_infer(raw2d, gamma, beta, mm, mv, act=s.act, out=y2d)
        return y

    def _wino_fwd_ok(self, net, s, n):
        key = (s.scope, n)
        if key not in self._wino_ok:
            c, h, w = s.in_shape_ref
            self._wino_ok[key] = s.scope in self._wino and self._wino[s.scope][0] is not None and \
                ops.wino_algo(n, h, w, c, s.out, s.R, s.stride, False) == ops.wino_kind(self._wino[s.scope][0])
        return self._wino_ok[key]

    def generate(self, z, is_training=False):
        """G(z) -> NHWC images (the fake half of D's input buffer when the batch is B).  Inference: spectral norms (if G
        has any) from the stored power-iteration vectors, not updated - as discriminate()"""
        if is_training:
            scales = self._scales
        else:
            scales = {s.scope: sn_power_iteration(self.gen, s, self.buf, update=False, out_zeroed=False)
                      for s in self.gen.specs if s.sn}
        x = z
        waited = False
        for i, s in enumerate(self.gen.specs):
            if s.sn and is_training and not (self._sn_fused and waited):
                ops.event_wait(_EV_SN_GEN0 + (0 if self._sn_fused else i), ops._stream())   # this layer's power iteration (_forward)
                waited = True
            if is_training and getattr(self, '_gen_wino_wait', None) == i:
                ops.event_wait(_EV_WINO_GEN, ops._stream())
            x = self._layer_forward(self.gen, s, x, is_training, scales.get(s.scope))
            if s.out_reshape is not None:
                x = x.view(_native_shape(s.out_shape_ref, x.shape[0]))
        return x

    def discriminate(self, x_nhwc):
        """D(x) in INFERENCE mode (my_sngan.py:558-560, `self.Dis(..., is_training=False)`): spectral norms from the stored
        power-iteration vectors WITHOUT updating them (UPDATE_OPS run in training sessions only); [n, d] scores.
        Rows are independent in inference, so the batch is walked in chunks of the engine's 2B-row buffers."""
        self._sync_ahead(invalidate=True)                # (writes sigma / scale of D and the chains' scratch)
        scales = {s.scope: (sn_power_iteration(self.dis, s, self.buf, update=False, out_zeroed=False) if s.sn else None)
                  for s in self.dis.specs}
        outs = []
        for i in range(0, x_nhwc.shape[0], 2 * self.B):
            x = x_nhwc[i:i + 2 * self.B].contiguous()
            for s in self.dis.specs:
                x = self._layer_forward(self.dis, s, x, False, scales[s.scope])
                if s.out_reshape is not None:
                    x = x.view(_native_shape(s.out_shape_ref, x.shape[0]))
            outs.append(x.clone())
        return torch.cat(outs, 0)

    # ---------------------------------------------------------------------------------------
    def _forward(self, z, real):
        B, b = self.B, self.buf
        # the spectral-norm power iteration depends on D's weights only, not on the batch: its ~50
        # small launches run on a second HIP stream underneath G's forward pass
        main = ops._stream()
        self._scales = {}
        for st in self._sn_raw:
            ops.stream_wait(st, main)
        if getattr(self, '_sn_zero', None):
            ops.memset_zero_multi(self._sn_zero, stream=self._sn_raw[0])
        # (G's first: its forward pass is what the main stream runs next)
        for net, ev0 in ((self.gen, _EV_SN_GEN0), (self.dis, _EV_SN0)):
            if self._sn_fused:
                # every stage of all the net's chains as one launch, on the first power-iteration stream; one event for the net
                chains, single = self._sn_chains[id(net)]
                for s in net.specs:
                    self._scales[s.scope] = net.state[s.scope + '#scale'] if s.sn else None
                if net is self.dis and self._ahead:
                    if self._ahead_inline():
                        # (a captured hipGraph is one step, closed: the iteration runs here, at the head of its own step, through
                        # the same shadow + commit; nothing is pending between steps)
                        self._ahead_tail(behind=None, keep=False, inline=True)
                    # this step's iteration ran at the tail of the previous step (_ahead_tail; step() primes the first one):
                    # scale, dsigma/dW and D's transformed weights are in place, the new vectors and the spectral norms wait in
                    # the shadow - one copy makes them the live ones, behind that tail on the same stream
                    ops.copy(net.sn_live.flat, net.sn_shadow, stream=self._sn_raw[0])
                    ops.event_record(ev0, self._sn_raw[0])
                    continue
                if any(s.sn for s in net.specs):
                    with torch.cuda.stream(self._sn_streams[0]):
                        chains.run(update=True)
                        for s in single:
                            self._sn_step(net, s)
                        ops.event_record(ev0, self._sn_raw[0])
                continue
            for i, s in enumerate(net.specs):
                k = i % len(self._sn_streams)
                with torch.cuda.stream(self._sn_streams[k]):
                    self._scales[s.scope] = self._sn_step(net, s) if s.sn else None
                    if s.sn:
                        ops.event_record(ev0 + i, self._sn_raw[k])
        if real.data_ptr() != b['dis_in'].data_ptr():
            ops.copy(b['dis_in'][:B], real)                                  # my_sngan.py:278: D sees [real ; fake]
        gen_wino = any(net is self.gen for _, _, net in self._wino.values())
        # G's transformed weights: waited for where the first layer that reads them starts (a dense first layer runs beside the
        # transform, which itself waits for the previous step's Adam)
        self._gen_wino_wait = None
        if gen_wino:
            first = next(i for i, s in enumerate(self.gen.specs) if s.scope in self._wino)
            if self._queue_opt and first > 0:
                self._gen_wino_wait = first
            else:
                ops.event_wait(_EV_WINO_GEN, main)
        self.generate(z, is_training=True)                                   # writes dis_in[B:]
        if self._wino:
            ops.event_wait(_EV_WINO_DIS, main)                               # D's transformed weights of this step
        x = b['dis_in']
        waited = False
        for i, s in enumerate(self.dis.specs):
            if s.sn and not (self._sn_fused and waited):
                # a layer waits for ITS power iteration only, not for both chains (measured: no difference at CIFAR B=64, where
                # the chains finish under G's forward pass; tried with it: D's real half as a separate half-batch pass on the
                # parameter-gradient stream underneath G's forward pass - 3.05 instead of 2.33 ms per step, dropped; again in round 4 with
                # the reduction-split launches and the split-K head: 1.92 instead of 1.87 ms - a half-batch pass of D takes 230-250 us
                # against 325 for the whole batch, more than G's forward pass leaves idle)
                ops.event_wait(_EV_SN0 + (0 if self._sn_fused else i), main)
                waited = True
            scale = self._scales[s.scope]
            x = self._layer_forward(self.dis, s, x, True, scale)
            if s.out_reshape is not None:
                x = x.view(_native_shape(s.out_shape_ref, x.shape[0]))
        if not self._queue_opt:
            for st in self._sn_raw:
                ops.stream_wait(main, st)                                    # the chains' tails (x <- normalised F^T(y))
        # (_queue_opt: nothing before the parameter gradients reads what the chains' tails write - the updated vectors,
        # dsigma/dW - so the weight-gradient stream joins them instead, at the start of the backward pass: _backward_dis)
        scores = x                                                           # [2B, d]; s_x = [:B], s_gen = [B:]
        self._loss.launch(scores, self.losses)
        return scores

    # ---------------------------------------------------------------------------------------
    def _backward_dis(self):
        """loss_dis -> D parameters (rows 0..2B-1) and loss_gen -> d(fake images) (rows 2B..3B-1, the fake
        half, input-gradients only), back-propagated through D together: one 3B-row dgrad launch per
        layer (the last B rows take their activation derivative from the fake half's activations),
        weight / bias gradients from the first 2B rows only."""
        B, b, net = self.B, self.buf, self.dis
        # the loss kernel wrote [dLd/ds_x (real rows, my_sngan.py:278-279) ; dLd/ds_gen ; dLg/ds_gen ; dLg/ds_x]:
        # its first 3B rows are the score gradient of the 3B-row backward pass
        specs = net.specs
        dz = b['mmd_grads'].view(4 * B, -1)[:3 * B]
        if self._queue_opt:
            # the power-iteration chains' tails: their readers are the gradient fix-ups and Adam, all of which run on the
            # weight-gradient stream or behind it (the main stream joins that stream before its own Adam; the exchange stream
            # IS the first power-iteration stream)
            for st in (self._sn_raw[:1] if self._sn_fused else self._sn_raw):
                ops.stream_wait(self._wg_raw, st)
        held = None
        for li in range(len(specs) - 1, -1, -1):
            s = specs[li]
            x_in = b['dis_in'] if li == 0 else b[specs[li - 1].scope + '#y']
            w = net.p(s.scope + '/kernel/kernel')
            gw = net.g(s.scope + '/kernel/kernel')
            scale = self._scales[s.scope]
            dz_main = dz[:2 * B]

            def param_grads(s=s, x_in=x_in, w=w, gw=gw, scale=scale, dz_main=dz_main):
                gb = net.g(s.scope + '/bias/bias') if s.has_bias else None
                # a spectrally normalised kernel: the arena keeps the RAW gradient plus <G, W>; the fix-up is
                # applied where the gradient is read (Network.opt / effective_grad) - no pass over the kernel for it
                dot = net.state[s.scope + '#dot'] if s.sn else None
                if s.op == 'd':
                    if gb is not None:
                        ops.colsum(dz_main.reshape(-1, dz_main.shape[-1]), out=gb)
                    ops.gemm(x_in.reshape(2 * B, -1), dz_main.reshape(2 * B, -1), trans_a=True, out=gw, out_zeroed=True)
                    if s.sn:
                        ops.dot(gw.view(-1), w.view(-1), out=dot)
                else:                                                        # bias gradient (and <G, W>) ride on the wgrad launch
                    ops.conv2d_wgrad(x_in, dz_main, s.R, s.stride, out=gw, dbias=gb, w=w if s.sn else None, dot=dot)
                    self._wgrad_run_after()                                  # (that launch carried, or followed, the previous reduction)
                if s.sn and not net.opt.fold_fixup:                          # data-parallel replicas fix up before the exchange
                    fix = lambda: ops.sn_wgrad_fixup(gw.view(-1), net.state[s.scope + '#dsigma'].view(-1), dot,
                                                     net.state[s.scope + '#sigma'], scale)
                    if s.op == 'd':
                        fix()
                    else:                # the fix-up reads the SUMMED gradient and <G, W>: behind the next weight-gradient launch, or a flush
                        self._wg_after.append((fix, torch.cuda.current_stream()))
            # D's dense head sits between the loss and the first convolution of the backward pass with nothing beside it: its
            # parameter gradients wait for the NEXT layer's hand-over to the weight-gradient stream (one marker in the main queue -
            # ~6 us of idle queue each - instead of two in a row; its input-gradient, which the main chain needs, goes first)
            hold = self._queue_opt and self._side_wgrad and s.op == 'd' and li == len(specs) - 1 and li > 0
            if held is not None:
                fn, lj = held
                held = None
                self._on_wg_stream(lambda fn=fn, pg=param_grads: (fn(), pg()), s)
                self._exchange(net, lj)
            elif not hold:
                self._on_wg_stream(param_grads, s)
            if hold:
                held = (param_grads, li)
            if li > 0:
                if not hold:
                    self._exchange(net, li)
                prev = specs[li - 1]
                dprev, yprev = b[prev.scope + '#dz'], b[prev.scope + '#y']
                if s.op == 'd':
                    ops.gemm(dz.reshape(3 * B, -1), w, trans_b=True, scale=scale, act=prev.act,
                             dact_of=yprev.view(2 * B, -1), dact_rows=2 * B, out=dprev.view(3 * B, -1))
                else:
                    ops.conv2d_dgrad(dz, w, (yprev.shape[1], yprev.shape[2]), s.stride, scale=scale, act=prev.act,
                                     dact_of=yprev, dact_batch=2 * B, out=dprev,
                                     wino=self._wino.get(s.scope, (None, None, None))[1])
                dz = dprev
            else:
                # below D l1 sits G's output: only the loss_gen rows go further, with G's last
                # activation derivative (tanh') unless G ends in BN (then bn_bwd applies it)
                gen_last = self.gen.specs[-1]
                act_prev = 'linear' if gen_last.bn else gen_last.act
                yprev, out, dzg = b['dis_in'][B:], b['d_fake'], dz[2 * B:]
                if s.op == 'd':
                    ops.gemm(dzg.reshape(B, -1), w, trans_b=True, scale=scale, act=act_prev,
                             dact_of=yprev.reshape(B, -1), out=out.view(B, -1))
                else:
                    ops.conv2d_dgrad(dzg, w, (yprev.shape[1], yprev.shape[2]), s.stride, scale=scale, act=act_prev,
                                     dact_of=yprev, out=out)
                # D's last bucket goes AFTER the launch above: D's early Adam (which follows that bucket on the exchange
                # stream) rewrites the kernel this input-gradient reads
                self._exchange(net, li)
        return b['d_fake']                                                   # gradient w.r.t. G's last pre-activation

    def _wgrad_run_after(self):
        """what waits for a finished (summed) weight gradient - queued by param_grads under data parallelism - is issued on
        the current stream; call behind a weight-gradient launch or a flush"""
        todo, self._wg_after = self._wg_after, []
        for fn, st in todo:                              # on the stream its gradient (and that gradient's reduction) was issued on
            with torch.cuda.stream(st):
                fn()

    def _wgrad_flush(self):
        ops.wgrad_flush()
        self._wgrad_run_after()

    def _on_wg_stream(self, fn, spec):
        """run the parameter-gradient launches of one layer on the weight-gradient stream (ordered after
        everything issued so far on the current stream).  That includes the thin first / last layers, whose weight
        gradients put their partial sums in the library's shared workspace, and the Winograd-domain weight gradients
        with their per-split slabs: in the backward pass this stream is the workspace's only user - every Winograd
        launch of the main stream gets weights transformed at step start, never an in-call transform."""
        if not self._side_wgrad:
            fn()
            return
        ops.stream_wait(self._wg_raw, ops._stream())
        with torch.cuda.stream(self._wg_stream):
            fn()

    def _join_wg_stream(self):
        self._wgrad_flush()                    # the last slab weight gradient of the pass: its reduction as a stand-alone launch
        if self._side_wgrad:
            ops.stream_wait(ops._stream(), self._wg_raw)

    def _backward_gen(self, dz, z):
        B, b, net = self.B, self.buf, self.gen
        specs = net.specs
        deferred = []
        for li in range(len(specs) - 1, -1, -1):
            s = specs[li]
            in_shape = _native_shape(s.in_shape_ref, B)
            x_in = (z if li == 0 else b[specs[li - 1].scope + '#y']).view(in_shape)
            w = net.p(s.scope + '/kernel/kernel')
            gw = net.g(s.scope + '/kernel/kernel')
            scale = self._scales[s.scope]                                    # None unless the layer is spectrally normalised
            if s.bn:                                                         # dz is d/d(BN output after act)
                raw, y = b[s.scope + '#raw'], b[s.scope + '#y']
                C = raw.shape[-1]
                draw = b[s.scope + '#dz']
                # (relu / lrelu / linear: the activation's sign is recomputed from the pre-BN values instead of reading y back)
                resign = self._bn_resign and s.act in ('linear', 'relu', 'lrelu')
                ops.bn_bwd(raw.view(-1, C), None if resign else y.view(-1, C), dz.view(-1, C), net.p(s.scope + '/BN/BN/gamma'),
                           b[s.scope + '#mean'], b[s.scope + '#invstd'], act=s.act, dgamma=net.g(s.scope + '/BN/BN/gamma'),
                           dbeta=net.g(s.scope + '/BN/BN/beta'), out=draw.view(-1, C), workspace=self._bn_totals[s.scope][1],
                           beta=net.p(s.scope + '/BN/BN/beta') if resign else None)
                dz = draw
            dz = dz.view(_native_shape(s.op_out_ref, B))

            def param_grads(s=s, x_in=x_in, gw=gw, dz=dz, w=w, scale=scale):
                gb = net.g(s.scope + '/bias/bias') if s.has_bias else None
                dot = net.state[s.scope + '#dot'] if s.sn else None          # raw gradient + <G, W>, as in D
                if gb is not None and s.op != 'c':
                    ops.colsum(dz.view(-1, dz.shape[-1]), out=gb)
                if s.op == 'd':
                    ops.gemm(x_in, dz, trans_a=True, out=gw, out_zeroed=True)   # the gradient arena was zeroed
                    if s.sn:
                        ops.dot(gw.view(-1), w.view(-1), out=dot)
                elif s.op == 'c':
                    ops.conv2d_wgrad(x_in, dz, s.R, s.stride, out=gw, dbias=gb, w=w if s.sn else None, dot=dot)
                else:                                                        # tc: W[R,R,Cout,Cin]; y = dgrad(v, W)
                    ops.conv2d_wgrad(dz, x_in, s.R, s.stride, out=gw, w=w if s.sn else None, dot=dot)
                if s.op != 'd':
                    self._wgrad_run_after()
                if s.sn and not net.opt.fold_fixup:
                    fix = lambda: ops.sn_wgrad_fixup(gw.view(-1), net.state[s.scope + '#dsigma'].view(-1), dot,
                                                     net.state[s.scope + '#sigma'], scale)
                    if s.op == 'd':
                        fix()
                    else:
                        self._wg_after.append((fix, torch.cuda.current_stream()))
            if li < self._gen_tail_on_main and li > 0:
                # the tail of G's backward pass: the input-gradient chain of the main stream ends at layer 1 while the
                # weight-gradient stream still holds the gradients of the layers above - the last layers' parameter gradients
                # go behind their own input-gradient on the main stream instead (the workspace is handed over by
                # workspace_acquire: one user at a time)
                deferred.append((param_grads, li))
            elif li == 0 and deferred:
                for fn, lj in deferred:
                    fn()
                    self._exchange(net, lj, on_main=True)
                param_grads()
                self._exchange(net, li, on_main=True)
            else:
                self._on_wg_stream(param_grads, s)
                self._exchange(net, li)
            if li > 0:
                prev = specs[li - 1]
                # a BN layer below gets d/d(its activated output) and applies act' itself in bn_bwd;
                # otherwise the epilogue multiplies by act'(y_prev) and the result is d/d(pre-activation)
                yprev = b[prev.scope + '#y'].view(in_shape)
                # (a linear layer below has derivative 1: no dact either - which also lets a launch with few tiles split its
                # reduction into the zeroed buffer: G l2's input-gradient, 128 tiles, 80 -> 45 us)
                act_prev, dact = ('linear', None) if (prev.bn or prev.act == 'linear') else (prev.act, yprev)
                dprev = b[prev.scope + ('#dy' if prev.bn else '#dz')].view(in_shape)
                if s.op == 'd':
                    # dprev is NOT on the step's zero list: no out_zeroed, so no split-K accumulation into last
                    # step's values (a dense layer above a BN dense layer meets every other split condition)
                    ops.gemm(dz, w, trans_b=True, scale=scale, act=act_prev, dact_of=dact, out=dprev)
                elif s.op == 'c':
                    ops.conv2d_dgrad(dz, w, (in_shape[1], in_shape[2]), s.stride, scale=scale, act=act_prev, dact_of=dact,
                                     out=dprev, wino=self._wino.get(s.scope, (None, None, None))[1])
                else:                                                        # d/dv of dgrad(v, W) = conv(dz, W)
                    # few tiles (M = B*h*w is small at the top of G): if the epilogue is linear let the
                    # kernel split its K = R*R*Cout reduction into a buffer zeroed at step start
                    zeroed = dact is None and act_prev == 'linear' and dprev.data_ptr() in self._zeroed_ptrs
                    ops.conv2d_fwd(dz, w, s.stride, scale=scale, act=act_prev, dact_of=dact, out=dprev, out_zeroed=zeroed,
                                   wino=self._wino.get(s.scope, (None, None, None))[0])
                dz = dprev

    # ---------------------------------------------------------------------------------------
    # data-parallel gradient exchange (the reference's dormant tower helper, graph_func.py:69-94)
    # ---------------------------------------------------------------------------------------
    # ---------------------------------------------------------------------------------------
    # the step boundary, pipelined (VERDICT r04 "next" #2): D's power iterations and D's Winograd weight transform depend on
    # D's weights only, and those are final as soon as D's early Adam has run - ~270 us before the step ends
    # ---------------------------------------------------------------------------------------
    def _dis_sn_zero(self):
        """what D's power iterations accumulate into: zeroed in front of them"""
        return [self.dis.sn_scratch.flat] + list(self._sn_chains[id(self.dis)][0].zero_each_step)

    def _ahead_inline(self):
        return self.launch_mode == 'graph' and self.dist_group is None

    def _ahead_tail(self, behind=None, prezeroed=True, keep=True, inline=False):
        """D's power iteration and transformed weights FOR THE NEXT STEP, on the first power-iteration stream behind D's Adam
        of this one (`behind`: the stream that Adam was issued on).  The iteration reads the live vectors and writes scale,
        dsigma/dW (read next by the next step's Adam) and - into the shadow - the new vectors and the spectral norms, which the
        next step's head commits: between steps `in_rand` and sigma read as if the iteration ran inside the next step
        (math_func.py:661-672, 739-744)."""
        if not self._ahead or (self._ahead_inline() and not inline):
            return
        sn0 = self._sn_raw[0]
        if behind is not None and behind != sn0:
            ops.stream_wait(sn0, behind)
        with torch.cuda.stream(self._sn_streams[0]):
            if keep:
                # scale and dsigma/dW as this step's Adam read them, for effective_grad() (tests, inspection): the iteration
                # below replaces them.  23 MB for CIFAR's D, on this side stream beside G's backward pass
                ops.copy(self._sn_prev[0], self.dis.sn_scales.flat, stream=sn0)
                ops.copy(self._sn_prev[1], self.dis.sn_scratch.flat, stream=sn0)
                ops.event_record(_EV_AHEAD_KEPT, sn0)
            if prezeroed:
                ops.memset_zero_multi(self._dis_sn_zero(), stream=sn0)
            self._sn_chains[id(self.dis)][0].run(update=True)
            self._wino_jobs[1].run(stream=sn0)
        self._ahead_valid = not inline
        self._tail_issued = not inline                   # (the step body DID leave a tail: step() checks it, below)
        self.dis.readout = self._sn_readout if keep else None

    def _prime_ahead(self):
        """the tail a previous step would have left: before the first step, and after anything that changed D's weights or
        vectors from outside (set_variables, load_state_dict, a snapshot restored).  Issued eagerly, outside any recording."""
        if not self._ahead or self._ahead_valid or self._ahead_inline():
            return
        with self._handle:
            # (the live values are where the commit of the next step's head takes them FROM the shadow: the un-touched entries
            # of the shadow must equal the live ones)
            self.dis.sn_shadow.copy_(self.dis.sn_live.flat)
            self._ahead_tail(behind=ops._stream(), prezeroed=False, keep=False)
            ops.stream_wait(ops._stream(), self._sn_raw[0])

    def touch(self):
        """call BEFORE writing this engine's weight / state tensors directly (net.params, net.state[...] - set_variables and
        load_state_dict do it themselves): waits for the side-stream tail of the last step and discards what it prepared for
        the next one from the old weights (D's spectral norms, dsigma/dW, transformed weights); the next step prepares again"""
        self._sync_ahead(invalidate=True)
        self.dis.readout = None

    def _sync_ahead(self, invalidate=False):
        """before anything outside a step touches D's weights or power-iteration state: the tail of the last step may still be
        running on its side stream.  invalidate: what it produced no longer belongs to the state (the next step primes again)."""
        if self._ahead:
            ops.stream_wait(ops._stream(), self._sn_raw[0])
            if invalidate:
                self._ahead_valid = False

    def _dp_active(self):
        return self.dist_group is not None and (self.world > 1 or self._dp_force)   # MMDGAN_DP_FORCE=1: one-rank plumbing test

    def _exchange(self, net, li, on_main=False):
        """called right after the parameter gradients of layer `li` of `net` have been issued: if that completes an
        exchange bucket (_make_buckets), its SUM all-reduce starts NOW on the exchange stream and travels underneath
        the backward kernels still to come - only the last bucket of G (its first dense layer, 4 MB for CIFAR) is
        exposed.  The collectives are issued as blocking ones on a stream of our own choosing - ProcessGroupNCCL then
        runs them there, not on its internal stream, whose hardware queue it shares with whichever of our streams
        happens to map to it (streams.py).  The stream used is the first power-iteration stream: idle from the start
        of D's forward pass to the next step, on a queue of its own.  Averaging is Adam's grad_scale = 1/world."""
        if not self._dp_active():
            return
        bucket = next((b for b in self._grad_buckets[id(net)] if b[0] == li), None)
        if bucket is None:
            return
        _, lo, hi = bucket
        self._wgrad_flush()                  # (a slab weight gradient of the bucket may still be waiting for its reduction)
        # the bucket is complete once the parameter-gradient stream has drained what it holds now (and the main stream
        # has reached this point, for gradients that stay there)
        if self._side_wgrad:
            # every parameter gradient is issued there, and that stream was ordered behind the main stream (batch-norm
            # gradients of this layer included) when the layer's launches went in (_on_wg_stream): no second marker in the
            # main queue - six of them per step cost the one-rank exchange 0.05 ms
            ops.stream_wait(self._comm_raw, self._wg_raw)
        if on_main or not self._side_wgrad:               # (the tail of G's backward pass puts its gradients on the main stream)
            ops.stream_wait(self._comm_raw, ops._stream())
        lib = ops.require_device()
        if self._recording and self._dp_backend != 'capi':
            lib.mmdgan_plan_mark()                       # the collective is not the library's: a segment boundary
            self._plan_collectives.append((net, lo, hi))
        last = bucket is self._grad_buckets[id(net)][-1]
        if self.bucket_probe is not None and not self._recording:
            # tools/scale_predict.py: when (on the exchange stream, i.e. behind the gradients it waits for) each bucket is READY
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self._comm_stream)
            self.bucket_probe.append((net.name, 4 * (hi - lo), ev))
        probe = self.exchange_probe if (last and net is self.gen and not self._recording) else None
        if probe is not None:                            # tools/scale.sh: how long G's last bucket takes and how much of it is exposed
            probe['bucket_bytes'] = 4 * (hi - lo)
            probe['start'] = torch.cuda.Event(enable_timing=True)
            probe['start'].record(self._comm_stream)
        self._issue_collective(net, lo, hi)
        if probe is not None:
            probe['end'] = torch.cuda.Event(enable_timing=True)
            probe['end'].record(self._comm_stream)
        if last and net is self.dis and self._early_d_adam:
            # D's exchange is complete and nothing in G's backward pass reads D's weights: its Adam runs on the exchange
            # stream, beside G's backward pass, instead of at the tail of the step - behind everything the main stream
            # holds now (the last reader of D's weights, the input-gradient of D's first layer, was issued just before)
            ops.stream_wait(self._comm_raw, ops._stream())
            with torch.cuda.stream(self._comm_stream):
                self.dis.opt.step(self.lr_d, grad_scale=1.0 / self.world)
            self._d_updated_early = True
            self._ahead_tail(behind=self._comm_raw)

    def _issue_collective(self, net, lo, hi):
        if self._dp_backend == 'capi':
            ops.check(ops.require_device().mmdgan_allreduce_bucket(net.grads.data_ptr() + 4 * lo, hi - lo, self._comm_raw),
                      'allreduce_bucket')
            return
        from . import dist as mdist
        with torch.cuda.stream(self._comm_stream):
            mdist.allreduce_sum_(net.grads[lo:hi], self.dist_group)

    def _update(self):
        gs = 1.0 / self.world
        main = ops._stream()
        if self._dp_active():
            if self.bucket_probe is not None and not self._recording:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self.bucket_probe.append(('main_ready', 0, ev))          # the main stream has nothing left but the updates
            if self.exchange_probe is not None and not self._recording:
                self.exchange_probe['main_ready'] = torch.cuda.Event(enable_timing=True)
                self.exchange_probe['main_ready'].record()       # the main stream has nothing left but Adam
            ops.stream_wait(main, self._comm_raw)        # all buckets (and D's early Adam) have landed
        if not self._d_updated_early:
            self.dis.opt.step(self.lr_d, grad_scale=gs)
        self.gen.opt.step(self.lr_g, grad_scale=gs)

    def _step_body(self, z, real):
        self._tail_issued = False                        # set by _ahead_tail; a replayed plan / graph keeps its recording's value
        lib = ops.require_device()
        lib.mmdgan_set_outputs_prezeroed(1)
        main = ops._stream()
        if self.bucket_probe is not None and not self._recording:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.bucket_probe.append(('step_start', 0, ev))
        try:
            # the two gradient arenas (61 MB of memset) are first touched in the backward pass: zero them
            # on the parameter-gradient stream, underneath the forward pass
            arenas = (self.gen.grads, self.dis.grads) if self._side_wgrad else ()
            if arenas:
                ops.stream_wait(self._wg_raw, main)                          # after the previous step's Adam
                with torch.cuda.stream(self._wg_stream):
                    # G's transformed weights first (its forward pass starts right away and waits on this event),
                    # then the memsets, then D's (needed after G's forward / in the backward pass)
                    # (one launch per network: ops.WinoTransforms)
                    self._wino_jobs[0].run()
                    ops.event_record(_EV_WINO_GEN, self._wg_raw)
                    ops.memset_zero_multi(list(arenas))
                    if not self._ahead:
                        self._wino_jobs[1].run()
                    # (the step counts and bias-corrected learning rates of both updates: no gradient needed, so here)
                    ops.adam_prepare_multi([(self.dis.opt, self.lr_d), (self.gen.opt, self.lr_g)])
                    ops.event_record(_EV_WINO_DIS, self._wg_raw)
            else:                                            # no side stream (MMDGAN_SIDE_WGRAD=0): the same work on the main one
                self._wino_jobs[0].run()
                self._wino_jobs[1].run()
                ops.event_record(_EV_WINO_GEN, main)
                ops.event_record(_EV_WINO_DIS, main)
            # the small scratch buffers of the step (power-iteration scratch, batch-norm totals, split-K outputs): one launch.
            # The power iterations' scratch (dsigma/dW of every normalised kernel: 22 MB for CIFAR's D) is zeroed on THEIR stream,
            # in front of the chains (_forward) - the main stream's first layer does not wait for it
            small = [t for t in self._zero_each_step if not any(t is a for a in arenas)]
            self._sn_zero = []
            if self._queue_opt and self._sn_fused:
                sn_flat = [net.sn_scratch.flat for net in (self.gen, self.dis)] + [t for c in self._sn_chains.values() for t in c[0].zero_each_step]
                self._sn_zero = [t for t in small if any(t is f for f in sn_flat)]
                small = [t for t in small if not any(t is f for f in sn_flat)]
                if self._ahead:                          # D's scratch is zeroed where D's iteration runs: _ahead_tail
                    self._sn_zero = [t for t in self._sn_zero if not any(t is f for f in self._dis_sn_zero())]
            if small:
                ops.memset_zero_multi(small)
            self._in_step = True
            self._d_updated_early = False
            # the Winograd-domain weight gradients of the backward pass form a chain on the weight-gradient stream: each sums
            # its predecessor's slabs in its own prologue instead of a bandwidth-only launch in between (mmdgan_wgrad_defer)
            ops.wgrad_defer(self._wgrad_defer)
            self._forward(z, real)
            if arenas and not (self._queue_opt and self._wino):
                # (with transformed weights in the step the main stream has already waited for _EV_WINO_DIS, recorded behind
                # everything the weight-gradient stream does at step start)
                ops.stream_wait(main, self._wg_raw)
            dz = self._backward_dis()
            if self._early_d_adam and not self._dp_active() and self._side_wgrad:
                # D's gradients are complete once the parameter-gradient stream has drained what it holds now and
                # the main stream has reached this point (thin layers); nothing in G's backward pass reads D's
                # weights, so D's Adam runs there, beside G's backward pass, instead of at the tail of the step
                self._wgrad_flush()
                ops.stream_wait(self._wg_raw, main)
                with torch.cuda.stream(self._wg_stream):
                    self.dis.opt.step(self.lr_d, grad_scale=1.0)
                self._d_updated_early = True
                self._ahead_tail(behind=self._wg_raw)
            self._backward_gen(dz, z)
            self._join_wg_stream()
            if self._ahead and self._d_updated_early and not self._ahead_inline():
                ops.event_wait(_EV_AHEAD_KEPT, main)     # (readers of the last step's scale / dsigma follow the main stream)
            self._update()
        finally:
            self._in_step = False
            self._wg_after = []
            lib.mmdgan_wgrad_defer(0)
            lib.mmdgan_set_outputs_prezeroed(0)

    def step(self, real_nhwc=None, z=None, uni=None):
        """one training step; returns nothing on the host (losses stay in self.losses on the device:
        [0] loss_gen, [1] loss_dis, [2..6] e_kxx e_kxy e_kyy e_kxx_b e_kyy_b, pre-update values).
        uni: the `*_mix` losses' uniform draw of this step ([B]; default: sampled like z)"""
        if z is None:
            self._static_z.normal_(generator=self._z_gen)                    # my_sngan.py:123-124
        else:
            self._static_z.copy_(z)
        self._loss.draw(self._z_gen, uni)                                    # math_func.py:2079 (the *_mix coin)
        mode = self.launch_mode
        if mode == 'graph' and self.dist_group is not None:
            mode = 'eager'                               # a collective cannot sit inside the captured graph
        if (self.lr_d, self.lr_g) != self._baked_lr:     # a captured graph / recorded plan holds the learning rates by value
            self._baked_lr = (self.lr_d, self.lr_g)
            self._drop_recordings()
        self._prime_ahead()
        with self._handle:                               # this engine's workspace / prezeroed mode / plans
            if mode == 'eager':
                # the batch goes straight into the first half of D's input buffer (one copy, not two)
                self._step_body(self._static_z, real_nhwc if real_nhwc is not None else self._static_real)
            else:
                if real_nhwc is not None:
                    self._static_real.copy_(real_nhwc)                       # graph / plan read this buffer
                if mode == 'graph':
                    if self._graph is None:
                        self._capture()
                    else:
                        self._graph.replay()
                else:
                    self._plan_step()
        if self._ahead and not self._ahead_inline():
            # (host-side facts about what the step just issued - set here, not inside the recorded body: a replayed plan or
            # graph runs the tail's launches without running _ahead_tail.)  The tail is issued where D's early Adam is; a
            # schedule whose body - run or recorded - skipped it would commit a stale shadow next step: fail here instead
            if not self._tail_issued:
                raise RuntimeError('pipelined step boundary: the step body issued no tail for the next step '
                                   '(D\'s early Adam did not run where _ahead_tail is issued)')
            self._ahead_valid = True
            self.dis.readout = self._sn_readout
        self.global_step += 1                                                # tied to the D update, my_sngan.py:424

    def _plan_step(self):
        """record the step once (an ordinary eager step that the library notes down, include/mmdgan_hip.h "Launch
        plans"), replay it from one C call afterwards.  Under data parallelism the plan is cut where the exchange
        collectives go; they are issued between the segments, on the exchange stream, exactly as in the eager step."""
        lib = ops.require_device()
        main = ops._stream()
        if self._plan is not None and self._plan_stream != main:
            lib.mmdgan_plan_destroy(self._plan)          # recorded for another stream: record again
            self._plan = None
        if self._plan is None:
            self._plan_collectives = []
            self._handle.forget_workspace_users()        # (step start: every stream of the previous step has been joined)
            ops.check(lib.mmdgan_plan_begin(), 'plan_begin')
            self._recording = True
            try:
                self._step_body(self._static_z, self._static_real)
            except Exception:
                lib.mmdgan_plan_abort()
                raise
            finally:
                self._recording = False
            import ctypes
            pid = ctypes.c_int(-1)
            ops.check(lib.mmdgan_plan_end(ctypes.byref(pid)), 'plan_end')
            self._plan, self._plan_stream = pid.value, main
            return
        if not self._plan_collectives:
            ops.check(lib.mmdgan_plan_replay(self._plan, -1), 'plan_replay')
            return
        for i, (net, lo, hi) in enumerate(self._plan_collectives):
            ops.check(lib.mmdgan_plan_replay(self._plan, i), 'plan_replay')
            self._issue_collective(net, lo, hi)
        ops.check(lib.mmdgan_plan_replay(self._plan, len(self._plan_collectives)), 'plan_replay')

    def _capture(self):
        # warm-up on a side stream (allocates lazily-created buffers), then capture one step
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        snap = self._snapshot()
        with torch.cuda.stream(s):
            self._step_body(self._static_z, self._static_real)
        torch.cuda.current_stream().wait_stream(s)
        self._restore(snap)
        self._prime_ahead()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._step_body(self._static_z, self._static_real)
        self._restore(snap)
        self._prime_ahead()
        self._graph.replay()

    def _snapshot(self):
        self._sync_ahead()
        out = []
        for net in (self.gen, self.dis):
            out.append([net.params.clone(), net.adam_m.clone(), net.adam_v.clone(), net.opt.step_counter.clone(),
                        {k: v.clone() for k, v in net.state.items()}])
        out.append(self._loss.state.clone() if self._loss.mix else None)       # the *_mix coin's moving averages
        return out

    def _restore(self, snap):
        self._sync_ahead()
        for net, (p, m, v, t, st) in zip((self.gen, self.dis), snap):
            net.params.copy_(p); net.adam_m.copy_(m); net.adam_v.copy_(v); net.opt.step_counter.copy_(t)
            for k, val in st.items():
                net.state[k].copy_(val)
        if snap[-1] is not None:
            self._loss.state.copy_(snap[-1])
        self._ahead_valid = False

    # ---------------------------------------------------------------------------------------
    # reference-layout import / export (checkpoint keys follow TF scopes)
    # ---------------------------------------------------------------------------------------
    def variable_names(self, trainable_only=False):
        return self.gen.variable_names(trainable_only) + self.dis.variable_names(trainable_only)

    def _net_of(self, name):
        return self.gen if name.startswith('gen/') else self.dis

    def set_variables(self, values):
        self._sync_ahead(invalidate=True)
        self.dis.readout = None
        for k, v in values.items():
            self._net_of(k).set_variable(k, v)

    def get_variables(self, names=None, grad=False):
        names = names if names is not None else self.variable_names(trainable_only=grad)
        return OrderedDict((k, self._net_of(k).get_variable(k, grad=grad)) for k in names)

    def set_adam_state(self, m, v, t):
        """Adam moments (reference names and layouts, as set_variables takes them) and the step count both
        optimisers have taken: resume mid-run from state recorded elsewhere"""
        for k in m:
            net = self._net_of(k)
            net.arena.view(k, net.adam_m).copy_(torch.as_tensor(net._to_native(k, m[k]), device=self.device))
            net.arena.view(k, net.adam_v).copy_(torch.as_tensor(net._to_native(k, v[k]), device=self.device))
        for net in (self.gen, self.dis):
            net.opt.step_counter.fill_(int(t))
        self._drop_recordings()

    def sigmas(self):
        return OrderedDict((s.scope, float(net.state[s.scope + '#sigma'].item()))
                           for net in (self.dis, self.gen) for s in net.specs if s.sn)

    def get_adam_state(self):
        """(m, v, t): the Adam moments by variable name in the reference's layouts, and the step count - what set_adam_state takes"""
        m, v = OrderedDict(), OrderedDict()
        for net in (self.gen, self.dis):
            for k in net.variable_names(trainable_only=True):
                m[k] = net._to_ref(k, net.arena.view(k, net.adam_m).detach().cpu().numpy())
                v[k] = net._to_ref(k, net.arena.view(k, net.adam_v).detach().cpu().numpy())
        return m, v, int(self.dis.opt.step_counter.item())

    def state_dict(self):
        """format 2: the Adam moments travel BY VARIABLE NAME in the reference's layouts, like the variables - a checkpoint
        does not depend on how the engine lays its arenas out (format 1 stored the flat arenas, which the `#sn_dot` scratch
        entry at their head shifted)"""
        m, v, t = self.get_adam_state()
        return {'format': 2, 'global_step': self.global_step, 'variables': self.get_variables(),
                'loss_state': self._loss.state_dict(), 'adam_m': m, 'adam_v': v, 'adam_t': t}

    def load_state_dict(self, sd):
        self.set_variables(sd['variables'])
        self.global_step = int(sd['global_step'])
        self._loss.load_state_dict(sd.get('loss_state', {}))
        if sd.get('format', 1) >= 2:
            self.set_adam_state(sd['adam_m'], sd['adam_v'], sd['adam_t'])
        else:                                            # format 1: flat arenas of the layout that wrote them
            for tag, net in (('gen', self.gen), ('dis', self.dis)):
                if sd[tag + '/adam_m'].numel() != net.adam_m.numel():
                    raise ValueError('checkpoint format 1 holds the Adam moments of %s as a flat arena of %d floats, this build\'s '
                                     'arena has %d: re-save it with the build that wrote it' %
                                     (tag, sd[tag + '/adam_m'].numel(), net.adam_m.numel()))
                net.adam_m.copy_(sd[tag + '/adam_m'])
                net.adam_v.copy_(sd[tag + '/adam_v'])
                net.opt.step_counter.fill_(int(sd[tag + '/adam_t']))
        self._drop_recordings()

    def plan_kernels(self):
        """the kernel launches of the recorded step, in issue order: [(kernel, workgroups, threads, stream number)] -
        launch_mode 'plan', after the first step (ops.plan_kernels).  The reference's step is one fixed graph
        (graph_func.py:851-854); this is the build's: tests and bench.py compare it with tests/golden/production_kernels.json"""
        if self._plan is None:
            raise RuntimeError('no recorded plan: launch_mode must be "plan" and one step must have run')
        with self._handle:
            return ops.plan_kernels(self._plan)

    def _drop_recordings(self):
        """forget the captured graph and the recorded plan (both are re-made by the next step)"""
        self._graph = None
        if self._plan is not None:
            with self._handle:
                ops.require_device().mmdgan_plan_destroy(self._plan)
            self._plan = None
