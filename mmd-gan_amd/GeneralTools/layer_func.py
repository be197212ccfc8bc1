"""Net / Routine: the reference's architecture-dict front end (layer_func.py:2111-2494), kept as
an API so that `Net(design, net_name, data_format, num_class)` + `Routine(net).add_input_layers /
seq_links / add_output_layers / __call__` read as they do in my_sngan.py:85-108.

Sequential nets of default-type layers with ops d / c / tc (spectral norm and batch norm included) run on the hand-scheduled
layer chain; nets with residual blocks ('type': 'res' / 'res_i' / 'res_v1', layer_func.py:2043-2067), identity layers, scaling
ops or an input reshape are lowered to primitive ops (mmdgan_hip.tape) - every architecture SNGan accepts also runs here;
anything else raises the reference's error for an unsupported op or type.  Tensors crossing this
API are NCHW like the reference's (misc_fun.py:50-51); inside they are NHWC and every op is a HIP
kernel (mmdgan_hip.ops).  Training uses mmdgan_hip.engine.GanEngine (preallocated buffers, one
hipGraph); Routine.__call__ is the eager inference / inspection path.
"""
import numpy as np
import torch

from GeneralTools.misc_fun import FLAGS
from mmdgan_hip import ops
from mmdgan_hip.engine import Network, build_specs, _native_shape, sn_power_iteration, sn_scratch_buffers


class Net(object):
    primitive, lowered = False, None     # the design needs primitive ops / its mmdgan_hip.tape.NetForward (set by Routine)

    def __init__(self, net_design, net_name='net', data_format=None, num_class=0):
        if num_class not in (0, 1):
            raise NotImplementedError('{}: conditional layers are outside the hot path'.format(net_name))
        if data_format not in (None, 'channels_first', 'NCHW'):
            raise NotImplementedError('{}: the API seam is NCHW (FLAGS.IMAGE_FORMAT default)'.format(net_name))
        self.net_def, self.net_name, self.num_layers = net_design, net_name, len(net_design)
        self.specs = None            # filled by Routine.add_input_layers (shape inference needs the input shape)
        self.network = None

    @property
    def layers(self):
        return self.specs


class Routine(object):
    def __init__(self, net_object):
        self.net = net_object
        self.layer_indices, self.output_layer_indices, self.output_added = [], [], False
        self._sn_bufs = {}

    def add_input_layers(self, input_shape, out_layer_indices):
        """input_shape = [batch, features] or [batch, C, H, W]; only dims [1:] matter (layer_func.py:694)."""
        if out_layer_indices != [0]:
            raise NotImplementedError('only sequential routines are on the hot path')
        from mmdgan_hip.tape import needs_primitive_ops
        self.net.in_shape_ref = list(input_shape[1:])
        self.net.primitive = needs_primitive_ops(self.net.net_def)
        if self.net.primitive:
            self.net.specs = list(self.net.net_def)      # lowered on first use (needs the device)
        else:
            self.net.specs = build_specs(self.net.net_def, list(input_shape[1:]), self.net.net_name,
                                         FLAGS.SPECTRAL_NORM_MODE)
        self.layer_indices.append(0)

    def seq_links(self, in_layer_indices):
        if self.net.specs is None:
            raise NotImplementedError('Input layer {} has not been defined yet.'.format(in_layer_indices[0]))
        if list(in_layer_indices) != list(range(self.net.num_layers)):
            raise NotImplementedError('only sequential routines are on the hot path')
        self.layer_indices = list(in_layer_indices)

    def add_output_layers(self, in_layer_indices):
        for idx in in_layer_indices:
            if idx in self.output_layer_indices:
                raise AttributeError('Layer {} has already been added as output layer.'.format(idx))
            self.output_layer_indices.append(idx)
        self.output_added = True

    def _ensure_network(self, device):
        if self.net.network is None:
            if self.net.primitive:                       # residual blocks / scaling ops: the primitive-op lowering
                from mmdgan_hip.tape import NetForward
                self.net.lowered = NetForward(self.net.net_def, self.net.in_shape_ref, self.net.net_name, device,
                                              FLAGS.SPECTRAL_NORM_MODE, FLAGS.WEIGHT_INITIALIZER)
                self.net.network = self.net.lowered.net  # same variable interface: set_variable / get_variable / state
            else:
                self.net.network = Network(self.net.specs, device, np.random.RandomState(0), FLAGS.WEIGHT_INITIALIZER)
        return self.net.network

    def __call__(self, routine_inputs, is_training=True):
        """eager forward; accepts a tensor or {'x': tensor} (NCHW or [B,F]) and returns {'x': ...}."""
        if not self.output_added:
            raise NotImplementedError('Output layer has not been defined.')
        x = routine_inputs['x'] if isinstance(routine_inputs, dict) else routine_inputs
        net = self._ensure_network(x.device)
        if self.net.lowered is not None:
            assert list(x.shape[1:]) == self.net.in_shape_ref, \
                '{}: the input shape {} does not match existed shape {}.'.format(self.net.net_name, list(x.shape[1:]),
                                                                                self.net.in_shape_ref)
            y = self.net.lowered(ops.nchw_to_nhwc(x.contiguous()) if x.dim() == 4 else x.contiguous(), is_training)
            # a 2-D result is the lowered net's cached buffer (keyed by batch size): the next call with that batch size would
            # overwrite what this one returned - s_x = D(x)['x']; s_gen = D(G(z))['x'] - so it leaves as a copy
            return {'x': ops.nhwc_to_nchw(y.contiguous()) if y.dim() == 4 else y.clone()}
        specs = net.specs
        assert list(x.shape[1:]) == specs[0].in_shape_ref, \
            '{}: the input shape {} does not match existed shape {}.'.format(specs[0].scope, list(x.shape[1:]),
                                                                            specs[0].in_shape_ref)
        x = ops.nchw_to_nhwc(x.contiguous()) if x.dim() == 4 else x.contiguous()
        n = x.shape[0]
        for s in specs:
            x = x.view(_native_shape(s.in_shape_ref, n))
            w = net.p(s.scope + '/kernel/kernel')
            bias = net.p(s.scope + '/bias/bias') if s.has_bias else None
            scale = None
            if s.sn:
                # _get_weight_norm_ (layer_func.py:785-825): sigma from one power-iteration step on the stored vector,
                # W_eff = W * act_k / sigma (:886-887).  The vector's update is an UPDATE_OP (math_func.py:744), which
                # the reference's sessions run in training only - as this eager path does with BN's moving statistics
                if s.scope not in self._sn_bufs:
                    self._sn_bufs[s.scope] = sn_scratch_buffers(net, s, x.device)
                scale = sn_power_iteration(net, s, self._sn_bufs[s.scope], update=bool(is_training), out_zeroed=False)
            act = 'linear' if s.bn else s.act
            if s.op == 'd':
                y = ops.gemm(x, w, bias=bias, scale=scale, act=act)
            elif s.op == 'c':
                y = ops.conv2d_fwd(x, w, s.stride, bias=bias, scale=scale, act=act)
            else:
                y = ops.conv2d_dgrad(x, w, (x.shape[1] * s.stride, x.shape[2] * s.stride), s.stride, bias=bias, scale=scale,
                                     act=act)
            if s.bn:
                y2 = y.view(-1, y.shape[-1])
                g, b_ = net.p(s.scope + '/BN/BN/gamma'), net.p(s.scope + '/BN/BN/beta')
                mm, mv = net.state[s.scope + '/BN/BN/moving_mean'], net.state[s.scope + '/BN/BN/moving_variance']
                if is_training:
                    y2 = ops.bn_fwd_train(y2, g, b_, mm, mv, act=s.act, unbiased=(y.dim() == 4), new_moving_mean=mm,
                                          new_moving_var=mv)[0]
                else:
                    y2 = ops.bn_fwd_infer(y2, g, b_, mm, mv, act=s.act)
                y = y2.view(y.shape)
            x = y
        x = x.view(_native_shape(specs[-1].out_shape_ref, n))
        return {'x': ops.nhwc_to_nchw(x) if x.dim() == 4 else x}

    def apply(self, routine_inputs, is_training=True):
        return self.__call__(routine_inputs, is_training)
