#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the REFERENCE's own code.

TEST INFRASTRUCTURE.  Runs in the build container only: imports
/root/reference/GeneralTools/{misc_fun,math_func,layer_func}.py unmodified
through `oracle/tf1_shim.py` (TensorFlow itself is absent) and records inputs
and outputs of the hot-path functions as small fixtures.  The fixtures are
data (seeded inputs + reference outputs), never reference source text.

    python oracle/make_golden.py            # writes tests/golden/

Each fixture carries fp32 outputs (`*_f32`, what the reference computes) and
fp64 outputs (`*_f64`, same code with the shim in double precision: the value
every fp32 implementation is measured against).
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
REFERENCE = '/root/reference'

sys.path.insert(0, HERE)
import tf1_shim as tf  # noqa: E402

tf.install()
sys.path.insert(0, REFERENCE)
from GeneralTools.misc_fun import FLAGS  # noqa: E402
from GeneralTools import math_func as ref_math  # noqa: E402
from GeneralTools import layer_func as ref_layer  # noqa: E402

FLAGS.SILENT_MODE = True
DT = {'f32': torch.float32, 'f64': torch.float64}


def npy(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------
# 1. pairwise distance + rep / rmb losses (math_func.py:767, 1288, 1356, 2088)
# ---------------------------------------------------------------------------
def mmd_inputs(B, d, sig_gen, sig_x, offset, seed):
    rs = np.random.RandomState(seed)
    s_gen = (rs.randn(B, d) * sig_gen).astype(np.float32)
    s_x = (rs.randn(B, d) * sig_x + offset).astype(np.float32)
    return s_gen, s_x


def run_mmd(loss_type, s_gen_np, s_x_np, rep_weights, tag):
    out = {}
    B = s_gen_np.shape[0]
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        sg = torch.tensor(s_gen_np, dtype=dt, requires_grad=True)
        sx = torch.tensor(s_x_np, dtype=dt, requires_grad=True)
        dgg, dgd, ddd = ref_math.get_squared_dist(sg, sx)
        lg, ld = ref_math.GANLoss(False).apply(sg, sx, loss_type, batch_size=B, d=s_gen_np.shape[1],
                                               rep_weights=list(rep_weights))
        glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True)
        gld = torch.autograd.grad(ld, [sg, sx])
        out.update({'loss_gen_' + key: npy(lg), 'loss_dis_' + key: npy(ld),
                    'dLg_dsgen_' + key: npy(glg[0]), 'dLg_dsx_' + key: npy(glg[1]),
                    'dLd_dsgen_' + key: npy(gld[0]), 'dLd_dsx_' + key: npy(gld[1])})
        if B <= 64:
            out.update({'dist_gg_' + key: npy(dgg), 'dist_gd_' + key: npy(dgd), 'dist_dd_' + key: npy(ddd)})
        m = float(B)
        k = lambda dist: ref_math.matrix_mean_wo_diagonal(torch.exp(-dist / 2.0), m)
        out.update({'e_kxx_' + key: npy(k(dgg)), 'e_kxy_' + key: npy(k(dgd)), 'e_kyy_' + key: npy(k(ddd))})
        if key == 'f32':
            out['mask_gg_lt_lb'] = npy(dgg < 0.25)
            out['mask_dd_gt_ub'] = npy(ddd > 4.0)
            out['mask_gd_gt_ub'] = npy(dgd > 4.0)
            off = ~np.eye(B, dtype=bool)
            margin = min(np.abs(npy(dgg)[off] - 0.25).min(), np.abs(npy(ddd)[off] - 4.0).min(),
                         np.abs(npy(dgd) - 4.0).min())
            out['threshold_margin'] = np.float64(margin)
    out.update({'s_gen': s_gen_np, 's_x': s_x_np, 'rep_weights': np.asarray(rep_weights, np.float64),
                'loss_type': np.asarray(loss_type)})
    return out


def make_mmd():
    cases = []
    # (B, sigma_gen, sigma_x, offset) - typical D-output scales of SURVEY 8(d)
    for B in (8, 64, 128):
        for loss_type in ('rep', 'rmb'):
            for (sg, sx, off) in ((0.25, 0.3, 0.1), (0.05, 0.05, 0.02), (1.0, 1.0, 0.2)):
                cases.append((B, loss_type, sg, sx, off, (0.0, -1.0)))
    cases.append((64, 'rep', 0.25, 0.3, 0.1, (-1.0, -2.0)))       # math_func.py:2116 alternative weights
    cases.append((64, 'rmb', 0.5, 0.6, 0.1, (-1.0, -2.0)))
    cases.append((64, 'rmb', 0.5, 0.6, 0.1, (1.0, 0.0)))          # w0>0 branch: k_xy upper bound
    n = 0
    for (B, loss_type, sg, sx, off, w) in cases:
        seed = 0
        while True:                                               # reject near-threshold seeds
            s_gen, s_x = mmd_inputs(B, 16, sg, sx, off, 1000 * n + seed)
            fx = run_mmd(loss_type, s_gen, s_x, w, '')
            if fx['threshold_margin'] > 1e-4:
                break
            seed += 1
        name = 'mmd_{}_B{}_c{:02d}.npz'.format(loss_type, B, n)
        np.savez_compressed(os.path.join(OUT, name), **fx)
        n += 1
    print('mmd fixtures:', n)


# ---------------------------------------------------------------------------
# 1b. SURVEY 8(f) row 1: the other in-kernel losses - mixture-Gaussian mmd_g (math_func.py:1435-1473, 2160-2173),
#     mgb (:2175-2193), hinge (:2137-2143), logistic (:2128-2135) - through the reference's own GANLoss
# ---------------------------------------------------------------------------
def run_loss_next(loss_type, s_gen_np, s_x_np):
    out = {}
    B = s_gen_np.shape[0]
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        sg = torch.tensor(s_gen_np, dtype=dt, requires_grad=True)
        sx = torch.tensor(s_x_np, dtype=dt, requires_grad=True)
        lg, ld = ref_math.GANLoss(False).apply(sg, sx, loss_type, batch_size=B, d=s_gen_np.shape[1])
        glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True, allow_unused=True)
        gld = torch.autograd.grad(ld, [sg, sx], allow_unused=True)
        z = lambda g, like: npy(g) if g is not None else np.zeros_like(npy(like))
        out.update({'loss_gen_' + key: npy(lg), 'loss_dis_' + key: npy(ld),
                    'dLg_dsgen_' + key: z(glg[0], sg), 'dLg_dsx_' + key: z(glg[1], sx),
                    'dLd_dsgen_' + key: z(gld[0], sg), 'dLd_dsx_' + key: z(gld[1], sx)})
        if key == 'f32' and loss_type == 'mgb':
            dgg, dgd, ddd = ref_math.get_squared_dist(sg, sx)
            off = ~np.eye(B, dtype=bool)
            out['threshold_margin'] = np.float64(min(np.abs(npy(dgg)[off] - 0.25).min(), np.abs(npy(ddd)[off] - 0.25).min(),
                                                     np.abs(npy(dgd) - 4.0).min()))
    out.update({'s_gen': s_gen_np, 's_x': s_x_np, 'loss_type': np.asarray(loss_type)})
    return out


def make_loss_next():
    n = 0
    for loss_type in ('mmd_g', 'mgb', 'hinge', 'logistic'):
        for B in (8, 64):
            for (sg, sx, off) in ((0.25, 0.3, 0.1), (1.0, 1.0, 0.2)):
                seed = 0
                while True:                                       # reject near-threshold seeds (mgb clamps)
                    s_gen, s_x = mmd_inputs(B, 16, sg, sx, off, 5000 + 1000 * n + seed)
                    fx = run_loss_next(loss_type, s_gen, s_x)
                    if fx.get('threshold_margin', 1.0) > 1e-4:
                        break
                    seed += 1
                np.savez_compressed(os.path.join(OUT, 'lossx_{}_B{}_c{:02d}.npz'.format(loss_type, B, n)), **fx)
                n += 1
    print('next-row loss fixtures:', n)


# ---------------------------------------------------------------------------
# 1c. SURVEY 8(f) row 1, the `*_mix` family: 'mmd_g_mix' / 'fixed_g_mix' (GANLoss._mmd_g_mix_, math_func.py:2195-2228)
#     and 'sgm' (_single_mmd_g_mix_, :2230-2263) with their coin (get_mix_coin :2061-2085), the group masks
#     (slice_pairwise_distance :2038-2058) and the moving-average state (:1979-2035).  The uniform draw is recorded
#     as an INPUT (`uni`), the two state variables are set to non-trivial values before the recorded call, and the
#     boolean masks are captured where the reference builds them (the arguments of mat_slice).
# ---------------------------------------------------------------------------
def run_mix(loss_type, s_gen_np, s_x_np, state_in, seed, threshold=None):
    out = {}
    B = s_gen_np.shape[0]
    captured = {}
    real_mat_slice = ref_math.mat_slice

    def spy(mat, row_index, col_index=None, name='slice'):
        captured.setdefault('calls', []).append((npy(row_index), npy(col_index) if col_index is not None else None))
        return real_mat_slice(mat, row_index, col_index, name)
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        kwargs = {} if threshold is None else {'mix_threshold': threshold}
        # a first call creates the two non-trainable variables (zero); then give them the fixture's state
        sg = torch.tensor(s_gen_np, dtype=dt, requires_grad=True)
        sx = torch.tensor(s_x_np, dtype=dt, requires_grad=True)
        ref_math.GANLoss(False).apply(sg, sx, loss_type, batch_size=B, d=s_gen_np.shape[1], **kwargs)
        tf.STATE.update_ops = []
        names = sorted(tf.STATE.variables)
        assert names == ['mmd_g_mix/coin/gen_average', 'mmd_g_mix/coin/prob'], names
        with torch.no_grad():
            tf.STATE.variables['mmd_g_mix/coin/gen_average'].copy_(torch.tensor(float(state_in[0]), dtype=dt))
            tf.STATE.variables['mmd_g_mix/coin/prob'].copy_(torch.tensor(float(state_in[1]), dtype=dt))
        tf.STATE.rng = np.random.RandomState(seed)
        captured.clear()
        ref_math.mat_slice = spy
        try:
            lg, ld = ref_math.GANLoss(False).apply(sg, sx, loss_type, batch_size=B, d=s_gen_np.shape[1], **kwargs)
        finally:
            ref_math.mat_slice = real_mat_slice
        glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True)
        gld = torch.autograd.grad(ld, [sg, sx])
        out.update({'loss_gen_' + key: npy(lg), 'loss_dis_' + key: npy(ld),
                    'dLg_dsgen_' + key: npy(glg[0]), 'dLg_dsx_' + key: npy(glg[1]),
                    'dLd_dsgen_' + key: npy(gld[0]), 'dLd_dsx_' + key: npy(gld[1])})
        tf.run_update_ops()
        out['state_out_' + key] = np.asarray([float(tf.STATE.variables['mmd_g_mix/coin/gen_average']),
                                              float(tf.STATE.variables['mmd_g_mix/coin/prob'])], np.float64)
        calls = captured['calls']                                   # dist_g1, dist_g2, dist_g1g2 (:2054-2056)
        assert len(calls) == 3 and calls[0][1] is None and calls[1][1] is None
        g1, g2 = calls[0][0], calls[1][0]
        assert np.array_equal(calls[2][0], g1) and np.array_equal(calls[2][1], g2)
        masks = {'uni': tf.STATE.last_uniform.copy(), 'mix_group_1': g1.astype(bool), 'mix_group_2': g2.astype(bool),
                 'mix_indices': g1[:B].astype(bool)}
        assert np.array_equal(g1[B:], ~g1[:B]) and np.array_equal(g2, ~g1)
        if key == 'f32':
            out.update(masks)
        else:                                                       # the same coin in both precisions
            for k, v in masks.items():
                assert np.array_equal(out[k], v), k
    out.update({'s_gen': s_gen_np, 's_x': s_x_np, 'loss_type': np.asarray(loss_type),
                'state_in': np.asarray(state_in, np.float32),
                'mix_threshold': np.asarray(-1.0 if threshold is None else threshold),
                'margin': np.float64(np.abs(out['uni'] - np.float32(state_in[1])).min())})
    return out


def make_mix():
    n = 0
    for loss_type in ('mmd_g_mix', 'sgm'):
        for B in (8, 64):
            for (sg, sx, off, state, thr) in ((0.25, 0.3, 0.1, (1.5, 0.3), None), (1.0, 1.0, 0.2, (0.05, 0.45), None),
                                              (0.5, 0.6, 0.1, (0.6, 0.0), 0.5), (0.5, 0.6, 0.1, (2.0, 0.5), None)):
                state = (np.float32(state[0]), np.float32(state[1]))
                s_gen, s_x = mmd_inputs(B, 16, sg, sx, off, 9000 + 100 * n)
                fx = run_mix(loss_type, s_gen, s_x, state, seed=400 + n, threshold=thr)
                assert fx['margin'] > 1e-6 or state[1] in (0.0,), fx['margin']       # no coin on the threshold
                np.savez_compressed(os.path.join(OUT, 'lossmix_{}_B{}_c{:02d}.npz'.format(loss_type, B, n)), **fx)
                n += 1
    fx = run_mix('fixed_g_mix', *mmd_inputs(8, 16, 0.25, 0.3, 0.1, 9990), (np.float32(1.2), np.float32(0.25)), seed=77)
    np.savez_compressed(os.path.join(OUT, 'lossmix_fixed_g_mix_B8_c{:02d}.npz'.format(n)), **fx)
    print('mix-loss fixtures:', n + 1)


# ---------------------------------------------------------------------------
# 2. layers through the reference's Net / Routine (layer_func.py)
# ---------------------------------------------------------------------------
def build_routine(designs, net_name, input_shape):
    net = ref_layer.Net(designs, net_name=net_name, data_format=FLAGS.IMAGE_FORMAT, num_class=0)
    r = ref_layer.Routine(net)
    r.add_input_layers([64] + list(input_shape), [0])
    r.seq_links(list(range(net.num_layers)))
    r.add_output_layers([net.num_layers - 1])
    return r


def snapshot():
    return {k: npy(v).copy() for k, v in tf.STATE.variables.items()}


def run_net_case(designs, net_name, input_shape, batch, seed, n_steps=2):
    """forward + backward of a small net with a random upstream gradient, `n_steps` consecutive
    evaluations (SN in_rand / BN moving stats are updated between them)."""
    rs = np.random.RandomState(seed)
    x_np = rs.uniform(-1, 1, size=[batch] + list(input_shape)).astype(np.float32)
    out = {'x': x_np}
    init = None
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        tf.STATE.rng = np.random.RandomState(seed + 1)
        r = build_routine(designs, net_name, input_shape)
        x = torch.tensor(x_np, dtype=dt, requires_grad=True)
        y = r({'x': x}, is_training=True)['x']                     # creates the variables
        if init is None:
            init = snapshot()                                      # fp32 initial values
            for k, v in init.items():
                out['init/' + k] = v.astype(np.float32)
            dy_np = np.random.RandomState(seed + 2).randn(*y.shape).astype(np.float32)
            out['dy'] = dy_np
        # restart from the recorded fp32 initial values in this dtype
        for k, v in tf.STATE.variables.items():
            with torch.no_grad():
                v.copy_(torch.tensor(init[k], dtype=dt))
        tf.STATE.update_ops = []
        for step in range(n_steps):
            r = build_routine(designs, net_name, input_shape)
            x = torch.tensor(x_np, dtype=dt, requires_grad=True)
            y = r({'x': x}, is_training=True)['x']
            names = list(tf.STATE.trainable)
            vs = [tf.STATE.variables[n] for n in names]
            grads = torch.autograd.grad((y * torch.tensor(out['dy'], dtype=dt)).sum(), [x] + vs)
            pre = 'step{}/'.format(step)
            out[pre + 'y_' + key] = npy(y).astype(np.float32)
            if key == 'f64':                      # fp64 truth, stored as float32 to keep fixtures small
                out[pre + 'dx_f64'] = npy(grads[0]).astype(np.float32)
                for n, g in zip(names, grads[1:]):
                    out[pre + 'grad/' + n + '_f64'] = npy(g).astype(np.float32)
            for layer in r.net.layers:
                kn = layer.ops['kernel'].kernel_norm
                if kn is not None:
                    out[pre + 'sigma/' + layer.layer_scope + '_' + key] = npy(kn)
            tf.run_update_ops()
            for k, v in tf.STATE.variables.items():
                if (k.endswith('in_rand') or '/moving_' in k) and key == 'f64':
                    out[pre + 'after/' + k + '_f64'] = npy(v).astype(np.float32)
    return out


def make_layers(only=None):
    ak = float(np.power(64.0, 0.125))
    cases = {
        # the 'padding' and 'dilation' keys of the layer dict (layer_func.py:541-556, 912-916; math_func.py:172-193): 'VALID'
        # convs with and without spectral norm (k3 s1, k4 s2, k3 s2), dilated k3 convs ('SAME' and 'VALID'; no spectral norm:
        # the reference's SpectralNorm sends dilated kernels through NHWC-only atrous ops), dilation ignored at stride 2 ('SAME'
        # only: with 'VALID' the reference's shape inference counts the ignored dilation and trips its own output check)
        'dis_valid_dil': ([{'name': 'l1_v', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'padding': 'VALID'},
                           {'name': 'l2_dil', 'out': 16, 'act': 'lrelu', 'dilation': 2},
                           {'name': 'l3_vs2', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'kernel': 4, 'strides': 2,
                            'padding': 'VALID'},
                           {'name': 'l4_dilv', 'out': 24, 'act': 'relu', 'dilation': 2, 'padding': 'VALID'},
                           {'name': 'l5_s2d', 'out': 32, 'act': 'lrelu', 'strides': 2, 'dilation': 2,
                            'out_reshape': [4 * 5 * 32]},
                           {'name': 'l6_s', 'out': 16, 'op': 'd', 'act_k': ak, 'bias': 'b', 'w_nm': 's'}],
                          'dis', [3, 26, 30], 6),
        # D-style SN conv layers: k3s1 (use_u) and k4s2 (not use_u), + dense SN head with C,H,W flatten
        'dis_small': ([{'name': 'l1_f32', 'out': 8, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's'},
                       {'name': 'l2_ds', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'kernel': 4, 'strides': 2},
                       {'name': 'l3', 'out': 16, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's'},
                       {'name': 'l4_ds', 'out': 32, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'kernel': 4, 'strides': 2,
                        'out_reshape': [4 * 4 * 32]},
                       {'name': 'l5_s', 'out': 16, 'op': 'd', 'act_k': ak, 'bias': 'b', 'w_nm': 's'}],
                      'dis', [3, 16, 16], 6),
        # G-style: dense -> reshape -> tc+BN+relu x2 -> conv+tanh
        'gen_small': ([{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'act': 'linear', 'act_nm': None,
                        'out_reshape': [32, 4, 4]},
                       {'name': 'l2_up', 'out': 16, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
                       {'name': 'l3_up', 'out': 8, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
                       {'name': 'l4_t', 'out': 3, 'act': 'tanh'}],
                      'gen', [24], 6),
        # STL-style first G layer: dense + BN + relu on a 2-D tensor (my_test_stl.py:12)
        'gen_stl_head': ([{'name': 'l1', 'out': 16 * 3 * 3, 'op': 'd', 'act': 'relu', 'act_nm': 'bn',
                           'out_reshape': [16, 3, 3]},
                          {'name': 'l2_up', 'out': 8, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4,
                           'strides': 2}],
                         'gen', [20], 5),
        # single full-width-ish SN layers to exercise channel counts that are not tile multiples
        'dis_odd': ([{'name': 'l1', 'out': 24, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's'},
                     {'name': 'l2', 'out': 40, 'act': 'lrelu', 'act_k': ak, 'w_nm': 's', 'kernel': 4, 'strides': 2}],
                    'dis', [5, 12, 12], 3),
    }
    for name, (designs, net_name, in_shape, batch) in cases.items():
        if only is not None and name not in only:
            continue
        fx = run_net_case(designs, net_name, in_shape, batch, seed=zlib.crc32(name.encode()) % 1000)
        fx['designs_repr'] = np.asarray(repr(designs))
        fx['input_shape'] = np.asarray(in_shape)
        np.savez_compressed(os.path.join(OUT, 'net_{}.npz'.format(name)), **fx)
    print('net fixtures:', len(cases))


# ---------------------------------------------------------------------------
# 3. full G+D step on a width/8 CIFAR-shaped net, 3 consecutive steps
# ---------------------------------------------------------------------------
from tiny_arch import (tiny_architecture, tiny_gsn_architecture, tiny_res_architecture, tiny_res_ps_architecture,  # noqa: E402
                       tiny_res_bil_architecture, tiny_res_max_architecture, tiny_res_bic_architecture,  # noqa: E402
                       tiny_res_tc_architecture)  # noqa: E402


def tf_adam_inplace(var, g, m, v, t, lr, b1=0.5, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (graph_func.py:525-526 hyper-parameters).  graph_func.py needs
    tf.contrib + sessions and is not importable through the shim, so these 4 lines are restated."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).add_(g * g, alpha=1.0 - b2)
    with torch.no_grad():
        var.sub_(lr_t * m / (torch.sqrt(v) + eps))


def make_step(loss_type, B=8, n_steps=3, lr=(5e-4, 2e-4), store_grads=True, sn_mode='default', arch_fn=None,
              tag=None):
    """sn_mode='sn_paper': the flattened-matrix power iteration of layer_func.py:811-814."""
    FLAGS.SPECTRAL_NORM_MODE = sn_mode
    arch = (arch_fn or tiny_architecture)()
    out = {'lr': np.asarray(lr), 'loss_type': np.asarray(loss_type), 'B': np.asarray(B)}
    rs = np.random.RandomState(77)
    zs = rs.randn(n_steps, B, arch['code'][0][0]).astype(np.float32)
    reals = rs.uniform(-1, 1, size=(n_steps, B) + tuple(arch['input'][0])).astype(np.float32)
    out['z'], out['real'] = zs, reals
    init = None
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        tf.STATE.rng = np.random.RandomState(5)

        def build():
            g = build_routine(arch['generator'], 'gen', [arch['code'][0][0]])
            d = build_routine(arch['discriminator'], 'dis', list(arch['input'][0]))
            return g, d
        G, D = build()
        D({'x': torch.cat([torch.tensor(reals[0], dtype=dt), G({'x': torch.tensor(zs[0], dtype=dt)})['x']], 0)})
        if init is None:
            init = snapshot()
            for k, v in init.items():
                out['init/' + k] = v.astype(np.float32)
        for k, v in tf.STATE.variables.items():
            with torch.no_grad():
                v.copy_(torch.tensor(init[k], dtype=dt))
        tf.STATE.update_ops = []
        adam = {}
        for step in range(n_steps):
            G, D = build()                                           # "re-trace the graph"
            z, real = torch.tensor(zs[step], dtype=dt), torch.tensor(reals[step], dtype=dt)
            gen = G({'x': z}, is_training=True)['x']                 # my_sngan.py:277
            dis_out = D({'x': torch.cat([real, gen], 0)}, is_training=True)['x']     # my_sngan.py:278
            s_x, s_gen = tf.split(dis_out, 2, 0)                     # my_sngan.py:279
            lg, ld = ref_math.GANLoss(False).apply(s_gen, s_x, loss_type, batch_size=B, d=16,
                                                   rep_weights=[0.0, -1.0])          # my_sngan.py:284-286
            vd = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, 'dis')
            vg = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, 'gen')
            gd = torch.autograd.grad(ld, vd, retain_graph=True)      # my_sngan.py:302
            gg = torch.autograd.grad(lg, vg)                         # my_sngan.py:304
            pre = 'step{}/'.format(step)
            out[pre + 'loss_gen_' + key], out[pre + 'loss_dis_' + key] = npy(lg), npy(ld)
            out[pre + 's_x_' + key], out[pre + 's_gen_' + key] = npy(s_x), npy(s_gen)
            if key == 'f64' and step in (0, n_steps - 1) and store_grads:
                out[pre + 'gen_f64'] = npy(gen).astype(np.float32)
                for v, g in list(zip(vd, gd)) + list(zip(vg, gg)):
                    out[pre + 'grad/' + v.tf_name + '_f64'] = npy(g).astype(np.float32)
            for layer in list(D.net.layers) + list(G.net.layers):
                for op_name, op in layer.ops.items():
                    if getattr(op, 'kernel_norm', None) is not None:
                        scope = layer.layer_scope if op_name == 'kernel' else layer.layer_scope + '/' + op_name
                        out[pre + 'sigma/' + scope + '_' + key] = npy(op.kernel_norm)
            # apply both Adam updates, then the UPDATE_OPS (values were computed before any write)
            for lr_i, vs, gs in ((lr[0], vd, gd), (lr[1], vg, gg)):
                for v, g in zip(vs, gs):
                    m, vv = adam.setdefault(v.tf_name, (torch.zeros_like(v), torch.zeros_like(v)))
                    tf_adam_inplace(v, g, m, vv, step + 1, lr_i)
            tf.run_update_ops()
        if key == 'f64':
            for k, v in tf.STATE.variables.items():
                out['final/' + k + '_f64'] = npy(v).astype(np.float32)
    tag = tag or (loss_type if sn_mode == 'default' else loss_type + '_pim')
    out['sn_mode'] = np.asarray(sn_mode)
    FLAGS.SPECTRAL_NORM_MODE = 'default'
    np.savez_compressed(os.path.join(OUT, 'step_tiny_{}.npz'.format(tag)), **out)
    print('step fixture:', tag)


# ---------------------------------------------------------------------------
# 3b. the same step, recorded AFTER a warm-up: the reference starts its spectral-norm vectors un-normalised
#     (math_func.py:565-567), so the first step's gradients are ~1e-9 - at Adam's eps, where rounding noise decides
#     the update - and any two fp32 implementations part ways there.  Here the reference code first runs `warm` steps
#     (in fp64), the state it reached (variables, Adam moments, step count; rounded to fp32) is the fixture's starting
#     point, and the next `n_steps` steps are recorded from it in fp32 and fp64: gradients are O(1e-3) and a free-running
#     implementation can be held to 1e-4 on every step.
# ---------------------------------------------------------------------------
ACT_MARGIN_MIN = 1e-5            # SURVEY 8(c)'s rejection rule ("reject seeds with margin < 1e-5") applied to activations
ACT_MARGIN_TARGET = 3e-5         # what the repair of the recorded steps' inputs aims for


def make_step_warm(loss_type, warm=20, B=8, n_steps=3, lr=(5e-4, 2e-4), arch_fn=None, tag=None, sn_mode='default',
                   data_seed=99):
    """data_seed: of the synthetic z / real batches.  The fixture reports `act_margin`, the smallest |relu / lrelu
    PRE-activation| of the recorded steps relative to its layer's largest: where that is within fp32 resolution (~1e-7)
    an fp32 and an fp64 evaluation of the SAME algebra decide the activation differently and one element of every gradient
    below moves by O(1) of itself - a fixture with such an element tests luck, not arithmetic.  A fixture is REJECTED
    (assert) below ACT_MARGIN_MIN.  Drawing seeds cannot get there: the three recorded steps evaluate ~1.2 M activations, so
    ~100 of them lie within 1e-5 of zero for ANY draw; instead the inputs of the recorded steps (synthetic anyway) are
    moved by the minimum-norm perturbation that pushes exactly those pre-activations out to ACT_MARGIN_TARGET
    (_repair_step_inputs: ~1e-5 per input entry), step by step along the trajectory."""
    FLAGS.SPECTRAL_NORM_MODE = sn_mode
    arch = (arch_fn or tiny_architecture)()
    out = {'lr': np.asarray(lr), 'loss_type': np.asarray(loss_type), 'B': np.asarray(B), 'warm': np.asarray(warm),
           'sn_mode': np.asarray(sn_mode), 'data_seed': np.asarray(data_seed)}
    rs = np.random.RandomState(data_seed)
    zs = rs.randn(warm + n_steps, B, arch['code'][0][0]).astype(np.float32)
    reals = rs.uniform(-1, 1, size=(warm + n_steps, B) + tuple(arch['input'][0])).astype(np.float32)

    def build():
        g = build_routine(arch['generator'], 'gen', [arch['code'][0][0]])
        d = build_routine(arch['discriminator'], 'dis', list(arch['input'][0]))
        return g, d

    def one_step(step, dt, adam, record=None, key=None):
        G, D = build()                                               # "re-trace the graph"
        z, real = torch.tensor(zs[step], dtype=dt), torch.tensor(reals[step], dtype=dt)
        gen = G({'x': z}, is_training=True)['x']                     # my_sngan.py:277
        dis_out = D({'x': torch.cat([real, gen], 0)}, is_training=True)['x']     # my_sngan.py:278
        s_x, s_gen = tf.split(dis_out, 2, 0)                         # my_sngan.py:279
        lg, ld = ref_math.GANLoss(False).apply(s_gen, s_x, loss_type, batch_size=B, d=16,
                                               rep_weights=[0.0, -1.0])          # my_sngan.py:284-286
        vd = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, 'dis')
        vg = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, 'gen')
        gd = torch.autograd.grad(ld, vd, retain_graph=True)          # my_sngan.py:302
        gg = torch.autograd.grad(lg, vg)                             # my_sngan.py:304
        if record is not None:
            pre = 'step{}/'.format(step - warm)
            record[pre + 'loss_gen_' + key], record[pre + 'loss_dis_' + key] = npy(lg), npy(ld)
            record[pre + 's_x_' + key], record[pre + 's_gen_' + key] = npy(s_x), npy(s_gen)
            if key == 'f64' and step in (warm, warm + n_steps - 1):   # gradients of the first and the last recorded step
                for v, g in list(zip(vd, gd)) + list(zip(vg, gg)):
                    record[pre + 'grad/' + v.tf_name + '_f64'] = npy(g).astype(np.float32)
            for layer in list(D.net.layers) + list(G.net.layers):
                for op_name, op in layer.ops.items():
                    if getattr(op, 'kernel_norm', None) is not None:
                        scope = layer.layer_scope if op_name == 'kernel' else layer.layer_scope + '/' + op_name
                        record[pre + 'sigma/' + scope + '_' + key] = npy(op.kernel_norm)
        for lr_i, vs, gs in ((lr[0], vd, gd), (lr[1], vg, gg)):
            for v, g in zip(vs, gs):
                m, vv = adam.setdefault(v.tf_name, (torch.zeros_like(v), torch.zeros_like(v)))
                tf_adam_inplace(v, g, m, vv, step + 1, lr_i)
        tf.run_update_ops()
        if record is not None and key == 'f64' and step == warm + n_steps - 1:
            for k, v in tf.STATE.variables.items():                  # every variable after the last recorded step
                record['final/' + k + '_f64'] = npy(v).astype(np.float32)

    # phase 1: the warm-up, once, in fp64
    tf.set_dtype(torch.float64)
    tf.STATE.reset()
    tf.STATE.rng = np.random.RandomState(5)
    G, D = build()
    D({'x': torch.cat([torch.tensor(reals[0], dtype=torch.float64), G({'x': torch.tensor(zs[0], dtype=torch.float64)})['x']], 0)})
    tf.STATE.update_ops = []
    adam = {}
    for step in range(warm):
        one_step(step, torch.float64, adam)
    start = {k: npy(v).astype(np.float32) for k, v in tf.STATE.variables.items()}
    start_m = {k: npy(m).astype(np.float32) for k, (m, _) in adam.items()}
    start_v = {k: npy(v).astype(np.float32) for k, (_, v) in adam.items()}
    for k, v in start.items():
        out['init/' + k] = v
    for k in start_m:
        out['adam_m/' + k], out['adam_v/' + k] = start_m[k], start_v[k]
    out['adam_t'] = np.asarray(warm)
    # phase 1b: move the recorded steps' inputs off every activation knife edge (restatement, fp64, from the same state)
    out['repair'] = np.asarray(_repair_recorded_inputs(arch, loss_type, lr, sn_mode, start, start_m, start_v, warm, zs, reals, n_steps))
    out['z'], out['real'] = zs[warm:], reals[warm:]
    # phase 2: the recorded steps, from the fp32-rounded state, in both precisions
    for key, dt in DT.items():
        tf.set_dtype(dt)
        tf.STATE.reset()
        tf.STATE.rng = np.random.RandomState(5)
        G, D = build()
        D({'x': torch.cat([torch.tensor(reals[0], dtype=dt), G({'x': torch.tensor(zs[0], dtype=dt)})['x']], 0)})
        for k, v in tf.STATE.variables.items():
            with torch.no_grad():
                v.copy_(torch.tensor(start[k], dtype=dt))
        tf.STATE.update_ops = []
        adam = {k: (torch.tensor(start_m[k], dtype=dt), torch.tensor(start_v[k], dtype=dt)) for k in start_m}
        for step in range(warm, warm + n_steps):
            one_step(step, dt, adam, record=out, key=key)
    FLAGS.SPECTRAL_NORM_MODE = 'default'
    tag = tag or loss_type
    out['act_margin'] = np.asarray(_activation_margin(arch, out, sn_mode))
    assert float(out['act_margin']) >= ACT_MARGIN_MIN, ('fixture rejected: activation margin', tag, float(out['act_margin']))
    np.savez_compressed(os.path.join(OUT, 'step_warm_{}.npz'.format(tag)), **out)
    gmax = max(float(np.abs(v).max()) for k, v in out.items() if k.startswith('step0/grad/'))
    print('warm-start step fixture:', tag, 'largest step-0 gradient entry %.3g' % gmax, 'activation margin %.2e' % float(out['act_margin']))


def _margin_of(pres):
    """smallest |pre-activation| / largest |pre-activation| of its tensor, over a list of relu / lrelu inputs"""
    return min(float(p.detach().abs().min() / p.detach().abs().max()) for p in pres)


def _repair_step_inputs(gan, z32, real32, target=None, max_iter=12, log=None):
    """moves (z, real) of ONE step - fp32 arrays, in place - until every relu / lrelu pre-activation of `gan`'s forward
    pass on them is at least `target` of its tensor's largest away from zero.  Gauss-Newton on the few violating elements:
    with r = pre-activation / scale of the k violators and J = dr / d(z, real) (k backward passes), the minimum-norm
    step J^T (J J^T)^-1 (r_wanted - r) moves each of them out to 1.5 x target on its own side and everything else by far
    less than the band's width, so the few elements it pushes INTO the band are dealt with by the next iteration.
    Returns (iterations, margin before, margin after, largest |input change|)."""
    target = ACT_MARGIN_TARGET if target is None else target
    z0, r0 = z32.copy(), real32.copy()
    before = None
    for it in range(max_iter + 1):
        zt = torch.tensor(z32, dtype=torch.float64, requires_grad=True)
        rt = torch.tensor(real32, dtype=torch.float64, requires_grad=True)
        col = {}
        gan.forward_losses(zt, rt, collect=col)
        pres = col['pre_acts']
        margin = _margin_of(pres)
        before = margin if before is None else before
        rel, want = [], []
        for p in pres:
            r = p / p.detach().abs().max()
            sel = (r.detach().abs() < target).reshape(-1).nonzero().reshape(-1)
            if sel.numel():
                rv = r.reshape(-1)[sel]
                rel.append(rv)
                sgn = torch.where(rv.detach() >= 0, torch.ones_like(rv), -torch.ones_like(rv)).detach()
                want.append(sgn * 1.5 * target)
        if not rel:
            break
        assert it < max_iter, ('activation-margin repair did not converge', margin)
        rel, want = torch.cat(rel), torch.cat(want)
        k = rel.numel()
        jz = torch.zeros(k, zt.numel(), dtype=torch.float64)
        jr = torch.zeros(k, rt.numel(), dtype=torch.float64)
        for i in range(k):
            gz, gr = torch.autograd.grad(rel[i], (zt, rt), retain_graph=True, allow_unused=True)
            if gz is not None:
                jz[i] = gz.reshape(-1)
            if gr is not None:
                jr[i] = gr.reshape(-1)
        J = torch.cat([jz, jr], 1)
        A = J @ J.T
        A = A + 1e-12 * torch.diag(A).max() * torch.eye(k, dtype=torch.float64)
        delta = J.T @ torch.linalg.solve(A, (want - rel.detach()))
        z32 += delta[:zt.numel()].reshape(z32.shape).numpy().astype(np.float32)
        real32 += delta[zt.numel():].reshape(real32.shape).numpy().astype(np.float32)
        if log is not None:
            log('    repair iteration %d: %d pre-activations within %.0e, margin %.2e, step max %.2e' % (it, k, target, margin, float(delta.abs().max())))
    moved = max(float(np.abs(z32 - z0).max()), float(np.abs(real32 - r0).max()))
    return it, before, margin, moved


def _repair_recorded_inputs(arch, loss_type, lr, sn_mode, start, start_m, start_v, warm, zs, reals, n_steps):
    """the recorded steps' inputs moved off the activation knife edges, one step after the other along the fp64 trajectory of
    the restatement from the fixture's (fp32-rounded) starting state.  zs / reals are modified in place; returns, per step,
    (iterations, margin before, margin after, largest input change)."""
    import restatement as R
    gan = R.OracleGan(arch, loss_type, tuple(lr), dtype=torch.float64, params=start, sn_mode=sn_mode)
    gan.set_adam_state(start_m, start_v, warm)
    rep = []
    for step in range(warm, warm + n_steps):
        rep.append(_repair_step_inputs(gan, zs[step], reals[step], log=print))
        gan.step(torch.tensor(zs[step], dtype=torch.float64), torch.tensor(reals[step], dtype=torch.float64))
        print('  step %d inputs: %d repair iteration(s), activation margin %.2e -> %.2e, largest input change %.2e'
              % ((step - warm,) + tuple(rep[-1])))
    return rep


def _activation_margin(arch, fx, sn_mode):
    """smallest |relu / lrelu PRE-activation| / max|pre-activation of that tensor| over the recorded steps, from the
    restatement run in fp64 on the fixture's own state and inputs (every activation, those inside residual blocks too)"""
    import restatement as R
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    gan = R.OracleGan(arch, str(fx['loss_type']), tuple(fx['lr']), dtype=torch.float64, params=init, sn_mode=sn_mode)
    gan.set_adam_state({k[len('adam_m/'):]: v for k, v in fx.items() if k.startswith('adam_m/')},
                       {k[len('adam_v/'):]: v for k, v in fx.items() if k.startswith('adam_v/')}, int(fx['adam_t']))
    margin = np.inf
    for step in range(fx['z'].shape[0]):
        z, real = torch.tensor(fx['z'][step], dtype=torch.float64), torch.tensor(fx['real'][step], dtype=torch.float64)
        col = {}
        with torch.no_grad():
            gan.forward_losses(z, real, collect=col)
        margin = min(margin, _margin_of(col['pre_acts']))
        gan.step(z, real)
    return margin


# ---------------------------------------------------------------------------
# 5. eval helpers: NumPy FID on supplied pool3 features (graph_func.py:1733-1745,
#    math_func.py:56-67, 2671-2699) and the sprite grid (graph_func.py:222-266)
# ---------------------------------------------------------------------------
def import_reference_graph_func():
    """graph_func.py imports tf.contrib.gan, the timeline client and pyplot at module level; none of them is touched
    by the NumPy functions recorded here, so empty modules stand in for them at import time."""
    import types
    for name in ('tensorflow.contrib', 'tensorflow.contrib.gan', 'tensorflow.python', 'tensorflow.python.client',
                 'tensorflow.python.client.timeline'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['tensorflow.contrib'].gan = sys.modules['tensorflow.contrib.gan']
    sys.modules['tensorflow.python.client'].timeline = sys.modules['tensorflow.python.client.timeline']
    # its input_func import (default arguments use tf.uint8 / tf.string dtypes the shim does not carry) likewise
    stub = types.ModuleType('GeneralTools.input_func')
    stub.ReadTFRecords = None
    sys.modules.setdefault('GeneralTools.input_func', stub)
    import matplotlib
    matplotlib.use('Agg')
    from GeneralTools import graph_func as ref_graph
    return ref_graph


def make_eval():
    ref_graph = import_reference_graph_func()
    out = {}
    rs = np.random.RandomState(2024)
    mix = rs.randn(48, 48) * 0.3
    x = (rs.randn(300, 48) @ mix + rs.randn(48) * 0.2).astype(np.float64)
    y = (rs.randn(260, 48) @ (mix + rs.randn(48, 48) * 0.05) + 0.1).astype(np.float64)
    mu_x, cov_x = ref_math.mean_cov_np(x)
    mu_y, cov_y = ref_math.mean_cov_np(y)
    out.update({'x': x, 'y': y, 'mu_x': mu_x, 'cov_x': cov_x, 'mu_y': mu_y, 'cov_y': cov_y,
                'sqrt_cov_x': ref_math.sqrt_sym_mat_np(cov_x),
                'trace_sqrt_product': np.float64(ref_math.trace_sqrt_product_np(cov_x, cov_y)),
                'fid': np.float64(ref_graph.GenerativeModelMetric.my_fid_from_pool3(x, y)),
                'fid_from_stats': np.float64(ref_graph.GenerativeModelMetric.my_fid_from_pool3([mu_x, cov_x], y)),
                'fid_self': np.float64(ref_graph.GenerativeModelMetric.my_fid_from_pool3(x, x))})
    # rank-deficient covariance (fewer samples than features): the eps cut of sqrt_sym_mat_np matters
    z = rs.randn(20, 48)
    out['z'] = z
    out['fid_rank_deficient'] = np.float64(ref_graph.GenerativeModelMetric.my_fid_from_pool3(z, y))
    np.savez_compressed(os.path.join(OUT, 'eval_fid.npz'), **out)

    # MeshCode (math_func.py:220-340): the deterministic code layouts eval_sampling can ask for
    mc = {}
    support = rs.randn(4, 5).astype(np.float32)
    for key, dt in DT.items():
        tf.set_dtype(dt)
        code = ref_math.MeshCode(5, mesh_num=(3, 4))
        mc['sine_' + key] = npy(code.by_sine(z_support=torch.tensor(support, dtype=dt)))
    tf.set_dtype(torch.float32)
    z, gx, gy = ref_math.MeshCode(2, mesh_num=(3, 4)).simple_grid()
    z2, _, _ = ref_math.MeshCode(2, mesh_num=(2, 5)).simple_grid(np.array([[-2.0, 0.5], [1.0, 3.0]], dtype=np.float32))
    mc.update({'support': support, 'grid_z': z, 'grid_x': gx, 'grid_y': gy, 'grid2_z': z2})
    np.savez_compressed(os.path.join(OUT, 'eval_meshcode.npz'), **mc)

    # sprite: capture what write_sprite hands to scipy.misc.imsave (removed from SciPy long ago)
    import types
    import scipy
    captured = {}
    misc = types.ModuleType('scipy.misc')
    misc.imsave = lambda path, arr: captured.__setitem__('arr', np.array(arr))
    sys.modules['scipy.misc'] = misc
    scipy.misc = misc
    sp = {}
    cases = {'rgb_auto': (rs.uniform(-1, 1, (10, 6, 5, 3)), None, False),
             'rgb_mesh': (rs.uniform(-1, 1, (10, 6, 5, 3)), (2, 5), False),
             'rgb_invert': (rs.uniform(0, 1, (6, 4, 4, 3)), [3, 2], True),
             'gray3d': (rs.uniform(-1, 1, (9, 5, 7)), None, False),
             'gray4d': (rs.uniform(-1, 1, (4, 5, 7, 1)), (2, 2), False)}
    for name, (img, mesh, inv) in cases.items():
        ref_graph.write_sprite('unused.png', img.astype(np.float32), mesh_num=mesh, if_invert=inv)
        sp[name + '/images'] = img.astype(np.float32)
        sp[name + '/mesh'] = np.asarray(mesh if mesh is not None else [-1, -1])
        sp[name + '/invert'] = np.asarray(inv)
        sp[name + '/sprite'] = captured['arr']
    np.savez_compressed(os.path.join(OUT, 'eval_sprite.npz'), **sp)
    print('eval fixtures: fid %.6f, sprites %s' % (out['fid'], [sp[k + '/sprite'].shape for k in cases]))


# ---------------------------------------------------------------------------
# 6. initialisers: what the reference's weight_initializer / bias_initializer return in each
#    FLAGS.WEIGHT_INITIALIZER mode (layer_func.py:14-80) - sample statistics of the draws, per activation and kernel
#    shape.  The random stream is the shim's, so the statistics (not the samples) are the fixture.
# ---------------------------------------------------------------------------
def make_init_stats():
    tf.set_dtype(torch.float64)
    out, n = {}, 0
    shapes = {'dense': [512, 384], 'conv3': [3, 3, 64, 128], 'tconv4': [4, 4, 32, 256]}     # tc kernels are [k,k,Cout,Cin]
    for mode in ('default', 'sn_paper', 'pg_paper'):
        FLAGS.WEIGHT_INITIALIZER = mode
        for act in ('relu', 'lrelu', 'tanh', 'linear'):
            for sname, shape in shapes.items():
                tf.STATE.rng = np.random.RandomState(1000 + n)
                w = npy(ref_layer.weight_initializer(act)(shape))
                key = '{}/{}/{}/'.format(mode, act, sname)
                out[key + 'std'], out[key + 'absmax'], out[key + 'mean'] = np.float64(w.std()), np.float64(np.abs(w).max()), np.float64(w.mean())
                out[key + 'kurt'] = np.float64(((w - w.mean()) ** 4).mean() / w.var() ** 2)   # 1.8 uniform, ~2.1 truncated normal
                out[key + 'shape'] = np.asarray(shape)
                n += 1
    FLAGS.WEIGHT_INITIALIZER = 'default'
    for act, scale in (('relu', 0.25), ('linear', 4.0), ('relu', 0.0)):                    # init_w_scale (layer_func.py:719-720)
        tf.STATE.rng = np.random.RandomState(2000 + n)
        w = npy(ref_layer.weight_initializer(act, scale)(shapes['conv3']))
        key = 'default/{}/conv3/scale{}/'.format(act, scale)
        out[key + 'std'], out[key + 'absmax'] = np.float64(w.std()), np.float64(np.abs(w).max())
        n += 1
    tf.STATE.rng = np.random.RandomState(3000)
    b = npy(ref_layer.bias_initializer(1e-5)([4096]))
    out['bias/std'], out['bias/absmax'] = np.float64(b.std()), np.float64(np.abs(b).max())
    out['bias0/absmax'] = np.float64(np.abs(npy(ref_layer.bias_initializer(0.0)([64]))).max())
    try:
        FLAGS.WEIGHT_INITIALIZER = 'no_such_mode'
        ref_layer.weight_initializer('relu')
        out['unknown_mode_error'] = np.asarray('')
    except NotImplementedError as err:
        out['unknown_mode_error'] = np.asarray(str(err))
    FLAGS.WEIGHT_INITIALIZER = 'default'
    tf.set_dtype(torch.float32)
    np.savez_compressed(os.path.join(OUT, 'init_stats.npz'), **out)
    print('initialiser fixture: %d cases' % n)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if '--only-init' in sys.argv:
        make_init_stats()
        sys.exit(0)
    if '--only-mix' in sys.argv:
        make_mix()
        sys.exit(0)
    if '--only-warm' in sys.argv:
        make_step_warm('rep')
        make_step_warm('rep', arch_fn=tiny_res_architecture, tag='res_rep')
        make_step_warm('rep', sn_mode='sn_paper', tag='rep_pim')
        make_step_warm('rep', arch_fn=tiny_gsn_architecture, tag='gsn_rep')
        sys.exit(0)
    if '--only-gsn' in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(4)
        make_step('rep', arch_fn=tiny_gsn_architecture, tag='gsn_rep')
        make_step('rmb', sn_mode='sn_paper', arch_fn=tiny_gsn_architecture, tag='gsn_rmb_pim')
        make_step_warm('rep', arch_fn=tiny_gsn_architecture, tag='gsn_rep')
        sys.exit(0)
    if '--only-tc' in sys.argv:                                      # transposed convs inside residual blocks (layer_func.py:1725-1727)
        torch.manual_seed(0)
        torch.set_num_threads(4)
        make_step('rep', arch_fn=tiny_res_tc_architecture, tag='res_tc_rep', store_grads=True)
        sys.exit(0)
    if '--only-valid' in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(4)
        make_layers(only=('dis_valid_dil',))
        sys.exit(0)
    if '--only-bic' in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(4)
        make_step('rep', arch_fn=tiny_res_bic_architecture, tag='res_bic_rep', store_grads=True)
        sys.exit(0)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    make_mmd()
    if '--only-next' in sys.argv:
        make_loss_next()
        make_step('rep', sn_mode='sn_paper')
        make_step('rep', arch_fn=tiny_res_architecture, tag='res_rep')
        make_step('rmb', arch_fn=tiny_res_ps_architecture, tag='res_ps_rmb')
        make_step('rep', arch_fn=tiny_res_bil_architecture, tag='res_bil_rep', store_grads=True)
        make_step('rep', arch_fn=tiny_res_max_architecture, tag='res_max_rep')
        make_eval()
        sys.exit(0)
    make_loss_next()
    make_layers()
    make_step('rep')
    make_step('rmb', store_grads=False)
    make_step('rep', sn_mode='sn_paper')
    make_step('rep', arch_fn=tiny_res_architecture, tag='res_rep')
    make_step('rmb', arch_fn=tiny_res_ps_architecture, tag='res_ps_rmb')
    make_step('rep', arch_fn=tiny_res_bil_architecture, tag='res_bil_rep')
    make_step('rep', arch_fn=tiny_res_bic_architecture, tag='res_bic_rep')
    make_step('rep', arch_fn=tiny_res_max_architecture, tag='res_max_rep')
    make_step('rep', arch_fn=tiny_res_tc_architecture, tag='res_tc_rep')
    make_step('rep', arch_fn=tiny_gsn_architecture, tag='gsn_rep')
    make_step('rmb', sn_mode='sn_paper', arch_fn=tiny_gsn_architecture, tag='gsn_rmb_pim')
    make_eval()
    make_init_stats()
    make_mix()
    make_step_warm('rep')
    make_step_warm('rep', arch_fn=tiny_res_architecture, tag='res_rep')
    make_step_warm('rep', sn_mode='sn_paper', tag='rep_pim')
    make_step_warm('rep', arch_fn=tiny_gsn_architecture, tag='gsn_rep')
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('tests/golden total bytes:', total)
