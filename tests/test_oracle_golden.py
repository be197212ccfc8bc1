"""The oracle (oracle/restatement.py, oracle/mmd_oracle.c) against the golden vectors that
oracle/make_golden.py produced by running the reference's own code.  CPU only."""
import os

import numpy as np
import pytest
import torch

from helpers import RTOL, designs_of, golden, load, rel_err
from oracle import c_oracle, restatement as R

MMD = golden('mmd_*.npz')
NETS = golden('net_*.npz')


@pytest.mark.parametrize('path', MMD, ids=[p.split('/')[-1] for p in MMD])
@pytest.mark.parametrize('prec', ['f32', 'f64'])
def test_mmd_restatement_matches_reference(path, prec):
    fx = load(path)
    dt = torch.float32 if prec == 'f32' else torch.float64
    sg = torch.tensor(fx['s_gen'], dtype=dt, requires_grad=True)
    sx = torch.tensor(fx['s_x'], dtype=dt, requires_grad=True)
    B = sg.shape[0]
    lg, ld, stats = R.gan_loss(sg, sx, str(fx['loss_type']), B, tuple(fx['rep_weights']))
    tol = 2e-5 if prec == 'f32' else 1e-12
    assert abs(float(lg) - float(fx['loss_gen_' + prec])) <= tol * max(abs(float(fx['loss_gen_' + prec])), 1e-3)
    assert abs(float(ld) - float(fx['loss_dis_' + prec])) <= tol * max(abs(float(fx['loss_dis_' + prec])), 1e-3)
    glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True)
    gld = torch.autograd.grad(ld, [sg, sx], allow_unused=True)
    for got, key in ((glg[0], 'dLg_dsgen_'), (glg[1], 'dLg_dsx_'), (gld[0], 'dLd_dsgen_'), (gld[1], 'dLd_dsx_')):
        got = np.zeros_like(fx[key + prec]) if got is None else got.numpy()
        assert rel_err(got, fx[key + prec]) <= (1e-4 if prec == 'f32' else 1e-10)
    if prec == 'f32':                                   # bit-exact index masks
        m = R.mmd_masks(sg.detach(), sx.detach())
        assert np.array_equal(m['xx_lt_lb'].numpy(), fx['mask_gg_lt_lb'])
        assert np.array_equal(m['yy_gt_ub'].numpy(), fx['mask_dd_gt_ub'])
        assert np.array_equal(m['xy_gt_ub'].numpy(), fx['mask_gd_gt_ub'])


LOSSX = golden('lossx_*.npz')


@pytest.mark.parametrize('path', LOSSX, ids=[p.split('/')[-1] for p in LOSSX])
@pytest.mark.parametrize('prec', ['f32', 'f64'])
def test_next_row_losses_restatement_matches_reference(path, prec):
    """'mmd_g', 'mgb', 'hinge', 'logistic' of the reference's GANLoss (fixtures from
    oracle/make_golden.py:make_loss_next)."""
    fx = load(path)
    dt = torch.float32 if prec == 'f32' else torch.float64
    sg = torch.tensor(fx['s_gen'], dtype=dt, requires_grad=True)
    sx = torch.tensor(fx['s_x'], dtype=dt, requires_grad=True)
    lg, ld, _ = R.gan_loss(sg, sx, str(fx['loss_type']), sg.shape[0])
    tol = 2e-5 if prec == 'f32' else 1e-12
    assert abs(float(lg) - float(fx['loss_gen_' + prec])) <= tol * max(abs(float(fx['loss_gen_' + prec])), 1e-3)
    assert abs(float(ld) - float(fx['loss_dis_' + prec])) <= tol * max(abs(float(fx['loss_dis_' + prec])), 1e-3)
    glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True, allow_unused=True)
    gld = torch.autograd.grad(ld, [sg, sx], allow_unused=True)
    for got, key in ((glg[0], 'dLg_dsgen_'), (glg[1], 'dLg_dsx_'), (gld[0], 'dLd_dsgen_'), (gld[1], 'dLd_dsx_')):
        got = np.zeros_like(fx[key + prec]) if got is None else got.numpy()
        assert rel_err(got, fx[key + prec]) <= (1e-4 if prec == 'f32' else 1e-10)


@pytest.mark.parametrize('path', MMD, ids=[p.split('/')[-1] for p in MMD])
def test_mmd_c_oracle_matches_reference(path):
    fx = load(path)
    w = tuple(float(v) for v in fx['rep_weights'])
    for prec, dt in (('f32', np.float32), ('f64', np.float64)):
        out = c_oracle.mmd(fx['s_gen'], fx['s_x'], str(fx['loss_type']), w, dtype=dt)
        ltol = 5e-4 if prec == 'f32' else 1e-11   # two correct fp32 evaluations of the Gram form differ by this much
        # the losses are differences of O(1) kernel means (cancellation): an fp32 sequential sum
        # over B*B terms carries ~1e-5 of the mean's scale, whatever the loss's own size
        escale = max(float(fx['e_kxx_' + prec]), float(fx['e_kyy_' + prec]), float(fx['e_kxy_' + prec]))
        floor = 1e-5 * escale if prec == 'f32' else 1e-13
        for name in ('loss_gen', 'loss_dis'):
            ref = float(fx[name + '_' + prec])
            assert abs(float(out[name]) - ref) <= ltol * abs(ref) + floor, (name, prec)
        gtol = 1e-4 if prec == 'f32' else 1e-9
        for i, key in enumerate(('dLg_dsgen_', 'dLg_dsx_', 'dLd_dsgen_', 'dLd_dsx_')):
            assert rel_err(out['grads'][i], fx[key + prec]) <= gtol, (key, prec)
        assert rel_err(out['stats'][:3], [fx['e_kxx_' + prec], fx['e_kxy_' + prec], fx['e_kyy_' + prec]]) <= ltol
    # masks: the C oracle in fp32 reproduces the reference's fp32 clamp sets exactly when the
    # threshold margin exceeds fp32 rounding of the distance (fixtures guarantee > 1e-4)
    out = c_oracle.mmd(fx['s_gen'], fx['s_x'], str(fx['loss_type']), w, dtype=np.float32)
    assert np.array_equal(out['masks'][0], fx['mask_gg_lt_lb'])
    assert np.array_equal(out['masks'][1], fx['mask_gd_gt_ub'])
    assert np.array_equal(out['masks'][2], fx['mask_dd_gt_ub'])


@pytest.mark.parametrize('path', NETS, ids=[p.split('/')[-1] for p in NETS])
def test_net_restatement_matches_reference(path):
    fx = load(path)
    designs = designs_of(fx)
    net_name = [k for k in fx if k.startswith('init/')][0].split('/')[1]
    specs = R.build_net(designs, [int(v) for v in fx['input_shape']], net_name)
    for prec, dt in (('f32', torch.float32), ('f64', torch.float64)):
        params = {k[len('init/'):]: torch.tensor(v, dtype=dt) for k, v in fx.items() if k.startswith('init/')}
        names = R.trainable_names(params)
        for step in range(2):
            pre = 'step%d/' % step
            leaves = {n: params[n].clone().requires_grad_(True) for n in names}
            p = dict(params)
            p.update(leaves)
            x = torch.tensor(fx['x'], dtype=dt, requires_grad=True)
            col = {}
            y, updates = R.net_forward(specs, p, x, True, col)
            tol = 5e-5 if prec == 'f32' else 1e-6        # f64 fixtures are stored as float32
            assert rel_err(y.detach().numpy(), fx[pre + 'y_' + prec]) <= tol
            for k in fx:
                if k.startswith(pre + 'sigma/') and k.endswith(prec):
                    scope = k[len(pre + 'sigma/'):-len('_' + prec)]
                    assert abs(float(col[scope + '/sigma']) - float(fx[k])) <= tol * float(fx[k])
            if prec == 'f64':
                g = torch.autograd.grad((y * torch.tensor(fx['dy'], dtype=dt)).sum(), [x] + [leaves[n] for n in names])
                assert rel_err(g[0].numpy(), fx[pre + 'dx_f64']) <= 1e-6
                for n, gi in zip(names, g[1:]):
                    assert rel_err(gi.numpy(), fx[pre + 'grad/' + n + '_f64']) <= 1e-6, n
                for n, v in updates.items():
                    assert rel_err(v.numpy(), fx[pre + 'after/' + n + '_f64']) <= 1e-6, n
            for n, v in updates.items():
                params[n] = v


@pytest.mark.parametrize('loss_type', ['rep', 'rmb', 'rep_pim', 'res_rep', 'res_ps_rmb', 'res_bil_rep', 'res_bic_rep', 'res_max_rep',
                                       'res_tc_rep', 'gsn_rep', 'gsn_rmb_pim'])
def test_full_step_restatement_matches_reference(loss_type):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    fx = load(golden('step_tiny_%s.npz' % loss_type)[0])
    from tiny_arch import (tiny_architecture, tiny_gsn_architecture, tiny_res_architecture, tiny_res_ps_architecture,
                           tiny_res_bil_architecture, tiny_res_max_architecture, tiny_res_bic_architecture,
                           tiny_res_tc_architecture)
    # 'res_': the ResNet-shaped pair - every kind of residual block of layer_func.py:1687-1842
    is_res = loss_type.startswith('res_')
    res_arch = {'res_rep': tiny_res_architecture, 'res_ps_rmb': tiny_res_ps_architecture, 'res_bil_rep': tiny_res_bil_architecture,
                'res_max_rep': tiny_res_max_architecture, 'res_bic_rep': tiny_res_bic_architecture,
                # transposed convs inside the blocks (layer_func.py:1725-1727), one of them spectrally normalised
                'res_tc_rep': tiny_res_tc_architecture}
    # 'gsn_': spectral norm in the generator too, transposed-conv kernels included (math_func.py:512-528)
    arch = res_arch[loss_type]() if is_res else (tiny_gsn_architecture() if loss_type.startswith('gsn_') else tiny_architecture())
    # '_pim': FLAGS.SPECTRAL_NORM_MODE = 'sn_paper' in the reference run (layer_func.py:811-814)
    sn_mode = str(fx['sn_mode']) if 'sn_mode' in fx else 'default'
    loss_type = str(fx['loss_type'])
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    for prec, dt in (('f32', torch.float32), ('f64', torch.float64)):
        gan = R.OracleGan(arch, loss_type, tuple(fx['lr']), dtype=dt, params=init, sn_mode=sn_mode)
        for step in range(3):
            z, real = torch.tensor(fx['z'][step], dtype=dt), torch.tensor(fx['real'][step], dtype=dt)
            pre = 'step%d/' % step
            if prec == 'f64' and any(k.startswith(pre + 'grad/') for k in fx):
                lg, ld, stats, upd, gd, gg, aux = gan.grads(z, real)
                for n, g in list(gd.items()) + list(gg.items()):
                    ref = fx[pre + 'grad/' + n + '_f64']
                    if step == 0:
                        # un-normalised SN start vectors make D's step-0 outputs
                        # ~1e-14, the kernel values 1 - O(1e-28) and every gradient <1e-12, gated by the sign of
                        # rounding-noise distances (max(.,0) of +-1e-28), in the reference too: only magnitude is checkable
                        if not is_res and sn_mode == 'default' and np.abs(ref).max() < 1e-12:
                            assert np.abs(g.numpy()).max() < 1e-12, n
                            continue
                        # (the residual net's shortcuts keep the step-0 scores at ~1e-4, the flattened-kernel norms of
                        # 'sn_paper' are ~30x smaller than PICO's: their step-0 gradients are small but ordinary)
                    # dL/d(last bias) is exactly 0 analytically (the loss sees only score
                    # differences): allow an absolute floor tied to the net's gradient scale
                    gscale = max(np.abs(fx[pre + 'grad/' + m + '_f64']).max() for m in (gd if n in gd else gg))
                    assert np.abs(g.numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-9 * gscale, (step, n)
            lg, ld = gan.step(z, real)
            for got, name in ((lg, 'loss_gen_'), (ld, 'loss_dis_')):
                ref = float(fx[pre + name + prec])
                tol = 2e-4 if prec == 'f32' else 1e-9
                assert abs(got - ref) <= tol * max(abs(ref), 1e-6), (step, name, got, ref)
        if prec == 'f64':
            for k, v in fx.items():
                if k.startswith('final/'):
                    n = k[len('final/'):-len('_f64')]
                    assert rel_err(gan.params[n].numpy(), v) <= 2e-6, n


@pytest.mark.parametrize('tag', ['rep', 'res_rep', 'rep_pim', 'gsn_rep'])
def test_warm_start_restatement_matches_reference(tag):
    """the fixtures recorded after 20 warm-up steps of the reference code (oracle/make_golden.py:make_step_warm): from
    their state (variables, Adam moments, step count) the restatement's free-running fp64 trajectory reproduces
    losses, every gradient and every final variable of the reference run to 1e-6 - no step-0 noise regime here, so the
    bounds are rounding-level on every step"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    from tiny_arch import tiny_architecture, tiny_gsn_architecture, tiny_res_architecture
    fx = load(golden('step_warm_%s.npz' % tag)[0])
    arch = tiny_res_architecture() if tag.startswith('res_') else (tiny_gsn_architecture() if tag.startswith('gsn_') else tiny_architecture())
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    m = {k[len('adam_m/'):]: v for k, v in fx.items() if k.startswith('adam_m/')}
    v2 = {k[len('adam_v/'):]: v for k, v in fx.items() if k.startswith('adam_v/')}
    gan = R.OracleGan(arch, str(fx['loss_type']), tuple(fx['lr']), dtype=torch.float64, params=init,
                      sn_mode=str(fx['sn_mode']))
    gan.set_adam_state(m, v2, int(fx['adam_t']))
    n_steps = fx['z'].shape[0]
    margin = np.inf
    for step in range(n_steps):
        z, real = torch.tensor(fx['z'][step], dtype=torch.float64), torch.tensor(fx['real'][step], dtype=torch.float64)
        pre = 'step%d/' % step
        # SURVEY 8(c)'s rejection rule applied to activations: no relu / lrelu pre-activation of a recorded step within 1e-5
        # of its tensor's largest (recomputed here, not read from the fixture) - an fp32 engine cannot decide one differently
        col = {}
        with torch.no_grad():
            gan.forward_losses(z, real, collect=col)
        margin = min([margin] + [float(p.abs().min() / p.abs().max()) for p in col['pre_acts']])
        if any(k.startswith(pre + 'grad/') for k in fx):
            lg, ld, stats, upd, gd, gg, aux = gan.grads(z, real)
            for n, g in list(gd.items()) + list(gg.items()):
                ref = fx[pre + 'grad/' + n + '_f64']
                gscale = max(np.abs(fx[pre + 'grad/' + k + '_f64']).max() for k in (gd if n in gd else gg))
                assert np.abs(g.numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-9 * gscale, (step, n)
        lg, ld = gan.step(z, real)
        for got, name in ((lg, 'loss_gen_f64'), (ld, 'loss_dis_f64')):
            ref = float(fx[pre + name])
            assert abs(got - ref) <= 1e-9 * max(abs(ref), 1e-6), (step, name, got, ref)
    for k, v in fx.items():
        if k.startswith('final/'):
            n = k[len('final/'):-len('_f64')]
            assert rel_err(gan.params[n].numpy(), v) <= 2e-6, n
    assert margin >= 1e-5 and abs(margin - float(fx['act_margin'])) <= 1e-3 * margin, (margin, float(fx['act_margin']))


MIX = golden('lossmix_*.npz')


@pytest.mark.parametrize('path', MIX, ids=[p.split('/')[-1] for p in MIX])
def test_mix_loss_restatement_matches_reference(path):
    """'mmd_g_mix' / 'fixed_g_mix' / 'sgm' (math_func.py:2195-2263): with the recorded uniform draw and state as inputs,
    the restatement's coin, both group masks (bit for bit), losses, gradients and the state after the UPDATE_OPS"""
    fx = load(path)
    thr = float(fx['mix_threshold'])
    for prec, dt, tol in (('f32', torch.float32, 2e-5), ('f64', torch.float64, 1e-12)):
        sg = torch.tensor(fx['s_gen'], dtype=dt, requires_grad=True)
        sx = torch.tensor(fx['s_x'], dtype=dt, requires_grad=True)
        lg, ld, info = R.gan_loss_mix(sg, sx, str(fx['loss_type']), sg.shape[0], fx['uni'], tuple(fx['state_in']),
                                      None if thr < 0 else thr)
        for k in ('mix_indices', 'mix_group_1', 'mix_group_2'):
            assert np.array_equal(info[k].numpy(), fx[k]), k
        scale = max(abs(float(fx['loss_gen_f64'])), abs(float(fx['loss_dis_f64'])), 1e-3)
        assert abs(float(lg) - float(fx['loss_gen_' + prec])) <= tol * scale
        assert abs(float(ld) - float(fx['loss_dis_' + prec])) <= tol * scale
        glg = torch.autograd.grad(lg, [sg, sx], retain_graph=True)
        gld = torch.autograd.grad(ld, [sg, sx])
        for g, key in zip(list(glg) + list(gld), ('dLg_dsgen_', 'dLg_dsx_', 'dLd_dsgen_', 'dLd_dsx_')):
            assert np.abs(g.numpy() - fx[key + prec]).max() <= tol * max(np.abs(fx[key + prec]).max(), 1e-6), key
        assert np.abs(np.asarray(info['new_state']) - fx['state_out_' + prec]).max() <= 1e-6


def test_reference_fp32_noise_floor():
    """Why the GPU loss tolerance carries an absolute floor: the reference's OWN fp32 evaluation
    (torch-CPU under the shim) misses its fp64 evaluation by more than 1e-4 of the loss on the
    small-loss fixtures, i.e. by ~1e-4 of the kernel-mean scale.  The HIP kernel accumulates the
    kernel sums in double and skips the diagonal, so it is held to 1e-4*|loss| + 4e-7*scale
    (tests/test_ops_gpu.py::loss_tol), far tighter than fp32-vs-fp32 agreement."""
    worst_rel_loss, worst_rel_scale = 0.0, 0.0
    for path in MMD:
        fx = load(path)
        escale = max(float(fx['e_kxx_f64']), float(fx['e_kxy_f64']), float(fx['e_kyy_f64']))
        for name in ('loss_gen', 'loss_dis'):
            err = abs(float(fx[name + '_f32']) - float(fx[name + '_f64']))
            worst_rel_scale = max(worst_rel_scale, err / escale)
            if float(fx[name + '_f64']) != 0.0:
                worst_rel_loss = max(worst_rel_loss, err / abs(float(fx[name + '_f64'])))
    assert worst_rel_loss > 1e-4          # the reference's fp32 path itself is outside 1e-4
    assert worst_rel_scale < 1e-3


@pytest.mark.parametrize('h,w,oh,ow', [(3, 3, 6, 6), (6, 6, 12, 12), (12, 12, 6, 6), (6, 6, 2, 2), (5, 7, 11, 4), (4, 4, 1, 1), (1, 1, 3, 3)])
def test_bicubic_restatement(h, w, oh, ow):
    """'bic' = tf.image.resize_bicubic(align_corners=True), TF 1.x legacy kernel.  No TF here, so the pin is three-fold: the
    restatement's gather form equals the TF shim's dense-matrix form (written separately, the one the golden fixtures were
    generated through) to rounding; every row of weights sums to 1 within the table's fp32 rounding; and both agree with
    torch's bicubic (same Keys kernel A = -0.75, exact fractions instead of TF's 1/1024 grid) to the grid's resolution."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    import tf1_shim
    rs = np.random.RandomState(h * 100 + ow)
    x = torch.tensor(rs.randn(2, 3, h, w), dtype=torch.float64)
    got = R.bicubic_resize(x, (oh, ow))
    via_shim = tf1_shim.image.resize_bicubic(x.permute(0, 2, 3, 1), (oh, ow), align_corners=True).permute(0, 3, 1, 2)
    assert float((got - via_shim).abs().max()) <= 1e-12 * float(x.abs().max())
    for out_n, in_n in ((oh, h), (ow, w)):
        idx, wgt = R.bicubic_taps(out_n, in_n)
        assert np.abs(wgt.astype(np.float64).sum(1) - 1).max() <= 4e-7
        assert idx.min() >= 0 and idx.max() <= in_n - 1
    if oh > 1 and ow > 1:
        ref = torch.nn.functional.interpolate(x, size=(oh, ow), mode='bicubic', align_corners=True)
        assert float((got - ref).abs().max()) <= 4e-3 * float(x.abs().max())      # 1/2048 in position x the kernel's slope
        assert torch.allclose(got[:, :, 0, 0], x[:, :, 0, 0], atol=1e-6) and torch.allclose(got[:, :, -1, -1], x[:, :, -1, -1], atol=1e-6)


def test_forced_activation_masks_are_a_no_op_on_the_oracles_own_decisions():
    """OracleGan.grads(masks=...) (the fp32 floor of the step tests, helpers.fp32_floor): forcing every relu / lrelu to the
    sign decisions the SAME evaluation takes anyway changes nothing, for plain layers and for residual blocks; forcing one
    different decision does change the gradients (the option is live)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    from tiny_arch import tiny_architecture, tiny_res_architecture
    for arch in (tiny_architecture(), tiny_res_architecture()):
        B = 4
        ora = R.OracleGan(arch, 'rep', (5e-4, 2e-4), dtype=torch.float64, seed=3)
        rs = np.random.RandomState(1)
        z = torch.tensor(rs.randn(B, arch['code'][0][0]))
        real = torch.tensor(rs.uniform(-1, 1, (B,) + tuple(arch['input'][0])))
        ora.step(z, real)                                # normalised SN vectors: gradients above rounding noise
        ora.step(z, real)
        seen = {'gen': [], 'dis': []}
        orig = R._act

        def spy(x, name, mask=None, audit=None):
            y = orig(x, name, mask, audit)
            if name in ('relu', 'lrelu'):
                seen[spy.net].append((y.detach() > 0))
            return y
        # record the decisions of a plain evaluation, net by net (G runs first, then D on [real ; fake])
        R._act = spy
        try:
            spy.net = 'gen'
            gen, _ = R.net_forward(ora.gen_specs, ora.params, z, True)
            spy.net = 'dis'
            R.net_forward(ora.dis_specs, ora.params, torch.cat([real, gen], 0), True)
        finally:
            R._act = orig
        assert seen['gen'] and seen['dis']
        plain = ora.grads(z, real)
        forced = ora.grads(z, real, masks=seen)
        for a, b in zip(list(plain[4].values()) + list(plain[5].values()), list(forced[4].values()) + list(forced[5].values())):
            assert float((a - b).abs().max()) <= 1e-12 * max(float(a.abs().max()), 1e-30)
        audited = dict(seen, audit=[])                   # the audit of forced decisions (helpers.assert_knife_edges_only): none differ
        ora.grads(z, real, masks=audited)
        assert len(audited['audit']) == len(seen['gen']) + len(seen['dis']) and all(a[:2] == (0, 0.0) and a[2] > 0 for a in audited['audit'])
        flipped = {k: [m.clone() for m in v] for k, v in seen.items()}
        flipped['dis'][0].view(-1)[::7] ^= True
        flipped['audit'] = []
        other = ora.grads(z, real, masks=flipped)
        n_forced = int(flipped['dis'][0].numel() + 6) // 7
        first_d = flipped['audit'][len(seen['gen'])]      # (the layers above see other inputs now: their counts are whatever)
        assert first_d[0] == n_forced and first_d[1] > 1e-3   # ... and these are no knife edges
        name = next(iter(plain[4]))
        assert float((other[4][name] - plain[4][name]).abs().max()) > 1e-6 * float(plain[4][name].abs().max())
