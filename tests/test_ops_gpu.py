"""GPU parity tests: every C-ABI op against the oracle (oracle/restatement.py, oracle/mmd_oracle.c)
and against the golden vectors generated from the reference.  The oracle is the checker only."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import RTOL, elementwise_err, golden, load, rel_err
from oracle import c_oracle, restatement as R

pytestmark = pytest.mark.gpu

MMD = golden('mmd_*.npz')


@pytest.fixture(scope='module')
def ops():
    from mmdgan_hip import ops as o
    o.require_device()
    return o


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def nhwc(t):          # NCHW torch/np -> NHWC cuda
    return dev(np.transpose(np.asarray(t), (0, 2, 3, 1)))


def to_nchw(t):
    return np.transpose(t.cpu().numpy(), (0, 3, 1, 2))


# ---------------------------------------------------------------------------------------------
# fused pairwise / MMD loss
# ---------------------------------------------------------------------------------------------
def loss_tol(ref64, escale):
    """north_star: loss within 1e-4 relative.  The losses are differences of kernel means of size
    `escale`; fp32 inputs/exp leave an absolute floor of a few ulp of escale that no fp32
    implementation (the reference included) can beat - see test_reference_fp32_noise_floor."""
    return RTOL * abs(ref64) + 4e-7 * escale


@pytest.mark.parametrize('path', MMD, ids=[p.split('/')[-1] for p in MMD])
def test_mmd_loss_matches_reference_golden(ops, path):
    fx = load(path)
    w = tuple(float(v) for v in fx['rep_weights'])
    out = ops.mmd_loss(dev(fx['s_gen']), dev(fx['s_x']), str(fx['loss_type']), w, need_grads=True, need_masks=True,
                       need_dist=True)
    sc = out['scalars'].cpu().numpy().astype(np.float64)
    escale = max(float(fx['e_kxx_f64']), float(fx['e_kxy_f64']), float(fx['e_kyy_f64']))
    for idx, name in ((0, 'loss_gen'), (1, 'loss_dis')):
        ref = float(fx[name + '_f64'])
        assert abs(sc[idx] - ref) <= loss_tol(ref, escale), (name, sc[idx], ref)
    for idx, name in ((2, 'e_kxx'), (3, 'e_kxy'), (4, 'e_kyy')):
        ref = float(fx[name + '_f64'])
        assert abs(sc[idx] - ref) <= 1e-5 * abs(ref) + 1e-12, (name, sc[idx], ref)
    g = out['grads'].cpu().numpy()
    for i, key in enumerate(('dLg_dsgen_f64', 'dLg_dsx_f64', 'dLd_dsgen_f64', 'dLd_dsx_f64')):
        assert rel_err(g[i], fx[key]) <= RTOL, key
    # pairwise index masks: bit-exact against the reference's fp32 clamp sets
    m = out['masks'].cpu().numpy()
    assert np.array_equal(m[0], fx['mask_gg_lt_lb'])
    assert np.array_equal(m[1], fx['mask_gd_gt_ub'])
    assert np.array_equal(m[2], fx['mask_dd_gt_ub'])
    if 'dist_gg_f64' in fx:
        d = out['dist'].cpu().numpy()
        for i, key in enumerate(('dist_gg_f64', 'dist_gd_f64', 'dist_dd_f64')):
            assert np.max(np.abs(d[i] - fx[key])) <= 1e-5 * max(1.0, np.max(fx[key])), key
        assert np.all(np.diagonal(d[0]) == 0.0) and np.all(np.diagonal(d[2]) == 0.0)


@pytest.mark.parametrize('B,d,loss_type', [(64, 16, 'rep'), (64, 16, 'rmb'), (100, 16, 'rmb'), (257, 7, 'rep'),
                                           (1024, 16, 'rmb'), (64, 40, 'rep'), (2, 16, 'rep'), (3, 1, 'rmb')])
def test_mmd_loss_matches_c_oracle(ops, B, d, loss_type):
    rs = np.random.RandomState(B * 31 + d)
    s_gen = (rs.randn(B, d) * 0.25).astype(np.float32)
    s_x = (rs.randn(B, d) * 0.3 + 0.1).astype(np.float32)
    ref = c_oracle.mmd(s_gen, s_x, loss_type, (0.0, -1.0), dtype=np.float64)
    out = ops.mmd_loss(dev(s_gen), dev(s_x), loss_type, (0.0, -1.0), need_grads=True)
    sc = out['scalars'].cpu().numpy().astype(np.float64)
    escale = float(np.max(ref['stats'][:3]))
    assert abs(sc[0] - ref['loss_gen']) <= loss_tol(ref['loss_gen'], escale)
    assert abs(sc[1] - ref['loss_dis']) <= loss_tol(ref['loss_dis'], escale)
    assert rel_err(sc[2:7], ref['stats']) <= 1e-5
    g = out['grads'].cpu().numpy()
    for i in range(4):
        assert rel_err(g[i], ref['grads'][i]) <= RTOL, i


LOSSX = golden('lossx_*.npz')
NEXT_LOSSES = ('mmd_g', 'mgb', 'hinge', 'logistic')


def _check_loss_against(out, lg, ld, grads_ref, pairwise):
    sc = out['scalars'].cpu().numpy().astype(np.float64)
    # MMD losses are differences of O(1) kernel means; the score losses are plain means
    escale = float(np.max(sc[2:5])) if pairwise else 0.0
    assert abs(sc[0] - lg) <= loss_tol(lg, escale), (sc[0], lg)
    assert abs(sc[1] - ld) <= loss_tol(ld, escale), (sc[1], ld)
    g = out['grads'].cpu().numpy()
    for i in range(4):
        assert rel_err(g[i], grads_ref[i]) <= RTOL, i


@pytest.mark.parametrize('path', LOSSX, ids=[p.split('/')[-1] for p in LOSSX])
def test_next_row_losses_match_reference_golden(ops, path):
    """SURVEY 8(f) row 1 ('mmd_g', 'mgb', 'hinge', 'logistic') against the reference's own GANLoss outputs."""
    fx = load(path)
    lt = str(fx['loss_type'])
    out = ops.mmd_loss(dev(fx['s_gen']), dev(fx['s_x']), lt, need_grads=True)
    _check_loss_against(out, float(fx['loss_gen_f64']), float(fx['loss_dis_f64']),
                        [fx[k] for k in ('dLg_dsgen_f64', 'dLg_dsx_f64', 'dLd_dsgen_f64', 'dLd_dsx_f64')],
                        lt in ('mmd_g', 'mgb'))


@pytest.mark.parametrize('B,d', [(64, 16), (100, 1), (257, 7), (1024, 16), (2, 3)])
@pytest.mark.parametrize('loss_type', NEXT_LOSSES)
def test_next_row_losses_match_oracle(ops, B, d, loss_type):
    rs = np.random.RandomState(B * 31 + d)
    s_gen = (rs.randn(B, d) * 0.6).astype(np.float32)          # distances straddle both mgb bounds
    s_x = (rs.randn(B, d) * 0.5 + 0.2).astype(np.float32)
    sg = torch.tensor(s_gen, dtype=torch.float64, requires_grad=True)
    sx = torch.tensor(s_x, dtype=torch.float64, requires_grad=True)
    lg, ld, _ = R.gan_loss(sg, sx, loss_type, B)
    g = list(torch.autograd.grad(lg, [sg, sx], retain_graph=True, allow_unused=True))
    g += list(torch.autograd.grad(ld, [sg, sx], allow_unused=True))
    g = [np.zeros((B, d)) if t is None else t.numpy() for t in g]
    out = ops.mmd_loss(dev(s_gen), dev(s_x), loss_type, need_grads=True)
    _check_loss_against(out, float(lg), float(ld), g, loss_type in ('mmd_g', 'mgb'))
    h = ops.mmd_loss(dev(s_gen), dev(s_x), loss_type, grads_dis_first=True)['grads']
    for slot, src in enumerate((3, 2, 0, 1)):
        assert torch.equal(h[slot], out['grads'][src]), (loss_type, slot)


def test_losses_at_random_sizes(ops):
    """every loss of the pairwise kernel and the two score losses at 48 drawn (batch, score width) pairs - batches 2 ... 400
    that are no tile multiple, widths 1 ... 48 - against the restatement in float64 with autograd: both losses, the kernel
    means, all four gradient blocks.  (The fixed cases above pin the reference's own sizes; this walks the ragged ones.)"""
    rs = np.random.RandomState(77)
    for i in range(48):
        B, d = int(rs.randint(2, 401)), int(rs.randint(1, 49))
        loss_type = ('rep', 'rmb', 'mmd_g', 'mgb', 'hinge', 'logistic')[i % 6]
        spread = float(rs.choice([0.05, 0.3, 0.8]))
        s_gen = (rs.randn(B, d) * spread).astype(np.float32)
        s_x = (rs.randn(B, d) * spread + 0.2 * spread).astype(np.float32)
        sg = torch.tensor(s_gen, dtype=torch.float64, requires_grad=True)
        sx = torch.tensor(s_x, dtype=torch.float64, requires_grad=True)
        lg, ld, _ = R.gan_loss(sg, sx, loss_type, B)
        g = list(torch.autograd.grad(lg, [sg, sx], retain_graph=True, allow_unused=True))
        g += list(torch.autograd.grad(ld, [sg, sx], allow_unused=True))
        g = [np.zeros((B, d)) if t is None else t.numpy() for t in g]
        out = ops.mmd_loss(dev(s_gen), dev(s_x), loss_type, need_grads=True)
        try:
            _check_loss_against(out, float(lg.detach()), float(ld.detach()), g, loss_type not in ('hinge', 'logistic'))
        except AssertionError as e:
            raise AssertionError((B, d, loss_type, spread)) from e


MIX = golden('lossmix_*.npz')


@pytest.mark.parametrize('path', MIX, ids=[p.split('/')[-1] for p in MIX])
def test_mix_losses_match_reference_golden(ops, path):
    """'mmd_g_mix' / 'fixed_g_mix' / 'sgm' (math_func.py:2195-2263) with the reference's uniform draw and state injected:
    mix_indices and both concatenated group masks BIT-EXACT (np.array_equal), losses and the four gradients at 1e-4,
    the two state variables after the UPDATE_OPS, in both gradient orders"""
    fx = load(path)
    thr = float(fx['mix_threshold'])
    B = fx['s_gen'].shape[0]
    for dis_first in (False, True):
        state = dev(fx['state_in'])
        out = ops.mmd_mix_loss(dev(fx['s_gen']), dev(fx['s_x']), dev(fx['uni']), state, str(fx['loss_type']),
                               None if thr < 0 else thr, need_masks=True, grads_dis_first=dis_first)
        for k in ('mix_indices', 'mix_group_1', 'mix_group_2'):
            assert np.array_equal(out['masks'][k].cpu().numpy(), fx[k]), k
        sc = out['scalars'].cpu().numpy().astype(np.float64)
        # the losses are differences of kernel means (five of them summed for the mixture): floor as loss_tol
        escale = max(sc[2:5].max(), 1.0)
        assert abs(sc[0] - float(fx['loss_gen_f64'])) <= loss_tol(float(fx['loss_gen_f64']), escale), (sc[0], fx['loss_gen_f64'])
        assert abs(sc[1] - float(fx['loss_dis_f64'])) <= loss_tol(float(fx['loss_dis_f64']), escale), (sc[1], fx['loss_dis_f64'])
        assert sc[5] == np.float32(fx['state_in'][0]) and sc[6] == np.float32(fx['state_in'][1])       # as used
        assert int(sc[7]) == int(fx['mix_indices'].sum())
        g = out['grads'].cpu().numpy()
        order = ('dLd_dsx', 'dLd_dsgen', 'dLg_dsgen', 'dLg_dsx') if dis_first else ('dLg_dsgen', 'dLg_dsx', 'dLd_dsgen', 'dLd_dsx')
        gscale = max(np.abs(fx[k + '_f64']).max() for k in order)
        for i, key in enumerate(order):
            ref = fx[key + '_f64']
            assert np.abs(g[i] - ref).max() <= RTOL * np.abs(ref).max() + 1e-6 * gscale, key
        got_state = state.cpu().numpy().astype(np.float64)
        assert np.abs(got_state - fx['state_out_f64']).max() <= 1e-6, (got_state, fx['state_out_f64'])


def test_mix_loss_at_larger_batches_and_errors(ops):
    """the coin and the group bookkeeping against the oracle restatement where the fixtures do not reach: ragged
    batches, all / no rows mixed, B in the thousands; and the entry's argument checks"""
    rs = np.random.RandomState(11)
    for B, d, loss_type, prob in ((100, 7, 'sgm', 0.3), (257, 16, 'mmd_g_mix', 0.5), (1024, 16, 'mmd_g_mix', 0.1),
                                  (64, 16, 'sgm', 0.0), (64, 3, 'mmd_g_mix', 1.5), (2, 1, 'sgm', 0.4)):
        sg = (rs.randn(B, d) * 0.4).astype(np.float32)
        sx = (rs.randn(B, d) * 0.5 + 0.1).astype(np.float32)
        uni = rs.uniform(0, 1, B).astype(np.float32)
        st_in = (np.float32(0.7), np.float32(prob))
        tg, tx = torch.tensor(sg, dtype=torch.float64, requires_grad=True), torch.tensor(sx, dtype=torch.float64, requires_grad=True)
        lg, ld, info = R.gan_loss_mix(tg, tx, loss_type, B, uni, st_in)
        gld = torch.autograd.grad(ld, [tg, tx])
        state = dev(np.asarray(st_in))
        out = ops.mmd_mix_loss(dev(sg), dev(sx), dev(uni), state, loss_type, need_masks=True)
        for k in ('mix_indices', 'mix_group_1', 'mix_group_2'):
            assert np.array_equal(out['masks'][k].cpu().numpy(), info[k].numpy()), (B, k)
        sc = out['scalars'].cpu().numpy().astype(np.float64)
        escale = max(sc[2:5].max(), 1.0)
        assert abs(sc[0] - float(lg)) <= loss_tol(float(lg), escale) and abs(sc[1] - float(ld)) <= loss_tol(float(ld), escale)
        g = out['grads'].cpu().numpy()
        gscale = max(float(gld[0].abs().max()), float(gld[1].abs().max()), 1e-12)
        assert np.abs(g[2] - gld[0].numpy()).max() <= RTOL * gscale and np.abs(g[3] - gld[1].numpy()).max() <= RTOL * gscale
        assert np.abs(state.cpu().numpy() - np.asarray(info['new_state'])).max() <= 1e-6
    z = dev(np.zeros((4, 2), np.float32))
    with pytest.raises(ValueError, match='use mmd_mix_loss'):
        ops.mmd_loss(z, z, 'sgm')
    with pytest.raises(NotImplementedError, match='Not implemented.'):
        ops.mmd_mix_loss(z, z, dev(np.zeros(4)), dev(np.zeros(2)), 'rep')
    with pytest.raises(ValueError, match='batch_size'):
        ops.mmd_mix_loss(z[:1], z[:1], dev(np.zeros(1)), dev(np.zeros(2)), 'sgm')


def test_mmd_size_independent_properties(ops):
    """at the benchmark's full sweep sizes: permutation invariance, x<->y symmetry of loss_gen,
    zero loss_gen for identical sets, translation invariance."""
    rs = np.random.RandomState(3)
    B, d = 4096, 16
    a = (rs.randn(B, d) * 0.25).astype(np.float32)
    b = (rs.randn(B, d) * 0.3 + 0.1).astype(np.float32)
    base = ops.mmd_loss(dev(a), dev(b), 'rep', need_grads=False)['scalars'].cpu().numpy()
    perm = rs.permutation(B)
    # the cross block drops pairs (i,i): permuting BOTH sets identically keeps every term
    p = ops.mmd_loss(dev(a[perm]), dev(b[perm]), 'rep', need_grads=False)['scalars'].cpu().numpy()
    assert abs(p[0] - base[0]) <= 2e-6 * max(abs(base[0]), 1e-3) + 1e-7
    sw = ops.mmd_loss(dev(b), dev(a), 'rep', need_grads=False)['scalars'].cpu().numpy()
    assert abs(sw[0] - base[0]) <= 2e-6 * max(abs(base[0]), 1e-3) + 1e-7       # loss_gen symmetric
    assert abs(sw[2] - base[4]) <= 1e-6 and abs(sw[4] - base[2]) <= 1e-6       # e_kxx <-> e_kyy
    same = ops.mmd_loss(dev(a), dev(a), 'rep', need_grads=False)['scalars'].cpu().numpy()
    # identical sets: e_kxx = e_kyy, e_kxy = the same sum (the dropped diagonal is the same set)
    assert abs(same[0]) <= 1e-6
    sh = ops.mmd_loss(dev(a + 0.5), dev(b + 0.5), 'rep', need_grads=False)['scalars'].cpu().numpy()
    assert abs(sh[0] - base[0]) <= 1e-4 * abs(base[0]) + 1e-6


def test_mmd_grads_dis_first_order(ops):
    rs = np.random.RandomState(3)
    a, b = dev(rs.randn(64, 16).astype(np.float32)), dev(rs.randn(64, 16).astype(np.float32))
    for lt in ('rep', 'rmb'):
        g = ops.mmd_loss(a, b, lt)['grads']
        h = ops.mmd_loss(a, b, lt, grads_dis_first=True)['grads']
        for slot, src in enumerate((3, 2, 0, 1)):
            assert torch.equal(h[slot], g[src]), (lt, slot)


def test_mmd_errors(ops):
    z = torch.zeros(8, 16).cuda()
    with pytest.raises(NotImplementedError, match='Not implemented.'):          # math_func.py:2651
        ops.mmd_loss(z, z, 'wasserstein')
    with pytest.raises(ValueError):                                               # no pairwise matrices to return
        ops.mmd_loss(z, z, 'hinge', need_dist=True)
    with pytest.raises(ValueError, match=r'w\[0\]-w\[1\] must be 1'):            # math_func.py:1340
        ops.mmd_loss(z, z, 'rep', rep_weights=(0.5, 0.0))
    with pytest.raises(ValueError):
        ops.mmd_loss(z[:1], z[:1], 'rep')


# ---------------------------------------------------------------------------------------------
# convolution family vs the oracle's linear operators
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, C, K, R, stride
    (2, 8, 8, 3, 8, 3, 1), (3, 12, 12, 5, 24, 3, 1), (3, 12, 12, 24, 40, 4, 2), (2, 9, 7, 4, 6, 3, 1),
    (4, 32, 32, 3, 64, 3, 1),            # D l1
    (4, 32, 32, 64, 3, 3, 1),            # G l5
    (4, 32, 32, 64, 128, 4, 2),          # D l2
    (4, 16, 16, 128, 128, 3, 1),         # D l3
    (8, 8, 8, 256, 512, 4, 2),           # D l6
    (8, 4, 4, 512, 512, 3, 1),           # D l7
    (1, 16, 16, 128, 128, 3, 1),         # SN batch-1
    (1, 4, 4, 512, 512, 3, 1),
    (2, 24, 24, 64, 128, 4, 2),          # STL-like
    (3, 6, 6, 256, 256, 3, 1),
    # few tiles -> 8-wave K-group kernels, split reductions with a ragged last split
    (32, 16, 16, 64, 128, 3, 1), (16, 8, 8, 128, 128, 4, 2), (8, 4, 4, 256, 256, 3, 1), (5, 10, 10, 64, 64, 3, 1),
    # thin layers on the MFMA patch-GEMM kernels: ragged pixel tiles, partial row bands, wide 32/64/128,
    # 5x5 taps, 1..3 thin channels, STL / 64x64 image sizes
    (3, 10, 12, 3, 32, 3, 1), (2, 7, 6, 64, 2, 3, 1), (2, 48, 48, 3, 64, 3, 1), (2, 48, 48, 64, 3, 3, 1),
    (1, 64, 64, 128, 3, 3, 1), (1, 64, 64, 3, 128, 3, 1), (2, 9, 10, 1, 64, 5, 1), (2, 11, 8, 32, 1, 5, 1),
]


def ew_floor(ops, case, dgrad, base):
    """floor_frac of helpers.elementwise_err for the kernel the LIBRARY picks for this geometry on its own: `base` (entries
    above it are held to 1e-4 of themselves, the ones below to base * 1e-4 of the tensor's scale), 5e-2 where that is
    F(4x4,3x3) - its transforms carry constants up to 8 and its rounding is 3-5e-6 of the output scale whatever the entry's
    size (tools/wino43_gate.py; measured 1.4-4.9e-6 in test_conv2d_winograd_f43_path), against 5e-7 for F(2x2,3x3)"""
    N, H, W, C, K, R, s = case
    return 5e-2 if ops.wino_algo(N, H, W, C, K, R, s, dgrad) == ops.WINO_F43 else base


def conv_data(case, seed=0):
    N, H, W, C, K, R, s = case
    rs = np.random.RandomState(seed)
    x = rs.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    w = (rs.randn(R, R, C, K) / np.sqrt(R * R * C)).astype(np.float32)
    b = (rs.randn(K) * 0.1).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_fwd(ops, case):
    N, H, W, C, K, ksz, s = case
    x, w, b = conv_data(case)
    sc = np.float32(0.37)
    for act in ('linear', 'lrelu', 'tanh'):
        ref = R._act(R.conv2d_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), s) * float(sc)
                     + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), act).numpy()
        y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act=act)
        assert rel_err(to_nchw(y), ref) <= RTOL, act
        # ... and element by element: 1e-4 of EACH activation above 2 % of the tensor's scale (helpers.elementwise_err; the
        # absolute floor 2e-6 of the scale covers the longest reductions here, 4096 terms: measured 1.1e-6 at a zero crossing)
        ff = ew_floor(ops, case, False, 2e-2)
        assert elementwise_err(to_nchw(y), ref, floor_frac=ff) <= RTOL, (act, elementwise_err(to_nchw(y), ref, ff))


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_dgrad_and_wgrad(ops, case):
    N, H, W, C, K, ksz, s = case
    x, w, _ = conv_data(case, 1)
    P, Q = -(-H // s), -(-W // s)
    rs = np.random.RandomState(5)
    dy = rs.randn(N, K, P, Q).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    yt = R.conv2d_same(xt, wt, s)
    gx, gw = torch.autograd.grad((yt * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, wt])
    dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s)
    assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
    dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s)
    assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
    db = torch.full((K,), 7.0, device='cuda')                      # conv2d_wgrad_bias: + column sums of dy, overwritten
    dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, dbias=db)
    assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
    assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum((0, 2, 3))) <= RTOL
    # with a registered library workspace the thin layers take the partial-sum (MFMA) kernels
    ops.set_workspace()
    try:
        dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s)
        assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    # backward epilogue form: scale * dgrad * lrelu'(y_prev)
    yprev = rs.randn(N, C, H, W).astype(np.float32)
    ref = 0.5 * gx.numpy() * np.where(yprev > 0, 1.0, 0.1)
    dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev))
    assert rel_err(to_nchw(dx2), ref) <= RTOL


# the benchmark's own launches, at full size (too large for the CPU oracle): size-independent properties of a
# linear operator and its two gradients
FULL_SIZE = [
    # N, H, W, C, K, R, stride                                    (rows 3B = 192 are the batched backward pass)
    (128, 32, 32, 64, 128, 4, 2), (192, 32, 32, 64, 128, 4, 2),   # CIFAR D l2: F(2x2,2x2) kernels
    (128, 16, 16, 128, 128, 3, 1), (192, 16, 16, 128, 128, 3, 1),  # CIFAR D l3: F(2x2,3x3) kernels
    (128, 8, 8, 256, 512, 4, 2), (128, 4, 4, 512, 512, 3, 1),     # CIFAR D l6, l7
    (128, 32, 32, 3, 64, 3, 1),                                   # CIFAR D l1 (thin)
    (64, 32, 32, 64, 128, 4, 2),                                  # G l4_up as the conv it is the input-gradient of
    (256, 64, 64, 64, 128, 4, 2),                                 # CelebA D l2 at batch 2B = 256: 268 MB of activations
    (64, 64, 64, 128, 128, 3, 1),                                 # the ResNet-SN config's largest 3x3
    # the ResNet-SN discriminator's first block at its bench batch 32 (3B = 96 rows): 64 -> 64 channels, the 4x4 stride-2
    # kernel its folded avg-pool makes.  Its input-gradient has a 64-channel reduction = work items of TWO pipeline stages in
    # the F(2x2,2x2) kernel, and each workgroup walks six of them: the launch that stored tiles at another item's addresses
    # (csrc/conv_wino2.hip, OB_SLOTS) - wrong by 100 % in L2, and caught by this identity
    (96, 64, 64, 64, 64, 4, 2), (64, 64, 64, 64, 64, 4, 2),
]


@pytest.mark.parametrize('case', FULL_SIZE, ids=[str(c) for c in FULL_SIZE])
def test_conv_full_size_adjointness_and_linearity(ops, case):
    """<conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)> (one bilinear form, three kernels), conv linear in x;
    every launch goes through the library's own dispatch (Winograd / implicit GEMM / thin) for that size"""
    N, H, W, C, K, ksz, s = case
    g = torch.Generator(device='cuda').manual_seed(N + C)
    P, Q = -(-H // s), -(-W // s)
    x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    x2 = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    w = torch.randn(ksz, ksz, C, K, device='cuda', generator=g) / float(np.sqrt(ksz * ksz * C))
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g)
    y = ops.conv2d_fwd(x, w, s)
    dx = ops.conv2d_dgrad(dy, w, (H, W), s)
    dw = ops.conv2d_wgrad(x, dy, ksz, s)

    def dot(a, b):
        return float((a.double() * b.double()).sum())
    form = dot(y, dy)
    scale = float(y.double().norm() * dy.double().norm())                # |<y, dy>| <= scale; the form itself is ~scale/sqrt(n)
    assert abs(dot(x, dx) - form) <= 1e-6 * scale, (form, dot(x, dx), scale)
    assert abs(dot(w, dw) - form) <= 1e-6 * scale, (form, dot(w, dw), scale)
    lin = ops.conv2d_fwd(x + 2.0 * x2, w, s)
    ref = y + 2.0 * ops.conv2d_fwd(x2, w, s)
    assert float((lin - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # the batch is a batch: every image's result is independent of its neighbours (first / last image alone)
    for i in (0, N - 1):
        alone = ops.conv2d_fwd(x[i:i + 1].contiguous(), w, s)
        assert float((alone[0] - y[i]).abs().max()) <= 1e-4 * float(y[i].abs().max())


def _config_conv_shapes(batches=None):
    """every convolution geometry of every BASELINE.json config at its per-GPU bench batch, with the batch sizes its launches
    run at: (config, N, H, W, C, K, R, stride).  DCGAN dicts: D layers at 2B (forward, weight gradient) and 3B (the joint
    input-gradient pass) rows, G's transposed layers as the conv they are the input-gradient of, at B rows.  ResNet-SN: the
    primitive-op engine's kernels (a block's scaling op folded into its 3x3 conv makes a 4x4 stride-2 geometry)."""
    import configs
    out = []
    batches = batches or {'cifar': 64, 'stl': 64, 'celeba': 128, 'lsun_resnet': 32}
    for name in ('cifar', 'stl', 'celeba'):
        B = batches[name]
        arch, _ = configs.CONFIGS[name]()
        c, h, _w = arch['input'][0]
        for d in arch['discriminator']:
            if d.get('op', 'c') != 'c':
                continue
            R, s, k = d.get('kernel', 3), d.get('strides', 1), d['out']
            out += [(name, n, h, h, c, k, R, s) for n in (2 * B, 3 * B)]
            c, h = k, -(-h // s)
        c, h = None, None
        for d in arch['generator']:
            if d.get('op', 'c') == 'd':
                c, h = d['out_reshape'][0], d['out_reshape'][1]
            elif d.get('op') == 'tc':                    # conv geometry: input = the layer's output
                out.append((name, B, h * d['strides'], h * d['strides'], d['out'], c, d['kernel'], d['strides']))
                c, h = d['out'], h * d['strides']
            else:
                out.append((name, B, h, h, c, d['out'], d.get('kernel', 3), d.get('strides', 1)))
                c = d['out']
    from mmdgan_hip.tape import _Net
    arch, _ = configs.lsun_resnet()
    B = batches['lsun_resnet']
    for net_name, designs, in_ref, batches in (('gen', arch['generator'], [arch['code'][0][0]], (B,)),
                                               ('dis', arch['discriminator'], list(arch['input'][0]), (2 * B, 3 * B))):
        net = _Net(designs, in_ref, net_name, torch.device('cuda'), np.random.RandomState(0), 'default')
        for k in net.kernels:
            if k.op not in ('c', 'tc'):
                continue
            R, s = (4, 2) if k.fold is not None else (k.R, k.stride)
            if k.op == 'c' and k.fold != 'unpool':
                c, h, kk = k.in_ref[0], k.in_ref[1], k.out
            else:                                        # transposed form: the conv whose input is the layer's output
                c, h, kk = k.out, k.out_ref[1], k.in_ref[0]
            out += [('lsun_resnet', n, h, h, c, kk, R, s) for n in batches]
    return sorted(set(out))


@pytest.mark.parametrize('which', ['bench', 'ragged'])
def test_conv_shapes_of_every_config_at_their_bench_batch(ops, which):
    """('ragged': the same geometries at batch sizes that are no multiple of anything - 24 / 24 / 40 / 12 per GPU - so tile
    blocks end ragged and the persistent kernels' work-item counts per workgroup differ from the bench's.)
    the adjointness identity  <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)>  on EVERY convolution geometry of
    every BASELINE.json config at the batch sizes its launches run at (2B / 3B rows in D, B in G), with the weights handed over
    transformed wherever the library's selection takes a Winograd kernel - i.e. each launch as the engines issue it, under
    whatever kernel selection this process runs (tests/test_production_gpu.py repeats the step itself under the production
    one).  Three kernels computing one bilinear form disagree when any of them mis-addresses a tile, whatever the size."""
    shapes = _config_conv_shapes(None if which == 'bench' else {'cifar': 24, 'stl': 24, 'celeba': 40, 'lsun_resnet': 12})
    assert len(shapes) >= 60
    bad = []
    ops.set_workspace(128 << 20)                         # what an engine's handle registers (slab / partial-sum paths need it)
    for (cfg, N, H, W, C, K, R, s) in shapes:
        g = torch.Generator(device='cuda').manual_seed(N * 7 + C + K)
        P, Q = -(-H // s), -(-W // s)
        x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
        w = torch.randn(R, R, C, K, device='cuda', generator=g) / float(np.sqrt(R * R * C))
        dy = torch.randn(N, P, Q, K, device='cuda', generator=g)
        wino_ok = R in (3, 4)
        uf = ops.wino_transform(w, False) if wino_ok and ops.wino_eligible(N, H, W, C, K, R, s, False) else None
        ud = ops.wino_transform(w, True) if wino_ok and ops.wino_eligible(N, H, W, C, K, R, s, True) else None
        y = ops.conv2d_fwd(x, w, s, wino=uf)
        dx = ops.conv2d_dgrad(dy, w, (H, W), s, wino=ud)
        dw = ops.conv2d_wgrad(x, dy, R, s)

        def dot(a, b):
            return float((a.double() * b.double()).sum())
        form, scale = dot(y, dy), float(y.double().norm() * dy.double().norm())
        e1, e2 = abs(dot(x, dx) - form) / scale, abs(dot(w, dw) - form) / scale
        # (a mis-addressed tile is an error of 1e-3 and up; rounding is what the bar leaves room for: 1e-6, and 5e-6 where the
        # weight gradient takes F(4x4,3x3) - its transforms carry constants up to 8 and one accumulator takes up to 2048
        # products: measured 2.8e-6 on CelebA's 16x16x256 layer at 384 rows, 1-1.6e-6 element-wise in its own parity test)
        bar2 = 5e-6 if ops.wgrad_algo(N, H, W, C, K, R, s) == ops.WINO_F43 else 1e-6
        if not (e1 <= 1e-6 and e2 <= bar2):
            bad.append(((cfg, N, H, W, C, K, R, s), e1, e2))
        del x, w, dy, y, dx, dw
    ops.require_device().mmdgan_set_workspace(None, 0)
    assert not bad, bad


ADD_CASES = [
    # N, H, W, C, K, R, stride                          which kernel takes it under the test thresholds
    (16, 16, 16, 64, 128, 3, 1),                          # F(2x2,3x3), unsplit: in the kernel's stores
    (4, 8, 8, 128, 128, 3, 1),                            # F(2x2,3x3) with the reduction split over slabs: in the slab pass
    (16, 16, 16, 64, 128, 4, 2),                          # F(2x2,2x2): the entry's axpby pass (unsplit) / the slab pass
    (8, 16, 16, 64, 64, 1, 1),                            # 1x1 shortcut conv: implicit GEMM
    (8, 16, 16, 3, 64, 1, 1), (8, 32, 32, 3, 64, 3, 1),   # thin input: direct / thin kernels
    (5, 9, 7, 5, 6, 3, 1),                                # generic direct kernel
]


@pytest.mark.parametrize('case', ADD_CASES, ids=[str(c) for c in ADD_CASES])
def test_conv_with_an_addend_equals_conv_plus_addend(ops, case):
    """mmdgan_conv2d_fwd_add / _dgrad_add: out = epilogue(conv) + addend, whichever kernel the geometry takes (its own
    epilogue or the entry's axpby pass), with an activation / activation derivative in front of the sum, with transformed
    weights handed in - against the same launch without the addend.  The addend may not be the output buffer."""
    N, H, W, C, K, R, s = case
    g = torch.Generator(device='cuda').manual_seed(N + C + K)
    P, Q = -(-H // s), -(-W // s)
    x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    w = torch.randn(R, R, C, K, device='cuda', generator=g) / float(np.sqrt(R * R * C))
    b = torch.randn(K, device='cuda', generator=g) * 0.1
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g)
    ay = torch.randn(N, P, Q, K, device='cuda', generator=g)
    ax = torch.randn(N, H, W, C, device='cuda', generator=g)
    sc = torch.tensor([0.37], device='cuda')
    ops.set_workspace(128 << 20)
    try:
        for wino in (False, True):
            ok = wino and R in (3, 4)
            uf = ops.wino_transform(w, False) if ok and ops.wino_eligible(N, H, W, C, K, R, s, False) else None
            ud = ops.wino_transform(w, True) if ok and ops.wino_eligible(N, H, W, C, K, R, s, True) else None
            if wino and uf is None and ud is None:
                continue
            y0 = ops.conv2d_fwd(x, w, s, bias=b, scale=sc, act='lrelu', wino=uf)
            y1 = ops.conv2d_fwd(x, w, s, bias=b, scale=sc, act='lrelu', wino=uf, addend=ay)
            assert torch.allclose(y1, y0 + ay, rtol=1e-6, atol=1e-6), ('fwd', wino)
            d0 = ops.conv2d_dgrad(dy, w, (H, W), s, scale=sc, act='lrelu', dact_of=x, wino=ud)
            d1 = ops.conv2d_dgrad(dy, w, (H, W), s, scale=sc, act='lrelu', dact_of=x, wino=ud, addend=ax)
            assert torch.allclose(d1, d0 + ax, rtol=1e-6, atol=1e-6), ('dgrad', wino)
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    lib = ops.require_device()
    import ctypes
    gm = ops.geom(N, H, W, C, K, R, s)
    for bad in (None, y0.data_ptr()):                    # no addend; the addend is the output
        rc = lib.mmdgan_conv2d_fwd_add(ctypes.byref(gm), x.data_ptr(), w.data_ptr(), None, None, 0, None, 0, bad, y0.data_ptr(), None)
        assert rc == -1 and b'addend' in lib.mmdgan_last_error()


def test_random_gemm_shapes(ops):
    """300 random dense-layer products through mmdgan_gemm - both operand forms, bias, scale, activation forward and its
    derivative backward with the 3B-row wrap, the split-K licence (out_zeroed) on a zeroed output, the skinny-N MFMA route
    (N = 16, K % 256 == 0) - against a float64 product of the same operands."""
    rs = np.random.RandomState(20260929)
    dims = [1, 2, 3, 5, 8, 16, 17, 31, 32, 48, 64, 65, 96, 100, 128, 192, 256, 384, 512, 1000, 1024, 2048, 4096, 8192]
    bad = []
    for i in range(300):
        M, N, K = (int(dims[rs.randint(len(dims))]) for _ in range(3))
        if rs.rand() < 0.2:
            N, K = 16, int(rs.choice([1024, 2048, 4608, 8192, 18432]))
            M = int(rs.choice([16, 48, 64, 128, 192, 256]))
        if M * N > 4e6 or M * K > 16e6 or N * K > 16e6:
            continue
        ta, tb = bool(rs.rand() < 0.3), bool(rs.rand() < 0.3)
        g = torch.Generator(device='cuda').manual_seed(1000 + i)
        a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
        b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
        bias = torch.randn(N, device='cuda', generator=g)
        sc = torch.tensor([0.7], device='cuda')
        ref = ((a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())) * 0.7 + bias.double()
        mode = i % 3
        if mode == 0:                                    # forward form
            got = ops.gemm(a, b, ta, tb, bias=bias, scale=sc, act='lrelu')
            want = torch.where(ref > 0, ref, 0.1 * ref)
        elif mode == 1:                                  # linear, into a zeroed output: the launch may split K and accumulate
            out = torch.zeros(M, N, device='cuda')
            got = ops.gemm(a, b, ta, tb, bias=bias, scale=sc, out=out, out_zeroed=True)
            want = ref
        else:                                            # backward form, the operand holding 2/3 of the rows where M allows
            rows = 2 * M // 3 if (M % 3 == 0 and M >= 3) else M
            y = torch.randn(rows, N, device='cuda', generator=g)
            full = torch.cat([y, y[rows - (M - rows):]], 0) if rows < M else y
            got = ops.gemm(a, b, ta, tb, bias=bias, scale=sc, act='lrelu', dact_of=y, dact_rows=rows if rows < M else 0)
            want = ref * torch.where(full > 0, 1.0, 0.1).double()
        err = float((got.double() - want).abs().max() / (want.abs().max() + 1e-30))
        if not err <= 2e-5:                              # (fp32 sums of up to 18432 terms against float64)
            bad.append(((M, N, K, ta, tb, mode), err))
    assert not bad, bad[:8]


@pytest.mark.parametrize('where', ['this process', 'production selection'])
def test_random_conv_geometries(where):
    """150 random geometries (tests/conv_fuzz.py, fixed seed: channels 3 ... 512, sizes 4 ... 64 also non-square, batches
    1 ... 200, kernels 1 / 2 / 3 / 4 / 5, strides 1 / 2) plus a 96-case sweep of the Winograd kernels' channel classes at
    sizes with many work items per workgroup, through the library's dispatch: adjointness of the three kernels,
    caller-transformed weights against the library's own transform, the fused epilogues (with the 3B-row wrap) against
    the linear launch finished in torch, batch independence.  Once under this process's
    thresholds and once in a subprocess with no MMDGAN_* variable set - the selection bench.py runs."""
    import json
    import subprocess
    import conv_fuzz
    if where == 'this process':
        out = conv_fuzz.run(20260929, 150)
    else:
        env = {k: v for k, v in os.environ.items() if not k.startswith('MMDGAN_')}
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'conv_fuzz.py'), '20260929', '150'], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out['env'] == []
    assert out['cases'] == 150 + 96 and not out['bad'], out['bad'][:5]


SPARSE_TAP_CASES = [(4, 16, 16, 16, 8, 1, 2), (3, 9, 12, 8, 8, 2, 3), (2, 8, 8, 64, 64, 1, 2), (5, 7, 7, 3, 16, 1, 3)]


@pytest.mark.parametrize('case', SPARSE_TAP_CASES, ids=[str(c) for c in SPARSE_TAP_CASES])
def test_conv2d_kernel_smaller_than_its_stride(ops, case):
    """kernels smaller than their stride (the 1x1 stride-2 transposed conv a residual block on 'tc' can have as its shortcut,
    layer_func.py:1725-1745): taps skip input pixels, the input-gradient leaves pixels no tap reaches at zero (+ bias).  All three
    kernels and the fused epilogues against the oracle's conv and its autograd."""
    N, H, W, C, K, ksz, s = case
    x, w, b = conv_data(case, 5)
    P, Q = -(-H // s), -(-W // s)
    rs = np.random.RandomState(11)
    dy = rs.randn(N, K, P, Q).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    yt = R.conv2d_same(xt, wt, s)
    assert tuple(yt.shape[2:]) == (P, Q)
    gx, gw = torch.autograd.grad((yt * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, wt])
    sc = np.float32(0.8)
    ref = R._act(yt.detach() * float(sc) + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), 'lrelu').numpy()
    y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='lrelu')
    assert rel_err(to_nchw(y), ref) <= RTOL
    dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s)
    assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
    bc = (rs.randn(C) * 0.1).astype(np.float32)                     # the transposed-conv forward form
    y2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, bias=dev(bc), act='relu')
    assert rel_err(to_nchw(y2), np.maximum(gx.numpy() + bc.reshape(1, -1, 1, 1), 0)) <= RTOL
    dbias = torch.zeros(K, device='cuda')
    dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, dbias=dbias)
    assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
    assert rel_err(dbias.cpu().numpy(), dy.sum((0, 2, 3))) <= RTOL


WINO2_CASES = [(16, 16, 16, 64, 128, 4, 2), (30, 12, 12, 32, 64, 4, 2), (6, 8, 16, 64, 64, 4, 2), (9, 4, 4, 128, 64, 4, 2),
               (33, 8, 8, 96, 192, 4, 2), (5, 12, 8, 64, 128, 4, 2), (9, 4, 4, 128, 256, 4, 2), (70, 8, 8, 64, 128, 4, 2)]


@pytest.mark.parametrize('case', WINO2_CASES, ids=[str(c) for c in WINO2_CASES])
def test_conv2d_winograd_stride2_path(ops, case):
    """4x4 / stride-2 layers: F(2x2,2x2) on the parity decomposition - forward (4 input-parity segments),
    input-gradient / transposed-conv forward (4 output-parity phases), 3B-row dact wrap, ragged tile blocks."""
    N, H, W, C, K, ksz, s = case
    x, w, b = conv_data(case, 3)
    P, Q = H // 2, W // 2
    rs = np.random.RandomState(10)
    dy = rs.randn(N, K, P, Q).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64)
    yt = R.conv2d_same(xt, wt, s)
    wt.requires_grad_(True)
    gx, gw = torch.autograd.grad((R.conv2d_same(xt, wt, s) * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, wt])
    wt = wt.detach()
    if C % 64 == 0 and K % 128 == 0:                     # F(2x2,2x2)-domain weight gradient, four tap-parity problems
        dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s)
        assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
    fwd_ok, bwd_ok = ops.wino_eligible(N, H, W, C, K, ksz, s, False), ops.wino_eligible(N, H, W, C, K, ksz, s, True)
    assert fwd_ok or bwd_ok
    uf, ub = ops.wino_transform(dev(w), False), ops.wino_transform(dev(w), True)
    assert uf.shape == (4, 9, C, K) and ub.shape == (4, 9, K, C)
    sc = np.float32(0.61)
    if fwd_ok:
        for act in ('linear', 'lrelu'):
            ref = R._act(yt.detach() * float(sc) + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), act).numpy()
            y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act=act, wino=uf)
            assert rel_err(to_nchw(y), ref) <= RTOL, act
    if bwd_ok:
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
        assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
        bc = (rs.randn(C) * 0.1).astype(np.float32)                 # transposed-conv forward form: relu(dgrad + bias)
        y2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, bias=dev(bc), act='relu', wino=ub)
        assert rel_err(to_nchw(y2), np.maximum(gx.numpy() + bc.reshape(1, -1, 1, 1), 0)) <= RTOL
        if N % 3 == 0:
            B = N // 3
            yprev = rs.randn(2 * B, C, H, W).astype(np.float32)
            mask = np.where(np.concatenate([yprev, yprev[B:]], 0) > 0, 1.0, 0.1)
            dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev),
                                   dact_batch=2 * B, wino=ub)
            assert rel_err(to_nchw(dx2), 0.5 * gx.numpy() * mask) <= RTOL
    # library-side transform (workspace) takes the same kernels
    ops.set_workspace()
    try:
        y = ops.conv2d_fwd(nhwc(x), dev(w), s)
        assert rel_err(to_nchw(y), yt.detach().numpy()) <= RTOL
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s)
        assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
        # transformed weights + a workspace: a launch with few workgroups cuts its reduction into parts (partial sums in
        # workspace slabs, a second pass sums them and applies the epilogue) - every case here is below the 384-workgroup line
        if fwd_ok:
            ref = R._act(yt.detach() * float(sc) + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), 'lrelu').numpy()
            y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='lrelu', wino=uf)
            assert rel_err(to_nchw(y), ref) <= RTOL
        if bwd_ok:
            y2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, bias=dev(bc), act='relu', wino=ub)
            assert rel_err(to_nchw(y2), np.maximum(gx.numpy() + bc.reshape(1, -1, 1, 1), 0)) <= RTOL
            if N % 3 == 0:
                dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev),
                                       dact_batch=2 * B, wino=ub)
                assert rel_err(to_nchw(dx2), 0.5 * gx.numpy() * mask) <= RTOL
        if C % 64 == 0 and K % 128 == 0:       # weight gradient: per-split partial slabs in the workspace + one reduction pass
            dw2 = torch.full((ksz, ksz, C, K), float('nan'), device='cuda')      # no zeroing needed, and twice the same bits
            ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, out=dw2)
            assert rel_err(dw2.cpu().numpy(), gw.numpy()) <= RTOL
            dw3 = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s)
            assert torch.equal(dw2, dw3)
            db = torch.full((K,), float('nan'), device='cuda')                   # the bias gradient rides along in the same kernel
            dw4 = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, dbias=db)
            assert torch.equal(dw2, dw4)
            assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum((0, 2, 3))) <= RTOL
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)


WINO_CASES = [(16, 16, 16, 128, 128, 3, 1), (32, 8, 8, 64, 128, 3, 1), (24, 4, 4, 256, 128, 3, 1), (9, 12, 12, 32, 64, 3, 1), (130, 4, 4, 64, 64, 3, 1),
              (5, 14, 18, 40, 192, 3, 1), (9, 10, 12, 64, 64, 3, 1), (7, 6, 10, 32, 256, 3, 1), (67, 4, 4, 96, 128, 3, 1)]


@pytest.mark.parametrize('case', WINO_CASES, ids=[str(c) for c in WINO_CASES])
def test_conv2d_winograd_path(ops, case):
    """3x3 / stride-1 layers with the library workspace registered run Winograd F(2x2,3x3) (forward and
    input-gradient, ragged last tile block, H != W, 3B-row dact wrap); same oracle, same tolerance."""
    N, H, W, C, K, ksz, s = case
    x, w, b = conv_data(case, 2)
    rs = np.random.RandomState(9)
    dy = rs.randn(N, K, H, W).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64)
    yt = R.conv2d_same(xt, wt, s)
    wt.requires_grad_(True)
    gx, gw = torch.autograd.grad((R.conv2d_same(xt, wt, s) * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, wt])
    wt = wt.detach()
    if C % 32 == 0:                                      # Winograd-domain weight gradient (+ fused-API bias gradient)
        db = torch.empty(K, device='cuda')
        dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, dbias=db)
        assert rel_err(dw.cpu().numpy(), gw.numpy()) <= RTOL
        assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum((0, 2, 3))) <= RTOL
    ops.set_workspace()
    try:
        sc = np.float32(0.37)
        for act in ('linear', 'lrelu'):
            ref = R._act(yt.detach() * float(sc) + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), act).numpy()
            y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act=act)
            assert rel_err(to_nchw(y), ref) <= RTOL, act
            ff = ew_floor(ops, case, False, 1e-2)        # (with a workspace the library's own choice is F(4x4,3x3) where H, W % 4 == 0)
            assert elementwise_err(to_nchw(y), ref, floor_frac=ff) <= RTOL, (act, elementwise_err(to_nchw(y), ref, ff))   # element by element
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s)
        assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
        if C % 32 == 0:        # weight gradient with the workspace: (K % 128 == 0) per-split slabs + one reduction pass, bias gradient fused
            dw2, db2 = torch.full((ksz, ksz, C, K), float('nan'), device='cuda'), torch.full((K,), float('nan'), device='cuda')
            ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, out=dw2, dbias=db2)
            assert rel_err(dw2.cpu().numpy(), gw.numpy()) <= RTOL
            assert rel_err(db2.cpu().numpy(), dy.astype(np.float64).sum((0, 2, 3))) <= RTOL
            if K % 128 == 0:
                assert torch.equal(dw2, ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s))      # deterministic, with or without dbias
        if N % 3 == 0:                                   # [2B ; B] rows against 2B activations
            B = N // 3
            yprev = rs.randn(2 * B, C, H, W).astype(np.float32)
            mask = np.where(np.concatenate([yprev, yprev[B:]], 0) > 0, 1.0, 0.1)
            dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev),
                                   dact_batch=2 * B)
            assert rel_err(to_nchw(dx2), 0.5 * gx.numpy() * mask) <= RTOL
        # weights transformed by the caller + a workspace: a launch with few workgroups (every case here) cuts its channel
        # reduction into parts (>= 64 channels each) over workspace slabs, a second pass sums them and applies the epilogue
        if ops.wino_eligible(N, H, W, C, K, ksz, s, False):
            uf = ops.wino_transform(dev(w), False)
            ref = R._act(yt.detach() * float(sc) + torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1), 'lrelu').numpy()
            y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='lrelu', wino=uf)
            assert rel_err(to_nchw(y), ref) <= RTOL
            assert torch.equal(y, ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='lrelu', wino=uf))
        if ops.wino_eligible(N, H, W, C, K, ksz, s, True):
            ub = ops.wino_transform(dev(w), True)
            dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
            assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
            if N % 3 == 0:
                dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev),
                                       dact_batch=2 * B, wino=ub)
                assert rel_err(to_nchw(dx2), 0.5 * gx.numpy() * mask) <= RTOL
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    # caller-side transform (no workspace): mmdgan_wino_transform + MMDGAN_ACT_FLAG_W_WINOGRAD
    assert ops.wino_eligible(N, H, W, C, K, ksz, s, False)                  # every case here is, in the forward direction
    uf, ub = ops.wino_transform(dev(w), False), ops.wino_transform(dev(w), True)
    assert uf.shape == (16, C, K) and ub.shape == (16, K, C)
    y = ops.conv2d_fwd(nhwc(x), dev(w), s, wino=uf)
    assert rel_err(to_nchw(y), yt.detach().numpy()) <= RTOL
    if ops.wino_eligible(N, H, W, C, K, ksz, s, True):                      # needs C % 64 == 0
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
        assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL
    else:
        with pytest.raises(ValueError, match='WINOGRAD'):
            ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
    assert not ops.wino_eligible(N, H, W, C, K, 5, 1, False)
    with pytest.raises(ValueError, match='WINOGRAD'):
        ops.conv2d_fwd(nhwc(x), dev(np.zeros((5, 5, C, K), np.float32)), 1, wino=uf)


W43_CASES = [(16, 16, 16, 128, 128, 3, 1), (32, 8, 8, 64, 128, 3, 1), (48, 4, 4, 256, 128, 3, 1), (9, 12, 12, 32, 64, 3, 1),
             (6, 8, 12, 48, 96, 3, 1), (130, 4, 4, 64, 64, 3, 1), (6, 16, 16, 64, 32, 3, 1), (3, 24, 24, 128, 128, 3, 1)]


@pytest.mark.parametrize('case', W43_CASES, ids=[str(c) for c in W43_CASES])
def test_conv2d_winograd_f43_path(ops, case):
    """3x3 / stride-1 layers whose H and W are multiples of 4 through F(4x4,3x3) (csrc/conv_wino43.hip; layer_func.py:912-916):
    forward and input-gradient with weights transformed by the caller (MMDGAN_ACT_FLAG_W_WINOGRAD43) - full, ragged and
    single tile blocks, H != W, channel counts that are not 32-multiples on the reduction side, bias / scale / activation,
    the 3B-row dact wrap, an addend - then the reduction split over workspace slabs and the library's own transform.
    Same fp64 oracle and norm-wise bar (1e-4) as the F(2x2,3x3) cases.  Element-wise (helpers.elementwise_err): every entry above
    5 % of the tensor's scale within 1e-4 of ITSELF, the entries below within 5e-6 of the scale - F(4x4,3x3)'s transforms carry
    constants up to 8 and its rounding is 3-5e-6 of the output scale whatever the entry's size (tools/wino43_gate.py,
    profiles/r05_wino43_gate.txt; F(2x2,3x3): 5e-7, held to a 1e-6 floor)."""
    N, H, W, C, K, ksz, s = case
    x, w, b = conv_data(case, 4)
    rs = np.random.RandomState(11)
    dy = rs.randn(N, K, H, W).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64)
    yt = R.conv2d_same(xt, wt, s)
    gx, = torch.autograd.grad((yt * torch.tensor(dy, dtype=torch.float64)).sum(), [xt])
    yt = yt.detach()
    assert ops.wino_algo(N, H, W, C, K, ksz, s, False) == ops.WINO_F43       # (tests/conftest.py: MMDGAN_WINO43=2)
    uf = ops.wino_transform(dev(w), False, algo=ops.WINO_F43)
    assert uf.shape == (36, C, K) and ops.wino_kind(uf) == ops.WINO_F43
    sc = np.float32(0.37)
    bt = torch.tensor(b, dtype=torch.float64).reshape(1, -1, 1, 1)

    def check_fwd(tag):
        for act in ('linear', 'lrelu', 'tanh'):
            ref = R._act(yt * float(sc) + bt, act).numpy()
            y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act=act, wino=uf)
            assert rel_err(to_nchw(y), ref) <= RTOL, (tag, act, rel_err(to_nchw(y), ref))
            assert elementwise_err(to_nchw(y), ref, floor_frac=5e-2) <= RTOL, (tag, act, elementwise_err(to_nchw(y), ref, 5e-2))
        y2 = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='tanh', wino=uf)
        assert torch.equal(y, y2), tag                                       # no atomics: the same bits every run
        y = ops.conv2d_fwd(nhwc(x), dev(w), s, wino=uf)                      # no bias, no scale
        assert rel_err(to_nchw(y), yt.numpy()) <= RTOL, tag
        print('F(4x4,3x3) %s %s: forward error %.2e of the output scale (entry-wise max)' % (case, tag, rel_err(to_nchw(y), yt.numpy())))
        addend = rs.randn(N, K, H, W).astype(np.float32)
        y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), act='lrelu', wino=uf, addend=nhwc(addend))
        assert rel_err(to_nchw(y), R._act(yt + bt, 'lrelu').numpy() + addend) <= RTOL, tag
    check_fwd('caller-transformed')
    dgrad_ok = ops.wino_algo(N, H, W, C, K, ksz, s, True) == ops.WINO_F43     # needs K % 8 == 0, K >= 32, C % 32 == 0
    assert dgrad_ok == (C % 32 == 0 and K % 16 == 0)
    ub = ops.wino_transform(dev(w), True, algo=ops.WINO_F43) if dgrad_ok else torch.zeros(36, K, C, device='cuda')

    def check_dgrad(tag):
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
        assert rel_err(to_nchw(dx), gx.numpy()) <= RTOL, tag
        assert elementwise_err(to_nchw(dx), gx.numpy(), floor_frac=5e-2) <= RTOL, (tag, elementwise_err(to_nchw(dx), gx.numpy(), 5e-2))
        if N % 3 == 0:                                   # [2B ; B] rows against 2B activations
            B = N // 3
            yprev = np.random.RandomState(5).randn(2 * B, C, H, W).astype(np.float32)
            mask = np.where(np.concatenate([yprev, yprev[B:]], 0) > 0, 1.0, 0.1)
            dx2 = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, scale=dev([0.5]), act='lrelu', dact_of=nhwc(yprev), dact_batch=2 * B,
                                   wino=ub)
            assert rel_err(to_nchw(dx2), 0.5 * gx.numpy() * mask) <= RTOL, tag
    if dgrad_ok:
        check_dgrad('caller-transformed')
    else:
        with pytest.raises(ValueError, match='WINOGRAD43'):
            ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s, wino=ub)
    # with a workspace: launches below one workgroup per CU (every case here) cut their channel reduction into slabs where
    # the parts keep >= 64 channels; and the library's own route (no tensor handed in) transforms into the workspace
    ops.set_workspace()
    try:
        check_fwd('workspace')
        if dgrad_ok:
            check_dgrad('workspace')
        y = ops.conv2d_fwd(nhwc(x), dev(w), s, bias=dev(b), scale=dev([sc]), act='lrelu')
        assert rel_err(to_nchw(y), R._act(yt * float(sc) + bt, 'lrelu').numpy()) <= RTOL
        if dgrad_ok:
            assert rel_err(to_nchw(ops.conv2d_dgrad(nhwc(dy), dev(w), (H, W), s)), gx.numpy()) <= RTOL
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    # a geometry F(4x4,3x3) does not take: H not a multiple of 4, or another kernel size
    assert ops.wino_algo(N, 6, W, C, K, ksz, s, False) != ops.WINO_F43
    with pytest.raises(ValueError, match='WINOGRAD43'):
        ops.conv2d_fwd(nhwc(x[:, :, :2]), dev(w), s, wino=uf)


W43W_CASES = [(16, 16, 16, 128, 128, 3, 1), (32, 8, 8, 64, 128, 3, 1), (130, 4, 4, 64, 64, 3, 1), (6, 20, 24, 32, 96, 3, 1),
              (33, 8, 8, 64, 32, 3, 1), (3, 24, 24, 128, 128, 3, 1), (65, 4, 8, 256, 64, 3, 1), (4, 32, 32, 64, 64, 3, 1)]


@pytest.mark.parametrize('case', W43W_CASES, ids=[str(c) for c in W43W_CASES])
def test_conv2d_winograd_f43_weight_gradient(ops, case):
    """the weight gradient of the same layers in the F(4x4,3x3) domain (csrc/conv_wino43w.hip; layer_func.py:912-916's kernels
    under tf.gradients): dW = G^T [sum over 4x4 tiles (B^T d B) (.) (A dY A^T)] G with per-split slabs in the workspace and the
    slab reduction - full and ragged windows of 8 tiles, tile ranges split over workgroups and not, H != W, one to eight
    channel blocks a side; the bias gradient and the spectral-norm scalar <dW, W> riding along; bit-reproducible; the same bits
    whether the reduction runs as its own pass or as the prologue of the next weight-gradient launch.  fp64 oracle, the
    norm-wise 1e-4 bar of every conv test, and an entry-wise floor of 5e-6 of the tensor's scale (tools/wino43_gate.py measured
    1.5-4e-6 for this transform pair against fp64; F(2x2,3x3): 5-9e-7)."""
    N, H, W, C, K, ksz, s = case
    x, w, _ = conv_data(case, 6)
    rs = np.random.RandomState(13)
    dy = rs.randn(N, K, H, W).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    gw, = torch.autograd.grad((R.conv2d_same(xt, wt, s) * torch.tensor(dy, dtype=torch.float64)).sum(), [wt])
    gw, gb = gw.numpy(), dy.astype(np.float64).sum((0, 2, 3))
    tun = ops.tuning()
    assert tun['wino43_wgrad'][0] >= 1 and N * (H // 4) * (W // 4) >= tun['wino43_wgrad_min_tiles'][0], tun   # (tests/conftest.py)
    ops.set_workspace()
    try:
        dw, db = torch.full((3, 3, C, K), float('nan'), device='cuda'), torch.full((K,), float('nan'), device='cuda')
        ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, out=dw, dbias=db)
        err = rel_err(dw.cpu().numpy(), gw)
        print('F(4x4,3x3) weight gradient %s: error %.2e of the tensor scale (entry-wise max)' % (case, err))
        assert err <= RTOL
        assert elementwise_err(dw.cpu().numpy(), gw, floor_frac=5e-2) <= RTOL, elementwise_err(dw.cpu().numpy(), gw, 5e-2)
        assert rel_err(db.cpu().numpy(), gb) <= RTOL
        assert torch.equal(dw, ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s))           # deterministic, with or without dbias
        # <dW, W> on the way (mmdgan_conv2d_wgrad_sn)
        dot = torch.full((1,), float('nan'), device='cuda')
        dw2 = ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, w=dev(w), dot=dot)
        assert torch.equal(dw2, dw)
        want = float((gw * w.astype(np.float64)).sum())
        assert abs(float(dot.item()) - want) <= 1e-4 * max(abs(want), float(np.abs(gw).max()) * float(np.abs(w).max())), (dot.item(), want)
        # the reduction as the prologue of the NEXT weight-gradient launch: a chain of three, the same bits
        ops.wgrad_defer(True)
        try:
            outs = [torch.full((3, 3, C, K), float('nan'), device='cuda') for _ in range(3)]
            dbs = [torch.full((K,), float('nan'), device='cuda') for _ in range(3)]
            for o, b_ in zip(outs, dbs):
                ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s, out=o, dbias=b_)
            ops.wgrad_flush()
        finally:
            ops.wgrad_defer(False)
        for o, b_ in zip(outs, dbs):
            assert torch.equal(o, dw) and torch.equal(b_, db)
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    # no workspace: no slabs - the call still answers (another kernel), same oracle
    assert rel_err(ops.conv2d_wgrad(nhwc(x), nhwc(dy), ksz, s).cpu().numpy(), gw) <= RTOL


def test_conv2d_winograd_f43_weight_gradient_under_load(ops):
    """the same kernel with the chip BUSY: 200 back-to-back launches per geometry, every other one with a forward convolution
    running on a second stream, the first one right after a synchronise (cold caches), with and without the bias gradient -
    every run bit-identical to the first and within the fp64 bar.  Round 6 shipped (for an hour) a version whose requests
    were inline assembly with hand-placed waits: hipcc copied registers whose request was still in flight, so a cold or
    contended run multiplied stale values - one run in ten wrong by 30-60 % on whole 32 x 32 blocks, never in a quiet
    parity test.  This is the test that would have caught it (tools/scratch history: profiles/r06_wino43w_ablation.txt)."""
    side = torch.cuda.Stream()
    big, wb = torch.randn(64, 64, 64, 64, device='cuda'), torch.randn(3, 3, 64, 64, device='cuda')
    ops.set_workspace(128 << 20)
    try:
        for (N, H, W, C, K) in [(4, 32, 32, 64, 64), (16, 16, 16, 128, 128), (48, 64, 64, 256, 32), (130, 4, 4, 64, 64), (33, 8, 8, 64, 32)]:
            assert ops.wgrad_algo(N, H, W, C, K, 3, 1) == ops.WINO_F43
            g = torch.Generator(device='cuda').manual_seed(N + C)
            x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
            dy = torch.randn(N, H, W, K, device='cuda', generator=g)
            wz = torch.zeros(K, C, 3, 3, device='cuda', dtype=torch.float64, requires_grad=True)
            gref, = torch.autograd.grad((torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wz, padding=1) *
                                         dy.permute(0, 3, 1, 2).double()).sum(), [wz])
            want = gref.permute(2, 3, 1, 0).contiguous()
            scale = float(want.abs().max())
            db = torch.empty(K, device='cuda')
            torch.cuda.synchronize()
            first = ops.conv2d_wgrad(x, dy, 3, 1).clone()
            assert float((first.double() - want).abs().max()) <= RTOL * scale, 'the first (cold) run'
            for rep in range(200):
                if rep % 2:
                    with torch.cuda.stream(side):
                        ops.conv2d_fwd(big, wb, 1)
                dw = ops.conv2d_wgrad(x, dy, 3, 1, dbias=db if rep % 3 == 0 else None)
                assert torch.equal(first, dw), ((N, H, W, C, K), rep, float((dw.double() - want).abs().max()) / scale)
            torch.cuda.synchronize()
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)


def test_winograd_weight_transforms_of_many_kernels_in_one_launch(ops):
    """mmdgan_wino_transform_multi: 3x3 and 4x4 kernels, both forms, ragged 3x3 channel blocks, more jobs than one table
    holds (24) - every transformed tensor bit-equal to its one-kernel launch"""
    rs = np.random.RandomState(21)
    shapes = [(3, 128, 128), (4, 64, 128), (3, 40, 192), (4, 32, 64), (3, 64, 64), (4, 128, 256), (3, 8, 32)]
    jobs, want = [], []
    for rep in range(2):
        for R, C, K in shapes:
            w = dev(rs.randn(R, R, C, K).astype(np.float32))
            for dgrad in (False, True):
                if R == 3 and (K if dgrad else C) % 8:
                    continue
                want.append(ops.wino_transform(w, dgrad))
                jobs.append((w, torch.full_like(want[-1], float('nan')), dgrad))
                if R == 3 and (C if dgrad else K) % 32 == 0:        # ... and in the F(4x4,3x3) layout
                    want.append(ops.wino_transform(w, dgrad, algo=ops.WINO_F43))
                    jobs.append((w, torch.full_like(want[-1], float('nan')), dgrad))
    assert len(jobs) > 24
    ops.WinoTransforms(jobs).run()
    torch.cuda.synchronize()
    for (w, u, dgrad), ref in zip(jobs, want):
        assert torch.equal(u, ref), (tuple(w.shape), dgrad)
    ops.WinoTransforms([]).run()                                   # nothing to do: no launch, no error
    bad = ops.WinoTransforms([(dev(rs.randn(4, 4, 32, 64).astype(np.float32)), torch.empty(4, 9, 32, 64, device='cuda'), False)])
    bad.table[0].C = 24                                            # 4x4 kernels: multiples of 32 only
    with pytest.raises(ValueError, match='multiples of 32'):
        bad.run()


@pytest.mark.parametrize('case', [(4, 4, 4, 512, 256, 4, 2), (3, 8, 8, 256, 128, 4, 2), (2, 16, 16, 128, 64, 4, 2),
                                  (3, 4, 4, 32, 16, 4, 2), (2, 3, 3, 16, 8, 4, 2)],
                         ids=lambda c: str(c))
def test_conv2d_transpose_forward_form(ops, case):
    """G 'tc' layers (layer_func.py:917-928): tf.nn.conv2d_transpose == dgrad of the conv whose
    kernel is [R,R,Cout_tc,Cin_tc]."""
    N, h, w_, cin, cout, R_, s = case
    rs = np.random.RandomState(11)
    v = rs.randn(N, cin, h, w_).astype(np.float32)
    k = (rs.randn(R_, R_, cout, cin) / np.sqrt(R_ * R_ * cin / 4)).astype(np.float32)
    ref = torch.relu(R.conv2d_transpose_same(torch.tensor(v, dtype=torch.float64), torch.tensor(k, dtype=torch.float64),
                                             (h * s, w_ * s), s)).numpy()
    y = ops.conv2d_dgrad(nhwc(v), dev(k), (h * s, w_ * s), s, act='relu')
    assert rel_err(to_nchw(y), ref) <= RTOL


def test_dact_batch_wrap(ops):
    """3B-row backward: the last B output images take act' from the LAST B images of a 2B-image
    activation tensor (conv dgrad on the MFMA path, on the direct path, and gemm)."""
    rs = np.random.RandomState(4)
    for (B, H, C, K, ksz, st) in ((4, 8, 64, 128, 3, 1), (2, 6, 5, 7, 3, 1), (4, 8, 64, 64, 4, 2),
                                  (4, 8, 3, 64, 3, 1), (4, 8, 64, 3, 3, 1)):       # + the thin MFMA kernels
        P = -(-H // st)
        dy = rs.randn(3 * B, K, P, P).astype(np.float32)
        w = (rs.randn(ksz, ksz, C, K) / np.sqrt(ksz * ksz * C)).astype(np.float32)
        yprev = rs.randn(2 * B, C, H, H).astype(np.float32)
        xt = torch.zeros(3 * B, C, H, H, dtype=torch.float64, requires_grad=True)
        (R.conv2d_same(xt, torch.tensor(w, dtype=torch.float64), st) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
        mask = np.where(np.concatenate([yprev, yprev[B:]], 0) > 0, 1.0, 0.1)
        ref = xt.grad.numpy() * mask
        dx = ops.conv2d_dgrad(nhwc(dy), dev(w), (H, H), st, act='lrelu', dact_of=nhwc(yprev), dact_batch=2 * B)
        assert rel_err(to_nchw(dx), ref) <= RTOL, (B, H, C, K)
    a, bm = rs.randn(12, 16).astype(np.float32), rs.randn(40, 16).astype(np.float32)
    y2 = rs.randn(8, 40).astype(np.float32)
    ref = (a.astype(np.float64) @ bm.T.astype(np.float64)) * np.where(np.concatenate([y2, y2[4:]], 0) > 0, 1.0, 0.0)
    c = ops.gemm(dev(a), dev(bm), trans_b=True, act='relu', dact_of=dev(y2), dact_rows=8)
    assert rel_err(c.cpu().numpy(), ref) <= RTOL
    with pytest.raises(ValueError, match='dact_rows'):
        ops.gemm(dev(a), dev(bm), trans_b=True, act='relu', dact_of=dev(y2), dact_rows=3)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,ta,tb', [(64, 8192, 128, False, False), (128, 16, 8192, False, False),
                                         (128, 8192, 16, False, True), (8192, 16, 128, True, False),
                                         (128, 8192, 64, True, False), (1, 8192, 16, False, True),
                                         (1, 16, 8192, False, False), (37, 53, 29, False, False),
                                         (37, 53, 29, True, True),
                                         # the skinny-N MFMA kernel of D's head (N = 16, M % 16 == 0, K % 256 == 0) and its neighbours
                                         (256, 16, 18432, False, False), (16, 16, 1024, False, False), (48, 16, 4608, False, False),
                                         (128, 16, 8192 + 128, False, False), (120, 16, 8192, False, False),
                                         (128, 8192, 128, False, False), (32, 1024, 64, False, False), (96, 2048, 256, False, False),
                                         (64, 4160, 100, False, False),
                                         # short reductions (K <= 128), every operand form, ragged edges
                                         (70, 1000, 100, True, True), (64, 8192, 128, False, True), (130, 1090, 128, True, False),
                                         (64, 2049, 7, False, False)])
def test_gemm(ops, M, N, K, ta, tb):
    rs = np.random.RandomState(M + N + K)
    a = rs.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    b = rs.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    A = a.T if ta else a
    Bm = b.T if tb else b
    ref = (A.astype(np.float64) @ Bm.astype(np.float64)) * 0.7 + bias
    c = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), scale=dev([0.7]))
    assert rel_err(c.cpu().numpy(), ref) <= RTOL
    c2 = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), scale=dev([0.7]), act='relu')
    assert rel_err(c2.cpu().numpy(), np.maximum(ref, 0)) <= RTOL
    # backward form: (scale * A B + bias) * lrelu'(y), y holding fewer rows than the output (the 3B-row wrap)
    if M % 3 == 0 or M >= 64:
        rows = (2 * M) // 3 if M % 3 == 0 else M
        y = rs.randn(rows, N).astype(np.float32)
        full = np.concatenate([y, y[rows - (M - rows):]], 0) if rows < M else y
        c3 = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), scale=dev([0.7]), act='lrelu', dact_of=dev(y), dact_rows=rows if rows < M else 0)
        assert rel_err(c3.cpu().numpy(), ref * np.where(full > 0, 1.0, 0.1)) <= RTOL


@pytest.mark.parametrize('M,N,K,ta', [(64, 8192, 128, False), (128, 8192, 64, True), (64, 2048, 128, False), (192, 4096, 84, True),
                                       (64, 12288, 100, False), (128, 2048, 64, False), (256, 16384, 8, True), (32, 16384, 128, False), (16, 2048, 64, True),
                                       (80, 4096, 32, False)])
def test_gemm_short_reduction_panels(ops, M, N, K, ta):
    """the whole-K panel kernels (csrc/gemm.hip: gemm_npanel_kernel - G's first layer and its weight gradient;
    gemm_mpanel16_kernel - the weight gradient of D's head) against fp64, with and without bias / scale / activation"""
    rs = np.random.RandomState(M + N + K)
    a = rs.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    b = rs.randn(K, N).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ b.astype(np.float64)
    assert rel_err(ops.gemm(dev(a), dev(b), ta, False).cpu().numpy(), ref) <= RTOL
    assert rel_err(ops.gemm(dev(a), dev(b), ta, False, bias=dev(bias), scale=dev([0.7]), act='lrelu').cpu().numpy(),
                   np.where(ref * 0.7 + bias > 0, 1.0, 0.1) * (ref * 0.7 + bias)) <= RTOL
    # the transposed problem: long M, sixteen columns (A^T B with A [K, M])
    b16 = rs.randn(K, 16).astype(np.float32)
    a_t = rs.randn(K, N).astype(np.float32)
    got = ops.gemm(dev(a_t), dev(b16), True, False)
    assert rel_err(got.cpu().numpy(), a_t.T.astype(np.float64) @ b16.astype(np.float64)) <= RTOL


def test_colsum_dot_layout(ops):
    rs = np.random.RandomState(0)
    x = rs.randn(5000, 70).astype(np.float32)
    assert rel_err(ops.colsum(dev(x)).cpu().numpy(), x.astype(np.float64).sum(0)) <= 1e-5
    a, b = rs.randn(100003).astype(np.float32), rs.randn(100003).astype(np.float32)
    assert abs(float(ops.dot(dev(a), dev(b)).item()) - float(a.astype(np.float64) @ b.astype(np.float64))) <= 1e-3
    t = rs.randn(3, 5, 7, 9).astype(np.float32)
    y = ops.nchw_to_nhwc(dev(t))
    assert np.array_equal(y.cpu().numpy(), np.transpose(t, (0, 2, 3, 1)))
    assert np.array_equal(ops.nhwc_to_nchw(y).cpu().numpy(), t)


def _check_bn_bwd_without_y(ops, x, dy, gamma, beta, smean, sinv, act, dx, dgamma, dbeta):
    """mmdgan_bn_bwd with y = NULL (what the engines call): the sign of relu / lrelu recomputed from x must be the forward
    entry's decision at EVERY element - one differing decision moves that element of dx by its whole |dy| - so the result
    equals the one computed from y up to the order of the fp64 atomics; tanh needs y itself and is refused"""
    if act == 'tanh':
        with pytest.raises(ValueError):
            ops.bn_bwd(x, None, dy, gamma, smean, sinv, act=act, beta=beta)
        return
    dx2, dgamma2, dbeta2 = ops.bn_bwd(x, None, dy, gamma, smean, sinv, act=act, beta=beta)
    assert rel_err(dx2.cpu().numpy(), dx.cpu().numpy()) <= 1e-6
    assert rel_err(dgamma2.cpu().numpy(), dgamma.cpu().numpy()) <= 1e-6
    assert rel_err(dbeta2.cpu().numpy(), dbeta.cpu().numpy()) <= 1e-6


def test_batch_norm_backward_without_y_at_exact_zeros(ops):
    """forward values that are EXACTLY zero (x at the batch mean, beta = 0) and denormal-small ones: relu / lrelu output 0 or
    -0 there, the derivative taken from y says "not positive", and so must the recomputed sign"""
    for act in ('relu', 'lrelu'):
        for C in (4, 7):                                 # the float4 and the scalar kernels
            x = np.zeros((3, C), np.float32)
            x[:, :] = np.array([1.5, 2.0, 1.0], np.float32)[:, None]       # mean 1.5 exactly: row 0 sits on it
            x[:, 1] *= 1e-20                                               # ... and a column whose values vanish against eps
            gamma, beta = np.ones(C, np.float32), np.zeros(C, np.float32)
            beta[2] = 1e-42                                                # a denormal forward value
            dy = np.ones((3, C), np.float32)
            z = np.zeros(C, np.float32)
            y, smean, sinv, _, _ = ops.bn_fwd_train(dev(x), dev(gamma), dev(beta), dev(z), dev(z + 1), act=act)
            assert (y[0].cpu().numpy()[[0, 3]] == 0).all()
            ref = ops.bn_bwd(dev(x), y, dev(dy), dev(gamma), smean, sinv, act=act)
            got = ops.bn_bwd(dev(x), None, dev(dy), dev(gamma), smean, sinv, act=act, beta=dev(beta))
            for r, g in zip(ref, got):
                assert np.array_equal(r.cpu().numpy(), g.cpu().numpy()), (act, C)


@pytest.mark.parametrize('rows,C,act,four_d', [(64 * 64, 256, 'relu', True), (64 * 1024, 64, 'relu', True),
                                               (64, 2304, 'relu', False), (100, 7, 'linear', True)])
def test_batch_norm(ops, rows, C, act, four_d):
    rs = np.random.RandomState(rows + C)
    x = (rs.randn(rows, C) * 1.7 + 0.4).astype(np.float32)
    gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.randn(C).astype(np.float32) * 0.2
    mm, mv = rs.randn(C).astype(np.float32) * 0.1, rs.uniform(0.5, 2, C).astype(np.float32)
    dy = rs.randn(rows, C).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mean = xt.mean(0)
    var = ((xt - mean) ** 2).mean(0)
    yt = R._act((xt - mean) / torch.sqrt(var + R.BN_EPS) * gt + bt, act)
    gx, gg, gb = torch.autograd.grad((yt * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, gt, bt])
    y, smean, sinv, nmm, nmv = ops.bn_fwd_train(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), act=act, unbiased=four_d)
    assert rel_err(y.cpu().numpy(), yt.detach().numpy()) <= RTOL
    var_u = var.detach().numpy() * (rows / (rows - 1.0)) if four_d else var.detach().numpy()
    assert rel_err(nmm.cpu().numpy(), mm * 0.99 + mean.detach().numpy() * 0.01) <= 1e-5
    assert rel_err(nmv.cpu().numpy(), mv * 0.99 + var_u * 0.01) <= 1e-5
    dx, dgamma, dbeta = ops.bn_bwd(dev(x), y, dev(dy), dev(gamma), smean, sinv, act=act)
    assert rel_err(dx.cpu().numpy(), gx.numpy()) <= RTOL
    assert rel_err(dgamma.cpu().numpy(), gg.numpy()) <= RTOL
    assert rel_err(dbeta.cpu().numpy(), gb.numpy()) <= RTOL
    _check_bn_bwd_without_y(ops, dev(x), dev(dy), dev(gamma), dev(beta), smean, sinv, act, dx, dgamma, dbeta)
    yi = ops.bn_fwd_infer(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), act=act)
    ref = R._act((torch.tensor(x, dtype=torch.float64) - torch.tensor(mm, dtype=torch.float64))
                 / torch.sqrt(torch.tensor(mv, dtype=torch.float64) + R.BN_EPS) * gt.detach() + bt.detach(), act)
    assert rel_err(yi.cpu().numpy(), ref.numpy()) <= RTOL


def test_sn_helpers_and_adam(ops):
    rs = np.random.RandomState(9)
    v = rs.randn(32768).astype(np.float32)
    norm, vn = ops.sn_norm(dev(v))
    n64 = np.sqrt((v.astype(np.float64) ** 2).sum())
    assert abs(norm.item() - n64) <= 1e-6 * n64
    assert rel_err(vn.cpu().numpy(), v / (n64 + 1e-10)) <= 1e-6
    sc = ops.sn_scale(norm, 1.6818)
    assert abs(sc.item() - 1.6818 / n64) <= 1e-6 * (1.6818 / n64)
    g, ds, wgt = (rs.randn(5000).astype(np.float32) for _ in range(3))
    dot = ops.dot(dev(g), dev(wgt))
    sigma = dev([2.5])
    scale = ops.sn_scale(sigma, 1.5)
    out = ops.sn_wgrad_fixup(dev(g), dev(ds), dot, sigma, scale)
    ref = 0.6 * g - (0.6 / 2.5) * float(g.astype(np.float64) @ wgt.astype(np.float64)) * ds
    assert rel_err(out.cpu().numpy(), ref) <= 1e-5
    # TF-Adam, 3 steps, two tensors, against the oracle's AdamTF
    ps = [rs.randn(1000).astype(np.float32), rs.randn(77).astype(np.float32)]
    params = {'a': torch.tensor(ps[0]), 'b': torch.tensor(ps[1])}
    opt = R.AdamTF(['a', 'b'], params, 5e-4)
    dp = [dev(p) for p in ps]
    dg = [torch.zeros_like(p) for p in dp]
    grp = ops.AdamGroup(dp, dg, [torch.zeros_like(p) for p in dp], [torch.zeros_like(p) for p in dp])
    for step in range(1, 4):
        gs = [rs.randn(1000).astype(np.float32) * 1e-3, rs.randn(77).astype(np.float32)]
        for t, gnp in zip(dg, gs):
            t.copy_(torch.tensor(gnp))
        grp.step(5e-4, step)
        opt.apply(params, {'a': torch.tensor(gs[0]), 'b': torch.tensor(gs[1])})
    assert rel_err(dp[0].cpu().numpy(), params['a'].numpy()) <= 1e-6
    assert rel_err(dp[1].cpu().numpy(), params['b'].numpy()) <= 1e-6


def test_batch_norm_at_random_shapes(ops):
    """batch norm forward (training and inference) and backward at 30 drawn (rows, features) pairs - feature counts that are no
    multiple of 4 (the scalar kernels), a handful of rows, tens of thousands - against float64 autograd"""
    rs = np.random.RandomState(5)
    for i in range(30):
        C = int(rs.choice([1, 3, 5, 7, 8, 12, 16, 31, 32, 48, 64, 100, 128, 200, 256, 512, 1000]))
        rows = int(rs.choice([2, 3, 9, 64, 100, 1000, 4096, 10000, 40000]))
        if rows * C > 4e6:
            continue
        act = ('relu', 'lrelu', 'linear', 'tanh')[i % 4]
        x = (rs.randn(rows, C) * 1.3 + 0.3).astype(np.float32)
        gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), (rs.randn(C) * 0.2).astype(np.float32)
        mm, mv = (rs.randn(C) * 0.1).astype(np.float32), rs.uniform(0.5, 2, C).astype(np.float32)
        dy = rs.randn(rows, C).astype(np.float32)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
        bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
        mean = xt.mean(0)
        var = ((xt - mean) ** 2).mean(0)
        yt = R._act((xt - mean) / torch.sqrt(var + R.BN_EPS) * gt + bt, act)
        gx, gg, gb = torch.autograd.grad((yt * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, gt, bt])
        y, smean, sinv, nmm, nmv = ops.bn_fwd_train(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), act=act, unbiased=False)
        tag = (rows, C, act)
        assert rel_err(y.cpu().numpy(), yt.detach().numpy()) <= RTOL, tag
        assert rel_err(nmv.cpu().numpy(), mv * 0.99 + var.detach().numpy() * 0.01) <= 1e-5, tag
        dx, dgamma, dbeta = ops.bn_bwd(dev(x), y, dev(dy), dev(gamma), smean, sinv, act=act)
        _check_bn_bwd_without_y(ops, dev(x), dev(dy), dev(gamma), dev(beta), smean, sinv, act, dx, dgamma, dbeta)
        gscale = float(np.abs(gx.numpy()).max())
        assert np.abs(dx.cpu().numpy() - gx.numpy()).max() <= RTOL * gscale + 1e-6 * gscale, tag
        assert rel_err(dgamma.cpu().numpy(), gg.numpy()) <= RTOL + 1e-6, tag
        assert rel_err(dbeta.cpu().numpy(), gb.numpy()) <= RTOL + 1e-6, tag


def test_power_iterations_of_many_kernels_in_six_launches(ops):
    """mmdgan_sn_power_iteration (every stage of all chains as one launch) against the same chains issued per kernel
    through the batch-1 conv / gemm entries: both conv forms (3x3, 4x4 stride 2, 3x3 stride 2 with its one-sided 'SAME'
    padding, 1x1, channel counts that are no tile multiple), both dense forms, more kernels than one group holds (8);
    with and without the update; in prezeroed mode and with the entry zeroing its own accumulators."""
    rs = np.random.RandomState(31)
    convs = [(0, 16, 16, 64, 128, 3, 1), (1, 16, 16, 64, 128, 4, 2), (0, 32, 32, 3, 64, 3, 1), (1, 8, 8, 128, 256, 3, 1),
             (0, 8, 8, 24, 40, 4, 2), (1, 10, 6, 16, 32, 3, 2), (0, 6, 10, 32, 16, 3, 2), (0, 8, 8, 64, 96, 1, 1),
             (1, 4, 4, 512, 512, 3, 1), (0, 4, 4, 256, 512, 4, 2)]
    dense = [(2, 8192, 16), (3, 16, 8192), (2, 128, 1024), (3, 100, 37)]
    layers, want = [], []
    z = lambda *sh: torch.zeros(*sh, device='cuda')
    for form, H, W, C, K, R, st in convs:
        P, Q = -(-H // st), -(-W // st)
        w = dev((rs.randn(R, R, C, K) * 0.1).astype(np.float32))
        xin, uout = ([1, H, W, C], [1, P, Q, K]) if form == 0 else ([1, P, Q, K], [1, H, W, C])
        x = dev(rs.randn(*xin).astype(np.float32))
        x /= x.norm()
        layers.append(dict(w=w, x=x.clone(), u=z(*uout), un=z(*uout), xb=z(*xin), dsigma=z(R, R, C, K), sigma=z(1), scale=z(1),
                           xb_norm=z(1), act_k=1.7, form=form, H=H, W=W, C=C, K=K, R=R, stride=st))
        if form == 0:
            u = ops.conv2d_fwd(x, w, st)
        else:
            u = ops.conv2d_dgrad(x, w, (H, W), st)
        sig = u.norm()
        un = u / (sig + 1e-10)
        if form == 0:
            ds, xb = ops.conv2d_wgrad(x, un, R, st), ops.conv2d_dgrad(un, w, (H, W), st)
        else:
            ds, xb = ops.conv2d_wgrad(un, x, R, st), ops.conv2d_fwd(un, w, st)
        want.append((u, sig, un, ds, xb, xb / (xb.norm() + 1e-10)))
    for form, C, K in dense:
        w = dev((rs.randn(C, K) * 0.1).astype(np.float32))
        x = dev(rs.randn(1, C if form == 2 else K).astype(np.float32))
        x /= x.norm()
        nu = K if form == 2 else C
        layers.append(dict(w=w, x=x.clone(), u=z(1, nu), un=z(1, nu), xb=z(*x.shape), dsigma=z(C, K), sigma=z(1), scale=z(1),
                           xb_norm=z(1), act_k=0.9, form=form, C=C, K=K))
        u = x @ w if form == 2 else x @ w.t()
        sig = u.norm()
        un = u / (sig + 1e-10)
        ds, xb = (x.t() @ un, un @ w.t()) if form == 2 else (un.t() @ x, un @ w)
        want.append((u, sig, un, ds, xb, xb / (xb.norm() + 1e-10)))
    assert len(layers) > 8
    x0 = [L['x'].clone() for L in layers]
    chains = ops.SnChains(layers, 'cuda')

    def check(update):
        for L, (u, sig, un, ds, xb, xn), x_old in zip(layers, want, x0):
            tag = (L['form'], tuple(L['w'].shape))
            assert rel_err(L['u'].cpu().numpy(), u.cpu().numpy()) <= 1e-5, tag
            assert abs(float(L['sigma']) - float(sig)) <= 1e-5 * float(sig), tag
            assert abs(float(L['scale']) - L['act_k'] / float(sig)) <= 1e-5 * L['act_k'] / float(sig), tag
            assert rel_err(L['un'].cpu().numpy(), un.cpu().numpy()) <= 1e-5, tag
            if update:
                assert rel_err(L['dsigma'].cpu().numpy(), ds.cpu().numpy()) <= 1e-5, tag
                assert rel_err(L['xb'].cpu().numpy(), xb.cpu().numpy()) <= 1e-5, tag
                assert rel_err(L['x'].cpu().numpy(), xn.cpu().numpy()) <= 1e-5, tag
                assert abs(float(L['xb_norm']) - float(xb.norm())) <= 1e-5 * float(xb.norm()), tag
            else:
                assert torch.equal(L['x'], x_old), tag
    for L in layers:                                     # garbage in the accumulators: the entry zeroes them itself
        for k in ('u', 'xb', 'dsigma'):
            L[k].fill_(float('nan'))
    chains.col_flat.fill_(float('nan'))                  # the entry zeroes what it accumulates into: whatever was there before
    chains.run(update=False)
    check(False)
    chains.run(update=True)
    check(True)
    lib = ops.require_device()
    for L, x_old in zip(layers, x0):                     # prezeroed mode: the caller zeroes, the entry accumulates
        L['x'].copy_(x_old)
        for k in ('u', 'xb', 'dsigma'):
            L[k].zero_()
    chains.col_flat.fill_(float('nan'))                  # (the patch matrices that hold split products too - and only those)
    for t in chains.zero_each_step:
        t.zero_()
    lib.mmdgan_set_outputs_prezeroed(1)
    try:
        chains.run(update=True)
    finally:
        lib.mmdgan_set_outputs_prezeroed(0)
    check(True)


def test_batch_norm_statistics_ride_on_the_convolution(ops):
    """mmdgan_conv2d_fwd_stats / _dgrad_stats + mmdgan_bn_fwd_apply: the convolution's output bit for bit what the plain entry
    writes, and the batch norm behind it (normalised output, saved mean / inverse deviation, moving statistics) what
    mmdgan_bn_fwd_train gives on that output - on geometries whose launch ends in the slab pass (the statistics are formed
    there: G's transposed layers at batch 64, a 3x3 layer with few tiles) and on ones that take the separate statistics pass"""
    rs = np.random.RandomState(9)
    ops.set_workspace(256 << 20)
    try:
        #        N   H   C    K   R  s  transposed (the layer is the input-gradient form: G's 'tc' layers)
        cases = [(64, 8, 256, 512, 4, 2, True), (64, 16, 128, 256, 4, 2, True), (64, 32, 64, 128, 4, 2, True),
                 (32, 8, 128, 128, 3, 1, False), (16, 16, 64, 128, 4, 2, False), (8, 8, 24, 40, 3, 1, False)]
        for (N, H, C, K, R_, st, tc) in cases:
            P = H // st
            w = dev((rs.randn(R_, R_, C, K) / np.sqrt(R_ * R_ * C)).astype(np.float32))
            if tc:
                x = dev(rs.randn(N, P, P, K).astype(np.float32))
                ch = C
                uw = ops.wino_transform(w, True) if ops.wino_eligible(N, H, H, C, K, R_, st, True) else None
                run = lambda **kw: ops.conv2d_dgrad(x, w, (H, H), st, wino=uw, **kw)
            else:
                x = dev(rs.randn(N, H, H, C).astype(np.float32))
                ch = K
                uw = ops.wino_transform(w, False) if ops.wino_eligible(N, H, H, C, K, R_, st, False) else None
                run = lambda **kw: ops.conv2d_fwd(x, w, st, wino=uw, **kw)
            plain = run()
            totals = torch.zeros(ops.require_device().mmdgan_bn_workspace_bytes(ch) // 8, dtype=torch.float64, device='cuda')
            got = run(bn_totals=totals)
            assert torch.equal(got, plain), (N, H, C, K, R_, st, tc)
            gamma, beta = dev(rs.uniform(0.5, 1.5, ch).astype(np.float32)), dev(rs.randn(ch).astype(np.float32))
            mm, mv = dev(rs.randn(ch).astype(np.float32)), dev(rs.uniform(0.5, 2, ch).astype(np.float32))
            ref = ops.bn_fwd_train(plain.view(-1, ch), gamma, beta, mm, mv, act='relu')
            out = ops.bn_fwd_train(got.view(-1, ch), gamma, beta, mm, mv, act='relu', workspace=totals, have_totals=True)
            for a, b_, what in zip(out, ref, ('y', 'mean', 'invstd', 'moving mean', 'moving variance')):
                assert rel_err(a.cpu().numpy(), b_.cpu().numpy()) <= 2e-6, (what, N, H, C, K, R_, st, tc)
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)


def test_deferred_slab_reduction_gives_the_same_bits(ops):
    """mmdgan_wgrad_defer: a chain of slab weight gradients on one stream, each summing its predecessor's slabs in its own
    prologue (quad form: >= 8 slabs; plain form: fewer), the last one flushed - dw, the bias gradient and <dw, w> are
    BIT-identical to the same calls with their own reduction launches (the dot to rounding: it is accumulated with one atomic
    per workgroup in both forms).  Also: another workspace user on the stream (a thin-layer weight gradient), a call on a
    second stream and mmdgan_wgrad_defer(0) each issue what is pending; a workspace too small for two sets of slabs falls
    back to a launch per call; the outputs of a deferred call are NOT complete before one of those."""
    rs = np.random.RandomState(5)
    # (N, H, C, K, R, stride): 56 / 14 / 3 slabs on the 4x4 stride-2 kernel, 28 / 7 / 2 on the 3x3 one, then a thin layer
    chain = [(32, 16, 128, 128, 3, 1), (32, 16, 64, 128, 4, 2), (16, 16, 128, 256, 4, 2), (32, 8, 256, 256, 3, 1),
             (32, 8, 256, 512, 4, 2), (64, 4, 512, 512, 3, 1), (8, 32, 64, 128, 4, 2)]
    data = []
    for (N, H, C, K, R_, s) in chain:
        P = H // s
        data.append((dev(rs.randn(N, H, H, C).astype(np.float32)), dev(rs.randn(N, P, P, K).astype(np.float32)),
                     dev(rs.randn(R_, R_, C, K).astype(np.float32)), R_, s, K))

    def run(defer, extra=None):
        outs = []
        ops.wgrad_defer(defer)
        try:
            for i, (x, dy, w, R_, s, K) in enumerate(data):
                dw = torch.full_like(w, float('nan'))
                db = torch.full((K,), float('nan'), device='cuda')
                dot = torch.zeros(1, device='cuda')
                if i % 2:
                    ops.conv2d_wgrad(x, dy, R_, s, out=dw, dbias=db, w=w, dot=dot)
                else:
                    ops.conv2d_wgrad(x, dy, R_, s, out=dw, dbias=db)
                outs.append((dw, db, dot))
                if extra is not None:
                    extra(i, outs)
            ops.wgrad_flush()
        finally:
            ops.wgrad_defer(False)
        torch.cuda.synchronize()
        return outs

    def same(a, b, what):
        for i, ((dw, db, dot), (dw2, db2, dot2)) in enumerate(zip(a, b)):
            assert torch.isfinite(dw).all() and torch.isfinite(db).all(), (what, i)
            assert torch.equal(dw, dw2) and torch.equal(db, db2), (what, i)
            assert abs(dot.item() - dot2.item()) <= 1e-5 * float((dw.double() * data[i][2].double()).abs().sum()), (what, i)

    ops.set_workspace(256 << 20)
    try:
        base = run(False)
        # (the base itself - a launch per reduction - is what test_conv2d_dgrad_and_wgrad and the Winograd tests hold to fp64)
        same(run(True), base, 'deferred chain')
        same(run(True), base, 'deferred chain, again')
        # a deferred call's outputs are incomplete until something is issued behind it
        ops.wgrad_defer(True)
        try:
            x, dy, w, R_, s, K = data[1]
            dw = torch.full_like(w, float('nan'))
            ops.conv2d_wgrad(x, dy, R_, s, out=dw)
            torch.cuda.synchronize()
            assert torch.isnan(dw).all()
            ops.wgrad_flush()
            torch.cuda.synchronize()
            assert torch.equal(dw, base[1][0])
        finally:
            ops.wgrad_defer(False)
        # another workspace user of the stream in the middle of the chain (D's first, thin layer): it issues what is pending
        xt, dyt = dev(rs.randn(32, 32, 32, 3).astype(np.float32)), dev(rs.randn(32, 32, 32, 64).astype(np.float32))
        thin = ops.conv2d_wgrad(xt, dyt, 3, 1)

        def thin_in_between(i, outs):
            if i == 2:
                got = ops.conv2d_wgrad(xt, dyt, 3, 1)
                torch.cuda.synchronize()
                assert torch.equal(outs[2][0], base[2][0]) and torch.equal(got, thin)
        same(run(True, thin_in_between), base, 'thin layer in between')
        # a second stream takes over in the middle: the first stream's pending reduction is issued on the first stream
        side = torch.cuda.Stream()

        def switch_stream(i, outs):
            if i == 3:
                x, dy, w, R_, s, K = data[0]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    d2 = ops.conv2d_wgrad(x, dy, R_, s)
                    ops.wgrad_flush()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                assert torch.equal(d2, base[0][0]) and torch.equal(outs[3][0], base[3][0])
        same(run(True, switch_stream), base, 'second stream')
        # a workspace with room for ONE set of slabs only: every call sums its own (no prologue form), same bits
        ops.set_workspace(80 << 20)
        same(run(True), base, 'small workspace')
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)


def test_segmented_adam_folds_the_spectral_norm_fixup(ops):
    """mmdgan_adam_segments + mmdgan_conv2d_wgrad_sn: the gradient arena keeps the RAW gradient of a spectrally normalised
    kernel plus the scalar <G, W>; the optimiser reads  scale * G - (scale / sigma) * <G, W> * dsigma/dW  (SURVEY A.2).
    (a) the weight-gradient entry returns the plain weight gradient and the dot product, on the slab path (Winograd
    domain, dot formed by the reduction pass) and on the direct path; (b) three TF-Adam steps over an arena of four
    segments - an SN kernel, two plain ones of ragged size, one that starts un-aligned - against the oracle's AdamTF fed
    with the fixed-up gradient, untouched gaps between segments included."""
    rs = np.random.RandomState(21)
    ops.set_workspace()
    try:
        # (the last one: D's first layer - the thin MFMA kernel, whose partial-sum reduction forms <dw, w> itself)
        for (N, H, C, K, R_, s) in ((16, 16, 64, 128, 3, 1), (16, 16, 64, 128, 4, 2), (4, 8, 24, 40, 3, 1), (16, 32, 3, 64, 3, 1)):
            P = H // s
            x = dev(rs.randn(N, H, H, C).astype(np.float32))
            dy = dev(rs.randn(N, P, P, K).astype(np.float32))
            w = dev(rs.randn(R_, R_, C, K).astype(np.float32))
            plain = ops.conv2d_wgrad(x, dy, R_, s)
            dot = torch.full((1,), float('nan'), device='cuda')
            db = torch.empty(K, device='cuda')
            got = ops.conv2d_wgrad(x, dy, R_, s, dbias=db, w=w, dot=dot)
            assert rel_err(got.cpu().numpy(), plain.cpu().numpy()) <= 1e-6
            want = float((plain.double() * w.double()).sum())
            assert abs(dot.item() - want) <= 1e-5 * float((plain.double() * w.double()).abs().sum()), (N, H, C, K, R_)
            assert rel_err(db.cpu().numpy(), dy.double().sum((0, 1, 2)).cpu().numpy()) <= 1e-5
    finally:
        ops.require_device().mmdgan_set_workspace(None, 0)
    sizes = [(0, 4096 + 36), (4132, 77), (4212, 3), (4217, 1030)]          # (offset, n): gaps at 4209..4211 and 4215..4216
    total = 4217 + 1030 + 5
    p0 = rs.randn(total).astype(np.float32)
    p, g, m, v = dev(p0), torch.zeros(total, device='cuda'), torch.zeros(total, device='cuda'), torch.zeros(total, device='cuda')
    ds = dev(rs.randn(sizes[0][1]).astype(np.float32))
    sigma, scale, dot = dev([2.5]), dev([0.6]), torch.zeros(1, device='cuda')
    segs = [(sizes[0][0], sizes[0][1], dict(dsigma=ds, dot=dot, sigma=sigma, scale=scale))] + [(o, n, None) for o, n in sizes[1:]]
    opt_dev = ops.AdamArena(p, g, m, v, segs)
    names = ['s%d' % i for i in range(4)]
    params = {nm: torch.tensor(p0[o:o + n]) for nm, (o, n) in zip(names, sizes)}
    opt = R.AdamTF(names, params, 5e-4)
    for step in range(3):
        gnp = (rs.randn(total) * 1e-2).astype(np.float32)
        g.copy_(torch.tensor(gnp))
        d = float(rs.randn())
        dot.fill_(d)
        opt_dev.step(5e-4, grad_scale=0.5)
        eff = {nm: 0.5 * torch.tensor(gnp[o:o + n]) for nm, (o, n) in zip(names, sizes)}
        eff['s0'] = 0.5 * (0.6 * torch.tensor(gnp[:sizes[0][1]]) - (0.6 / 2.5) * d * ds.cpu())
        opt.apply(params, eff)
    got = p.cpu().numpy()
    for nm, (o, n) in zip(names, sizes):
        assert rel_err(got[o:o + n], params[nm].numpy()) <= 2e-6, nm
    for lo, hi in ((4209, 4212), (4215, 4217), (total - 5, total)):
        assert np.array_equal(got[lo:hi], p0[lo:hi])
    assert int(opt_dev.step_counter.item()) == 3
    # the step counts / learning rates of two arenas prepared by ONE launch (mmdgan_adam_prepare_multi), ahead of the updates:
    # the same parameters as two arenas that prepare inside their own step()
    pa, pb = [[dev(p0) for _ in range(2)] for _ in range(2)]
    za = [[torch.zeros(total, device='cuda') for _ in range(4)] for _ in range(2)]
    early = [ops.AdamArena(pa[i], g, za[i][0], za[i][1], segs) for i in range(2)]
    late = [ops.AdamArena(pb[i], g, za[i][2], za[i][3], segs) for i in range(2)]
    for _ in range(2):
        ops.adam_prepare_multi([(early[0], 5e-4), (early[1], 2e-4)])
        assert early[0].prepared and early[1].prepared
        for i, lr in enumerate((5e-4, 2e-4)):
            early[i].step(lr)
            late[i].step(lr)
            assert not early[i].prepared
    for i in range(2):
        assert int(early[i].step_counter.item()) == 2
        assert torch.equal(pa[i], pb[i]) and torch.equal(early[i].lr_t, late[i].lr_t)
    # fold_fixup = False (data-parallel replicas fix up before their all-reduce): every segment is read plainly
    p2 = dev(p0)
    plain = ops.AdamArena(p2, g, torch.zeros_like(m), torch.zeros_like(v), segs)
    plain.fold_fixup = False
    plain.step(5e-4, grad_scale=1.0)
    ref = {nm: torch.tensor(p0[o:o + n]) for nm, (o, n) in zip(names, sizes)}
    R.AdamTF(names, ref, 5e-4).apply(ref, {nm: g.cpu()[o:o + n] for nm, (o, n) in zip(names, sizes)})
    assert rel_err(p2.cpu().numpy()[:sizes[0][1]], ref['s0'].numpy()) <= 2e-6


# ---------------------------------------------------------------------------------------------
# elementwise pieces of the residual blocks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,h,w,c,f', [(4, 8, 8, 16, 2), (3, 6, 10, 3, 2), (2, 9, 6, 5, 3), (128, 64, 64, 64, 2), (1, 2, 2, 1, 2),
                                       (5, 4, 4, 8, 4)])
def test_resample_matches_the_reference_ops_and_their_gradients(ops, n, h, w, c, f):
    """'avg' = tf.nn.avg_pool(window = stride = f) and 'unpool' = nearest repeat (layer_func.py:1155-1163), and each
    one's gradient, against torch autograd on NCHW tensors"""
    rs = np.random.RandomState(n + h + c)
    x = rs.randn(n, c, h, w).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    pooled = torch.nn.functional.avg_pool2d(xt, f)
    got = to_nchw(ops.resample_down(nhwc(x), f))
    assert rel_err(got, pooled.detach().numpy()) <= 2e-6
    dy = rs.randn(*pooled.shape).astype(np.float32)
    gx, = torch.autograd.grad(pooled, xt, torch.tensor(dy, dtype=torch.float64))
    assert rel_err(to_nchw(ops.resample_up(nhwc(dy), f, scale=1.0 / (f * f))), gx.numpy()) <= 2e-6      # d avg
    up = xt.repeat_interleave(f, dim=2).repeat_interleave(f, dim=3)
    got_up = to_nchw(ops.resample_up(nhwc(x), f))
    assert np.array_equal(got_up, up.detach().numpy().astype(np.float32))                               # pure copies
    du = rs.randn(*up.shape).astype(np.float32)
    gx, = torch.autograd.grad(up, xt, torch.tensor(du, dtype=torch.float64))
    assert rel_err(to_nchw(ops.resample_down(nhwc(du), f, scale=1.0)), gx.numpy()) <= 2e-6              # d unpool
    # accumulate: out += ...
    base = nhwc(rs.randn(*pooled.shape).astype(np.float32))
    want = base + ops.resample_down(nhwc(x), f)
    ops.resample_down(nhwc(x), f, out=base, accumulate=True)
    assert torch.equal(base, want)


@pytest.mark.parametrize('n,h,w,c,f', [(2, 4, 4, 8, 2), (3, 5, 3, 3, 2), (2, 4, 6, 2, 3), (64, 16, 16, 64, 2), (1, 1, 1, 1, 2)])
def test_periodic_shuffle_is_the_reference_permutation(ops, n, h, w, c, f):
    """'ps' = tf.depth_to_space / tf.space_to_depth on NCHW tensors (layer_func.py:197-244): block-major channels.
    Pure permutations: exact, inverse of each other, each the other's gradient."""
    rs = np.random.RandomState(n + c)
    small = rs.randn(n, f * f * c, h, w).astype(np.float32)                     # NCHW, channel = (i*f + j)*c + ch
    big_ref = small.reshape(n, f, f, c, h, w).transpose(0, 3, 4, 1, 5, 2).reshape(n, c, h * f, w * f)
    big = ops.periodic_shuffle(nhwc(small), f, True)
    assert np.array_equal(to_nchw(big), big_ref)
    back = ops.periodic_shuffle(big, f, False)
    assert np.array_equal(to_nchw(back), small)
    # <shuffle(x), y> == <x, unshuffle(y)>: the inverse permutation is the adjoint
    y = rs.randn(*big_ref.shape).astype(np.float32)
    lhs = float((big_ref.astype(np.float64) * y).sum())
    rhs = float((small.astype(np.float64) * to_nchw(ops.periodic_shuffle(nhwc(y), f, False))).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), 1.0)


@pytest.mark.parametrize('n,c,h,w,oh,ow', [(2, 3, 4, 4, 8, 8), (3, 5, 6, 6, 3, 3), (2, 4, 6, 9, 2, 3), (2, 8, 3, 3, 12, 12),
                                           (1, 1, 1, 1, 2, 2), (2, 2, 5, 7, 5, 7), (4, 16, 16, 16, 32, 32), (2, 3, 4, 4, 1, 1)])
def test_bilinear_resize_and_its_gradient(ops, n, c, h, w, oh, ow):
    """'bil' = tf.image.resize_bilinear(align_corners=True) (layer_func.py:1128-1137) against the oracle's written-out
    interpolation in fp64, and the adjoint against autograd"""
    rs = np.random.RandomState(n + c + oh)
    x = rs.randn(n, c, h, w).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = R.bilinear_resize(xt, (oh, ow))
    got = ops.bilinear_resize(nhwc(x), (oh, ow))
    assert rel_err(to_nchw(got), ref.detach().numpy()) <= 2e-6
    dy = rs.randn(n, c, oh, ow).astype(np.float32)
    gx, = torch.autograd.grad(ref, xt, torch.tensor(dy, dtype=torch.float64))
    dx = ops.bilinear_resize_grad(nhwc(dy), (h, w))
    assert rel_err(to_nchw(dx), gx.numpy()) <= 5e-6
    # align_corners: the corner pixels are copied
    assert np.array_equal(to_nchw(got)[:, :, 0, 0], x[:, :, 0, 0])
    if oh > 1 and ow > 1:
        assert np.allclose(to_nchw(got)[:, :, -1, -1], x[:, :, -1, -1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('n,c,h,w,oh,ow', [(2, 3, 3, 3, 6, 6), (3, 5, 6, 6, 3, 3), (2, 4, 6, 9, 2, 3), (2, 8, 3, 3, 12, 12),
                                           (1, 1, 1, 1, 2, 2), (2, 2, 5, 7, 5, 7), (4, 16, 16, 16, 32, 32), (2, 3, 4, 4, 1, 1),
                                           (2, 3, 12, 12, 4, 4)])
def test_bicubic_resize_and_its_gradient(ops, n, c, h, w, oh, ow):
    """'bic' = tf.image.resize_bicubic(align_corners=True) (layer_func.py:1138-1147, TF 1.x legacy kernel: Keys cubic A = -0.75
    on a 1/1024 grid, clamped taps) against the oracle's written-out interpolation in fp64, and the adjoint against autograd"""
    rs = np.random.RandomState(n + c + oh)
    x = rs.randn(n, c, h, w).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = R.bicubic_resize(xt, (oh, ow))
    got = ops.bicubic_resize(nhwc(x), (oh, ow))
    assert rel_err(to_nchw(got), ref.detach().numpy()) <= 2e-6
    dy = rs.randn(n, c, oh, ow).astype(np.float32)
    gx, = torch.autograd.grad(ref, xt, torch.tensor(dy, dtype=torch.float64))
    dx = ops.bicubic_resize_grad(nhwc(dy), (h, w))
    assert rel_err(to_nchw(dx), gx.numpy()) <= 5e-6
    if oh > 1 and ow > 1:                      # align_corners: the corner pixels are copied (weights 0, 1, 0, 0 exactly)
        assert np.array_equal(to_nchw(got)[:, :, 0, 0], x[:, :, 0, 0])


@pytest.mark.parametrize('n,c,h,w,f', [(2, 3, 4, 4, 2), (3, 5, 6, 9, 3), (8, 64, 16, 16, 2), (1, 1, 2, 2, 2)])
def test_max_pool_and_its_gradient(ops, n, c, h, w, f):
    """'max' = tf.nn.max_pool(window = stride = f) (layer_func.py:1149-1153); ties (plenty after a relu) send the
    gradient to the first maximum of the window"""
    rs = np.random.RandomState(n + c)
    x = np.maximum(rs.randn(n, c, h, w), 0.0).astype(np.float32)                # relu output: windows of zeros tie
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = torch.nn.functional.max_pool2d(xt, f)
    got = ops.max_pool(nhwc(x), f)
    assert np.array_equal(to_nchw(got), ref.detach().numpy().astype(np.float32))
    dy = rs.randn(*ref.shape).astype(np.float32)
    gx, = torch.autograd.grad(ref, xt, torch.tensor(dy, dtype=torch.float64))
    dx = ops.max_pool(nhwc(x), f, dy=nhwc(dy))
    assert np.array_equal(to_nchw(dx), gx.numpy().astype(np.float32))


@pytest.mark.parametrize('n,h,c,k', [(2, 8, 8, 64), (3, 6, 5, 3), (4, 16, 64, 128), (2, 4, 128, 64), (1, 2, 3, 5)])
def test_scaling_folded_into_the_conv(ops, n, h, c, k):
    """avgpool/2 o conv3x3 and conv3x3 o unpool x2 as single 4x4 stride-2 (transposed) convs with composed kernels:
    same results as the two-op form, and the kernel-gradient adjoint"""
    rs = np.random.RandomState(n + c + k)
    w = dev((rs.randn(3, 3, c, k) / np.sqrt(9 * c)).astype(np.float32))
    x = dev(rs.uniform(-1, 1, (n, h, h, c)).astype(np.float32))
    b = dev((rs.randn(k) * 0.1).astype(np.float32))
    two = ops.resample_down(ops.conv2d_fwd(x, w, 1, bias=b), 2)
    one = ops.conv2d_fwd(x, ops.compose_scaled_conv(w, 'avg'), 2, bias=b)
    assert rel_err(one.cpu().numpy(), two.cpu().numpy()) <= 5e-6
    two = ops.conv2d_fwd(ops.resample_up(x, 2), w, 1, bias=b)
    one = ops.conv2d_dgrad(x, ops.compose_scaled_conv(w, 'unpool'), (2 * h, 2 * h), 2, bias=b)
    assert rel_err(one.cpu().numpy(), two.cpu().numpy()) <= 5e-6
    for mode in ('avg', 'unpool'):
        w4 = ops.compose_scaled_conv(w, mode)
        d4 = dev(rs.randn(*w4.shape).astype(np.float32))
        lhs = float((w4.double() * d4.double()).sum())
        rhs = float((w.double() * ops.compose_scaled_conv_grad(d4, mode).double()).sum())
        assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0), mode


@pytest.mark.parametrize('act', ['linear', 'relu', 'lrelu', 'tanh'])
def test_act_and_axpby(ops, act):
    rs = np.random.RandomState(5)
    x = rs.randn(3, 7, 5, 11).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = R._act(xt, act)
    y = ops.act_fwd(dev(x), act)
    assert rel_err(y.cpu().numpy(), yt.detach().numpy()) <= 2e-6
    dy = rs.randn(*x.shape).astype(np.float32)
    gx, = torch.autograd.grad(yt, xt, torch.tensor(dy, dtype=torch.float64), allow_unused=True)
    got = ops.act_bwd(dev(dy), y, act)
    assert rel_err(got.cpu().numpy(), gx.numpy()) <= 2e-6
    acc = dev(dy).clone()
    ops.act_bwd(dev(dy), y, act, out=acc, accumulate=True)
    assert rel_err(acc.cpu().numpy(), gx.numpy() + dy) <= 2e-6
    a, b = dev(x), dev(dy)
    assert torch.equal(ops.axpby(a, b), a + b)
    out = ops.axpby(a, b, 0.5, -2.0)
    assert rel_err(out.cpu().numpy(), 0.5 * x - 2.0 * dy) <= 1e-6
    ops.axpby(a, b, out=a)                                                                              # in place
    assert rel_err(a.cpu().numpy(), x + dy) <= 1e-6


def test_slice_and_space_batch_compose_valid_and_dilated_convs(ops):
    """mmdgan_strided_slice / mmdgan_space_batch (the 'padding': 'VALID' and 'dilation' keys of a conv layer as compositions
    around the 'SAME' kernels): the ops and their adjoints against indexing, and the composed convs against the oracle's
    conv with padding / dilation (layer_func.py:912-916)"""
    rs = np.random.RandomState(5)
    x = dev(rs.randn(3, 11, 14, 5).astype(np.float32))
    for off, step in ((1, 1), (1, 2), (0, 3), (2, 2)):
        P, Q = -(-(11 - 2 * off) // step), -(-(14 - 2 * off) // step)
        want = x[:, off:off + (P - 1) * step + 1:step, off:off + (Q - 1) * step + 1:step]
        got = ops.strided_slice(x, off, step, (P, Q))
        assert torch.equal(got, want.contiguous())
        g = dev(rs.randn(3, P, Q, 5).astype(np.float32))
        back = ops.strided_slice(g, off, step, None, adjoint_hw=(11, 14), out=torch.full((3, 11, 14, 5), float('nan'), device='cuda'))
        ref = torch.zeros(3, 11, 14, 5, device='cuda')
        ref[:, off:off + (P - 1) * step + 1:step, off:off + (Q - 1) * step + 1:step] = g
        assert torch.equal(back, ref)
    for d in (2, 3):
        xb = ops.space_batch(x, d)
        Hd, Wd = -(-11 // d), -(-14 // d)
        assert xb.shape == (3 * d * d, Hd, Wd, 5)
        pad = torch.zeros(3, Hd * d, Wd * d, 5, device='cuda')
        pad[:, :11, :14] = x
        want = pad.view(3, Hd, d, Wd, d, 5).permute(0, 2, 4, 1, 3, 5).reshape(3 * d * d, Hd, Wd, 5)
        assert torch.equal(xb, want)
        assert torch.equal(ops.space_batch(xb, d, hw=(11, 14)), x)
    # composed convs: N, H, W, C, K, R, stride, dilation, padding
    for N, H, W, C, K, Rk, st, dl, pad in ((4, 12, 10, 16, 32, 3, 1, 1, 'VALID'), (4, 13, 12, 16, 32, 4, 2, 1, 'VALID'),
                                           (3, 12, 9, 16, 32, 3, 1, 2, 'SAME'), (3, 14, 13, 16, 32, 3, 1, 3, 'VALID'),
                                           (2, 9, 9, 8, 8, 3, 2, 1, 'VALID')):
        xx = rs.randn(N, C, H, W).astype(np.float32)
        w = (rs.randn(Rk, Rk, C, K) * 0.1).astype(np.float32)
        ref = R.conv2d_same(torch.tensor(xx, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), st, dl, pad).numpy()
        xin = nhwc(xx)
        if dl > 1:
            xin = ops.space_batch(xin, dl)
        full = ops.conv2d_fwd(xin, dev(w), 1 if pad == 'VALID' else st)
        if dl > 1:
            full = ops.space_batch(full, dl, hw=(H, W))
        if pad == 'VALID':
            full = ops.strided_slice(full, dl * ((Rk - 1) // 2), st, ref.shape[2:])
        assert rel_err(to_nchw(full), ref) <= RTOL, (N, H, W, Rk, st, dl, pad)


def test_memset_zero_multi(ops):
    """several scratch buffers zeroed by one launch; odd sizes / alignments take the plain-memset route"""
    ts = [torch.randn(n, device='cuda') for n in (4, 1024, 100000, 12, 7, 0, 524288)]
    odd = torch.randn(64, device='cuda')[1:33]                    # 4-byte aligned only
    d64 = torch.randn(300, device='cuda', dtype=torch.float64)
    guard = torch.ones(8, device='cuda')
    many = [torch.randn(16, device='cuda') for _ in range(40)]    # more than one table's worth
    ops.memset_zero_multi(ts + [odd, d64] + many)
    torch.cuda.synchronize()
    for t in ts + [odd, d64] + many:
        assert float(t.abs().sum()) == 0.0
    assert float(guard.sum()) == 8.0
