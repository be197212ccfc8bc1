"""The HIP kernels against the per-layer fixtures the reference's own Net / Routine produced (tests/golden/net_*.npz,
oracle/make_golden.py:make_layers) - DIRECTLY, without the oracle restatement in between: SN conv k3s1 / k4s2, the SN
dense head behind a C,H,W flatten, transposed conv + BN, dense + BN, channel counts that are no tile multiples.
Two consecutive evaluations each (spectral-norm vectors and BN moving statistics are updated in between):
  * through the API mirror's Routine (GeneralTools/layer_func.py): outputs, and the UPDATE_OPS state afterwards;
  * through the primitive-op engine's forward / backward passes (mmdgan_hip/tape.py) with the fixture's upstream
    gradient dy: outputs, sigma of every SN kernel, dx, every parameter gradient, the state afterwards.
Bars: 1e-4 of the tensor's scale (BASELINE.json north_star), against the fp64 evaluation of the reference code."""
import numpy as np
import pytest
import torch

from helpers import RTOL, designs_of, golden, load, rel_err

pytestmark = pytest.mark.gpu

CASES = ['dis_small', 'gen_small', 'gen_stl_head', 'dis_odd', 'dis_valid_dil']


def _to_dev(a):
    a = np.asarray(a, np.float32)
    if a.ndim == 4:
        a = np.ascontiguousarray(a.transpose(0, 2, 3, 1))
    return torch.as_tensor(a).cuda()


def _to_ref(t):
    a = t.detach().cpu().numpy()
    return a.transpose(0, 3, 1, 2) if a.ndim == 4 else a


@pytest.mark.parametrize('case', CASES)
def test_routine_reproduces_the_reference_net_fixtures(case):
    from GeneralTools.layer_func import Net, Routine
    from GeneralTools.misc_fun import FLAGS
    fx = load(golden('net_%s.npz' % case)[0])
    designs, in_shape = designs_of(fx), [int(v) for v in fx['input_shape']]
    name = 'dis' if case.startswith('dis') else 'gen'
    net = Net(designs, net_name=name, data_format=FLAGS.IMAGE_FORMAT, num_class=0)
    r = Routine(net)
    r.add_input_layers([64] + in_shape, [0])
    r.seq_links(list(range(net.num_layers)))
    r.add_output_layers([net.num_layers - 1])
    x = torch.as_tensor(fx['x']).cuda()
    r(x, is_training=False)                                       # creates the variables (and must not touch the state)
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    assert sorted(init) == sorted(net.network.variable_names())   # the reference's variable names, all of them
    for k, v in init.items():
        net.network.set_variable(k, v)
    y_inf = r(x, is_training=False)['x'].cpu().numpy()            # inference: sigma from the stored vector, no update
    for k, v in init.items():
        if k.endswith('in_rand') or '/moving_' in k:
            assert np.array_equal(net.network.get_variable(k), v), k
    if name == 'dis':                                             # no BN: inference == training output of step 0
        assert rel_err(y_inf, fx['step0/y_f64']) <= RTOL
    for step in range(2):
        y = r({'x': x}, is_training=True)['x'].cpu().numpy()
        pre = 'step%d/' % step
        assert y.shape == fx[pre + 'y_f64'].shape
        assert rel_err(y, fx[pre + 'y_f64']) <= RTOL, (case, step, rel_err(y, fx[pre + 'y_f64']))
        for k, v in fx.items():
            if k.startswith(pre + 'after/'):
                got = net.network.get_variable(k[len(pre + 'after/'):-len('_f64')])
                assert rel_err(got, v) <= RTOL, (case, step, k)


def _pair(designs, name, in_shape):
    """the net under test with a minimal partner, as the architecture dict the engines take"""
    designs = [dict(d) for d in designs]
    if name == 'dis':
        if 'out_reshape' not in designs[-1] and designs[-1].get('op', 'c') != 'd':
            # the engines want a score vector: flatten the last feature map (C,H,W order, a pure re-indexing)
            c, h, w = in_shape
            for d in designs:
                s = d.get('strides', 1)
                c, h, w = d['out'], -(-h // s), -(-w // s)
            designs[-1]['out_reshape'] = [c * h * w]
        c, h, w = in_shape
        gen = [{'name': 'g1', 'out': c * h * w, 'op': 'd', 'out_reshape': [c, h, w]}]
        return {'input': [tuple(in_shape)], 'code': [(4, 'linear')], 'generator': gen, 'discriminator': designs}
    from mmdgan_hip.engine import build_specs
    out = build_specs(designs, in_shape, 'gen')[-1].out_shape_ref
    dis = [{'name': 'd1', 'out': 4, 'op': 'd', 'act_k': 1.0, 'w_nm': 's'}] if len(out) == 1 else \
        [{'name': 'd1', 'out': 8, 'act': 'lrelu', 'act_k': 1.0, 'w_nm': 's', 'out_reshape': [8 * out[1] * out[2]]},
         {'name': 'd2', 'out': 4, 'op': 'd', 'act_k': 1.0, 'w_nm': 's'}]
    return {'input': [tuple(out)], 'code': [(in_shape[0], 'linear')], 'generator': designs, 'discriminator': dis}


@pytest.mark.parametrize('case', CASES)
def test_forward_backward_kernels_reproduce_the_reference_net_fixtures(case):
    from mmdgan_hip.tape import TapeEngine
    fx = load(golden('net_%s.npz' % case)[0])
    designs, in_shape = designs_of(fx), [int(v) for v in fx['input_shape']]
    name = 'dis' if case.startswith('dis') else 'gen'
    batch = fx['x'].shape[0]
    eng = TapeEngine(_pair(designs, name, in_shape), 'rep', (1e-4, 1e-4), batch_size=batch)
    net = eng.dis if name == 'dis' else eng.gen
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    assert sorted(init) == sorted(net.variable_names())
    for k, v in init.items():
        net.set_variable(k, v)
    x = _to_dev(fx['x'])
    dy_ref = fx['dy']
    for step in range(2):
        pre = 'step%d/' % step
        for k in net.kernels:                                     # the power iterations (UPDATE_OPS included)
            if k.sn:
                eng._sn_step(net, k)
        vals = eng._forward(net, x, True, 'fx')
        y = vals[net.out_val]
        y_ref = fx[pre + 'y_f64']
        if y.dim() == 2 and y_ref.ndim == 4:
            # the engine's flatten of a feature map is a view of its NHWC memory (the permutation to the reference's
            # C,H,W order lives in the rows of the dense layer that follows - here there is none): un-flatten as NHWC
            got_y = y.cpu().numpy().reshape(batch, y_ref.shape[2], y_ref.shape[3], y_ref.shape[1]).transpose(0, 3, 1, 2)
        else:
            got_y = _to_ref(y) if y.dim() == 4 else y.cpu().numpy()
        assert rel_err(got_y, y_ref) <= RTOL, (case, step, 'y', rel_err(got_y, y_ref))
        for k, v in fx.items():                                   # sigma of every SN kernel
            if k.startswith(pre + 'sigma/') and k.endswith('_f64'):
                scope = k[len(pre + 'sigma/'):-len('_f64')]
                got = float(net.sn[scope + '/kernel']['sigma'].item())
                assert abs(got - float(v)) <= RTOL * abs(float(v)), (case, step, scope, got, float(v))
        # upstream gradient in the engine's layout
        if y.dim() == 4:
            dy = _to_dev(dy_ref)
        elif dy_ref.ndim == 4:                                    # the same NHWC flatten as above
            dy = torch.as_tensor(np.ascontiguousarray(dy_ref.transpose(0, 2, 3, 1).reshape(batch, -1))).cuda()
        else:
            dy = torch.as_tensor(np.ascontiguousarray(dy_ref)).cuda()
        net.grads.zero_()
        dx = eng._backward(net, vals, dy.contiguous(), 'fxb', param_grads=True, need_input_grad=True)
        torch.cuda.synchronize()
        assert rel_err(_to_ref(dx), fx[pre + 'dx_f64']) <= RTOL, (case, step, 'dx', rel_err(_to_ref(dx), fx[pre + 'dx_f64']))
        gscale = max(np.abs(v).max() for k, v in fx.items() if k.startswith(pre + 'grad/'))
        for k, v in fx.items():
            if k.startswith(pre + 'grad/'):
                vname = k[len(pre + 'grad/'):-len('_f64')]
                g = net.get_variable(vname, grad=True)
                # 1e-4 of the tensor's own scale; a floor of 1e-6 of the net's gradient scale for tensors whose
                # gradient is small against the rest (fp32 accumulation noise of the large ones' neighbours)
                assert np.abs(g - v).max() <= RTOL * np.abs(v).max() + 1e-6 * gscale, \
                    (case, step, vname, np.abs(g - v).max(), np.abs(v).max())
        for k, v in fx.items():
            if k.startswith(pre + 'after/'):
                got = net.get_variable(k[len(pre + 'after/'):-len('_f64')])
                assert rel_err(got, v) <= RTOL, (case, step, k)
