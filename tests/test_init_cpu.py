"""the three FLAGS.WEIGHT_INITIALIZER modes (reference layer_func.py:14-80) as the build draws them,
against sample statistics recorded from the reference's own weight_initializer / bias_initializer
(tests/golden/init_stats.npz, oracle/make_golden.py:make_init_stats).  The random streams differ (NumPy here, the
shim's there), so the comparison is statistical: standard deviation, extreme value, kurtosis (1.8 = uniform,
~2.37 = normal truncated at two sigma) of >= 70 k draws each."""
import numpy as np
import pytest

from helpers import golden, load
from mmdgan_hip import initializers as I


@pytest.fixture(scope='module')
def fx():
    return load(golden('init_stats.npz')[0])


@pytest.mark.parametrize('mode', ['default', 'sn_paper', 'pg_paper'])
@pytest.mark.parametrize('act', ['relu', 'lrelu', 'tanh', 'linear'])
def test_weight_initializer_distributions(fx, mode, act):
    for sname in ('dense', 'conv3', 'tconv4'):
        key = '{}/{}/{}/'.format(mode, act, sname)
        shape = [int(v) for v in fx[key + 'shape']]
        w = I.weight_initializer(np.random.RandomState(11), shape, act, mode)
        assert w.dtype == np.float32 and list(w.shape) == shape
        assert abs(w.std() / float(fx[key + 'std']) - 1.0) < 0.02, (key, w.std(), float(fx[key + 'std']))
        assert abs(np.abs(w).max() / float(fx[key + 'absmax']) - 1.0) < 0.01, key      # 2 sigma cut / uniform limit
        kurt = ((w - w.mean()) ** 4).mean() / w.var() ** 2
        assert abs(kurt - float(fx[key + 'kurt'])) < 0.05, (key, kurt)
        assert abs(w.mean()) < 0.02 * w.std()


def test_tc_kernels_take_fan_in_from_cout():
    """TF's fan rule reads shape[-2] as the input dimension of EVERY kernel: a transposed-conv kernel [k,k,Cout,Cin]
    is scaled by Cout (layer_func.py:34-36 with :595)"""
    k, cout, cin = 4, 32, 256
    w = I.weight_initializer(np.random.RandomState(0), [k, k, cout, cin], 'relu')
    assert abs(np.abs(w).max() - 2.0 * np.sqrt(2.0 / (k * k * cout))) < 1e-3           # cut at 2 * sqrt(2 / fan_in)
    assert I.fans([k, k, cout, cin]) == (k * k * cout, k * k * cin)
    assert I.fans([128, 8192]) == (128, 8192)


def test_init_w_scale_and_bias(fx):
    for act, scale in (('relu', 0.25), ('linear', 4.0), ('relu', 0.0)):
        key = 'default/{}/conv3/scale{}/'.format(act, scale)
        w = I.weight_initializer(np.random.RandomState(5), [3, 3, 64, 128], act, 'default', scale)
        if scale == 0.0:
            assert not w.any() and float(fx[key + 'absmax']) == 0.0                     # layer_func.py:28-29
            continue
        assert abs(w.std() / float(fx[key + 'std']) - 1.0) < 0.02, key
        assert abs(np.abs(w).max() / float(fx[key + 'absmax']) - 1.0) < 0.01, key
    b = I.bias_initializer(np.random.RandomState(6), [4096])
    assert abs(b.std() / float(fx['bias/std']) - 1.0) < 0.05 and np.abs(b).max() <= 2e-5    # trunc-normal(1e-5), :747
    assert not I.bias_initializer(np.random.RandomState(6), [64], 0.0).any()


def test_unknown_mode_raises_like_the_reference(fx):
    with pytest.raises(NotImplementedError) as err:
        I.weight_initializer(np.random.RandomState(0), [4, 4], 'relu', 'no_such_mode')
    assert str(err.value) == str(fx['unknown_mode_error'])                              # layer_func.py:64


@pytest.mark.gpu
def test_networks_draw_with_the_flagged_mode():
    """the flag reaches the variables of a Network (GanEngine, Routine); device arenas, so a GPU test"""
    import torch
    from mmdgan_hip.engine import Network, build_specs
    designs = [{'name': 'l1', 'out': 64, 'act': 'lrelu', 'act_k': 1.5, 'w_nm': 's'},
               {'name': 'l2', 'out': 64, 'act': 'lrelu', 'act_k': 1.5, 'w_nm': 's', 'kernel': 4, 'strides': 2,
                'init_w_scale': 0.0}]
    for mode, cut in (('default', 2.0 * np.sqrt(2.0 / 1.01 / 27.0)), ('sn_paper', 0.04), ('pg_paper', 2.0)):
        net = Network(build_specs(designs, [3, 16, 16], 'dis'), torch.device('cuda'), np.random.RandomState(3), mode)
        w = net.get_variable('dis/l1/kernel/kernel')
        assert abs(np.abs(w).max() - cut) < 0.03 * cut, (mode, np.abs(w).max())
        w2 = net.get_variable('dis/l2/kernel/kernel')
        assert (not w2.any()) == (mode == 'default')            # init_w_scale only exists in the default branch
        assert np.abs(net.get_variable('dis/l1/bias/bias')).max() <= 2e-5
    with pytest.raises(NotImplementedError, match='The initializer he is not implemented.'):
        Network(build_specs(designs, [3, 16, 16], 'dis'), torch.device('cuda'), np.random.RandomState(3), 'he')
