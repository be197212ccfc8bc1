"""The reference-facing Python API on the GPU: SNGan.training / eval_sampling through Agent with
synthetic data, GANLoss.apply, get_squared_dist, Net/Routine inference."""
import numpy as np
import pytest
import torch

import configs
from helpers import RTOL, rel_err
from oracle import restatement as R

pytestmark = pytest.mark.gpu


def test_sngan_training_and_sampling(tmp_path):
    from GeneralTools.misc_fun import FLAGS
    FLAGS.DEFAULT_OUT = str(tmp_path) + '/'
    FLAGS.SYNTHETIC_DATA = True
    FLAGS.SILENT_MODE = True
    from GeneralTools.graph_func import Agent
    from DeepLearning.my_sngan import SNGan
    arch, lr = configs.cifar()
    agent = Agent('cifar', 'unit', load_ckpt=True, do_save=True, query_step=5, print_loss=True)
    mdl = SNGan(arch, num_class=0, loss_type='rep', optimizer='adam')
    mdl.training('cifar', agent, 64 * 8, lr, end_lr=1e-7, max_step=10, batch_size=64)
    assert mdl.global_step == 10
    lg, ld = mdl.engine.losses[:2].tolist()
    assert np.isfinite(lg) and np.isfinite(ld)
    mdl.training('cifar', agent, 64 * 8, lr, end_lr=1e-7, max_step=10, batch_size=64)       # resumes at 10
    assert mdl.global_step == 20
    x = mdl.eval_sampling('cifar', 'unit', mesh_num=(4, 4), code_x=np.random.randn(16, 128).astype(np.float32))
    assert x.shape == (16, 3, 32, 32) and np.abs(x).max() <= 1.0
    with pytest.raises(NotImplementedError):
        mdl.mdl_score('cifar', 'unit', 64)
    FLAGS.SYNTHETIC_DATA = False
    FLAGS.SILENT_MODE = False


def test_ganloss_and_squared_dist_api():
    from GeneralTools.math_func import GANLoss, get_squared_dist
    rs = np.random.RandomState(0)
    sg = torch.as_tensor((rs.randn(64, 16) * 0.25).astype(np.float32)).cuda()
    sx = torch.as_tensor((rs.randn(64, 16) * 0.3 + 0.1).astype(np.float32)).cuda()
    for loss in ('rep', 'rmb', 'mmd_g', 'fixed_g', 'mgb', 'hinge', 'logistic', ''):
        lg, ld = GANLoss(False).apply(sg, sx, loss, batch_size=64, d=16, rep_weights=[0.0, -1.0])
        rg, rd, _ = R.gan_loss(sg.cpu().double(), sx.cpu().double(), loss, 64)
        assert abs(float(lg) - float(rg)) <= RTOL * abs(float(rg)) + 4e-7
        assert abs(float(ld) - float(rd)) <= RTOL * abs(float(rd)) + 4e-7
    with pytest.raises(NotImplementedError, match='Not implemented.'):
        GANLoss().apply(sg, sx, 'wasserstein', batch_size=64)
    dxx, dxy, dyy = get_squared_dist(sg, sx)
    ref = R.get_squared_dist(sg.cpu().double(), sx.cpu().double())
    for got, r in zip((dxx, dxy, dyy), ref):
        assert np.abs(got.cpu().numpy() - r.numpy()).max() <= 1e-5


def test_routine_inference_matches_oracle():
    from GeneralTools.layer_func import Net, Routine
    designs = [{'name': 'l1', 'out': 32 * 4 * 4, 'op': 'd', 'act': 'relu', 'act_nm': 'bn', 'out_reshape': [32, 4, 4]},
               {'name': 'l2_up', 'out': 16, 'op': 'tc', 'act': 'relu', 'act_nm': 'bn', 'kernel': 4, 'strides': 2},
               {'name': 'l3', 'out': 3, 'act': 'tanh'}]
    net = Net(designs, net_name='gen', data_format='channels_first', num_class=0)
    r = Routine(net)
    r.add_input_layers([64, 24], [0])
    r.seq_links(list(range(net.num_layers)))
    r.add_output_layers([net.num_layers - 1])
    z = torch.as_tensor(np.random.RandomState(1).randn(6, 24).astype(np.float32)).cuda()
    y = r({'x': z}, is_training=True)['x']
    assert tuple(y.shape) == (6, 3, 8, 8)
    specs = R.build_net(designs, [24], 'gen')
    params = {k: torch.tensor(net.network.get_variable(k), dtype=torch.float64) for k in net.network.variable_names()}
    params['gen/l1/BN/BN/moving_mean'] = torch.zeros(512, dtype=torch.float64)     # state before the training call
    params['gen/l1/BN/BN/moving_variance'] = torch.ones(512, dtype=torch.float64)
    params['gen/l2_up/BN/BN/moving_mean'] = torch.zeros(16, dtype=torch.float64)
    params['gen/l2_up/BN/BN/moving_variance'] = torch.ones(16, dtype=torch.float64)
    ref, _ = R.net_forward(specs, params, z.cpu().double(), True)
    assert rel_err(y.cpu().numpy(), ref.numpy()) <= RTOL
