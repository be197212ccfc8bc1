"""Host-side logic that needs no GPU: architecture-dict parsing and shape inference against the
oracle, error conventions of the reference API, layout permutations, Agent checkpoint rotation."""
import os

import numpy as np
import pytest
import torch

import configs
from mmdgan_hip import engine as E
from oracle import restatement as R


@pytest.mark.parametrize('name', ['cifar', 'stl', 'celeba', 'lsun'])
def test_specs_match_oracle_shape_inference(name):
    arch, lr = configs.CONFIGS[name]()
    for key, shape, net in (('generator', [arch['code'][0][0]], 'gen'), ('discriminator', list(arch['input'][0]), 'dis')):
        mine = E.build_specs(arch[key], shape, net)
        ref = R.build_net(arch[key], shape, net)
        assert len(mine) == len(ref)
        for a, b in zip(mine, ref):
            assert a.scope == b['scope'] and a.kernel_shape == b['kernel_shape']
            assert a.in_shape_ref == b['in_shape'] and a.out_shape_ref == b['out_shape']
            if a.sn:
                assert a.use_u == b['use_u'] and a.sn_x_ref == b['sn_x_shape']
    fg, fd = configs.flops_per_image(arch)
    expect = {'cifar': (206.96, 431.62), 'stl': (465.67, 971.15), 'celeba': (1092.09, 2296.38), 'lsun': (1092.09, 2296.38)}
    assert abs(fg / 1e6 - expect[name][0]) < 0.01 and abs(fd / 1e6 - expect[name][1]) < 0.01


def test_cifar_sn_vector_shapes_and_param_counts():
    arch, _ = configs.cifar()
    dis = E.build_specs(arch['discriminator'], [3, 32, 32], 'dis')
    # persistent power-iteration vector shapes, verified there by a probe of the reference
    assert [s.sn_x_ref for s in dis] == [[1, 3, 32, 32], [1, 128, 16, 16], [1, 128, 16, 16], [1, 256, 8, 8],
                                         [1, 256, 8, 8], [1, 512, 4, 4], [1, 512, 4, 4], [1, 16]]
    n_dis = sum(int(np.prod(s.kernel_shape)) + (s.channels if s.has_bias else 0) for s in dis)
    gen = E.build_specs(arch['generator'], [128], 'gen')
    n_gen = sum(int(np.prod(s.kernel_shape)) + (s.channels if s.has_bias else 0) + (2 * s.channels if s.bn else 0) for s in gen)
    assert n_dis == 5983760 and n_gen == 3811907                                                    # SURVEY A.1 totals
    assert dis[-1].row_perm is not None and gen[0].col_perm is not None


def test_error_conventions_follow_the_reference():
    ok = {'name': 'l', 'out': 8}
    with pytest.raises(AttributeError, match='not supported'):                 # layer_func.py:1275
        E.LayerSpec(dict(ok, op='sc'), 'net', [3, 8, 8])
    with pytest.raises(NotImplementedError, match='is not implemented'):       # layer_func.py:2067
        E.LayerSpec(dict(ok, type='res'), 'net', [3, 8, 8])
    with pytest.raises(NotImplementedError, match='Function swish is not implemented'):   # layer_func.py:149
        E.LayerSpec(dict(ok, act='swish'), 'net', [3, 8, 8])
    with pytest.raises(ValueError, match='numeric act_k'):                      # documented deviation
        E.LayerSpec(dict(ok, w_nm='s'), 'net', [3, 8, 8])
    with pytest.raises(AssertionError, match='does not match'):
        E.LayerSpec(dict(ok, op='d'), 'net', [3, 8, 8])
    s = E.LayerSpec(dict(ok, act_nm='bn'), 'net', [3, 8, 8])
    assert s.bn and not s.has_bias                                              # layer_func.py:1241-1242


def test_chw_permutation_roundtrip():
    c, h, w = 5, 3, 4
    perm = E._chw_perm(c, h, w)
    ref = np.arange(c * h * w, dtype=np.float32)                  # a [C,H,W]-ordered feature vector
    nat = ref[perm]                                               # [H,W,C]-ordered
    assert np.array_equal(nat.reshape(h, w, c), ref.reshape(c, h, w).transpose(1, 2, 0))
    back = np.empty_like(nat)
    back[perm] = nat
    assert np.array_equal(back, ref)


def test_flags_and_api_surface():
    from GeneralTools.misc_fun import FLAGS
    for attr in ('DEFAULT_IN', 'DEFAULT_OUT', 'IMAGE_FORMAT', 'IMAGE_FORMAT_ALIAS', 'WEIGHT_INITIALIZER',
                 'SPECTRAL_NORM_MODE', 'EPSI', 'SILENT_MODE', 'num_gpus'):
        assert hasattr(FLAGS, attr)
    assert FLAGS.IMAGE_FORMAT == 'channels_first' and FLAGS.EPSI == 1e-10
    FLAGS.NUM_GPUS = 4
    assert FLAGS.num_gpus == 4
    FLAGS.NUM_GPUS = 1
    from DeepLearning.my_sngan import SNGan
    arch, _ = configs.cifar()
    with pytest.raises(NotImplementedError, match='Not implemented.'):          # math_func.py:2651
        SNGan(arch, loss_type='no_such_loss')
    m = SNGan(arch, num_class=0, loss_type='rep', optimizer='adam', do_summary=True, do_summary_image=True,
              num_summary_image=8, image_transpose=False)
    assert (m.code_size, m.score_size, m.channels, m.height, m.width) == (128, 16, 3, 32, 32)
    with pytest.raises(AttributeError, match='max_step should be larger than step_per_epoch'):     # my_sngan.py:389-391
        m.training('cifar', None, 50000, [5e-4, 2e-4], max_step=10, batch_size=64)


def test_agent_checkpoint_rotation(tmp_path):
    from GeneralTools.misc_fun import FLAGS
    from GeneralTools.graph_func import Agent
    FLAGS.DEFAULT_OUT = str(tmp_path) + '/'
    FLAGS.SILENT_MODE = True

    class FakeEngine:
        global_step = 0

        def state_dict(self):
            return {'global_step': self.global_step}

        def load_state_dict(self, sd):
            self.global_step = sd['global_step']
    eng = FakeEngine()
    agent = Agent('unit', 'sub', load_ckpt=True, do_save=True, query_step=2, print_loss=False)
    assert os.path.isdir(os.path.join(str(tmp_path), 'unit_ckpt', 'sub'))                         # graph_func.py:172-178

    def step():
        eng.global_step += 1
    for _ in range(3):
        agent.train([step], lambda: (0.1, -0.2), eng, 5, step_per_epoch=2)
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'unit_ckpt', 'sub')))
    assert files == ['unit.ckpt-10', 'unit.ckpt-15']                                               # keep 2
    eng2 = FakeEngine()
    assert agent.load(eng2) and eng2.global_step == 15
    with pytest.raises(AssertionError, match='Model diverged'):                                    # graph_func.py:856
        agent.train([step], lambda: (float('nan'), 0.0), eng, 2)
    # a diverged model is never written, also when nothing is ever printed (query_step=None): the next run would reload it
    quiet = Agent('unit', 'nan', load_ckpt=False, do_save=True, query_step=None, print_loss=False)
    with pytest.raises(AssertionError, match='Model diverged'):
        quiet.train([step], lambda: (0.0, float('nan')), eng, 3)
    assert os.listdir(os.path.join(str(tmp_path), 'unit_ckpt', 'nan')) == []
    # data-parallel replicas hold identical variables: only rank 0 writes
    other = FakeEngine()
    other.rank = 1
    Agent('unit', 'rank1', do_save=True, query_step=None, print_loss=False).train([step], lambda: (0.1, 0.2), other, 2)
    assert os.listdir(os.path.join(str(tmp_path), 'unit_ckpt', 'rank1')) == []
    # checkpoints hold tensors and plain containers only (loaded with weights_only=True): variables round-trip as arrays
    class VarEngine(FakeEngine):
        def __init__(self):
            self.vars = {'gen/l1/kernel/kernel': np.arange(6, dtype=np.float32).reshape(2, 3)}

        def state_dict(self):
            return {'global_step': self.global_step, 'variables': dict(self.vars)}

        def load_state_dict(self, sd):
            self.global_step, self.vars = sd['global_step'], sd['variables']
    a, b = VarEngine(), VarEngine()
    a.global_step = 7
    b.vars = {}
    ag = Agent('unit', 'vars', load_ckpt=True, do_save=True, query_step=None, print_loss=False)
    ag.save(a)
    assert ag.load(b) and b.global_step == 7
    assert isinstance(b.vars['gen/l1/kernel/kernel'], np.ndarray) and np.array_equal(b.vars['gen/l1/kernel/kernel'], a.vars['gen/l1/kernel/kernel'])
    FLAGS.SILENT_MODE = False


def test_driver_table_and_residual_config():
    """drivers.py (the four experiment scripts as a table), the authored ResNet-SN dict and its FLOP accounting"""
    import configs
    import drivers
    from mmdgan_hip.tape import has_residual_blocks
    assert drivers.sub_folder_name('rep', [5e-4, 2e-4], 1.6817928, [0.0, -1.0]) == 'sngan_rep_5e-04_2e-04_k1.68_0.0_-1.0'
    assert drivers.sub_folder_name('hinge', [1e-4, 2e-4], 1.5157166, [0.0, -1.0]) == 'sngan_hinge_1e-04_2e-04_k1.52'
    assert drivers.EXPERIMENTS['celebA'].num_file * drivers.EXPERIMENTS['celebA'].per_file == 202599      # my_test_celebA.py:42
    assert drivers.EXPERIMENTS['lsun'].num_file * drivers.EXPERIMENTS['lsun'].per_file == 3033042         # my_test_lsun.py:42
    arch, lr = configs.lsun_resnet()
    assert has_residual_blocks(arch) and not has_residual_blocks(configs.cifar()[0])
    assert [d.get('type', 'default') for d in arch['discriminator']] == ['res_v1', 'res', 'res', 'res', 'res_i', 'default']
    fg, fd = configs.flops_per_image(arch)
    # by hand for the first D block (res_v1, 3 -> 64 at 64x64): 3x3 3->64 and 3x3 64->64 at 64x64, 1x1 3->64 at 32x32
    first = 2 * 9 * 3 * 64 * 64 * 64 + 2 * 9 * 64 * 64 * 64 * 64 + 2 * 3 * 64 * 32 * 32
    rest = configs.flops_per_image({'generator': arch['generator'], 'code': arch['code'], 'input': [(64, 32, 32)],
                                    'discriminator': arch['discriminator'][1:]})[1]
    assert fd == first + rest
    assert abs(fg / 1e9 - 3.91) < 0.01 and abs(fd / 1e9 - 1.88) < 0.01
    assert configs.flops_per_image(configs.cifar()[0]) == (206962688, 431620096)
