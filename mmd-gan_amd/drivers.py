"""Experiment driver shared by my_test_{cifar,stl,celebA,lsun}.py.

The reference has one ~100-line script per dataset (my_test_cifar.py etc.) that differ in a handful of constants;
here those constants are a table and the experiment - eight rounds of (train `save_per_step` steps, resume from the
checkpoint, write a 20 x 20 sprite of a fixed set of codes) - is one function driving the same API objects
(FLAGS, Agent, SNGan) with the same settings.  Inception / FID scoring needs the frozen Inception graph, which is not
in the repository: that call is left out.
"""
import argparse
import collections

import numpy as np

Experiment = collections.namedtuple('Experiment', 'config data_folder num_file per_file batch_size')

EXPERIMENTS = {                                      # reference script -> constants it sets
    'cifar': Experiment('cifar', 'cifar_NCHW/', 1, 50000, 64),       # my_test_cifar.py: one record file
    'stl': Experiment('stl', 'stl_NCHW/', 1, 100000, 64),            # my_test_stl.py
    'celebA': Experiment('celeba', 'celebA_NCHW/', 9, 22511, 64),    # my_test_celebA.py: celebA_000 ... celebA_008
    'lsun': Experiment('lsun', 'lsun_NCHW/', 61, 49722, 64),         # my_test_lsun.py: lsun_000 ... lsun_060
}


def sub_folder_name(loss_type, lr_list, act_k, rep_weights):
    """the reference's run-folder convention: loss, both learning rates, act_k and - for the repulsive losses - the
    two kernel weights"""
    name = 'sngan_{}_{:.0e}_{:.0e}_k{:.3g}'.format(loss_type, lr_list[0], lr_list[1], act_k)
    if loss_type in ('rep', 'rmb'):
        name += '_{:.1f}_{:.1f}'.format(*rep_weights)
    return name


def run(dataset, argv=None):
    ap = argparse.ArgumentParser(description='MMD-GAN training on {} with the HIP engine'.format(dataset))
    ap.add_argument('--synthetic', action='store_true', help='random images instead of the dataset files')
    ap.add_argument('--steps', type=int, default=12500, help='training steps per round (the reference: 12500)')
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--loss', default='rep', help="'rep', 'rmb', 'mmd_g', 'mgb', 'hinge', 'logistic'")
    ap.add_argument('--sn-mode', default='default', help="'default' (PICO) or 'sn_paper' (PIM)")
    args = ap.parse_args(argv)
    exp = EXPERIMENTS[dataset]

    from GeneralTools.misc_fun import FLAGS           # FLAGS are edited before anything else reads them
    FLAGS.DEFAULT_IN += exp.data_folder
    FLAGS.SYNTHETIC_DATA = bool(args.synthetic)
    FLAGS.SPECTRAL_NORM_MODE = args.sn_mode
    import configs
    from DeepLearning.my_sngan import SNGan
    from GeneralTools.graph_func import Agent

    architecture, lr_list = configs.CONFIGS[exp.config]()
    rep_weights = [0.0, -1.0]                         # weights of e_kxy and -e_kyy; they must differ by one
    act_k = architecture['discriminator'][0]['act_k']
    folder = sub_folder_name(args.loss, lr_list, act_k, rep_weights)
    codes = np.random.randn(400, architecture['code'][0][0]).astype(np.float32)      # the same codes every round

    # a single record file is named after the dataset, several are <dataset>_000, <dataset>_001, ...
    files = dataset if exp.num_file == 1 else ['{}_{:03d}'.format(dataset, i) for i in range(exp.num_file)]
    num_instance = exp.num_file * exp.per_file
    agent = Agent(files, folder, load_ckpt=True, do_save=True, query_step=1000, print_loss=True)
    model = SNGan(architecture, num_class=0, loss_type=args.loss, optimizer='adam', rep_weights=rep_weights)
    for _ in range(args.rounds):
        model.training(files, agent, num_instance, lr_list, end_lr=1e-7, max_step=args.steps,
                       batch_size=exp.batch_size, num_threads=7)
        model.eval_sampling(files, folder, mesh_num=(20, 20), mesh_mode=0, code_x=codes, do_sprite=True)
    print('Chunk of code finished.')
