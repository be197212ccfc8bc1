"""Counterpart of the reference's my_test_celebA.py: same call sequence and keyword arguments
(FLAGS edits before the other imports, Agent(...), SNGan(...), 8 x (training -> eval_sampling)),
running on the HIP engine.  Usage:  python my_test_celebA.py [--synthetic] [--steps N]"""
import sys

import numpy as np

from GeneralTools.misc_fun import FLAGS
FLAGS.DEFAULT_IN = FLAGS.DEFAULT_IN + 'celeba_NCHW/'
if '--synthetic' in sys.argv:
    FLAGS.SYNTHETIC_DATA = True          # no dataset ships with the repository
from GeneralTools.graph_func import Agent
from DeepLearning.my_sngan import SNGan
import configs

filename = 'celebA'
architecture, lr_list = configs.CONFIGS['celeba']()            # dict identical to the reference script's
act_k = architecture['discriminator'][0]['act_k']
debug_mode = False
optimizer = 'adam'
num_instance = 202599
save_per_step = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 12500
batch_size = 64
num_class = 0                                                  # my_test_stl.py:73 forgets to define it
end_lr = 1e-7
num_threads = 7
code_x = np.random.randn(400, 128).astype(np.float32)
loss_type = 'rep'                                              # 'rep' | 'rmb'
rep_weights = [0.0, -1.0]
sample_same_class = False
sub_folder = 'sngan_{}_{:.0e}_{:.0e}_k{:.3g}_{:.1f}_{:.1f}'.format(
    loss_type, lr_list[0], lr_list[1], act_k, rep_weights[0], rep_weights[1])

agent = Agent(
    filename, sub_folder, load_ckpt=True, do_trace=False,
    do_save=True, debug_mode=debug_mode, debug_step=400,
    query_step=1000, log_device=False, imbalanced_update=None,
    print_loss=True)

mdl = SNGan(
    architecture, num_class=num_class, loss_type=loss_type,
    optimizer=optimizer, do_summary=True, do_summary_image=True,
    num_summary_image=8, image_transpose=False)

for i in range(8):
    mdl.training(
        filename, agent, num_instance, lr_list, end_lr=end_lr, max_step=save_per_step,
        batch_size=batch_size, sample_same_class=sample_same_class, num_threads=num_threads)
    if debug_mode is not None:
        _ = mdl.eval_sampling(
            filename, sub_folder, mesh_num=(20, 20), mesh_mode=0, code_x=code_x,
            real_sample=False, do_embedding=False, do_sprite=True)
    # mdl.mdl_score(...) needs the frozen Inception graph, which is not part of this repository

print('Chunk of code finished.')
