"""SNGan: the reference's model class (DeepLearning/my_sngan.py:31-690) over the HIP engine.

Kept: constructor and method signatures (my_sngan.py:31-70, 364-365, 499-502, 602), the
architecture-dict and loss-name vocabulary, the step semantics (one simultaneous D+G update per
step from one forward pass) and the error behaviour for bad arguments.  Not kept:
TF graphs/sessions, summaries, the tfrecord reader and the Inception scorer (SURVEY.md section 8
marks them out of scope); `mdl_score` says so instead of pretending.
"""
import os
from math import gcd

import numpy as np
import torch

from GeneralTools.misc_fun import FLAGS
from GeneralTools.graph_func import prepare_folder, write_sprite_wrapper
from mmdgan_hip import ops
from mmdgan_hip.engine import GanEngine
from mmdgan_hip.tape import TapeEngine, needs_tape_engine


class SNGan(object):
    def __init__(self, architecture, num_class=0, loss_type='logistic', optimizer='adam', do_summary=True,
                 do_summary_image=True, num_summary_image=8, image_transpose=False, **kwargs):
        self.optimizer_type = ['sgd', 'momentum', 'adam', 'rmsprop']
        self.data_format = FLAGS.IMAGE_FORMAT
        self.architecture, self.loss_type, self.optimizer = architecture, loss_type, optimizer
        self.num_class = num_class
        self.channels, self.height, self.width = architecture['input'][0]
        self.input_size = int(np.prod(architecture['input'][0]))
        self.code_size = architecture['code'][0][0]
        self.score_size = architecture['discriminator'][-1]['out']
        self.do_summary, self.do_summary_image, self.num_summary_image = do_summary, do_summary_image, num_summary_image
        self.loss_names = '<loss_gen>, <loss_dis>'
        self.global_step = None
        self.step_per_epoch = None
        self.sample_same_class = False
        self.force_print = True
        self.rep_weights = kwargs['rep_weights'] if 'rep_weights' in kwargs else [0.0, -1.0]
        self.penalty_weight = kwargs['mmd_g_scale'] if 'mmd_g_scale' in kwargs else 0.1
        if image_transpose:
            raise NotImplementedError('image_transpose is outside the hot path')
        if num_class > 1:
            raise NotImplementedError('conditional models are outside the hot path')
        if loss_type not in ops.LOSS:
            raise NotImplementedError('Not implemented.')                      # math_func.py:2651
        if optimizer != 'adam':
            raise NotImplementedError('only the adam branch of opt_config is on the hot path (graph_func.py:518-527)')
        self.engine = None
        self.dist_group = kwargs.get('dist_group')

    # --------------------------------------------------------------------------------------
    def init_net(self, lr_list, batch_size, seed=0):
        """G and D with their optimisers (my_sngan.py:85-108, 412-415)."""
        if self.engine is None or self.engine.B != batch_size:
            # residual blocks / scaling ops / identity layers / batch norm in D: the primitive-op engine (mmdgan_hip/tape.py)
            engine_cls = TapeEngine if needs_tape_engine(self.architecture) else GanEngine
            self.engine = engine_cls(self.architecture, self.loss_type, lr_list, tuple(self.rep_weights),
                                    batch_size=batch_size, seed=seed, dist_group=self.dist_group,
                                    sn_mode=FLAGS.SPECTRAL_NORM_MODE,                # layer_func.py:802-814
                                    weight_init=FLAGS.WEIGHT_INITIALIZER,           # layer_func.py:27-64
                                    # (how a step reaches the GPU - eager issue, one hipGraph, the library's launch
                                    # plan - is the engine's `launch_mode`; MMDGAN_LAUNCH_MODE, mmdgan_hip/settings.py)
                                    )
            if self.dist_group is not None:
                # data-parallel replicas must start from the same variables, Adam moments, SN vectors and BN statistics
                from mmdgan_hip import dist as mdist
                mdist.broadcast_state(self.engine, self.dist_group)
        else:
            self.engine.lr_d, self.engine.lr_g = float(lr_list[0]), float(lr_list[1])
        return self.engine

    def get_data_batch(self, filename, batch_size, num_instance, file_repeat=1, num_threads=7, shuffle_file=False):
        """returns fn() -> NHWC fp32 batch on the device, values in [-1, 1] (my_sngan.py:348-354: ReadTFRecords ->
        shape2image -> next_batch; the uint8 records are decoded on the device, GeneralTools/input_func.py)."""
        if FLAGS.SYNTHETIC_DATA:
            dev = torch.device('cuda')
            gen = torch.Generator(device=dev)
            gen.manual_seed(1234 + (self.engine.rank if self.engine is not None else 0))   # a replica's own batches
            buf = torch.empty((batch_size, self.height, self.width, self.channels), device=dev)
            return lambda: buf.uniform_(-1, 1, generator=gen)
        from GeneralTools.input_func import ReadTFRecords
        try:
            self.training_data = ReadTFRecords(
                filename, self.channels * self.height * self.width, num_labels=0, x_dtype='string',
                batch_size=batch_size, file_repeat=file_repeat, num_threads=num_threads, shuffle_file=shuffle_file)
        except AssertionError as err:
            raise FileNotFoundError('{} (a uint8 [N, C*H*W] .npy of the same name is accepted too; '
                                    'FLAGS.SYNTHETIC_DATA = True needs no data)'.format(err))
        self.training_data.shape2image(self.channels, self.height, self.width)
        return lambda: self.training_data.next_batch()['x']

    # --------------------------------------------------------------------------------------
    def training(self, filename, agent, num_instance, lr_list, end_lr=1e-7, max_step=None, batch_size=64,
                 sample_same_class=False, num_threads=7, gpu='/gpu:0'):
        self.step_per_epoch = int(np.floor(num_instance / batch_size))
        self.sample_same_class = sample_same_class
        if max_step >= self.step_per_epoch:
            file_repeat = int(batch_size / gcd(num_instance, batch_size))     # my_sngan.py:383-385
        else:
            if isinstance(filename, str) or (isinstance(filename, (list, tuple)) and len(filename) == 1):
                raise AttributeError('max_step should be larger than step_per_epoch when there is a single file.')
            file_repeat = 1
        FLAGS.print('Num Instance: {}; Num Class: {}; Batch: {}; File_repeat: {}'.format(
            num_instance, self.num_class, batch_size, file_repeat))
        eng = self.init_net(lr_list, batch_size)
        next_batch = self.get_data_batch(filename, batch_size, num_instance, file_repeat, num_threads)
        FLAGS.print('Shape of input batch: {}'.format([batch_size, self.channels, self.height, self.width]))
        FLAGS.print('loss_list name: {}.'.format(self.loss_names))

        def step_fn():
            eng.step(next_batch())              # z ~ N(0,1) sampled on the device (my_sngan.py:123-124)

        def read_losses():
            lg, ld = eng.losses[:2].tolist()
            return lg, ld
        agent.train([step_fn], read_losses, eng, max_step, self.step_per_epoch, None, None,
                    force_print=self.force_print)
        self.global_step = eng.global_step
        self.force_print = False

    # --------------------------------------------------------------------------------------
    def eval_sampling(self, filename, sub_folder, mesh_num=None, mesh_mode=0, if_invert=False, code_x=None,
                      code_y=None, real_sample=False, sample_same_class=False, get_dis_score=True, do_sprite=True,
                      do_embedding=False, ckpt_file=None, num_threads=7):
        """my_sngan.py:499-581, signature and defaults as there: G(code_x) with BN moving statistics, clipped to [-1,1],
        written as the sprite <summary_folder>/<filename>_g_<sub_folder>_<step>_<mesh_mode>.png; with real_sample a batch
        of data is drawn too, written as ..._r_... (:567-571), and - when get_dis_score is set as well (:558) - D scores
        both in inference mode (`self.Dis(concat(data, gen), is_training=False)`, :559-560).  code_x defaults to
        MeshCode(...).get_batch(mesh_mode).  Returns the generated NCHW array (the reference returns nothing); the rest
        is left in self.eval_outputs = {x_gen, x_real, s_x, s_gen}.  The TensorBoard embedding (do_embedding) is not
        built.  The model evaluated is the one in memory (trained or loaded by an Agent): there is no `rollback`."""
        if self.engine is None:
            raise RuntimeError('eval_sampling: train (or load) a model first')
        if do_embedding:
            raise NotImplementedError('eval_sampling: the TensorBoard embedding (do_embedding) is not built')
        if ckpt_file is not None:
            raise NotImplementedError('eval_sampling: load older checkpoints through Agent(load_ckpt=True)')
        _, summary_folder, _ = prepare_folder(filename, sub_folder=sub_folder)
        if mesh_num is None:
            mesh_num = (10, 10)                                              # my_sngan.py:524-525
        elif code_x is not None:
            assert code_x.shape[0] == mesh_num[0] * mesh_num[1]               # my_sngan.py:526-527
        batch_size = mesh_num[0] * mesh_num[1]
        x_real_nhwc = None
        if real_sample:                                                       # my_sngan.py:538-541
            self.sample_same_class = sample_same_class
            x_real_nhwc = self.get_data_batch(filename, batch_size, batch_size, num_threads=num_threads)().clone()
        if code_x is None:                                                    # my_sngan.py:545-547
            from GeneralTools.math_func import MeshCode
            code_x = MeshCode(self.code_size, mesh_num=mesh_num).get_batch(mesh_mode, name='code_x')
        code_x = torch.as_tensor(np.asarray(code_x, np.float32)).cuda()
        gen_nhwc = []
        for i in range(0, code_x.shape[0], self.engine.B):
            z = code_x[i:i + self.engine.B].contiguous()
            gen_nhwc.append(self.engine.generate(z, is_training=False).clone().clamp_(-1, 1))   # :553-555
        gen_nhwc = torch.cat(gen_nhwc, 0)
        x_gen = ops.nhwc_to_nchw(gen_nhwc.contiguous()).cpu().numpy()
        self.eval_outputs = {'x_gen': x_gen, 'x_real': None, 's_x': None, 's_gen': None}
        if real_sample:
            self.eval_outputs['x_real'] = ops.nhwc_to_nchw(x_real_nhwc.contiguous()).cpu().numpy()
            if get_dis_score:                                                 # my_sngan.py:558-560
                scores = self.engine.discriminate(torch.cat([x_real_nhwc, gen_nhwc], 0))
                self.eval_outputs['s_x'] = scores[:batch_size].cpu().numpy()
                self.eval_outputs['s_gen'] = scores[batch_size:].cpu().numpy()
        if do_sprite:                                                         # my_sngan.py:566-575
            step = str(self.engine.global_step)
            if real_sample:
                write_sprite_wrapper(
                    self.eval_outputs['x_real'], mesh_num, filename, file_folder=summary_folder,
                    file_index='_r_' + sub_folder + '_' + step + '_' + str(mesh_mode),
                    if_invert=if_invert, image_format=FLAGS.IMAGE_FORMAT)
            write_sprite_wrapper(
                x_gen, mesh_num, filename, file_folder=summary_folder,
                file_index='_g_' + sub_folder + '_' + step + '_' + str(mesh_mode),
                if_invert=if_invert, image_format=FLAGS.IMAGE_FORMAT)
        return x_gen

    def mdl_score(self, filename, sub_folder, batch_size, num_batch=10, model='v1', ckpt_file=None, num_threads=7):
        raise NotImplementedError('mdl_score needs the frozen Inception-v1 graph (graph_func.py:1748-1799), which '
                                  'is not in the repository and is outside the hot path (SURVEY.md section 8)')
