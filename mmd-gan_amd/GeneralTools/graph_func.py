"""Agent: folders, checkpoints and the step loop (reference graph_func.py:161-180, 1145-1219,
851-874).  The TF session machinery is gone - SNGan.training hands the Agent a step function and
the Agent runs the loop with the reference's semantics: resume from the newest checkpoint, run
max_step steps, NaN assert, print every query_step, save at the last step (keep 2).
"""
import glob
import math
import os
import time

import torch

from GeneralTools.misc_fun import FLAGS


def prepare_folder(filename, sub_folder='', set_folder=True):
    """<DEFAULT_OUT>/<file>_ckpt/<sub>, <file>_log/<sub> (graph_func.py:161-180)."""
    if not isinstance(filename, str):
        filename = filename[0]
    ckpt_folder = os.path.join(FLAGS.DEFAULT_OUT, filename + '_ckpt', sub_folder)
    summary_folder = os.path.join(FLAGS.DEFAULT_OUT, filename + '_log', sub_folder)
    save_path = os.path.join(ckpt_folder, filename + '.ckpt')
    if set_folder:
        os.makedirs(ckpt_folder, exist_ok=True)
        os.makedirs(summary_folder, exist_ok=True)
    return ckpt_folder, summary_folder, save_path


class Agent(object):
    def __init__(self, filename, sub_folder, load_ckpt=False, do_trace=False, do_save=True, debug_mode=False,
                 debug_step=800, query_step=500, log_device=False, imbalanced_update=None, print_loss=True):
        self.ckpt_folder, self.summary_folder, self.save_path = prepare_folder(filename, sub_folder=sub_folder)
        self.load_ckpt, self.do_trace, self.do_save = load_ckpt, do_trace, do_save
        self.debug, self.debug_step, self.query_step = debug_mode, debug_step, query_step
        self.log_device, self.print_loss = log_device, print_loss
        if imbalanced_update is not None:
            raise NotImplementedError('imbalanced / dynamic update schedules are outside the hot path')
        self.imbalanced_update = None
        self.step_times = []

    # -- checkpoints (torch.save of the reference-layout state dict; TF ckpt format is out of scope)
    def latest_ckpt(self):
        files = glob.glob(self.save_path + '-*')
        return max(files, key=lambda f: int(f.rsplit('-', 1)[1])) if files else None

    def load(self, engine):
        path = self.latest_ckpt()
        if self.load_ckpt and path is not None:
            engine.load_state_dict(torch.load(path, map_location='cpu', weights_only=False))
            FLAGS.print('Model reloaded from {}'.format(path))
            return True
        return False

    def save(self, engine):
        path = '{}-{}'.format(self.save_path, engine.global_step)
        torch.save(engine.state_dict(), path)
        files = sorted(glob.glob(self.save_path + '-*'), key=lambda f: int(f.rsplit('-', 1)[1]))
        for old in files[:-2]:                                   # Saver(max_to_keep=2), graph_func.py:708-717
            os.remove(old)
        return path

    def train(self, op_list, loss_list, global_step, max_step, step_per_epoch=None, summary_op=None,
              summary_image_op=None, imbalanced_update=None, force_print=False):
        """op_list = [step_fn] (runs one G+D step), loss_list = fn() -> (loss_gen, loss_dis) host floats,
        global_step = the engine (its .global_step is the counter, my_sngan.py:424)."""
        engine, step_fn, read_losses = global_step, op_list[0], loss_list
        self.load(engine)
        start = time.time()
        for step in range(max_step):
            step_fn()
            last = step == max_step - 1
            if self.query_step is not None and (step % self.query_step == self.query_step - 1 or last):
                # losses stay on the device between query points: one host sync per query_step
                # (the reference syncs every step for its NaN assert, graph_func.py:856)
                lg, ld = read_losses()
                assert not (math.isnan(lg) or math.isnan(ld)), 'Model diverged with loss = NaN'
                if self.print_loss or force_print:
                    epoch = engine.global_step // step_per_epoch if step_per_epoch else 0
                    FLAGS.print('Epoch {}, global steps {}, loss_list {}'.format(
                        epoch, engine.global_step, ['{}'.format(['<{:.2f}>'.format(v) for v in (lg, ld)])]),
                        force_print=force_print)
            if last and self.do_save:
                self.save(engine)
        duration = time.time() - start
        FLAGS.print('Training for {} steps took {:.3f} sec.'.format(max_step, duration))
        return duration
