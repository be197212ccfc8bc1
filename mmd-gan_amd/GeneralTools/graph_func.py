"""Agent: folders, checkpoints and the step loop (reference graph_func.py:161-180, 1145-1219,
851-874).  The TF session machinery is gone - SNGan.training hands the Agent a step function and
the Agent runs the loop with the reference's semantics: resume from the newest checkpoint, run
max_step steps, NaN assert, print every query_step, save at the last step (keep 2).
"""
import glob
import math
import os
import time

import warnings

import numpy as np
import torch

from GeneralTools.math_func import mean_cov_np, trace_sqrt_product_np
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


def sprite_array(images, mesh_num=None, if_invert=False):
    """the uint8 [rows*H, cols*W, 3] mosaic write_sprite saves (graph_func.py:222-263): each image scaled by its own
    min and max to [0,1], optionally inverted, laid out row-major on the mesh (a square one, zero-padded, when no
    mesh is given), x*255 truncated to uint8."""
    img = np.asarray(images)
    if img.ndim == 3:
        img = img[..., np.newaxis]
    if img.shape[3] == 1:
        img = np.repeat(img, 3, axis=3)
    img = img.astype(np.float32)
    flat = img.reshape(img.shape[0], -1)
    img = img - flat.min(axis=1).reshape(-1, 1, 1, 1)
    img = img / img.reshape(img.shape[0], -1).max(axis=1).reshape(-1, 1, 1, 1)
    if if_invert:
        img = 1 - img
    if mesh_num is None:
        FLAGS.print('Mesh_num will be calculated as sqrt of batch_size')
        side = int(np.ceil(np.sqrt(img.shape[0])))
        mesh_num = (side, side)
        blank = np.zeros((side * side - img.shape[0],) + img.shape[1:], dtype=img.dtype)
        img = np.concatenate([img, blank], axis=0)
    rows, cols = tuple(mesh_num)
    n, h, w, c = img.shape
    assert rows * cols == n, 'mesh {}x{} does not hold {} images'.format(rows, cols, n)
    mosaic = np.empty((rows * h, cols * w, c), dtype=np.float32)
    for r in range(rows):
        for q in range(cols):
            mosaic[r * h:(r + 1) * h, q * w:(q + 1) * w] = img[r * cols + q]
    return (mosaic * 255).astype(np.uint8)


def write_sprite(sprite_path, images, mesh_num=None, if_invert=False):
    """channels_last images -> one PNG mosaic (graph_func.py:222-266; scipy.misc.imsave no longer exists, PIL writes
    the same uint8 array)."""
    from PIL import Image
    Image.fromarray(sprite_array(images, mesh_num, if_invert)).save(sprite_path)


def write_sprite_wrapper(images, mesh_num, filename, file_folder=None, file_index='', if_invert=False,
                         image_format='channels_last'):
    """graph_func.py:269-297: <file_folder>/<filename><file_index>.png, never overwriting an existing file."""
    if not isinstance(filename, str):
        filename = filename[0]
    if isinstance(mesh_num, list):
        mesh_num = tuple(mesh_num)
    if file_folder is None:
        file_folder = FLAGS.DEFAULT_OUT
    if image_format in {'channels_first', 'NCHW'}:
        images = np.transpose(images, axes=(0, 2, 3, 1))
    sprite_path = os.path.join(file_folder, filename + file_index + '.png')
    if os.path.isfile(sprite_path):
        warnings.warn('This file already exists: ' + sprite_path)
    else:
        write_sprite(sprite_path, images, mesh_num=mesh_num, if_invert=if_invert)
    return sprite_path


class GenerativeModelMetric(object):
    """Only the part of the reference class (graph_func.py:1595-2094) that needs no pretrained network: the Frechet
    distance between two sets of EXTERNALLY SUPPLIED pool3 features.  Everything that runs the frozen Inception-v1
    graph raises (the file is not part of the reference repository, Addon/inception_v1/ReadMe.md)."""

    def __init__(self, image_format=None, model='v1', model_path=None):
        self.image_format = FLAGS.IMAGE_FORMAT if image_format is None else image_format
        self.model, self.model_path = model, model_path

    @staticmethod
    def my_fid_from_pool3(x_pool3_np, y_pool3_np):
        """graph_func.py:1733-1745; either argument may be features [N, D] or a [mean, cov] pair."""
        x_mean, x_cov = x_pool3_np if isinstance(x_pool3_np, (list, tuple)) else mean_cov_np(x_pool3_np)
        y_mean, y_cov = y_pool3_np if isinstance(y_pool3_np, (list, tuple)) else mean_cov_np(y_pool3_np)
        return np.sum((x_mean - y_mean) ** 2) + np.trace(x_cov) + np.trace(y_cov) \
            - 2.0 * trace_sqrt_product_np(x_cov, y_cov)

    def inception_score_and_fid_v1(self, *args, **kwargs):
        raise NotImplementedError('needs the frozen Inception-v1 graph (graph_func.py:1748-1799), not in the repository')

    inception_v1 = inception_v1_one_batch = inception_score_and_fid_v1


class _KernelTimeline(object):
    """the reference's TimeLiner (graph_func.py:578-603) for this build: a Chrome trace of the traced steps' HIP kernels"""

    def __init__(self):
        from torch.profiler import ProfilerActivity, profile
        self._prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])

    def start(self):
        self._prof.__enter__()

    def save(self, trace_file):
        torch.cuda.synchronize()
        self._prof.__exit__(None, None, None)
        self._prof.export_chrome_trace(trace_file)
        return trace_file


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
        self.trace_file = None                                  # <summary_folder>/timeline.json after a do_trace run

    # -- checkpoints (torch.save of the reference-layout state dict; TF ckpt format is out of scope)
    def latest_ckpt(self):
        files = glob.glob(self.save_path + '-*')
        return max(files, key=lambda f: int(f.rsplit('-', 1)[1])) if files else None

    def load(self, engine):
        path = self.latest_ckpt()
        if self.load_ckpt and path is not None:
            # tensors and plain containers only: a checkpoint file is data, never code (weights_only=True refuses
            # anything that would need unpickling arbitrary objects)
            sd = torch.load(path, map_location='cpu', weights_only=True)
            for key in ('variables', 'adam_m', 'adam_v'):           # name -> tensor on disk, name -> numpy for the engine
                if isinstance(sd.get(key), dict):
                    sd[key] = {k: v.numpy() for k, v in sd[key].items()}
            engine.load_state_dict(sd)
            FLAGS.print('Model reloaded from {}'.format(path))
            return True
        return False

    def save(self, engine):
        """in a data-parallel run the replicas hold identical variables: rank 0 writes, the others return"""
        if getattr(engine, 'rank', 0) != 0:
            return None
        path = '{}-{}'.format(self.save_path, engine.global_step)
        sd = engine.state_dict()
        for key in ('variables', 'adam_m', 'adam_v'):            # numpy arrays -> tensors: loadable with weights_only=True
            if isinstance(sd.get(key), dict):
                sd[key] = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd[key].items()}
        torch.save(sd, path)
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
        tracer = None
        for step in range(max_step):
            if self.do_trace and step == max(max_step - 5, 0):
                # graph_func.py:996-998, 1015-1025: the LAST FIVE steps run traced and the per-step timelines are merged into
                # one Chrome trace, <summary_folder>/timeline.json (:1139-1141).  Here the tracer is the ROCm kernel
                # tracer behind torch.profiler: every HIP kernel launch of those steps with its stream, start and duration
                tracer = _KernelTimeline()
                tracer.start()
            step_fn()
            last = step == max_step - 1
            # the reference keys its periodic print on the GLOBAL step (graph_func.py:860), so a resumed run keeps the
            # cadence of the run it continues; the value it fetches next to the update is the pre-increment one
            gstep = engine.global_step - 1
            if self.query_step is not None and (gstep % self.query_step == self.query_step - 1 or last):
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
                # never write a diverged model: the next run would reload it (the reference asserts on every step,
                # graph_func.py:856; here the losses stay on the device between query points, so check now)
                lg, ld = read_losses()
                assert not (math.isnan(lg) or math.isnan(ld)), 'Model diverged with loss = NaN'
                self.save(engine)
        duration = time.time() - start
        FLAGS.print('Training for {} steps took {:.3f} sec.'.format(max_step, duration))
        if tracer is not None:
            self.trace_file = tracer.save(os.path.join(self.summary_folder, 'timeline.json'))
        return duration
