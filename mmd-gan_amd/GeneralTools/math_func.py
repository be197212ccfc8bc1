"""Loss-side API of the reference's math_func.py, served by the fused HIP kernel.

Kept names and signatures: get_squared_dist (math_func.py:767) and GANLoss (:2088) with
.apply(score_gen, score_data, loss_type, **kwargs) -> (loss_gen, loss_dis); matrix_mean_wo_diagonal
(:1048), mmd_g (:1288) and mmd_g_bounded (:1356) are fused inside the kernel and have no separate
entry (the B x B matrices they take are never materialised).  Inputs are [B, d] fp32 CUDA tensors; results are
device tensors (no host synchronisation).  Implemented loss names: the hot path's 'rep', 'rmb' and their
aliases (math_func.py:2644-2647) and, from SURVEY 8(f) row 1, 'mmd_g' / 'fixed_g', 'mgb', 'hinge' and
'logistic' / '' (:2602-2611), and the coin-mixed 'mmd_g_mix' / 'fixed_g_mix' / 'sgm' (:2613-2622; the uniform draw can be
injected as `uni=`, the boolean masks are exposed as .mix_indices / .mix_group_1 / .mix_group_2); every other name raises
exactly as the reference does for an unknown one.
"""
import numpy as np

from GeneralTools.misc_fun import FLAGS
from mmdgan_hip import ops

_NOT_ON_HOT_PATH = {'wasserstein', 'fixed_t', 'mmd_t', 'rand_g', 'rgb', 'rand_g_mix', 'sym_rg_mix', 'sym_rg',
                    'sym_rand_g', 'instance_noise', 'ins_noise', 'rep_gp', 'rep_ds', 'rmb_gp', 'rmb_ds', 'test'}

# the two non-trainable variables of get_mix_coin (math_func.py:2073-2078), 'mmd_g_mix/coin/gen_average' and
# 'mmd_g_mix/coin/prob': created once per device and shared by every GANLoss (tf.AUTO_REUSE), [loss_average, mix_prob]
_MIX_STATE = {}


def mix_state(device):
    import torch
    key = str(device)
    if key not in _MIX_STATE:
        _MIX_STATE[key] = torch.zeros(2, device=device)
    return _MIX_STATE[key]


def get_squared_dist(x, y=None, scale=None, z_score=False, mode='xxxyyy', name='squared_dist',
                     do_summary=False, scope_prefix=''):
    """pairwise squared distances in the reference's Gram form (math_func.py:799-840)."""
    if x.dim() > 2:
        raise AttributeError('get_dist: Input must be a matrix.')
    if scale is not None or z_score:
        raise NotImplementedError('get_squared_dist: scale / z_score are outside the hot path')
    if y is None:
        mode, y = 'xx', x
    if mode not in ('xx', 'xy', 'xxxy', 'xxxyyy'):
        raise AttributeError('Mode {} not supported'.format(mode))
    d = ops.mmd_loss(x.contiguous(), y.contiguous(), 'rep', need_grads=False, need_dist=True)['dist']
    return {'xx': d[0], 'xy': d[1], 'xxxy': (d[0], d[1]), 'xxxyyy': (d[0], d[1], d[2])}[mode]


class GANLoss(object):
    def __init__(self, do_summary=False):
        self.do_summary = do_summary
        self.score_gen = self.score_data = None
        self.batch_size = self.num_scores = None
        self.loss_gen = self.loss_dis = None
        self.dis_penalty = self.dis_scale = None
        self.debug_register = None
        self.repulsive_weights = [0.0, -1.0]          # math_func.py:2115
        self.grads = None                             # [4,B,d]: dLg/dsg, dLg/dsx, dLd/dsg, dLd/dsx
        self.stats = None                             # e_kxx, e_kxy, e_kyy, e_kxx_b, e_kyy_b

    def __call__(self, score_gen, score_data, loss_type='logistic', **kwargs):
        self.score_gen, self.score_data = score_gen, score_data
        for key, attr in (('batch_size', 'batch_size'), ('d', 'num_scores'), ('dis_penalty', 'dis_penalty'),
                          ('dis_scale', 'dis_scale'), ('rep_weights', 'repulsive_weights')):
            if key in kwargs:
                setattr(self, attr, kwargs[key])
        if loss_type in {'fixed_g', 'mmd_g', 'rep', 'rmb'}:                               # math_func.py:2589-2592
            assert self.batch_size is not None, 'GANLoss: batch_size must be provided'   # math_func.py:2592
        if loss_type not in ops.LOSS:
            if loss_type in _NOT_ON_HOT_PATH:
                raise NotImplementedError('Not implemented.')     # outside SURVEY section 8 scope
            raise NotImplementedError('Not implemented.')         # math_func.py:2651
        w = self.repulsive_weights
        if ops.LOSS[loss_type] <= 1:
            assert w[0] - w[1] == 1.0, 'w[0]-w[1] must be 1'      # math_func.py:1340
        if ops.is_mix_loss(loss_type):                            # math_func.py:2613-2622 -> :2195-2263
            import torch
            # the coin: tf.random_uniform([batch_size]) in the reference (:2079); `uni=` injects a recorded draw.  The
            # moving averages are updated in place by the call (the reference's UPDATE_OPS, run with every train step)
            uni = kwargs.get('uni')
            uni = torch.rand(score_gen.shape[0], device=score_gen.device) if uni is None else \
                torch.as_tensor(uni, dtype=torch.float32, device=score_gen.device).contiguous()
            out = ops.mmd_mix_loss(score_gen.contiguous(), score_data.contiguous(), uni, mix_state(score_gen.device),
                                   loss_type, kwargs.get('mix_threshold'), need_grads=True, need_masks=True)
            self.loss_gen, self.loss_dis = out['scalars'][0], out['scalars'][1]
            self.stats, self.grads = out['scalars'][2:7], out['grads']
            self.mix_indices, self.mix_group_1, self.mix_group_2 = (out['masks'][k] for k in
                                                                    ('mix_indices', 'mix_group_1', 'mix_group_2'))
            return self.loss_gen, self.loss_dis
        out = ops.mmd_loss(score_gen.contiguous(), score_data.contiguous(), loss_type, tuple(w), need_grads=True)
        self.loss_gen, self.loss_dis = out['scalars'][0], out['scalars'][1]
        self.stats, self.grads = out['scalars'][2:7], out['grads']
        if self.dis_penalty is not None and loss_type not in ('logistic', '', 'hinge'):   # :2128-2143 take none
            self.loss_dis = self.loss_dis + self.dis_penalty
        return self.loss_gen, self.loss_dis

    def apply(self, score_gen, score_data, loss_type='logistic', **kwargs):
        return self.__call__(score_gen, score_data, loss_type=loss_type, **kwargs)

    def get_register(self):
        registered_info, self.debug_register = self.debug_register, None
        return registered_info


# ------------------------------------------------------------------------------------------------
# evaluation helpers (host NumPy in the reference too)
# ------------------------------------------------------------------------------------------------
def mean_cov_np(x):
    """column means and unbiased covariance of a 2-D array (math_func.py:56-67)."""
    x = np.asarray(x)
    mu = np.mean(x, axis=0)
    centred = x - mu
    return mu, centred.T.dot(centred) / (x.shape[0] - 1.0)


def sqrt_sym_mat_np(mat, eps=None):
    """square root of a symmetric matrix as the reference defines it (math_func.py:2671-2683): U sqrt(S) V^T of
    the SVD with singular values below eps dropped.  For a symmetric matrix that is sum_i sign(l_i) sqrt|l_i| v_i v_i^T
    over its eigenpairs, which one symmetric eigendecomposition gives at a third of the SVD's cost."""
    if eps is None:
        eps = FLAGS.EPSI
    lam, vec = np.linalg.eigh((mat + mat.T) * 0.5)
    root = np.where(np.abs(lam) < eps, 0.0, np.sign(lam) * np.sqrt(np.abs(lam)))
    return (vec * root).dot(vec.T)


def trace_sqrt_product_np(cov1, cov2):
    """trace(sqrt(cov1 cov2)) through sqrt(cov1) cov2 sqrt(cov1) (math_func.py:2686-2699)."""
    root1 = sqrt_sym_mat_np(cov1)
    return np.trace(sqrt_sym_mat_np(root1.dot(cov2).dot(root1)))


class MeshCode(object):
    """codes for a grid of samples (math_func.py:220-340), as NumPy float32 arrays: 'random' N(0,1) draws, 'sine' a
    two-angle interpolation between four support codes, 'feature' one latent feature varied per row, and the plain
    2-D grid.  Row j * mesh_num[0] + i of the 'sine' batch is
        cos(phi_i) (cos(psi_j) z0 + sin(psi_j) z1) + sin(phi_i) (cos(psi_j) z2 + sin(psi_j) z3),
    phi_i over mesh_num[0] and psi_j over mesh_num[1] points of [0, pi/4]."""

    def __init__(self, code_length, mesh_num=None):
        self.D = code_length
        self.mesh_num = (10, 10) if mesh_num is None else tuple(mesh_num)

    def get_batch(self, mesh_mode, name=None):
        if mesh_mode in (0, 'random'):
            return self.by_random(name)
        if mesh_mode in (1, 'sine'):
            return self.by_sine(name=name)
        if mesh_mode in (2, 'feature'):
            return self.by_feature(name=name)
        raise AttributeError('mesh_mode is not supported.')                    # math_func.py:244

    def by_random(self, name=None):
        return np.random.randn(self.mesh_num[0] * self.mesh_num[1], self.D).astype(np.float32)

    def by_sine(self, z_support=None, name=None):
        z = np.random.randn(4, self.D) if z_support is None else np.asarray(z_support)
        z = z.astype(np.float32)
        phi = np.float32(np.pi / 4.0 * np.linspace(0.0, 1.0, self.mesh_num[0]))
        psi = np.float32(np.pi / 4.0 * np.linspace(0.0, 1.0, self.mesh_num[1]))
        low = np.cos(psi)[:, None] * z[0] + np.sin(psi)[:, None] * z[1]        # [mesh_num[1], D]
        high = np.cos(psi)[:, None] * z[2] + np.sin(psi)[:, None] * z[3]
        batch = np.cos(phi)[None, :, None] * low[:, None, :] + np.sin(phi)[None, :, None] * high[:, None, :]
        return batch.reshape(self.mesh_num[0] * self.mesh_num[1], self.D).astype(np.float32)

    def by_feature(self, grid=2.0, name=None):
        """feature f (a random one of the D, distinct per row group) takes mesh_num[1] values in [-grid, grid], all
        other features 0; the reference shuffles the columns with tf.random_shuffle"""
        values = np.float32(np.linspace(-grid, grid, self.mesh_num[1]))
        batch = np.zeros((self.mesh_num[0], self.mesh_num[1], self.D), np.float32)
        for f in range(min(self.mesh_num[0], self.D)):
            batch[f, :, f] = values
        batch = batch.reshape(-1, self.D)
        return batch[:, np.random.permutation(self.D)]

    def simple_grid(self, grid=None):
        if self.D != 2:
            raise AttributeError('Code length has to be two')                  # math_func.py:323
        if grid is None:
            grid = np.array([[-1.0, 1.0], [-1.0, 1.0]], dtype=np.float32)
        x = np.linspace(grid[0][0], grid[0][1], self.mesh_num[0])
        y = np.linspace(grid[1][0], grid[1][1], self.mesh_num[1])
        z = np.stack([np.repeat(x, self.mesh_num[1]), np.tile(y, self.mesh_num[0])], axis=1)
        return z, x, y
