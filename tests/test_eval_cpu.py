"""Eval helpers: NumPy FID on supplied pool3 features and the sprite mosaic, oracle and product
against fixtures recorded from the reference's own functions (oracle/make_golden.py:make_eval).  Host code in the
reference too - CPU only."""
import numpy as np
import pytest

from helpers import golden, load
from oracle import restatement as R
from GeneralTools import graph_func as G
from GeneralTools import math_func as M

FID = load(golden('eval_fid.npz')[0])
SPRITE = load(golden('eval_sprite.npz')[0])
CASES = ('rgb_auto', 'rgb_mesh', 'rgb_invert', 'gray3d', 'gray4d')


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def test_fid_oracle_matches_reference():
    assert rel(R.fid_from_pool3(FID['x'], FID['y']), float(FID['fid'])) <= 1e-12
    assert rel(R.fid_from_pool3([FID['mu_x'], FID['cov_x']], FID['y']), float(FID['fid_from_stats'])) <= 1e-12
    assert rel(R.fid_from_pool3(FID['z'], FID['y']), float(FID['fid_rank_deficient'])) <= 1e-12
    assert np.allclose(R.sqrt_sym_mat(FID['cov_x']), FID['sqrt_cov_x'], rtol=0, atol=1e-13)


def test_fid_product_matches_reference():
    mu, cov = M.mean_cov_np(FID['x'])
    assert np.allclose(mu, FID['mu_x'], rtol=0, atol=1e-14) and np.allclose(cov, FID['cov_x'], rtol=0, atol=1e-13)
    # a symmetric eigendecomposition instead of the reference's SVD: same matrix to rounding
    assert np.allclose(M.sqrt_sym_mat_np(FID['cov_x']), FID['sqrt_cov_x'], rtol=0, atol=1e-11)
    assert rel(M.trace_sqrt_product_np(FID['cov_x'], FID['cov_y']), float(FID['trace_sqrt_product'])) <= 1e-10
    fid = G.GenerativeModelMetric.my_fid_from_pool3
    assert rel(fid(FID['x'], FID['y']), float(FID['fid'])) <= 1e-9
    assert rel(fid([FID['mu_x'], FID['cov_x']], FID['y']), float(FID['fid_from_stats'])) <= 1e-9
    assert rel(fid(FID['x'], (FID['mu_y'], FID['cov_y'])), float(FID['fid'])) <= 1e-9
    assert rel(fid(FID['z'], FID['y']), float(FID['fid_rank_deficient'])) <= 1e-9        # 20 samples, 48 features
    assert abs(fid(FID['x'], FID['x'])) <= 1e-8                                           # the reference: -5.6e-11
    # properties at a size no fixture covers: symmetry, and the closed form for isotropic Gaussians' statistics
    rs = np.random.RandomState(1)
    a, b = rs.randn(4000, 256), rs.randn(3000, 256) * 1.5 + 0.25
    assert rel(fid(a, b), fid(b, a)) <= 1e-9
    d = 256
    assert rel(fid([np.zeros(d), np.eye(d)], [np.full(d, 0.25), 2.25 * np.eye(d)]), d * 0.0625 + d * (1 + 2.25 - 3.0)) <= 1e-12


def test_inception_paths_raise():
    m = G.GenerativeModelMetric()
    with pytest.raises(NotImplementedError, match='Inception'):
        m.inception_score_and_fid_v1(None, None)


@pytest.mark.parametrize('case', CASES)
def test_sprite_mosaic_is_bit_exact(case, tmp_path):
    mesh = SPRITE[case + '/mesh']
    mesh = None if mesh[0] < 0 else tuple(int(v) for v in mesh)
    ref = SPRITE[case + '/sprite']
    for fn in (R.sprite_grid, G.sprite_array):
        got = fn(SPRITE[case + '/images'], mesh, bool(SPRITE[case + '/invert']))
        assert got.dtype == np.uint8 and np.array_equal(got, ref), fn.__name__
    from PIL import Image
    path = str(tmp_path / 's.png')
    G.write_sprite(path, SPRITE[case + '/images'], mesh if case != 'rgb_invert' else list(mesh), bool(SPRITE[case + '/invert']))
    assert np.array_equal(np.asarray(Image.open(path)), ref)


def test_sprite_wrapper_names_and_never_overwrites(tmp_path):
    imgs = SPRITE['rgb_mesh/images'].transpose(0, 3, 1, 2)                                # NCHW in, as eval_sampling has it
    path = G.write_sprite_wrapper(imgs, [2, 5], ['cifar', 'x'], file_folder=str(tmp_path), file_index='_g_t_7_0',
                                  image_format='channels_first')
    assert path.endswith('cifar_g_t_7_0.png')
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path)), SPRITE['rgb_mesh/sprite'])
    with pytest.warns(UserWarning, match='already exists'):
        G.write_sprite_wrapper(imgs * 0, [2, 5], 'cifar', file_folder=str(tmp_path), file_index='_g_t_7_0',
                               image_format='channels_first')
    assert np.array_equal(np.asarray(Image.open(path)), SPRITE['rgb_mesh/sprite'])


def test_meshcode_matches_reference():
    """MeshCode.by_sine / simple_grid (math_func.py:220-340) against the reference's outputs; the random modes by shape"""
    mc = load(golden('eval_meshcode.npz')[0])
    code = M.MeshCode(5, mesh_num=(3, 4))
    got = code.by_sine(z_support=mc['support'])
    assert got.dtype == np.float32 and np.allclose(got, mc['sine_f64'], rtol=0, atol=3e-7)
    assert np.abs(got - mc['sine_f32']).max() <= 3e-7
    z, x, y = M.MeshCode(2, mesh_num=(3, 4)).simple_grid()
    assert np.array_equal(z, mc['grid_z']) and np.array_equal(x, mc['grid_x']) and np.array_equal(y, mc['grid_y'])
    z2, _, _ = M.MeshCode(2, mesh_num=(2, 5)).simple_grid(np.array([[-2.0, 0.5], [1.0, 3.0]], dtype=np.float32))
    assert np.array_equal(z2, mc['grid2_z'])
    assert code.get_batch(0).shape == (12, 5) and code.get_batch('sine').shape == (12, 5)
    f = M.MeshCode(6, mesh_num=(3, 4)).get_batch(2)
    assert f.shape == (12, 6) and np.all((f != 0).sum(axis=1) <= 1) and np.isclose(np.abs(f).max(), 2.0)
    with pytest.raises(AttributeError, match='mesh_mode is not supported'):
        code.get_batch(7)
    with pytest.raises(AttributeError, match='Code length has to be two'):
        code.simple_grid()
