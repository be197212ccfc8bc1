import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mmd-gan_amd'), os.path.join(ROOT, 'tests')]
from oracle import restatement as R
from test_step_gpu import mid_architecture, nhwc
from mmdgan_hip.engine import GanEngine
arch, B = mid_architecture(), 16
eng = GanEngine(arch, 'rep', (5e-4, 2e-4), batch_size=B, seed=3)
ora = R.OracleGan(arch, 'rep', (5e-4, 2e-4), dtype=torch.float64, params=eng.get_variables())
rs = np.random.RandomState(42)
def rel(a, b): return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
for step in range(3):
    z = rs.randn(B, 64).astype(np.float32); real = rs.uniform(-1, 1, (B, 3, 32, 32)).astype(np.float32)
    eng.set_variables({k: v.numpy() for k, v in ora.params.items()})
    zt, rt = torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64)
    if step == 2:
        col = {}
        lg, ld, stats, upd, aux = ora.forward_losses(zt.clone().requires_grad_(True), rt, col)
        names = [k for k in col if k.endswith('out_live')]
        g = torch.autograd.grad(lg, [col[k] for k in names], retain_graph=True)
        gl = dict(zip(names, g))
    ora.step(zt, rt)
    eng.step(nhwc(real), torch.as_tensor(z).cuda())
    if step == 2:
        b = eng.buf
        for sc in ['gen/l1', 'gen/l2_up', 'gen/l3_up', 'gen/l4_up']:
            ref = gl[sc + '/out_live'].numpy()
            got = b[sc + '#dy'].cpu().numpy()
            if ref.ndim == 4: ref = np.transpose(ref, (0, 2, 3, 1))
            else:
                got = got.reshape(B, 4, 4, 128).transpose(0, 3, 1, 2).reshape(B, -1)
            print(sc, 'dL/dy   rel', rel(got.reshape(ref.shape), ref), 'max', np.abs(ref).max())
            y_ref = col[sc + '/out'].numpy()
            y_got = b[sc + '#y'].cpu().numpy()
            if y_ref.ndim == 4: y_ref = np.transpose(y_ref, (0, 2, 3, 1))
            else: y_got = y_got.reshape(B, 4, 4, 128).transpose(0, 3, 1, 2).reshape(B, -1)
            print(sc, 'y       rel', rel(y_got.reshape(y_ref.shape), y_ref), 'mask mismatches', int(((y_got.reshape(y_ref.shape) > 0) != (y_ref > 0)).sum()), 'of', y_ref.size)
        # d/d(fake image) vs oracle
        ref = np.transpose(gl['gen/l5_t32/out_live'].numpy(), (0, 2, 3, 1))
        # engine d_fake is d/d(pre-tanh) = dL/dy * (1-y^2)
        y = np.transpose(col['gen/l5_t32/out'].numpy(), (0, 2, 3, 1))
        print('d_fake rel', rel(b['d_fake'].cpu().numpy(), ref * (1 - y * y)))
        for sc in ['dis/l1_f32', 'dis/l2_ds', 'dis/l3', 'dis/l4_ds']:
            refl = gl[sc + '/out_live'].numpy()[B:]            # fake half, d loss_gen / d y
            yy = col[sc + '/out'].numpy()[B:]
            refz = np.transpose(refl * np.where(yy > 0, 1.0, 0.1), (0, 2, 3, 1))
            print(sc, 'dz_g rel', rel(b[sc + '#dz_g'].cpu().numpy(), refz))
