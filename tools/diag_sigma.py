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
for step in range(2):
    z = rs.randn(B, 64).astype(np.float32); real = rs.uniform(-1, 1, (B, 3, 32, 32)).astype(np.float32)
    eng.set_variables({k: v.numpy() for k, v in ora.params.items()})
    col = {}
    ora.forward_losses(torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64), col)
    ora.step(torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64))
    eng.step(nhwc(real), torch.as_tensor(z).cuda())
    sg = eng.sigmas()
    for k, v in sg.items():
        print(step, k, v, float(col[k + '/sigma']))
    for sc in ['dis/l1_f32', 'dis/l2_ds', 'dis/l3', 'dis/l4_ds', 'dis/l5_s']:
        y = eng.buf[sc + '#y'].cpu().numpy(); r = col[sc + '/out'].numpy()
        if r.ndim == 4: r = r.transpose(0, 2, 3, 1)
        print('   ', sc, np.abs(y - r).max() / np.abs(r).max())
