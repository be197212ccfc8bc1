import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mmd-gan_amd'), os.path.join(ROOT, 'tests')]
from oracle import restatement as R
from test_step_gpu import mid_architecture, nhwc
from mmdgan_hip.engine import GanEngine
arch, B = mid_architecture(), 16
for loss_type in ['rep']:
    eng = GanEngine(arch, loss_type, (5e-4, 2e-4), batch_size=B, seed=3)
    ora = R.OracleGan(arch, loss_type, (5e-4, 2e-4), dtype=torch.float64, params=eng.get_variables())
    ora32 = R.OracleGan(arch, loss_type, (5e-4, 2e-4), dtype=torch.float32, params=eng.get_variables())
    rs = np.random.RandomState(42)
    for step in range(3):
        z = rs.randn(B, 64).astype(np.float32); real = rs.uniform(-1, 1, (B, 3, 32, 32)).astype(np.float32)
        eng.set_variables({k: v.numpy() for k, v in ora.params.items()})
        for k, v in ora.params.items(): ora32.params[k] = v.float()
        zt, rt = torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64)
        lg, ld, stats, upd, gd, gg, aux = ora.grads(zt, rt)
        _, _, _, _, gd32, gg32, _ = ora32.grads(zt.float(), rt.float())
        ora.step(zt, rt)
        eng.step(nhwc(real), torch.as_tensor(z).cuda())
        grads = eng.get_variables(grad=True)
        ref = dict(gd); ref.update(gg); ref32 = dict(gd32); ref32.update(gg32)
        print('step', step, 'losses', eng.losses[:2].cpu().numpy(), float(lg), float(ld))
        for n in grads:
            r = ref[n].numpy(); e = np.abs(grads[n] - r).max() / (np.abs(r).max() + 1e-30)
            e32 = np.abs(ref32[n].numpy() - r).max() / (np.abs(r).max() + 1e-30)
            print('  %-28s max|ref| %.3e  hip relerr %.2e   torch-fp32 relerr %.2e' % (n, np.abs(r).max(), e, e32))
