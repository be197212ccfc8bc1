"""run the tiny 3-step fixture scenario repeatedly: how much do gradients / final variables vary run to run?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), os.path.join(ROOT, 'oracle'), ROOT]
os.environ.setdefault('MMDGAN_WINO_MIN_TILES', '32'); os.environ.setdefault('MMDGAN_WINO2', '2')
from mmdgan_hip.engine import GanEngine
from tiny_arch import tiny_architecture
tag = sys.argv[1] if len(sys.argv) > 1 else 'rep_pim'
fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'step_tiny_%s.npz' % tag))
sn_mode = str(fx['sn_mode']) if 'sn_mode' in fx.files else 'default'
B = int(fx['B'])
init = {k[5:]: fx[k] for k in fx.files if k.startswith('init/')}
base = None
for run in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    eng = GanEngine(tiny_architecture(), str(fx['loss_type']), tuple(fx['lr']), batch_size=B, sn_mode=sn_mode)
    eng.set_variables(init)
    rec = {}
    for step in range(3):
        real = torch.as_tensor(np.ascontiguousarray(fx['real'][step].transpose(0, 2, 3, 1))).cuda()
        eng.step(real, torch.as_tensor(fx['z'][step]).cuda())
        for n, g in eng.get_variables(grad=True).items():
            rec['g%d/%s' % (step, n)] = g
        rec['loss%d' % step] = eng.losses.cpu().numpy().copy()
        for n, v in eng.sigmas().items():
            rec['s%d/%s' % (step, n)] = np.float64(v)
    for n, v in eng.get_variables().items():
        rec['final/' + n] = v
    if base is None:
        base = rec
        continue
    worst = sorted(((float(np.max(np.abs(np.asarray(rec[k], np.float64) - np.asarray(base[k], np.float64)))) /
                     (float(np.max(np.abs(np.asarray(base[k], np.float64)))) + 1e-30), k) for k in rec
                   if not k.startswith('g0/') and 'l8_s/bias' not in k), reverse=True)[:4]
    print('run %2d: ' % run + '  '.join('%s %.1e' % (k, v) for v, k in worst), flush=True)
