import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), ROOT]
os.environ.setdefault('MMDGAN_WINO_MIN_TILES', '32'); os.environ.setdefault('MMDGAN_WINO2', '2')
from test_res_gpu import mid_res_architecture
from mmdgan_hip.tape import TapeEngine
loss, sn = sys.argv[1], sys.argv[2]
arch, B = mid_res_architecture(), 16
engs = {}
for fold in ('0', '1'):
    os.environ['MMDGAN_TAPE_COMPOSE'] = fold
    engs[fold] = TapeEngine(arch, loss, (5e-4, 2e-4), batch_size=B, seed=3, sn_mode=sn)
engs['1'].set_variables(engs['0'].get_variables())
from oracle import restatement as R
ora = R.OracleGan(arch, loss, (5e-4, 2e-4), dtype=torch.float64, params=engs['0'].get_variables(), sn_mode=sn)
rs = np.random.RandomState(42)
for step in range(3):
    z = torch.as_tensor(rs.randn(B, 64).astype(np.float32)).cuda()
    real = torch.as_tensor(np.ascontiguousarray(rs.uniform(-1, 1, (B, 3, 32, 32)).astype(np.float32).transpose(0, 2, 3, 1))).cuda()
    pv = {k: v.numpy().copy() for k, v in ora.params.items()}
    for e in engs.values(): e.set_variables(pv)
    zt, rt = z.cpu().double(), real.cpu().permute(0, 3, 1, 2).double()
    res = ora.grads(zt, rt); ora.step(zt, rt)
    rg = dict(res[4]); rg.update(res[5])
    for e in engs.values(): e.step(real, z)
    g0, g1 = engs['0'].get_variables(grad=True), engs['1'].get_variables(grad=True)
    worst = sorted(((np.linalg.norm(g1[n] - g0[n]) / (np.linalg.norm(g0[n]) + 1e-12), n) for n in g0), reverse=True)[:6]
    for tag in ('0', '1'):
        g = engs[tag].get_variables(grad=True)
        w = sorted(((np.linalg.norm(g[n] - rg[n].numpy()) / (np.linalg.norm(rg[n].numpy()) + 1e-12), n) for n in g if float(rg[n].abs().max()) > 1e-7), reverse=True)[:3]
        print('   fold', tag, 'vs oracle:', '  '.join('%s %.1e' % (n, v) for v, n in w))
    if step == 2:
        for tag in ('0', '1'):
            g = engs[tag].get_variables(grad=True)['gen/l1/kernel/kernel'].astype(np.float64); r = rg['gen/l1/kernel/kernel'].numpy()
            e = np.abs(g - r); print('   fold', tag, 'gen/l1 grad: l2 %.2e, max err %.2e (max |ref| %.2e), entries with err > 1e-3 max: %d of %d, corr slope %.6f' % (np.linalg.norm(g - r) / np.linalg.norm(r), e.max(), np.abs(r).max(), int((e > 1e-3 * np.abs(r).max()).sum()), e.size, float((g * r).sum() / (r * r).sum())))
    print('step', step, 'loss', engs['0'].losses[:2].tolist(), engs['1'].losses[:2].tolist())
    print('   ' + '  '.join('%s %.1e' % (n, v) for v, n in worst))
    s0, s1 = engs['0'].sigmas(), engs['1'].sigmas()
    print('   sigma max rel diff', max(abs(s0[k] - s1[k]) / s0[k] for k in s0))
