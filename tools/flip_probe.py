#!/usr/bin/env python3
"""How much of a gradient's fp32-vs-fp64 difference is activation-mask flips?  CIFAR dict at batch 64, second
teacher-forced step: per D layer, the number of lrelu outputs whose SIGN differs between the engine (fp32) and the fp64
oracle, the relative error of the activations, and the L2 error of the layer's kernel / bias gradient.
    python tools/flip_probe.py [config] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs  # noqa: E402
from mmdgan_hip.engine import GanEngine  # noqa: E402
from oracle import restatement as R  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
arch, lr = configs.CONFIGS[config]()
c, h, w = arch['input'][0]
eng = GanEngine(arch, 'rep', tuple(lr), batch_size=B, seed=5)
ora = R.OracleGan(arch, 'rep', tuple(lr), dtype=torch.float64, params=eng.get_variables())
rs = np.random.RandomState(7)
for step in range(2):
    z = rs.randn(B, arch['code'][0][0]).astype(np.float32)
    real = rs.uniform(-1, 1, (B, c, h, w)).astype(np.float32)
    eng.set_variables({k: v.numpy().copy() for k, v in ora.params.items()})
    zt, rt = torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64)
    col = {}
    lg, ld, stats, upd, gd, gg, aux = ora.grads(zt, rt, collect=col)
    ora.step(zt, rt)
    eng.step(torch.as_tensor(np.ascontiguousarray(real.transpose(0, 2, 3, 1))).cuda(), torch.as_tensor(z).cuda())
grads = eng.get_variables(grad=True)
print('%-12s %10s %10s %12s %12s %12s' % ('layer', 'elements', 'sign flips', 'act rel err', 'dW l2 err', 'db l2 err'))
for s in eng.dis.specs:
    y = eng.buf[s.scope + '#y'].cpu().numpy()
    ref = col[s.scope + '/out'].numpy()
    if ref.ndim == 4:
        ref = ref.transpose(0, 2, 3, 1)
    y = y.reshape(ref.shape)
    flips = int(((y > 0) != (ref > 0)).sum())
    rel = np.abs(y - ref).max() / np.abs(ref).max()
    out = []
    for nm in ('/kernel/kernel', '/bias/bias'):
        r = gd[s.scope + nm].numpy()
        out.append(np.linalg.norm(grads[s.scope + nm].astype(np.float64) - r) / np.linalg.norm(r))
    print('%-12s %10d %10d %12.3g %12.3g %12.3g' % (s.scope, y.size, flips, rel, out[0], out[1]))
