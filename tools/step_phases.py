#!/usr/bin/env python3
"""where a training step's time goes, UN-profiled: wall-clock stamps (tools/clock_probe.hip) on the main stream at the
phase boundaries of GanEngine._step_body, eager issue, averaged over steps.  (A stamp on the main stream waits for what
the main stream waits for: the phases' ends include the joins with the side streams.)
    tools/step_phases.py [config]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
so = os.path.join(ROOT, 'tools', 'libclockprobe.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC',
                           os.path.join(ROOT, 'tools', 'clock_probe.hip'), '-o', so])
lib = ctypes.CDLL(so)
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
import configs  # noqa: E402
from mmdgan_hip.engine import GanEngine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
arch, lr = configs.CONFIGS[cfg]()
B = {'celeba': 128}.get(cfg, 64)
eng = GanEngine(arch, 'rep', lr, batch_size=B, seed=0, launch_mode='eager')
c, h, w = arch['input'][0]
real = torch.empty(B, h, w, c, device='cuda').uniform_(-1, 1)
NAMES = ['start', 'G forward', 'D forward + loss', 'D backward', 'G backward', 'update']
K = 40
stamps = torch.zeros(K, len(NAMES), 2, dtype=torch.int64, device='cuda')
state = {'k': -1, 'i': 0}


def stamp():
    if state['k'] >= 0:
        lib.clock_probe(stamps[state['k'], state['i']].data_ptr(), torch.cuda.current_stream().cuda_stream)
    state['i'] += 1


orig_gen, orig_fwd, orig_bd, orig_bg, orig_up = eng.generate, eng._forward, eng._backward_dis, eng._backward_gen, eng._update


def generate(z, is_training=False):
    out = orig_gen(z, is_training)
    if is_training:
        stamp()                                   # after G forward
    return out


def forward(z, real_):
    stamp()                                       # start (after the step's zeroing launches were issued)
    out = orig_fwd(z, real_)
    stamp()                                       # after D forward + loss
    return out


def bdis():
    out = orig_bd()
    stamp()
    return out


def bgen(dz, z):
    out = orig_bg(dz, z)
    eng._join_wg_stream()
    stamp()
    return out


def update():
    orig_up()
    stamp()


eng.generate, eng._forward, eng._backward_dis, eng._backward_gen, eng._update = generate, forward, bdis, bgen, update
for _ in range(20):
    state['i'] = 0
    eng.step(real)
for k in range(K):
    state['k'], state['i'] = k, 0
    eng.step(real)
torch.cuda.synchronize()
s = stamps.cpu().numpy()[:, :, 1].astype(np.float64) / 100.0        # wall clock: 100 MHz -> us
d = np.diff(s, axis=1)
step = np.diff(s[:, 0])
print('%s B=%d, eager issue, %d steps: step %.1f us (start to start)' % (cfg, B, K, step.mean()))
for i, n in enumerate(NAMES[1:]):
    print('  %-18s %8.1f us' % (n, d[:, i].mean()))
print('  %-18s %8.1f us' % ('(between steps)', step.mean() - d.sum(axis=1)[:-1].mean()))
