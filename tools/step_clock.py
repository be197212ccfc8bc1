#!/usr/bin/env python3
"""average SHADER CLOCK over training steps (clock64 / wall_clock64 stamps on the main stream, tools/clock_probe.hip): tells a
power-limited step (the part clocks down under two MFMA streams) from a latency-limited one.
    tools/step_clock.py [config]          env as bench.py (MMDGAN_SIDE_WGRAD=0: everything on one stream)"""
import ctypes
import os
import subprocess
import sys

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
eng = GanEngine(arch, 'rep', lr, batch_size=B, seed=0, launch_mode=os.environ.get('MMDGAN_LAUNCH_MODE', 'plan'))
c, h, w = arch['input'][0]
real = torch.empty(B, h, w, c, device='cuda').uniform_(-1, 1)
for _ in range(30):
    eng.step(real)
torch.cuda.synchronize()
stamps = torch.zeros(2, 2, dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
K = 50
lib.clock_probe(stamps[0].data_ptr(), st)
for _ in range(K):
    eng.step(real)
lib.clock_probe(stamps[1].data_ptr(), st)
torch.cuda.synchronize()
s = stamps.cpu().numpy()
dclk, dwall = int(s[1, 0] - s[0, 0]), int(s[1, 1] - s[0, 1])
print('%s B=%d: %d steps in %.3f ms (%.4f ms/step), average shader clock %.0f MHz' % (cfg, B, K, dwall / 1e5, dwall / 1e5 / K, dclk / dwall * 100.0))
