#!/usr/bin/env python3
"""What the first multi-GPU run should show (VERDICT r04 "next" #9): the data-parallel step with ONE rank - every stream
dependency, bucket boundary and flush of the N-rank step, the collectives themselves no-ops - timed per exchange bucket:
when (after the step's start) each bucket is ready on the exchange stream, and when the main stream has nothing left but the
updates.  On top of that an all-reduce cost model gives the part of the exchange nothing hides at N = 2 / 4 / 8:
    T(S, N) = alpha(N) + 2 (N - 1) / N * S / BW      alpha(N) = 12 us + 2 (N - 1) * 1.5 us (launch + ring hops)
    BW = 153 GB/s * 0.8 (one xGMI link: RCCL's ring is bound by it) - "ring"
    BW = (N - 1) * 153 GB/s * 0.6 (direct reduce-scatter + all-gather over all links of the fully connected node) - "all links"
Buckets are served in order on the exchange stream; D's Adam (its duration measured here as the gap it leaves) follows D's
last bucket there.  exposed = (end of the last bucket) - (main stream ready), floored at 0; predicted step = the single-process
step + the one-rank plumbing measured here + exposed.
    MMDGAN_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 \\
        tools/scale_predict.py [cifar|celeba|stl]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs  # noqa: E402
from mmdgan_hip import dist as mdist  # noqa: E402
from mmdgan_hip.engine import GanEngine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
arch, lr = configs.CONFIGS[cfg]()
B = {'celeba': 128}.get(cfg, 64)
loss = 'rmb' if cfg == 'stl' else 'rep'
torch.cuda.set_device(0)
group = mdist.init_process_group(0)
c, h, w = arch['input'][0]
real = torch.empty(B, h, w, c, device='cuda').uniform_(-1, 1)


def run(eng, n):
    for _ in range(n):
        eng.step(real)
    torch.cuda.synchronize()


def timed(eng, n=60):
    run(eng, 10)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    run(eng, n)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


single = GanEngine(arch, loss, tuple(lr), batch_size=B, seed=0, launch_mode='plan')
t_single = timed(single)
del single
eng = GanEngine(arch, loss, tuple(lr), batch_size=B, seed=0, launch_mode='plan', dist_group=group)
t_dp1 = timed(eng)
eng.launch_mode = 'eager'
run(eng, 5)
rows = []
for _ in range(30):
    eng.bucket_probe = []
    eng.step(real)
    torch.cuda.synchronize()
    pr = eng.bucket_probe
    start = [e for n, b, e in pr if n == 'step_start'][0]
    rows.append([(n, b, start.elapsed_time(e) * 1e3) for n, b, e in pr if n != 'step_start'])
eng.bucket_probe = None
names = [(n, b) for n, b, _ in rows[0]]
t = np.median(np.array([[r[2] for r in row] for row in rows]), axis=0)
print('# %s B=%d per GPU: single-process step %.1f us, one-rank data-parallel step %.1f us (plan replay; plumbing %.1f us)'
      % (cfg, B, t_single, t_dp1, t_dp1 - t_single))
print('# eager one-rank step, medians of 30: time after the step\'s start at which ...')
for (n, b), ti in zip(names, t):
    print('#   %-12s %8.2f MB  ready at %8.1f us' % (n, b / 1e6, ti))
buckets = [(n, b, ti) for (n, b), ti in zip(names, t) if n != 'main_ready']
main_ready = [ti for (n, b), ti in zip(names, t) if n == 'main_ready'][0]
d_adam = 36.0 if cfg == 'cifar' else 36.0 * sum(b for n, b, _ in buckets if n == 'dis') / 24e6     # (bandwidth-bound: scales with D's size)
print('# main stream ready for the updates at %.1f us; D\'s Adam on the exchange stream taken as %.0f us' % (main_ready, d_adam))
print('%-10s %3s %12s %12s %12s %10s' % ('model', 'N', 'exchange us', 'exposed us', 'step us', 'x 1 GPU'))
for model in ('ring', 'all links'):
    for N in (2, 4, 8):
        alpha = 12.0 + 2 * (N - 1) * 1.5
        bw = 153e9 * 0.8 if model == 'ring' else (N - 1) * 153e9 * 0.6
        fin, total = 0.0, 0.0
        last_net = None
        for n, b, ready in buckets:
            if last_net == 'dis' and n != 'dis':
                fin += d_adam
            T = alpha + 2.0 * (N - 1) / N * b / bw * 1e6
            fin = max(fin, ready) + T
            total += T
            last_net = n
        exposed = max(0.0, fin - main_ready)
        step = t_dp1 + exposed
        print('%-10s %3d %12.1f %12.1f %12.1f %10.2f' % (model, N, total, exposed, step, N * t_single / step))
import torch.distributed as tdist
tdist.destroy_process_group()
