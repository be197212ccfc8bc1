#!/usr/bin/env python3
"""the weight gradients of D's six conv layers (CIFAR, batch 2B = 128) as the backward pass issues them - l7, l6, ... l2 on one
stream - timed per chain: every launch followed by its own slab-reduction pass (mmdgan_wgrad_defer off), and as a chain in
which each launch sums its predecessor's slabs in its prologue (on; one flush at the end).  HIP events, 50 chains."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
ops.require_device()
ops.set_workspace(256 << 20)
N = 128
layers = [('l7', 4, 512, 512, 3, 1), ('l6', 8, 256, 512, 4, 2), ('l5', 8, 256, 256, 3, 1), ('l4', 16, 128, 256, 4, 2),
          ('l3', 16, 128, 128, 3, 1), ('l2', 32, 64, 128, 4, 2)]
data = []
for name, H, C, K, R, s in layers:
    P = H // s
    data.append((torch.randn(N, H, H, C, device='cuda'), torch.randn(N, P, P, K, device='cuda'), torch.empty(R, R, C, K, device='cuda'),
                 torch.empty(K, device='cuda'), R, s))


def chain(defer):
    ops.wgrad_defer(defer)
    for x, dy, dw, db, R, s in data:
        ops.conv2d_wgrad(x, dy, R, s, out=dw, dbias=db)
    ops.wgrad_flush()
    ops.wgrad_defer(False)


for defer in (False, True, False, True):
    for _ in range(5):
        chain(defer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        chain(defer)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print('defer %-5s: %.1f us per chain of six = %.1f us per layer' % (defer, t, t / 6))
