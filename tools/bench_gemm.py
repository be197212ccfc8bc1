#!/usr/bin/env python3
"""the dense products of the CIFAR step alone (HIP events, 200 launches each): G l1 forward, its weight gradient, the weight
gradient and input gradient of D's head.  MMDGAN_GEMM_PANEL=0 times the tiled kernel on the same calls."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
ops.require_device()
B = 64
cases = [('G l1 fwd  [64,128]x[128,8192]', (B, 128), (128, 8192), False, False, True),
         ('G l1 wgrad z^T[128,64]x[64,8192]', (B, 128), (B, 8192), True, False, False),
         ('D l8 wgrad x^T[8192,128]x[128,16]', (2 * B, 8192), (2 * B, 16), True, False, False),
         ('D l8 dgrad [192,16]x[16,8192]^T', (3 * B, 16), (8192, 16), False, True, False)]
for name, sa, sb, ta, tb, use_bias in cases:
    a, b = torch.randn(sa, device='cuda'), torch.randn(sb, device='cuda')
    M = sa[1] if ta else sa[0]
    N = sb[0] if tb else sb[1]
    out = torch.empty(M, N, device='cuda')
    bias = torch.randn(N, device='cuda') if use_bias else None
    for _ in range(20):
        ops.gemm(a, b, ta, tb, bias=bias, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.gemm(a, b, ta, tb, bias=bias, out=out)
    e1.record(); torch.cuda.synchronize()
    print('%-40s %.2f us per launch (back to back)' % (name, e0.elapsed_time(e1) / 200 * 1e3))
