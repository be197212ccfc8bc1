#!/usr/bin/env python3
"""the F(2x2,2x2) forward / input-gradient kernel alone (weights transformed beforehand, as the engine issues it):
correctness against torch's convolution on the GPU (a quick checker for kernel work - the parity tests are in tests/)
and HIP-event time per launch.  usage: tools/wino2_bench.py [B]      env: MMDGAN_WINO2=0 times the direct kernels instead"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
os.environ.setdefault('MMDGAN_WINO2', '2')
from mmdgan_hip import ops  # noqa: E402

ops.require_device()
ops.set_workspace()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CASES3 = [('D l3 fwd', 2 * B, 16, 128, 128, False), ('D l3 dgrad 3B', 3 * B, 16, 128, 128, True),
          ('D l5 fwd', 2 * B, 8, 256, 256, False), ('D l5 dgrad 3B', 3 * B, 8, 256, 256, True),
          ('D l7 fwd', 2 * B, 4, 512, 512, False), ('D l7 dgrad 3B', 3 * B, 4, 512, 512, True)]
CASES = [('D l2 fwd', 2 * B, 32, 64, 128, False), ('D l2 dgrad 3B', 3 * B, 32, 64, 128, True),
         ('D l4 fwd', 2 * B, 16, 128, 256, False), ('D l4 dgrad 3B', 3 * B, 16, 128, 256, True),
         ('D l6 fwd', 2 * B, 8, 256, 512, False), ('D l6 dgrad 3B', 3 * B, 8, 256, 512, True),
         ('G l4 tc fwd', B, 32, 64, 128, True), ('G l3 tc fwd', B, 16, 128, 256, True), ('G l2 tc fwd', B, 8, 256, 512, True),
         ('G l4 tc dgrad', B, 32, 64, 128, False), ('G l3 tc dgrad', B, 16, 128, 256, False), ('G l2 tc dgrad', B, 8, 256, 512, False)]
DIRECT = os.environ.get('MMDGAN_WINO2') == '0'                # A/B: the same launches on the direct implicit-GEMM kernels


def timeit(fn, reps=30, warm=200):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
for name, N, H, C, K, dgrad in CASES3:                       # F(2x2,3x3), stride 1
    w = torch.randn(3, 3, C, K, device='cuda') * 0.05
    wt = w.permute(3, 2, 0, 1).contiguous()
    u = None if os.environ.get('MMDGAN_WINO') == '0' else ops.wino_transform(w, dgrad)
    fl = 2.0 * N * H * H * K * 9 * C
    if not dgrad:
        x = torch.randn(N, H, H, C, device='cuda')
        y = torch.empty(N, H, H, K, device='cuda')
        run = lambda: ops.conv2d_fwd(x, w, 1, out=y, wino=u)
        run()
        ref = F.conv2d(x.permute(0, 3, 1, 2), wt, stride=1, padding=1).permute(0, 2, 3, 1)
        err = float((y - ref).abs().max() / ref.abs().max())
    else:
        dy = torch.randn(N, H, H, K, device='cuda')
        dx = torch.empty(N, H, H, C, device='cuda')
        run = lambda: ops.conv2d_dgrad(dy, w, (H, H), 1, out=dx, wino=u)
        run()
        ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), wt, stride=1, padding=1).permute(0, 2, 3, 1)
        err = float((dx - ref).abs().max() / ref.abs().max())
    t = timeit(run)
    print('%-14s N=%3d %2dx%2d C=%3d K=%3d: %7.1f us  %6.1f TF alg  %5.1f TF issued   max rel err %.2e %s' % (
        name, N, H, H, C, K, t * 1e3, fl / t / 1e9, fl * 16 / 36 / t / 1e9, err, '' if err < 1e-4 else '  <-- WRONG'))
for name, N, H, C, K, dgrad in CASES:
    P = H // 2
    w = torch.randn(4, 4, C, K, device='cuda') * 0.05
    wt = w.permute(3, 2, 0, 1).contiguous()                      # OIHW for torch
    u = None if DIRECT else ops.wino_transform(w, dgrad)
    fl = 2.0 * N * P * P * K * 16 * C
    if not dgrad:
        x = torch.randn(N, H, H, C, device='cuda')
        y = torch.empty(N, P, P, K, device='cuda')
        run = lambda: ops.conv2d_fwd(x, w, 2, out=y, wino=u)
        run()
        ref = F.conv2d(x.permute(0, 3, 1, 2), wt, stride=2, padding=1).permute(0, 2, 3, 1)
        err = float((y - ref).abs().max() / ref.abs().max())
    else:
        dy = torch.randn(N, P, P, K, device='cuda')
        dx = torch.empty(N, H, H, C, device='cuda')
        run = lambda: ops.conv2d_dgrad(dy, w, (H, H), 2, out=dx, wino=u)
        run()
        ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), wt, stride=2, padding=1).permute(0, 2, 3, 1)
        err = float((dx - ref).abs().max() / ref.abs().max())
    t = timeit(run)
    print('%-14s N=%3d %2dx%2d C=%3d K=%3d: %7.1f us  %6.1f TF alg  %5.1f TF issued   max rel err %.2e %s' % (
        name, N, H, H, C, K, t * 1e3, fl / t / 1e9, fl * 9 / 16 / t / 1e9, err, '' if err < 1e-4 else '  <-- WRONG'))
