#!/usr/bin/env python3
"""F(4x4,3x3) against F(2x2,3x3) on the 3x3 / stride-1 layers of the configs, launched the way the engines launch them
(transformed weights handed in, workspace registered so that small grids split their reduction): forward at D's batch 2B,
input-gradient at 3B rows with the activation derivative of 2B rows.  Times are hipGraph replays (short launches are
otherwise timed at what the Python wrapper costs).
    python tools/wino43_bench.py [config ...]        configs: cifar stl celeba resnet      env: BENCH_REPS (20)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
os.environ.setdefault('MMDGAN_WINO43', '2')
from mmdgan_hip import ops  # noqa: E402

ops.require_device()
ops.set_workspace()
# (name, B, H, C, K)
LAYERS = {'cifar': [('D l3', 64, 16, 128, 128), ('D l5', 64, 8, 256, 256), ('D l7', 64, 4, 512, 512)],
          'stl': [('D l3', 64, 24, 128, 128), ('D l5', 64, 12, 256, 256)],
          'celeba': [('D l3', 128, 32, 128, 128), ('D l5', 128, 16, 256, 256), ('D l7', 128, 8, 512, 512), ('D l9', 128, 4, 1024, 1024)],
          'resnet': [('res 64x64', 32, 64, 64, 64), ('res 32x32', 32, 32, 128, 128), ('res 16x16', 32, 16, 256, 256),
                     ('res 8x8', 32, 8, 512, 512)]}
REPS = int(os.environ.get('BENCH_REPS', '20'))


def timeit(fn):
    for _ in range(3):
        fn()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(REPS):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


print('%-8s %-10s %6s | %-29s | %-29s' % ('config', 'layer', 'GFLOP', 'forward 2B: F23 us, F43 us (x)', 'dgrad 3B: F23 us, F43 us (x)'))
for cfg in (sys.argv[1:] or ['cifar', 'stl', 'celeba', 'resnet']):
    for name, B, H, C, K in LAYERS[cfg]:
        nf, nb = 2 * B, 3 * B
        x = torch.randn(nf, H, H, C, device='cuda')
        w = torch.randn(3, 3, C, K, device='cuda') * 0.05
        y = torch.empty(nf, H, H, K, device='cuda')
        bias = torch.zeros(K, device='cuda')
        scale = torch.ones(1, device='cuda')
        dy = torch.randn(nb, H, H, K, device='cuda')
        dx = torch.empty(nb, H, H, C, device='cuda')
        cols = []
        for dgrad, n in ((False, nf), (True, nb)):
            t = {}
            for algo in (ops.WINO_F23, ops.WINO_F43):
                if algo == ops.WINO_F43 and ops.wino_algo(n, H, H, C, K, 3, 1, dgrad) != ops.WINO_F43:
                    t[algo] = float('nan')
                    continue
                u = ops.wino_transform(w, dgrad, algo=algo)
                if dgrad:
                    t[algo] = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, H), 1, scale=scale, act='lrelu', dact_of=x, dact_batch=nf, out=dx,
                                                              wino=u))
                else:
                    t[algo] = timeit(lambda: ops.conv2d_fwd(x, w, 1, bias=bias, scale=scale, act='lrelu', out=y, wino=u))
            cols.append('%8.1f %8.1f (%.2fx)' % (t[ops.WINO_F23], t[ops.WINO_F43], t[ops.WINO_F23] / t[ops.WINO_F43]))
        print('%-8s %-10s %6.2f | %-29s | %-29s' % (cfg, name, 2.0 * nf * H * H * 9 * C * K / 1e9, cols[0], cols[1]), flush=True)
