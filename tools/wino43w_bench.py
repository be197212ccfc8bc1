#!/usr/bin/env python3
"""The 3x3 / stride-1 weight gradients of the configs at D's batch 2B, alone, with their reduction pass and bias gradient
(mmdgan_conv2d_wgrad_bias, workspace registered): one process per kernel selection, because the library reads its switches
once - the parent runs itself with MMDGAN_WINO43_WGRAD=0 (F(2x2,3x3) slab kernel) and =2 (F(4x4,3x3), csrc/conv_wino43w.hip)
and prints both columns.  Times are hipGraph replays.
    python tools/wino43w_bench.py [config ...]       configs: cifar stl celeba resnet      env: BENCH_REPS (20)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, B, H, C, K)
LAYERS = {'cifar': [('D l3', 64, 16, 128, 128), ('D l5', 64, 8, 256, 256), ('D l7', 64, 4, 512, 512)],
          'stl': [('D l3', 64, 24, 128, 128), ('D l5', 64, 12, 256, 256)],
          'celeba': [('D l3', 128, 32, 128, 128), ('D l5', 128, 16, 256, 256), ('D l7', 128, 8, 512, 512), ('D l9', 128, 4, 1024, 1024)],
          'resnet': [('res 64x64', 32, 64, 64, 64), ('res 32x32', 32, 32, 128, 128), ('res 16x16', 32, 16, 256, 256),
                     ('res 8x8', 32, 8, 512, 512)]}
REPS = int(os.environ.get('BENCH_REPS', '20'))


def child(cfgs):
    import torch
    sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
    from mmdgan_hip import ops
    ops.require_device()
    ops.set_workspace(256 << 20)
    out = {}

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

    for cfg in cfgs:
        for name, B, H, C, K in LAYERS[cfg]:
            n = 2 * B
            x = torch.randn(n, H, H, C, device='cuda')
            dy = torch.randn(n, H, H, K, device='cuda')
            dw = torch.empty(3, 3, C, K, device='cuda')
            db = torch.empty(K, device='cuda')
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, 3, 1, out=dw, dbias=db))
            out['%s/%s' % (cfg, name)] = {'us': t, 'gflop': 2.0 * n * H * H * 9 * C * K / 1e9, 'sum': float(dw.double().abs().sum())}
    print(json.dumps(out))


if __name__ == '__main__':
    if os.environ.get('W43W_CHILD'):
        child(sys.argv[1:])
        sys.exit(0)
    cfgs = sys.argv[1:] or ['cifar', 'stl', 'celeba', 'resnet']
    res = {}
    for mode in ('0', '2'):
        env = dict(os.environ, W43W_CHILD='1', MMDGAN_WINO43_WGRAD=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + cfgs, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    print('%-8s %-10s %6s | weight gradient at 2B with reduction + bias gradient: F(2x2,3x3) us (TF), F(4x4,3x3) us (TF), x' % ('config', 'layer', 'GFLOP'))
    for key in res['0']:
        a, b = res['0'][key], res['2'][key]
        cfg, name = key.split('/')
        print('%-8s %-10s %6.2f | %8.1f (%5.1f) %8.1f (%5.1f)  %.2fx   |sum| %.6g / %.6g' % (
            cfg, name, a['gflop'], a['us'], a['gflop'] / a['us'] * 1e3, b['us'], b['gflop'] / b['us'] * 1e3, a['us'] / b['us'], a['sum'], b['sum']))
