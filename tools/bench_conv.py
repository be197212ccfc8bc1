#!/usr/bin/env python3
"""per-layer conv kernel timings (HIP events) for the CIFAR B=64 step: TFLOP/s per launch.

The layers are launched the way the engine launches them: with the Winograd-transformed weights handed in (`wino=`, made
once per step by mmdgan_wino_transform_multi, so the transform is not part of a layer's time and the launches with few
tile blocks may split their reduction over workspace slabs).  BENCH_OWN_TRANSFORM=1: the library transforms the weights
inside every call (what a caller without the transformed tensors gets; rounds 1-2 measured this)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
ops.require_device()
ops.set_workspace()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# (name, N, H, W, C, K, R, stride)
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''
LAYERS = [('D l2', 2 * B, 32, 32, 64, 128, 4, 2), ('D l3', 2 * B, 16, 16, 128, 128, 3, 1),
          ('D l4', 2 * B, 16, 16, 128, 256, 4, 2), ('D l5', 2 * B, 8, 8, 256, 256, 3, 1),
          ('D l6', 2 * B, 8, 8, 256, 512, 4, 2), ('D l7', 2 * B, 4, 4, 512, 512, 3, 1),
          # G tc layers expressed as the conv whose dgrad they are: conv input = tc output
          ('G l2 (tc)', B, 8, 8, 256, 512, 4, 2), ('G l3 (tc)', B, 16, 16, 128, 256, 4, 2),
          ('G l4 (tc)', B, 32, 32, 64, 128, 4, 2),
          ('D l1 thin', 2 * B, 32, 32, 3, 64, 3, 1), ('D l1 (B)', B, 32, 32, 3, 64, 3, 1), ('G l5 thin', B, 32, 32, 64, 3, 3, 1)]
def timeit(fn, reps=int(os.environ.get('BENCH_REPS', '20'))):
    # BENCH_GRAPH=1: the launches captured into a hipGraph and replayed - a launch below ~25 us is otherwise timed at what the
    # Python wrapper costs per call, not at what the kernel takes; BENCH_WARM: untimed launches first (the clocks ramp over ms)
    for _ in range(int(os.environ.get('BENCH_WARM', '3'))): fn()
    if os.environ.get('BENCH_GRAPH'):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            fn()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps): fn()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (5 * reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print('%-10s %8s | %14s | %14s | %14s' % ('layer', 'GFLOP', 'fwd us (TF)', 'dgrad us (TF)', 'wgrad us (TF)'))
tot = [0, 0, 0]
for name, N, H, W, C, K, R, s in LAYERS:
    if ONLY and ONLY not in name: continue
    P, Q = -(-H // s), -(-W // s)
    x = torch.randn(N, H, W, C, device='cuda'); w = torch.randn(R, R, C, K, device='cuda') * 0.05
    dy = torch.randn(N, P, Q, K, device='cuda'); y = torch.empty(N, P, Q, K, device='cuda')
    dx = torch.empty_like(x); dw = torch.empty_like(w); bias = torch.zeros(K, device='cuda')
    fl = 2.0 * N * P * Q * K * R * R * C
    uf = ud = None
    if not os.environ.get('BENCH_OWN_TRANSFORM'):
        uf = ops.wino_transform(w, False) if ops.wino_eligible(N, H, W, C, K, R, s, False) else None
        ud = ops.wino_transform(w, True) if ops.wino_eligible(N, H, W, C, K, R, s, True) else None
    if os.environ.get('BENCH_DGRAD_3B') and name.startswith('D l') and 'thin' not in name and '(B)' not in name:
        # the step's D backward: 3B rows (loss_dis rows 2B + loss_gen rows B), dact wraps to the last B images
        n3 = N + N // 2
        dy3 = torch.randn(n3, P, Q, K, device='cuda'); dx3 = torch.empty(n3, H, W, C, device='cuda')
        t3 = timeit(lambda: ops.conv2d_dgrad(dy3, w, (H, W), s, act='lrelu', dact_of=x, dact_batch=N, out=dx3, wino=ud))
        print('%-10s dgrad 3B rows: %7.1f us (%5.1f TF)' % (name, t3 * 1e3, 1.5 * fl / t3 / 1e9))
    t = [timeit(lambda: ops.conv2d_fwd(x, w, s, bias=bias, act='lrelu', out=y, wino=uf)),
         timeit(lambda: ops.conv2d_dgrad(dy, w, (H, W), s, act='lrelu', dact_of=x, out=dx, wino=ud)),
         timeit(lambda: ops.conv2d_wgrad(x, dy, R, s, out=dw))]
    for i in range(3): tot[i] += t[i]
    print('%-10s %8.2f | %7.1f (%5.1f) | %7.1f (%5.1f) | %7.1f (%5.1f)' % (
        name, fl / 1e9, t[0] * 1e3, fl / t[0] / 1e9, t[1] * 1e3, fl / t[1] / 1e9, t[2] * 1e3, fl / t[2] / 1e9))
print('sum us: fwd %.0f dgrad %.0f wgrad %.0f' % tuple(v * 1e3 for v in tot))
