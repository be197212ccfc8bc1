#!/usr/bin/env python3
"""The split-bf16 (bf16x6) convolution prototype (csrc/conv_bfx.hip, MMDGAN_BFX=2) against the shipped fp32-MFMA kernels:
error against an fp64 reference and time per launch, forward and input-gradient, on the CIFAR B=64 layer shapes.
Run twice (the switch is read once per process):  MMDGAN_BFX=2 python tools/bfx_probe.py ; MMDGAN_BFX=0 python tools/bfx_probe.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
from oracle import restatement as R
ops.require_device(); ops.set_workspace(64 << 20)
mode = os.environ.get('MMDGAN_BFX', '1')
def timeit(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('MMDGAN_BFX=%s   (error = max|y - y64| / max|y64|)' % mode)
print('%-8s %28s | %22s | %22s' % ('layer', 'shape N,H,W,C,K,R,s', 'fwd: err, us', 'dgrad: err, us'))
rs = np.random.RandomState(0)
for name, N, H, W, C, K, Rk, s in (('D l2', 128, 32, 32, 64, 128, 4, 2), ('D l3', 128, 16, 16, 128, 128, 3, 1), ('D l5', 128, 8, 8, 256, 256, 3, 1),
                                   ('D l7', 128, 4, 4, 512, 512, 3, 1)):
    P = -(-H // s)
    x = rs.uniform(-1, 1, (N, C, H, W)).astype(np.float32)
    w = (rs.randn(Rk, Rk, C, K) / np.sqrt(Rk * Rk * C)).astype(np.float32)
    dy = rs.randn(N, K, P, P).astype(np.float32)
    n_ref = 8                                            # fp64 reference on the first images only
    xt = torch.tensor(x[:n_ref], dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64)
    y64 = R.conv2d_same(xt, wt, s)
    gx, = torch.autograd.grad((y64 * torch.tensor(dy[:n_ref], dtype=torch.float64)).sum(), [xt])
    xd = torch.as_tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
    dyd = torch.as_tensor(np.ascontiguousarray(dy.transpose(0, 2, 3, 1))).cuda()
    wd = torch.as_tensor(w).cuda()
    y = ops.conv2d_fwd(xd, wd, s)
    dx = ops.conv2d_dgrad(dyd, wd, (H, W), s)
    ef = np.abs(y[:n_ref].cpu().numpy().transpose(0, 3, 1, 2) - y64.detach().numpy()).max() / np.abs(y64.detach().numpy()).max()
    eb = np.abs(dx[:n_ref].cpu().numpy().transpose(0, 3, 1, 2) - gx.numpy()).max() / np.abs(gx.numpy()).max()
    yo, dxo = torch.empty_like(y), torch.empty_like(dx)
    tf = timeit(lambda: ops.conv2d_fwd(xd, wd, s, out=yo))
    tb = timeit(lambda: ops.conv2d_dgrad(dyd, wd, (H, W), s, out=dxo))
    print('%-8s %28s | %10.2e %9.1f | %10.2e %9.1f' % (name, (N, H, W, C, K, Rk, s), ef, tf, eb, tb))
