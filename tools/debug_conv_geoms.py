#!/usr/bin/env python3
"""conv ops of the library at given geometries against torch-CPU fp64 (a debugging aid, not a test)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
from oracle import restatement as R
ops.require_device(); ops.set_workspace(128 << 20)
torch.manual_seed(0)
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())
# (name, N_fwd, N_dgrad, dact_batch, H, C, K, R, s)
GEOMS = [('D b1 k1 folded 4x4s2', 64, 96, 64, 64, 64, 64, 4, 2), ('D b1 k0 thin 3->64', 64, 32, 0, 64, 3, 64, 3, 1),
         ('D b1 sc 1x1 3->64', 64, 32, 0, 32, 3, 64, 1, 1), ('D b2 k0 3x3 64->128 @32', 64, 96, 64, 32, 64, 128, 3, 1),
         ('G last 3x3 64->3 @64', 32, 32, 0, 64, 64, 3, 3, 1), ('G b5 upconv 4x4s2 as conv 64<-128... C=64 K=128', 32, 32, 0, 64, 64, 128, 4, 2)]
for name, nf, nd, db, H, C, K, Rk, s in GEOMS:
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    P = -(-H // s)
    w = torch.randn(Rk, Rk, C, K) * 0.05
    for use_wino in (False, True):
        wd = w.cuda()
        # forward
        x = torch.randn(nf, C, H, H)
        ref = R.conv2d_same(x.double(), w.double(), s)
        uf = ops.wino_transform(wd, False) if use_wino and ops.wino_eligible(nf, H, H, C, K, Rk, s, False) else None
        y = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), wd, s, wino=uf)
        ef = rel(y.cpu().permute(0, 3, 1, 2), ref)
        # input gradient with the activation derivative of a wrapped operand
        dy = torch.randn(nd, K, P, P)
        yact = torch.randn(db if db else nd, C, H, H)
        dref = R.conv2d_transpose_same(dy.double(), w.double(), (H, H), s)
        full = torch.cat([yact, yact[-(nd - db):]], 0) if db else yact
        dref = dref * torch.where(full > 0, torch.ones_like(full), torch.full_like(full, 0.1)).double()
        ud = ops.wino_transform(wd, True) if use_wino and ops.wino_eligible(nd, H, H, C, K, Rk, s, True) else None
        dx = ops.conv2d_dgrad(dy.permute(0, 2, 3, 1).contiguous().cuda(), wd, (H, H), s, act='lrelu',
                              dact_of=yact.permute(0, 2, 3, 1).contiguous().cuda(), dact_batch=db, wino=ud)
        ed = rel(dx.cpu().permute(0, 3, 1, 2), dref)
        # weight gradient
        x2 = x.double().requires_grad_(False)
        w2 = w.double().clone().requires_grad_(True)
        dyf = torch.randn(nf, K, P, P)
        (R.conv2d_same(x2, w2, s) * dyf.double()).sum().backward()
        dw = ops.conv2d_wgrad(x.permute(0, 2, 3, 1).contiguous().cuda(), dyf.permute(0, 2, 3, 1).contiguous().cuda(), Rk, s)
        ew = rel(dw.cpu(), w2.grad)
        print('%-50s wino=%-5s (fwd %s dgrad %s)  fwd %.1e/%.1e  dgrad %.1e/%.1e  wgrad %.1e/%.1e' % (
            name, use_wino, uf is not None, ud is not None, ef[0], ef[1], ed[0], ed[1], ew[0], ew[1]), flush=True)
