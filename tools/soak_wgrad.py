# part of tools/soak_under_load.py (run that, or this file directly)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0])))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import numpy as np, torch
from mmdgan_hip import ops
ops.require_device(); ops.set_workspace(256 << 20)
shapes = [(128, 16, 16, 128, 128), (128, 8, 8, 256, 256), (128, 4, 4, 512, 512), (128, 24, 24, 128, 128), (64, 32, 32, 128, 128), (64, 64, 64, 64, 64),
          (256, 16, 16, 256, 256), (256, 8, 8, 512, 512), (67, 12, 20, 96, 160), (4, 32, 32, 64, 64), (130, 4, 4, 64, 64)]
side = torch.cuda.Stream(); side2 = torch.cuda.Stream()
big, wb = torch.randn(64, 64, 64, 64, device='cuda'), torch.randn(3, 3, 64, 64, device='cuda')
junk = torch.randn(64 << 20, device='cuda')
t0 = time.time(); total = 0; bad = 0
for (N, H, W, C, K) in shapes:
    g = torch.Generator(device='cuda').manual_seed(N + C + H)
    x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    dy = torch.randn(N, H, W, K, device='cuda', generator=g)
    w = torch.randn(3, 3, C, K, device='cuda', generator=g)
    db = torch.empty(K, device='cuda'); dot = torch.empty(1, device='cuda')
    torch.cuda.synchronize()
    first = ops.conv2d_wgrad(x, dy, 3, 1).clone()
    fdb = None
    for rep in range(600):
        if rep % 2:
            with torch.cuda.stream(side):
                ops.conv2d_fwd(big, wb, 1)
        if rep % 5 == 0:
            with torch.cuda.stream(side2):
                junk.mul_(1.0001)                       # a bandwidth hog
        if rep % 7 == 0:
            torch.cuda.synchronize()                    # cold restarts
        mode = rep % 4
        if mode == 0: dw = ops.conv2d_wgrad(x, dy, 3, 1)
        elif mode == 1: dw = ops.conv2d_wgrad(x, dy, 3, 1, dbias=db)
        elif mode == 2: dw = ops.conv2d_wgrad(x, dy, 3, 1, dbias=db, w=w, dot=dot)
        else:
            ops.wgrad_defer(True)
            dw = ops.conv2d_wgrad(x, dy, 3, 1, dbias=db); dw2 = ops.conv2d_wgrad(x, dy, 3, 1)
            ops.wgrad_flush(); ops.wgrad_defer(False)
            if not torch.equal(dw2, first): bad += 1
        total += 1
        if not torch.equal(dw, first):
            bad += 1
        if mode in (1, 2, 3):
            if fdb is None: fdb = db.clone()
            elif not torch.equal(db, fdb): bad += 1
    torch.cuda.synchronize()
    print((N, H, W, C, K), 'algo', ops.wgrad_algo(N, H, W, C, K, 3, 1), 'mismatches so far', bad, flush=True)
print('launch groups', total, 'mismatching results', bad, 'seconds %.0f' % (time.time() - t0))
