# part of tools/soak_under_load.py (run that, or this file directly)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0])))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import numpy as np, torch
from mmdgan_hip import ops
ops.require_device(); ops.set_workspace(256 << 20)
shapes = [(128, 16, 16, 128, 128, 3, 1), (192, 8, 8, 256, 256, 3, 1), (192, 4, 4, 512, 512, 3, 1), (128, 32, 32, 64, 128, 4, 2), (192, 16, 16, 128, 256, 4, 2), (64, 32, 32, 64, 64, 3, 1)]
side = torch.cuda.Stream(); side2 = torch.cuda.Stream()
big, dyb = torch.randn(128, 16, 16, 128, device='cuda'), torch.randn(128, 16, 16, 128, device='cuda')
junk = torch.randn(64 << 20, device='cuda')
bad = 0; total = 0
for (N, H, W, C, K, R, s) in shapes:
    g = torch.Generator(device='cuda').manual_seed(N + C + H)
    x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    w = torch.randn(R, R, C, K, device='cuda', generator=g) * 0.05
    P = H // s
    dy = torch.randn(N, P, P, K, device='cuda', generator=g)
    af, ab = ops.wino_algo(N, H, W, C, K, R, s, False), ops.wino_algo(N, H, W, C, K, R, s, True)
    uf = ops.wino_transform(w, False, algo=af) if af else None
    ub = ops.wino_transform(w, True, algo=ab) if ab else None
    torch.cuda.synchronize()
    y0 = ops.conv2d_fwd(x, w, s, act='lrelu', wino=uf).clone(); dx0 = ops.conv2d_dgrad(dy, w, (H, W), s, wino=ub).clone()
    for rep in range(400):
        if rep % 2:
            with torch.cuda.stream(side):
                ops.conv2d_wgrad(big, dyb, 3, 1)
        if rep % 5 == 0:
            with torch.cuda.stream(side2):
                junk.mul_(1.0001)
        if rep % 7 == 0:
            torch.cuda.synchronize()
        y = ops.conv2d_fwd(x, w, s, act='lrelu', wino=uf); dx = ops.conv2d_dgrad(dy, w, (H, W), s, wino=ub)
        total += 2
        bad += (not torch.equal(y, y0)) + (not torch.equal(dx, dx0))
    torch.cuda.synchronize()
    print((N, H, W, C, K, R, s), 'algos', af, ab, 'mismatches so far', bad, flush=True)
print('launches', total, 'mismatching results', bad)
