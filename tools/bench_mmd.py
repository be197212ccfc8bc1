"""pairwise / MMD loss kernel over the batch sweep of SURVEY 8(d): time, algorithmic bytes against the HBM peak, and
the pair evaluations per second that actually bound it (O(B^2 d) VALU + exp work on O(B d) bytes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
d = 16
print('# mmdgan_mmd_loss, d = 16, fp32; forward+backward in one launch ("fwd" = need_grads False)')
print('# algorithmic bytes: read 2*B*d*4, write 8 scalars (+ 4*B*d*4 gradients); pairs = 4*B^2 distance/kernel evaluations')
print('%6s %5s %10s %12s %14s %12s' % ('B', 'loss', 'us', 'alg. GB/s', '% of 8 TB/s', 'Gpair/s'))
for loss in ('rep', 'rmb', 'mmd_g'):
    for B in (64, 128, 256, 1024, 4096, 16384):
        g = torch.Generator(device='cuda').manual_seed(B)
        a = torch.randn(B, d, device='cuda', generator=g) * 0.25
        b = torch.randn(B, d, device='cuda', generator=g) * 0.3 + 0.1
        for grads in (True, False):
            if loss != 'rep' and not grads:
                continue
            reps = 200 if B <= 1024 else (40 if B <= 4096 else 8)
            for _ in range(3): ops.mmd_loss(a, b, loss, need_grads=grads)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ops.mmd_loss(a, b, loss, need_grads=grads)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            nbytes = 2 * B * d * 4 + 32 + (4 * B * d * 4 if grads else 0)
            print('%6d %5s %10.1f %12.3f %14.4f %12.1f  %s' % (B, loss, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80, 4.0 * B * B / us / 1e3,
                                                             'fwd+bwd' if grads else 'fwd'))
