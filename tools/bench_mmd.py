"""pairwise / MMD loss kernel over the batch sweep of SURVEY 8(d): time, algorithmic bytes against the HBM peak, and
the pair evaluations per second that actually bound it (O(B^2 d) VALU + exp work on O(B d) bytes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
d = 16
print('# mmdgan_mmd_loss, d = 16, fp32; forward+backward in one launch ("fwd" = need_grads False)')
print('# algorithmic bytes: read 2*B*d*4, write 8 scalars (+ 4*B*d*4 gradients); pairs = 4*B^2 distance/kernel evaluations')
# What bounds it at large B is neither HBM nor a matrix pipe but VALU issue (and the LDS port feeding it).  Instruction model of
# csrc/mmd.hip per (i, j) lane-iteration = 4 distance / kernel evaluations ("pairs"), d = 16, counted in fp32 VALU issue slots
# (a transcendental v_exp_f32 is quarter rate = 4 slots, expf = scale + v_exp):
#   forward : 6 dot products x 16 fma = 96, distances / clamps / sums ~ 24, 4 expf x 5 = 20         -> 140 slots = 35 per pair
#   backward: 16 x (4 sub + 8 fma) = 192, 8 coefficients ~ 24                                        -> +216     = +54 per pair
#   rmb     : two more expf on the bounded blocks + 4 compares                                        -> +14      = +3.5 per pair
#   mmd_g   : five Gaussians per evaluation: 4 x 4 more expf + 4 x 8 fma                              -> +112     = +28 per pair
# VALU issue peak: 256 CU x 4 SIMD x 32 lanes / clk x 2.4 GHz = 78.6 T lane-slots/s (the 157.3 TFLOP/s fp32 vector peak / 2).
# LDS: the 32 ds_read_b32 of x_j / y_j per tile and wave (row stride d + 1: conflict-free) take 2 clk each on the CU's one
# 128 B/clk port, 4 waves -> 256 clk per tile against ~280 clk of VALU issue per SIMD: the two limits are about equal, so the
# kernel cannot pass ~50 % of the VALU-issue column below without keeping x_j / y_j in registers across the four rows.
VALU_PEAK = 78.6e12
print('# VALU-issue column: fp32 VALU issue slots per pair evaluation (d = 16; v_exp_f32 quarter rate): rep forward 35, + backward 54;')
print('# rmb + 3.5, mmd_g (five Gaussians) + 28; against 78.6 T lane-slots/s (256 CU x 4 SIMD x 32 lanes x 2.4 GHz).  The LDS port')
print('# (32 ds_read_b32 of x_j / y_j per tile and wave) caps the kernel near 50 % of that column; below B ~ 1024 it is launch latency.')
SLOTS = {'rep': (35.0, 54.0), 'rmb': (38.5, 54.0), 'mmd_g': (63.0, 54.0)}
print('%6s %5s %10s %12s %14s %12s %16s' % ('B', 'loss', 'us', 'alg. GB/s', '% of 8 TB/s', 'Gpair/s', '% of VALU issue'))
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
            slots = SLOTS[loss][0] + (SLOTS[loss][1] if grads else 0.0)
            print('%6d %5s %10.1f %12.3f %14.4f %12.1f %16.1f  %s' % (B, loss, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80, 4.0 * B * B / us / 1e3,
                                                                     100.0 * 4.0 * B * B * slots / (us * 1e-6) / VALU_PEAK,
                                                                     'fwd+bwd' if grads else 'fwd'))
