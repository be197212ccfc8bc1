"""uint8-record decode kernel against the HBM roofline: algorithmic bytes = 1 B in + 4 B out per value"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip import ops
for n, c, h, w in ((64, 3, 32, 32), (128, 3, 64, 64), (4096, 3, 64, 64), (16384, 3, 64, 64)):
    u8 = torch.randint(0, 256, (n, c * h * w), dtype=torch.uint8, device='cuda')
    out = torch.empty(n, h, w, c, device='cuda')
    for _ in range(5): ops.u8_records_to_nhwc(u8, c, h, w, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.u8_records_to_nhwc(u8, c, h, w, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gb = u8.numel() * 5 / 1e9
    print('decode %6d x %dx%dx%d: %8.1f us  %7.1f GB/s  (%.1f%% of 8 TB/s)' % (n, c, h, w, us, gb / us * 1e6, gb / us * 1e6 / 80))
