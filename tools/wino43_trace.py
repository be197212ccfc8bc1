#!/usr/bin/env python3
"""Where one workgroup of the F(4x4,3x3) kernel spends its windows: shader-clock stamps written by a measurement build
(-DW43_TRACE=<workgroup id>, csrc/conv_wino43.hip) of consumer wave 0 and producer wave 4.
    here:    tools/wino43_trace.py build            -> tools/scratch/w43libs/trace.so
    GPU box: python tools/wino43_trace.py [dgrad]   (swaps the library in for this process's run, restores it)"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'mmd-gan_amd', 'lib', 'libmmdgan_hip.so')
TRACE = os.path.join(ROOT, 'tools', 'scratch', 'w43libs', 'trace.so')
if sys.argv[1:2] == ['build']:
    os.makedirs(os.path.dirname(TRACE), exist_ok=True)
    os.chdir(os.path.join(ROOT, 'mmd-gan_amd'))
    subprocess.check_call([sys.executable, 'build_ext.py'], stdout=subprocess.DEVNULL)
    subprocess.check_call('/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -DW43_TRACE=%s '
                          '%s -c csrc/conv_wino43.hip -o /tmp/w43_trace.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s '
                          '$(ls build/*.o | grep -v conv_wino43.o) /tmp/w43_trace.o' % (sys.argv[2] if len(sys.argv) > 2 else '17', ' '.join(sys.argv[3:]), TRACE), shell=True)
    sys.exit(0)
shutil.copy(LIB, '/tmp/lib_keep.so')
shutil.copy(TRACE, LIB)
try:
    sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
    os.environ['MMDGAN_WINO43'] = '2'
    import torch
    from mmdgan_hip import ops
    lib = ops.require_device()
    dgrad = 'dgrad' in sys.argv
    B, H, C, K = 64, 16, 128, 128
    n = 3 * B if dgrad else 2 * B
    x = torch.randn(n, H, H, C, device='cuda')
    w = torch.randn(3, 3, C, K, device='cuda') * 0.05
    y = torch.empty(n, H, H, K, device='cuda')
    u = ops.wino_transform(w, dgrad, algo=ops.WINO_F43)
    fn = (lambda: ops.conv2d_dgrad(x, w, (H, H), 1, out=y, wino=u)) if dgrad else (lambda: ops.conv2d_fwd(x, w, 1, act='lrelu', out=y, wino=u))
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_long * (8 * 256))()
    lib.mmdgan_w43_trace.restype = ctypes.c_int
    assert lib.mmdgan_w43_trace(buf, 8 * 256) == 0
    t = [[buf[r * 256 + j] for j in range(256)] for r in range(8)]
    S = C // 16
    t0 = t[0][0]
    print('window | consumer: start, MFMAs issued | producer: start, [unused], patches requested, transformed + V stored   (cycles since the first window)')
    for j in range(S):
        print('%6d | %8d %8d | %8d %8d %8d %8d' % (j, t[0][j] - t0, t[1][j] - t0, t[2][j] - t0, t[3][j] - t0, t[4][j] - t0, t[5][j] - t0))
    print('kernel entry %d; first barrier passed %d' % (t[7][0] - t0, t[7][1] - t0))
    print('after the last window %d; exchange written %d; barrier passed %d; kernel end %d' % (t[6][0] - t0, t[6][1] - t0, t[6][2] - t0, t[6][3] - t0))
finally:
    shutil.copy('/tmp/lib_keep.so', LIB)
