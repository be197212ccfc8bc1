#!/usr/bin/env python3
"""Where one workgroup of the F(4x4,3x3) weight-gradient kernel spends its windows: shader-clock stamps written by a
measurement build (-DW43W_TRACE=<workgroup id>, csrc/conv_wino43w.hip) of consumer wave 0 and producer wave 4.
    here:    tools/wino43w_trace.py build [wg] [extra hipcc flags]   -> tools/scratch/w43wlibs/trace.so
    GPU box: python tools/wino43w_trace.py [H C K]                   (swaps the library in for this process's run, restores it)"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'mmd-gan_amd', 'lib', 'libmmdgan_hip.so')
TRACE = os.path.join(ROOT, 'tools', 'scratch', 'w43wlibs', 'trace.so')
if sys.argv[1:2] == ['build']:
    os.makedirs(os.path.dirname(TRACE), exist_ok=True)
    os.chdir(os.path.join(ROOT, 'mmd-gan_amd'))
    subprocess.check_call([sys.executable, 'build_ext.py'], stdout=subprocess.DEVNULL)
    subprocess.check_call('/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -DW43W_TRACE=%s '
                          '%s -c csrc/conv_wino43w.hip -o /tmp/w43w_trace.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s '
                          '$(ls build/*.o | grep -v conv_wino43w.o) /tmp/w43w_trace.o' % (sys.argv[2] if len(sys.argv) > 2 else '17', ' '.join(sys.argv[3:]), TRACE), shell=True)
    sys.exit(0)
shutil.copy(LIB, '/tmp/lib_keep.so')
shutil.copy(TRACE, LIB)
try:
    sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
    os.environ['MMDGAN_WINO43_WGRAD'] = '2'
    import torch
    from mmdgan_hip import ops
    lib = ops.require_device()
    ops.set_workspace(256 << 20)
    H, C, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 128, 128)
    n = 128
    x = torch.randn(n, H, H, C, device='cuda')
    dy = torch.randn(n, H, H, K, device='cuda')
    dw = torch.empty(3, 3, C, K, device='cuda')
    for _ in range(20):
        ops.conv2d_wgrad(x, dy, 3, 1, out=dw)
    torch.cuda.synchronize()
    buf = (ctypes.c_long * (8 * 256))()
    lib.mmdgan_w43w_trace.restype = ctypes.c_int
    assert lib.mmdgan_w43w_trace(buf, 8 * 256) == 0
    t = [[buf[r * 256 + j] for j in range(256)] for r in range(8)]
    S = max(j for j in range(256) if t[0][j]) + 1
    t0 = t[0][0]
    print('%d windows; cycles since the consumer\'s first window' % S)
    print('window | consumer: start, MFMAs issued (d) | producer wave 4: kind, start, done (d) | window length')
    for j in range(S):
        nxt = t[0][j + 1] if j + 1 < S else t[6][0]
        kind, ps = ('off: B^T d B ', t[2][j]) if t[2][j] else ('duty: stores + M + requests', t[3][j])
        print('%6d | %8d %8d (%5d) | %-28s %8d %8d (%5d) | %6d' % (
            j, t[0][j] - t0, t[1][j] - t0, t[1][j] - t[0][j], kind, ps - t0, t[4][j] - t0, t[4][j] - ps, nxt - t[0][j]))
    print('kernel entry %d; consumers past the first barrier %d' % (t[7][0] - t0, t[7][1] - t0))
    print('after the last window %d; exchange written %d; barrier passed %d; kernel end %d' % (t[6][0] - t0, t[6][1] - t0, t[6][2] - t0, t[6][3] - t0))
finally:
    shutil.copy('/tmp/lib_keep.so', LIB)
