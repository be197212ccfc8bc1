"""find a fast and a slow placement of the same engine in one process, then run 30 steps of each back to back
(for a rocprofv3 --kernel-trace: the last 60 steps of the trace are 30 fast then 30 slow)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
arch, lr = configs.CONFIGS['cifar']()
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)
os.environ.setdefault('MMDGAN_SIDE_WGRAD', '0'); os.environ.setdefault('MMDGAN_SN_STREAMS', '1')


def run(eng, N=40):
    for _ in range(5): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


engs, keep = [], []
for nbytes in (0, 1 << 20, 3 << 20, 1 << 30):
    if nbytes:
        keep.append(torch.empty(nbytes, dtype=torch.uint8, device='cuda'))
    e = GanEngine(arch, 'rep', lr, batch_size=64, seed=0)
    engs.append((run(e), e))
engs.sort(key=lambda t: t[0])
print('placements (ms/step):', [round(t, 3) for t, _ in engs], flush=True)
fast, slow = engs[0][1], engs[-1][1]
torch.cuda.synchronize()
for e in (fast, slow):
    for _ in range(30): e.step(real)
    torch.cuda.synchronize()

def addrs(e):
    out = {k: v.data_ptr() for k, v in e.buf.items() if hasattr(v, 'data_ptr')}
    out['gen.params'] = e.gen.params.data_ptr(); out['dis.params'] = e.dis.params.data_ptr()
    out['gen.grads'] = e.gen.grads.data_ptr(); out['dis.grads'] = e.dis.grads.data_ptr()
    return out
import json
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'place'), exist_ok=True)
json.dump({'fast': addrs(fast), 'slow': addrs(slow), 'ms': [engs[0][0], engs[-1][0]]},
          open(os.path.join(ROOT, 'gpurun_out', 'place', 'addrs.json'), 'w'))
