"""eager issue vs hipGraph replay vs launch-plan replay in one process: CPU issue time and total time per step
    python tools/issue_time.py [config] [batch]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
config = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
B = int(sys.argv[2]) if len(sys.argv) > 2 else {'celeba': 128}.get(config, 64)
arch, lr = configs.CONFIGS[config]()
eng = GanEngine(arch, 'rep', lr, batch_size=B, seed=0)
c, h, w = arch['input'][0]
real = torch.empty(B, h, w, c, device='cuda').uniform_(-1, 1)
def run(tag, mode, N=100):
    eng.launch_mode = mode
    for _ in range(10): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()                   # host issue time: 20 steps into an empty queue (a longer run measures the
    for _ in range(20): eng.step(real)         # queue's back-pressure instead: the host blocks once it is ~20 steps ahead)
    issue = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    torch.cuda.synchronize()
    print('%-14s CPU issue %.3f ms/step, total %.3f ms/step' % (tag, issue * 1e3, (time.perf_counter() - t0) / N * 1e3))
for tag, m in (('eager', 'eager'), ('graph', 'graph'), ('plan', 'plan'), ('eager again', 'eager'), ('graph again', 'graph'), ('plan again', 'plan')):
    run(tag, m)
with eng._handle:
    from mmdgan_hip import ops
    lib = ops.require_device()
    print('plan: %d recorded nodes (launches, memsets, stream dependencies), %d segment(s)' % (lib.mmdgan_plan_nodes(eng._plan), lib.mmdgan_plan_segments(eng._plan)))
