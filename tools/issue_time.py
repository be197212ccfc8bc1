"""eager vs hipGraph replay in one process: CPU issue time and total time per step"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
arch, lr = configs.CONFIGS['cifar']()
eng = GanEngine(arch, 'rep', lr, batch_size=64, seed=0, use_graph=False)
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)
def run(tag, graph, N=50):
    eng.use_graph = graph
    for _ in range(5): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-14s CPU issue %.3f ms/step, total %.3f ms/step' % (tag, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
for tag, g in (('eager', False), ('graph', True), ('eager again', False), ('graph again', True), ('eager 3', False)):
    run(tag, g)
