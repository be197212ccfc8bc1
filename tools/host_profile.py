"""where the host time of one eager step goes (cProfile over 30 steps)"""
import cProfile, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
arch, lr = configs.CONFIGS['cifar']()
eng = GanEngine(arch, 'rep', lr, batch_size=64, seed=0, use_graph=False)
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)
for _ in range(5): eng.step(real)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30): eng.step(real)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
