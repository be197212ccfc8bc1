"""one-rank RCCL group on one GPU: what the gradient exchange plumbing costs per step (host issue time and total),
with the pieces switched on one at a time.  Run: MMDGAN_DP_FORCE=1 python tools/dp_probe.py"""
import os, sys, time, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ['MMDGAN_DP_FORCE'] = '1'
import configs
from mmdgan_hip.engine import GanEngine
from mmdgan_hip import dist as mdist
torch.cuda.set_device(0)
mdist.init_process_group(0) if os.environ.get('DP_PROBE_PLAIN') != '1' else dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
arch, lr = configs.CONFIGS['cifar']()
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)


def run(tag, eng, N=100):
    for _ in range(10): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-34s CPU issue %.3f ms/step, total %.3f ms/step' % (tag, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3), flush=True)


e0 = GanEngine(arch, 'rep', lr, batch_size=64, seed=0)
run('no group', e0)
e1 = GanEngine(arch, 'rep', lr, batch_size=64, seed=0, dist_group=dist.group.WORLD)
run('group, exchange on', e1)
e1._dp_force = False
run('group, exchange off', e1)
e1._dp_force = True
orig = mdist.allreduce_sum_
mdist.allreduce_sum_ = lambda flat, group=None, bucket_bytes=0: flat
run('group, all-reduce stubbed out', e1)
mdist.allreduce_sum_ = orig
for mb in (8, 128):
    mdist.DEFAULT_BUCKET_BYTES = mb << 20
    mdist.allreduce_sum_.__defaults__ = (None, mb << 20)
    run('group, %d MiB buckets' % mb, e1)
print('arena MB: G %.1f D %.1f' % (e1.gen.grads.numel() * 4 / 1e6, e1.dis.grads.numel() * 4 / 1e6))
run('no group again', e0)
run('group, exchange on again', e1)
run('no group 3', e0)
del e1
import gc; gc.collect(); torch.cuda.empty_cache()
run('no group, other engine freed', e0)
dist.destroy_process_group()
