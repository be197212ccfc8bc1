"""debug: per-tensor comparison of the 2-rank (gloo, one GPU) engine step with the local gradients"""
import os, sys, json, subprocess, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if 'RANK' not in os.environ:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0')) for r in (0, 1)]
    for p in ps: p.wait()
    sys.exit(0)
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), ROOT]
from mmdgan_hip.engine import GanEngine
from mmdgan_hip import dist as mdist
from test_step_gpu import mid_architecture
torch.cuda.set_device(0)
rank = int(os.environ['RANK'])
mdist.init_process_group(0, backend='gloo')
os.environ['MMDGAN_DP_BUCKET_MB'] = os.environ.get('BUCKET', '0.25')
arch, B, lr = mid_architecture(), 16, (5e-4, 2e-4)
def batch(r, k):
    rs = np.random.RandomState(100 + 10 * r + k)
    return (torch.as_tensor(rs.uniform(-1, 1, (B, 32, 32, 3)).astype(np.float32)).cuda(), torch.as_tensor(rs.randn(B, 64).astype(np.float32)).cuda())
def zr(r, k):
    rs = np.random.RandomState(100 + 10 * r + k)
    z = torch.as_tensor(rs.randn(B, 64).astype(np.float32)).cuda()
    real = torch.as_tensor(rs.uniform(-1, 1, (B, 32, 32, 3)).astype(np.float32)).cuda()
    return real, z
eng = GanEngine(arch, 'rep', lr, batch_size=B, seed=3 + rank, dist_group=dist.group.WORLD)
mdist.broadcast_state(eng, dist.group.WORLD)
eng.step(*zr(rank, 0)); torch.cuda.synchronize()
mid = eng.get_variables()
eng.step(*zr(rank, 1)); torch.cuda.synchronize()
if rank == 0:
    print('buckets', {n.specs[0].scope[:3]: eng._grad_buckets[id(n)] for n in (eng.gen, eng.dis)}, 'arena sizes', eng.gen.arena.size, eng.dis.arena.size)
    local = []
    for r in (0, 1):
        e = GanEngine(arch, 'rep', lr, batch_size=B, seed=3)
        e.set_variables(mid)
        e.step(*zr(r, 1))
        local.append(e.get_variables(grad=True))
    summed = eng.get_variables(grad=True)
    for n, g in summed.items():
        a, b = local[0][n].astype(np.float64), local[1][n].astype(np.float64)
        print('%-28s |g|=%.3e  vs sum %.3e  vs g0 %.3e  vs g1 %.3e  vs 2*sum %.3e' % (n, np.linalg.norm(g), np.linalg.norm(g - a - b) / (np.linalg.norm(a + b) + 1e-30), np.linalg.norm(g - a) / (np.linalg.norm(a) + 1e-30), np.linalg.norm(g - b) / (np.linalg.norm(b) + 1e-30), np.linalg.norm(g - 2 * (a + b)) / (np.linalg.norm(a + b) + 1e-30)))
dist.barrier(); dist.destroy_process_group()
