"""does a blocking (async_op=False) ProcessGroupNCCL collective run on the CURRENT stream or on an internal one?
Issue it on stream S behind a 4 ms spin and see which pool streams' queues are held up meanwhile."""
import os, sys, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
from mmdgan_hip.streams import shares_queue, _calibrate, distinct_queue_streams
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29579')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
CYC = _calibrate(torch.device('cuda', 0))
t = torch.zeros(1 << 20, device='cuda')
dist.all_reduce(t); torch.cuda.synchronize()
pool = [torch.cuda.Stream() for _ in range(8)]
for p in pool:
    with torch.cuda.stream(p): torch.cuda._sleep(10)
torch.cuda.synchronize()
S = distinct_queue_streams(1, torch.device('cuda', 0))[0]
print('S shares with pool:', [shares_queue(S, p) for p in pool])
for mode in ('async_op=False', 'async_op=True'):
    held = []
    for p in pool:
        torch.cuda.synchronize()
        es, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(S):
            es.record(S)
            torch.cuda._sleep(int(4 * CYC))
            w = dist.all_reduce(t, async_op=(mode == 'async_op=True'))
        with torch.cuda.stream(p):
            torch.cuda._sleep(10); eb.record(p)
        torch.cuda.synchronize()
        held.append(es.elapsed_time(eb) > 2.0)
    print(mode, 'pool streams held up:', held)
dist.destroy_process_group()
