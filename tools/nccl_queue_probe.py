"""one-rank RCCL group: which hardware queue does ProcessGroupNCCL's internal stream land on (normal vs
high-priority option), relative to the default stream and torch's pool streams?"""
import os, sys, time, torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29578')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
hp = len(sys.argv) > 1 and sys.argv[1] == 'hp'
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=hp)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0), pg_options=opts)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
CYC = 10_000_000 / e0.elapsed_time(e1)
src = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
dst = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
dist.all_gather_into_tensor(dst, src); torch.cuda.synchronize()
t0 = time.perf_counter(); dist.all_gather_into_tensor(dst, src); torch.cuda.synchronize()
print('1-rank all_gather of 1 GiB: %.3f ms' % ((time.perf_counter() - t0) * 1e3))


def shares_queue(a, b, ms=0.5):
    torch.cuda.synchronize()
    es, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        es.record(a); torch.cuda._sleep(int(ms * CYC)); ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(10); eb.record(b)
    torch.cuda.synchronize()
    return es.elapsed_time(eb) > 0.5 * es.elapsed_time(ea)


idle = torch.cuda.Stream()                          # the collective is issued from here, behind a long spin


def shares_with_nccl(x, ms=4.0):
    """RCCL's stream waits for `idle` (which spins for `ms`), so its queue is blocked that long"""
    torch.cuda.synchronize()
    es, eb, ee = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(idle):
        es.record(idle)
        torch.cuda._sleep(int(ms * CYC))
        w = dist.all_gather_into_tensor(dst[:1 << 20], src[:1 << 20], async_op=True)
    with torch.cuda.stream(x):
        torch.cuda._sleep(10); eb.record(x)
    w.wait(); ee.record(); torch.cuda.synchronize()
    return es.elapsed_time(eb) > 0.5 * ms, es.elapsed_time(eb), es.elapsed_time(ee)


main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(8)]
hi = torch.cuda.Stream(priority=-1)
print('high-priority option:', hp)
for name, s in [('main', main)] + [('pool%d' % i, p) for i, p in enumerate(pool)] + [('hi-prio', hi)]:
    shares_with_nccl(s)                               # warm-up: queue creation takes milliseconds
    sh = shares_with_nccl(s)
    print('%-8s shares queue with: main=%s idle=%s nccl=%s (tiny done at %.3f of %.3f ms)' % (
        name, name != 'main' and shares_queue(main, s), shares_queue(idle, s), sh[0], sh[1], sh[2]))
dist.destroy_process_group()
