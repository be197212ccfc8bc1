"""which of an engine's four streams share a hardware queue, next to its step time.  Two streams share a queue when
a short kernel on one cannot finish while a long spin kernel launched earlier on the other is still running."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
arch, lr = configs.CONFIGS['cifar']()
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)

# calibrate torch.cuda._sleep
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
CYC_PER_MS = 10_000_000 / e0.elapsed_time(e1)
print('sleep cycles per ms: %.0f' % CYC_PER_MS)


def shares_queue(a, b, ms=0.5):
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        es.record(a)
        torch.cuda._sleep(int(ms * CYC_PER_MS))
        ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(10)
        eb.record(b)
    torch.cuda.synchronize()
    return es.elapsed_time(eb) > 0.5 * es.elapsed_time(ea)


def run(eng, N=100):
    for _ in range(10): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


keep = []
for k in range(10):
    eng = GanEngine(arch, 'rep', lr, batch_size=64, seed=0)
    t = run(eng)
    st = [('main', torch.cuda.current_stream()), ('sn0', eng._sn_streams[0]), ('sn1', eng._sn_streams[1]), ('wg', eng._wg_stream)]
    pairs = [a + '=' + b for i, (a, sa) in enumerate(st) for (b, sb) in st[i + 1:] if shares_queue(sa, sb)]
    print('engine %d: %.3f ms/step   shared queues: %s' % (k, t, ' '.join(pairs) or 'none'), flush=True)
    keep.append(eng)
