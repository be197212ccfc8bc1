"""does the step time depend on where the caching allocator happens to put the engine's buffers?  Builds the same
engine several times in one process with differently-sized dummy allocations in front of it."""
import os, sys, time, gc, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]
import configs
from mmdgan_hip.engine import GanEngine
arch, lr = configs.CONFIGS['cifar']()
real = torch.empty(64, 32, 32, 3, device='cuda').uniform_(-1, 1)


def run(eng, N=100):
    for _ in range(10): eng.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.step(real)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


keep = []
for tag, nbytes in (('first', 0), ('second', 0), ('+1 MiB small-pool junk', 1 << 20), ('+3 MiB', 3 << 20), ('+20 MiB', 20 << 20),
                    ('+100 MiB', 100 << 20), ('+1 GiB', 1 << 30), ('again', 0)):
    if nbytes:
        keep.append(torch.empty(nbytes, dtype=torch.uint8, device='cuda'))
    eng = GanEngine(arch, 'rep', lr, batch_size=64, seed=0)
    t = [run(eng) for _ in range(2)]
    ptrs = [eng.gen.params.data_ptr(), eng.dis.params.data_ptr(), eng.gen.grads.data_ptr(), eng.dis.grads.data_ptr()]
    print('%-26s %.3f %.3f ms/step   arenas at %s' % (tag, t[0], t[1], ' '.join('%x' % p for p in ptrs)), flush=True)
    keep.append(eng)          # keep it alive so the next one lands elsewhere
for i, e in enumerate(keep):
    if isinstance(e, GanEngine):
        print('re-run engine %d: %.3f' % (i, run(e)), flush=True)
