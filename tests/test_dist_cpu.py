"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise the same bucketed gradient
averaging / state broadcast / shard helpers the GPU engine uses with RCCL (mmdgan_hip/dist.py is
device-agnostic; only the kernels need a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from mmdgan_hip import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    mdist.init_process_group(backend='gloo', rank=rank, world_size=world)       # (pins the bootstrap to `lo`: pin_loopback)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(100003, generator=g)                      # a gradient arena, different per replica
        mine = flat.clone()
        # bucketed async SUM + scale == mean over replicas (what Adam's grad_scale applies)
        works = mdist.allreduce_sum_async(flat, bucket_bytes=64 << 10)
        assert len(works) == len(mdist.buckets(flat.numel(), 64 << 10)) == 7
        mdist.wait_all(works)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        tdist.all_gather(gathered, mine)
        ref = torch.stack(gathered).sum(0)
        ok_sum = torch.allclose(flat, ref, rtol=0, atol=1e-5)
        # the blocking variant the engine issues on its own stream: same buckets, same sums
        ok_sum = ok_sum and torch.allclose(mdist.allreduce_sum_(mine.clone(), bucket_bytes=64 << 10), ref, rtol=0, atol=1e-5)
        avg = mdist.average_(mine.clone(), bucket_bytes=1 << 20)
        ok_avg = torch.allclose(avg, ref / world, rtol=0, atol=1e-5)

        # broadcast_state: every replica ends with rank 0's weights / moments / SN vectors
        class Opt:
            step_counter = torch.tensor([3 + rank], dtype=torch.int32)

        class Net:
            def __init__(self):
                self.params = torch.full((17,), float(rank))
                self.adam_m = torch.full((17,), 10.0 + rank)
                self.adam_v = torch.full((17,), 20.0 + rank)
                self.opt = Opt()
                self.state = {'b': torch.full((3,), 5.0 + rank), 'a': torch.full((2,), 7.0 + rank)}

        class Eng:
            gen, dis = Net(), Net()
        eng = Eng()
        mdist.broadcast_state(eng)
        ok_bc = all(float(n.params[0]) == 0.0 and float(n.adam_m[0]) == 10.0 and float(n.state['a'][0]) == 7.0
                    and int(n.opt.step_counter[0]) == 3 for n in (eng.gen, eng.dis))
        # the engine's per-layer exchange: the arena cut into buckets in backward order, each bucket all-reduced on its own
        # as the backward pass reaches its lowest layer == one all-reduce of the whole arena
        sizes = [7, 8192, 600, 20000, 36, 65000, 6268]                 # layers in forward order (sum = 100003)
        ranges, o = [], 0
        for n in sizes:
            ranges.append((o, o + n))
            o += n
        bl = mdist.layer_buckets(ranges, 16384)
        arena = mine.clone()
        for li, lo, hi in bl:                                           # issue order = backward order
            tdist.all_reduce(arena[lo:hi], op=tdist.ReduceOp.SUM)
        ok_sum = ok_sum and torch.allclose(arena, ref, rtol=0, atol=1e-5) and [b[0] for b in bl] == sorted([b[0] for b in bl], reverse=True)
        q.put((rank, ok_sum, ok_avg, ok_bc, mdist.shard_of(50000, rank, world)))
    finally:
        tdist.destroy_process_group()


def test_gloo_world2_gradient_average_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1:4] for r in results] == [(True, True, True)] * world
    assert results[0][4] == (0, 25000) and results[1][4] == (25000, 50000)


def test_layer_buckets_tile_the_arena_in_backward_order():
    ranges = [(0, 7), (8, 8200), (8200, 8800), (8800, 28800), (28800, 28836), (28836, 93836), (93836, 100104)]
    for target in (1, 5000, 16384, 10 ** 9):
        bl = mdist.layer_buckets(ranges, target)
        assert bl[0][2] == 100104 and bl[-1][1] == 0 and bl[-1][0] == 0                 # from the arena's end down to layer 0
        assert all(a[1] == b[2] or (a[1], b[2]) == (8, 7) for a, b in zip(bl, bl[1:]))     # contiguous (alignment gaps stay inside)
        assert [b[0] for b in bl] == sorted((b[0] for b in bl), reverse=True)
        assert all(hi - lo >= target or li == 0 for li, lo, hi in bl)
    assert len(mdist.layer_buckets(ranges, 10 ** 9)) == 1 and len(mdist.layer_buckets(ranges, 1)) == len(ranges)
    # the CIFAR nets of the shipped config: D [l8+l7 | l6 | l5..l1], G [l5..l2 | l1] at the default 8 MiB
    import configs
    from mmdgan_hip.engine import build_specs
    arch, _ = configs.cifar()
    for key, in_shape, want in (('discriminator', [3, 32, 32], [6, 5, 0]), ('generator', [128], [1, 0])):
        specs = build_specs(arch[key], in_shape, key[:3])
        sizes = [int(__import__('numpy').prod(s.kernel_shape)) + (s.channels if s.has_bias else 0) + (2 * s.channels if s.bn else 0)
                 for s in specs]
        rng, o = [], 0
        for n in sizes:
            rng.append((o, o + n))
            o += n
        assert [b[0] for b in mdist.layer_buckets(rng, (8 << 20) // 4)] == want, key


def test_buckets_cover_exactly():
    for n in (1, 7, 4096, 100003):
        b = mdist.buckets(n, 4096)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all(e - s <= 1024 for s, e in b)
    assert mdist.shard_of(10, 0, 3) == (0, 4) and mdist.shard_of(10, 2, 3) == (7, 10)


def test_one_node_rendezvous_is_pinned_to_loopback_by_default_only():
    """mdist.pin_loopback: a loopback rendezvous address names `lo` for gloo's and RCCL's bootstrap, a user's choice and a
    multi-node address are left alone."""
    if not os.path.isdir('/sys/class/net/lo'):
        pytest.skip('no loopback interface here')
    env = {'MASTER_ADDR': '127.0.0.1'}
    assert mdist.pin_loopback(env) == ['GLOO_SOCKET_IFNAME', 'NCCL_SOCKET_IFNAME'] and env['NCCL_SOCKET_IFNAME'] == 'lo'
    env = {'MASTER_ADDR': 'localhost', 'NCCL_SOCKET_IFNAME': 'eth0'}
    assert mdist.pin_loopback(env) == ['GLOO_SOCKET_IFNAME'] and env['NCCL_SOCKET_IFNAME'] == 'eth0'
    env = {'MASTER_ADDR': '10.0.0.7'}
    assert mdist.pin_loopback(env) == [] and 'GLOO_SOCKET_IFNAME' not in env


def test_bench_gpus_2_launches_itself_and_reaches_rendezvous():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver calls it): re-executes itself under
    torch.distributed.run, both ranks meet (gloo here - no GPU) and carry one all-reduce; stdout is ONE JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rendezvous-only'], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {'rendezvous': 'ok', 'n_gpus': 2, 'asked': 2, 'backend': 'gloo', 'sum': 3.0}


def test_init_process_group_leaves_foreign_rendezvous_alone(monkeypatch):
    """the loopback pin applies to a loopback rendezvous only - an init_method / store rendezvous (possibly on
    another host) must not get GLOO/NCCL_SOCKET_IFNAME=lo"""
    for k in ('MASTER_ADDR', 'GLOO_SOCKET_IFNAME', 'NCCL_SOCKET_IFNAME', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    seen = {}
    monkeypatch.setattr(tdist, 'init_process_group', lambda backend, **kw: seen.update(backend=backend, **kw))
    mdist.init_process_group(backend='gloo', init_method='tcp://10.1.2.3:29500', rank=0, world_size=2)
    assert seen['init_method'].startswith('tcp://10.') and 'GLOO_SOCKET_IFNAME' not in os.environ and 'MASTER_ADDR' not in os.environ
    monkeypatch.setenv('WORLD_SIZE', '16')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    mdist.init_process_group(backend='gloo')
    assert 'GLOO_SOCKET_IFNAME' not in os.environ
    monkeypatch.setenv('WORLD_SIZE', '8')
    mdist.init_process_group(backend='gloo')
    assert os.environ.get('GLOO_SOCKET_IFNAME') == 'lo'
    monkeypatch.delenv('GLOO_SOCKET_IFNAME')
    monkeypatch.delenv('NCCL_SOCKET_IFNAME', raising=False)
