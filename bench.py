#!/usr/bin/env python3
"""Benchmark of the MMD-GAN hot path on MI355X: one step = one G+D training step (G fwd, D fwd on
[real;fake], SN power iteration, rep loss, both backward passes, both TF-Adam updates, SN/BN state
updates) of the CIFAR-10-shaped 32x32 DCGAN-SN at batch 64 per GPU (BASELINE.json configs[1]),
synthetic data resident in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N --steps K --warmup W          (no launcher: re-executes itself under torch.distributed.run,
                                                            one rank per GPU, loopback rendezvous on a free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0: metric images/sec (whole job), ms_per_step, `roofline` (fp32-MFMA
bound: algorithmic FLOPs over HIP-event time for one launch of the dominant kernel, with HBM bytes from the
committed PMC passes, and `roofline.whole_step` = B*(3 F_G + 7 F_D) over the step time) and `cpu_baseline` (the oracle restatement timed on the
host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'mmd-gan_amd'), ROOT]

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md (f32-input MFMA = vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='cifar', choices=['cifar', 'stl', 'celeba', 'lsun_resnet'])
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: 64; 128 for celeba; 32 for lsun_resnet = 256 on 8 GPUs)')
    ap.add_argument('--loss', default=None, choices=['rep', 'rmb'],
                    help="default: the loss BASELINE.json names for the config ('rmb' for stl, 'rep' otherwise)")
    ap.add_argument('--launch-mode', default='auto', choices=['auto', 'eager', 'graph', 'plan'],
                    help="how a step reaches the GPU (mmdgan_hip/engine.py): 'eager' ~200 library calls from Python, 'graph' "
                         "one captured hipGraph, 'plan' the library's recorded launch plan replayed from one C call; "
                         "'auto' (default) times a few untimed steps of each during warm-up and keeps the fastest")
    ap.add_argument('--no-graph', action='store_true', help="same as --launch-mode eager")
    ap.add_argument('--graph', action='store_true', help="same as --launch-mode graph")
    ap.add_argument('--repeats', type=int, default=5,
                    help='the timed region (exactly --steps steps between barriers) is run this many times; the line reports '
                         'the MEDIAN region and every region\'s ms/step (SURVEY 8(d): median of 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5, help='timed CPU steps at the best thread count (at least 5)')
    ap.add_argument('--probe-only', action='store_true',
                    help='one step, then only the dominant-kernel probe (for rocprofv3: its kernel stats row is then '
                         'exactly the launches that roofline.dominant_kernel times)')
    ap.add_argument('--probe-reps', type=int, default=20)
    ap.add_argument('--probe-warm', type=int, default=400, help='untimed launches before the timed ones of --probe-only')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='bring the process group up (RCCL on GPUs, gloo without), all-reduce one number, print one JSON line '
                         'and stop: what the launch path of --gpus N can be checked with on a box without N GPUs')
    ap.add_argument('--engine', default='auto', choices=['auto', 'tape'],
                    help="'tape': run a DCGAN config on the primitive-op engine too (it is what residual-block configs use)")
    args = ap.parse_args()
    if args.loss is None:
        args.loss = 'rmb' if args.config == 'stl' else 'rep'
    return args


def dominant_kernel_probe(eng, reps=20, warm=3):
    """HIP-event timing of one launch of the kernel with the largest share of the step
    (profiles/*_kernel_stats_single_stream.txt): the input-gradient of D's first 4x4 stride-2 layer over the 3B
    rows of the batched backward pass (CIFAR: D l2, wino2_kernel, 1536 workgroups), issued exactly as the
    engine issues it.  `flops` are the algorithmic FLOPs of that input-gradient (the F(2x2,2x2) kernel issues
    9/16 of them as MFMAs)."""
    from mmdgan_hip import ops
    specs = eng.dis.specs
    li = next((i for i, s in enumerate(specs) if i > 0 and s.op == 'c' and s.R == 4 and s.stride == 2), None)
    if li is None:
        return None
    s, prev = specs[li], specs[li - 1]
    c, h, w = s.in_shape_ref
    B = eng.B
    dz, yprev, dprev = eng.buf[s.scope + '#dz'], eng.buf[prev.scope + '#y'], eng.buf[prev.scope + '#dz']
    wgt, scale = eng.dis.p(s.scope + '/kernel/kernel'), eng._scales[s.scope]
    wino = eng._wino.get(s.scope, (None, None, None))[1]
    flops = 2.0 * 3 * B * (h // 2) * (w // 2) * s.R * s.R * c * s.out

    def launch():
        ops.conv2d_dgrad(dz, wgt, (h, w), s.stride, scale=scale, act=prev.act, dact_of=yprev, dact_batch=2 * B, out=dprev,
                         wino=wino)
    for _ in range(warm):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {'kernel': 'conv2d_dgrad(%s: %dx%dx%d <- %d 4x4/2, %d rows; %s)' % (
                s.scope, h, w, c, s.out, 3 * B, 'wino2_kernel F(2x2,2x2)' if wino is not None else 'library choice'),
            # algorithmic bytes: dy [3B,h/2,w/2,K] + the activations act'(y) for 3B rows + the transformed weights read, dx written
            'alg_bytes_read': 4.0 * (3 * B * (h // 2) * (w // 2) * s.out + 3 * B * h * w * c + 36 * c * s.out),
            'alg_bytes_write': 4.0 * 3 * B * h * w * c,
            'mfma_share': 9.0 / 16 if wino is not None else 1.0,       # F(2x2,2x2): 9 multiplies per 16 algorithmic ones
            'flops': flops, 'ms': ms, 'tflops': flops / ms / 1e9}


def dominant_kernel_probe_tape(eng, reps=20, warm=3):
    """the residual-block configs: the kernel family with the largest share of their step is the Winograd-domain weight
    gradient of the blocks' 3x3 convolutions (profiles/*_resnet_kernel_stats.txt; since round 6 `wino43_wgrad_kernel`).  Timed here on
    the first un-folded 3x3 kernel of D whose channels are tile multiples, at D's batch 2B, on stand-alone buffers of
    its shapes (the kernel's time does not depend on the data).  FLOPs are the algorithmic ones of the weight gradient."""
    from mmdgan_hip import ops
    k = next((k for k in eng.dis.kernels if k.op == 'c' and k.R == 3 and k.stride == 1 and k.fold is None
              and k.kernel_shape[2] % 32 == 0 and k.kernel_shape[3] % 64 == 0), None)
    if k is None:
        return None
    c, h, w = k.in_ref
    n, kout = 2 * eng.B, k.out
    x = torch.randn(n, h, w, c, device='cuda')
    dy = torch.randn(n, h, w, kout, device='cuda')
    dw = torch.empty(3, 3, c, kout, device='cuda')
    flops = 2.0 * n * h * w * 9 * c * kout

    def launch():
        ops.conv2d_wgrad(x, dy, 3, 1, out=dw)
    for _ in range(warm):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    f43 = ops.wgrad_algo(n, h, w, c, kout, 3, 1) == ops.WINO_F43       # (bench.py registers a workspace: the library's own choice)
    which = ('wino43_wgrad_kernel + slab_reduce_kernel F(4x4,3x3)' if f43 else
             ('wino_wgrad_slab_kernel + slab_reduce_kernel' if kout % 128 == 0 else 'wino_wgrad_kernel') + ' F(2x2,3x3)')
    return {'kernel': 'conv2d_wgrad(%s: %dx%dx%d -> %d 3x3, %d rows; %s)' % (k.scope, h, w, c, kout, n, which),
            'alg_bytes_read': 4.0 * n * h * w * (c + kout), 'alg_bytes_write': 4.0 * 9 * c * kout,
            # MFMA multiplies issued per algorithmic one: F(4x4,3x3) 36 per 144, F(2x2,3x3) 16 per 36
            'mfma_share': 36.0 / 144 if f43 else 16.0 / 36,
            'flops': flops, 'ms': ms, 'tflops': flops / ms / 1e9}


def committed_profile(key, per_config=True):
    """the newest profiles/rNN_dominant_kernel_<config>.json (or rNN_<key>.json): {} if this config was never profiled"""
    import glob
    pat = 'r[0-9][0-9]_dominant_kernel_%s.json' % key if per_config else 'r[0-9][0-9]_%s.json' % key
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)))
    if not hits:
        return {}
    with open(hits[-1]) as f:
        d = json.load(f)
    d['_file'] = 'profiles/' + os.path.basename(hits[-1])
    return d


def loss_rel_err(eng, arch, lr, loss, B, when):
    """the second half of BASELINE.json's metric ("MMD-loss rel-err vs ref"): one more step of the timed engine on a known
    batch, against the CPU oracle (the checker, fp64) started from the engine's variables before that step - the losses
    of the HIP path relative to the reference algorithm on identical inputs, at the bench configuration itself.
    An MMD loss is a difference of kernel means (e_kxx + e_kyy - 2 e_kxy): `conditioning` = |loss_gen| / (e_kxx + e_kyy +
    2 e_kxy) says how much of the means cancels at this point of the training, i.e. by how much the relative error of the
    loss exceeds that of the means (`kernel_means`), which is what fp32 arithmetic bounds."""
    from oracle import restatement as R
    ora = R.OracleGan(arch, loss, tuple(lr), dtype=torch.float64, params=eng.get_variables())
    rs = np.random.RandomState(4321)
    c, h, w = arch['input'][0]
    z = rs.randn(B, arch['code'][0][0]).astype(np.float32)
    real = rs.uniform(-1, 1, (B, c, h, w)).astype(np.float32)
    with torch.no_grad():
        lg, ld, stats, _, _ = ora.forward_losses(torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64))
    eng.step(torch.as_tensor(np.ascontiguousarray(real.transpose(0, 2, 3, 1))).cuda(), torch.as_tensor(z).cuda())
    got = eng.losses.cpu().numpy().astype(np.float64)
    out = {'loss_gen': abs(got[0] - float(lg)) / abs(float(lg)), 'loss_dis': abs(got[1] - float(ld)) / abs(float(ld)),
           'values': {'loss_gen': [float(got[0]), float(lg)], 'loss_dis': [float(got[1]), float(ld)]}, 'when': when}
    if all(k in stats for k in ('kxx', 'kxy', 'kyy')) and got.shape[0] >= 5:
        means = {k: float(stats[k]) for k in ('kxx', 'kxy', 'kyy')}
        out['kernel_means'] = {k: abs(got[2 + i] - means[k]) / abs(means[k]) for i, k in enumerate(('kxx', 'kxy', 'kyy'))}
        out['conditioning'] = abs(float(lg)) / (means['kxx'] + means['kyy'] + 2 * means['kxy'])
    return out


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(arch, lr, loss, B, steps, budget_s=60.0):
    """the oracle restatement (fp32 torch-CPU) of the same step on the host cores ("N = all host cores and
    N = 1, core count and CPU model printed"): a reported baseline, not the optimisation target.  A batch of 64 32x32
    images does not scale over torch-CPU's thread pool (128 threads measured SLOWER than one), so "all cores" alone is
    not a baseline: the step is timed at threads in {1, 8, 16, 32, 64, all} - one WARMED step each (a first step at a
    new thread count pays oneDNN's primitive creation and the allocator: un-warmed sweeps picked different winners from
    run to run), cheapest first, while the sample stays bounded (`budget_s`).  The best count then runs `steps` (>= 5)
    more steps, each timed on its own: `value` is B over the MEDIAN step, `best` B over the fastest."""
    from oracle import restatement as R
    threads = torch.get_num_threads()
    rs = np.random.RandomState(1234)
    real = torch.tensor(rs.uniform(-1, 1, (B,) + tuple(arch['input'][0])).astype(np.float32))
    z = torch.tensor(rs.randn(B, arch['code'][0][0]).astype(np.float32))

    def timed(n_threads, n_steps, warm):
        """seconds of each of n_steps steps after `warm` untimed ones, and what the warm-up cost"""
        torch.set_num_threads(n_threads)
        try:
            gan = R.OracleGan(arch, loss, tuple(lr), dtype=torch.float32, seed=0)
            t0 = time.perf_counter()
            for _ in range(warm):
                gan.step(z, real)                       # warm-up (allocator, oneDNN primitive cache)
            t_warm = time.perf_counter() - t0
            out = []
            for _ in range(n_steps):
                t0 = time.perf_counter()
                gan.step(z, real)
                out.append(time.perf_counter() - t0)
            return out, t_warm
        finally:
            torch.set_num_threads(threads)
    what = 'G+D steps of the same %dx%d B=%d workload, oracle/restatement.py fp32 on torch-CPU' % (
        arch['input'][0][1], arch['input'][0][2], B)
    t_start = time.perf_counter()
    counts = sorted({n for n in (1, 8, 16, 32, 64, threads) if n <= threads}, key=lambda n: (n == 1, -n))   # one thread last: the slowest
    sweep, last = {}, 0.0
    for n in counts:
        if time.perf_counter() - t_start + 1.5 * last > budget_s and sweep:
            sweep[n] = None                              # not run: the sample would leave its bound
            continue
        t0 = time.perf_counter()
        (dt1,), _ = timed(n, 1, 1)
        last = time.perf_counter() - t0
        sweep[n] = B / dt1
    best = max((n for n in sweep if sweep[n]), key=lambda n: sweep[n])
    steps = max(5, steps)
    per_step, _ = timed(best, steps, 1)
    med = float(np.median(per_step))
    return {'value': B / med, 'unit': 'images/sec', 'cores': best, 'kind': 'port', 'cpu_model': cpu_model(),
            'host_cpus': os.cpu_count(), 'torch_threads_default': threads,
            'best': B / min(per_step), 'steps_ms': [round(t * 1e3, 1) for t in per_step],
            'sample': '%d %s, each timed on its own after one warm step, %d threads = the best of the sweep; value = B / '
                      'median step (%.3f s), best = B / fastest step; sweep = one warmed step per thread count'
                      % (steps, what, best, med),
            'thread_sweep': {str(n): (None if v is None else round(v, 2)) for n, v in sorted(sweep.items())},
            'all_threads': None if sweep.get(threads) is None else {'value': sweep[threads], 'cores': threads},
            'single_thread': None if sweep.get(1) is None else {'value': sweep[1], 'unit': 'images/sec', 'cores': 1,
                                                                 'sample': '1 %s, 1 thread, after one warm step' % what}}


def kernel_set_check(eng, config, loss, B):
    """do the kernels of THIS run's step equal the ones the parity tests ran (tests/test_production_gpu.py)?  One step
    is recorded as a launch plan and its kernel list (mmdgan_plan_describe) compared with the committed
    tests/golden/production_kernels.json - the list those tests assert, under the same default environment."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import shipped_step
    expected = shipped_step.expected_kernels(config, loss, B)
    keep = eng.launch_mode
    try:
        eng.launch_mode = 'plan'
        for _ in range(2):
            eng.step()
        got = shipped_step.kernel_multiset(eng.plan_kernels())
    finally:
        eng.launch_mode = keep
    env = sorted(k for k in os.environ if k.startswith('MMDGAN_'))
    diff = None if expected is None else sorted(k for k in set(got) | set(expected) if got.get(k) != expected.get(k))
    return {'launches_per_step': sum(got.values()), 'distinct': len(got), 'fixture': 'tests/golden/production_kernels.json',
            'case': shipped_step.case_key(config, loss, B), 'equals_tested_set': None if expected is None else not diff,
            'differs_in': diff[:8] if diff else None, 'mmdgan_env': env}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it (the way the driver calls N = 1): the same command line
    again under torch.distributed.run, one rank per GPU of this node, rendezvous over loopback on a port the kernel just
    handed out.  The ranks inherit stdout, so rank 0's JSON line is this process's one line; the exit code is theirs."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC: what RCCL needs between the ranks of one node here
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    from mmdgan_hip import dist as mdist
    mdist.pin_loopback(env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rendezvous_only(args, world, rank, local_rank):
    """--rendezvous-only: the process group of an N-rank run comes up and carries one collective"""
    import torch.distributed as dist
    from mmdgan_hip import dist as mdist
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)                                        # gloo / RCCL banners go to stderr: stdout carries one line
    gpu = torch.cuda.is_available() and torch.cuda.device_count() > local_rank
    if gpu:
        torch.cuda.set_device(local_rank)
    group = mdist.init_process_group(local_rank, backend='nccl' if gpu else 'gloo')
    t = torch.tensor([float(rank + 1)], device='cuda' if gpu else 'cpu')
    dist.all_reduce(t, group=group)
    ok = float(t.item()) == world * (world + 1) / 2
    backend = dist.get_backend(group)
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps({'rendezvous': 'ok' if ok else 'wrong sum', 'n_gpus': world, 'asked': args.gpus,
                          'backend': backend, 'sum': float(t.item())}), flush=True)
    sys.exit(0 if ok else 1)


def main():
    args = parse()
    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if not launched and (args.gpus > 1 or os.environ.get('MMDGAN_DP_FORCE') == '1'):
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        sys.exit('bench.py --gpus %d under a launcher that started %d rank(s): one rank per GPU' % (args.gpus, world))
    if args.rendezvous_only:
        rendezvous_only(args, world, rank, local_rank)
    torch.cuda.set_device(local_rank)
    group = None
    # stdout carries ONE JSON line.  RCCL prints a version banner to stdout when its first communicator comes up (rank 0), so
    # until the line is due file descriptor 1 points at stderr
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(_real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)
    # MMDGAN_DP_FORCE=1 under torch.distributed.run with one rank: time the exchange plumbing on a 1-GPU box
    force_dp = os.environ.get('MMDGAN_DP_FORCE') == '1' and 'RANK' in os.environ
    if world > 1 or force_dp:
        from mmdgan_hip import dist as mdist
        group = mdist.init_process_group(local_rank)     # RCCL over xGMI, its stream on a hardware queue of its own

    import configs
    from mmdgan_hip.engine import GanEngine
    arch, lr = configs.CONFIGS[args.config]()
    B = args.batch or {'celeba': 128, 'lsun_resnet': 32}.get(args.config, 64)
    from mmdgan_hip.tape import TapeEngine, has_residual_blocks
    tape = has_residual_blocks(arch) or args.engine == 'tape'   # residual blocks: the primitive-op engine (eager or plan)
    if tape:
        GanEngine = TapeEngine                           # noqa: F811
    # the engine starts in eager mode (its lazily-created buffers are then allocated on the stream that uses
    # them); the hipGraph, if wanted, is captured after the warm-up steps
    eng = GanEngine(arch, args.loss, lr, batch_size=B, seed=0, dist_group=group)
    if group is not None:
        from mmdgan_hip import dist as mdist
        mdist.broadcast_state(eng, group)                # identical weights / SN vectors on every replica
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234 + rank)
    c, h, w = arch['input'][0]
    real = torch.empty(B, h, w, c, device='cuda').uniform_(-1, 1, generator=gen)    # synthetic, resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.probe_only:
        eng.launch_mode = 'eager'
        eng.step(real)                                   # fills the activations / gradients the probe reads
        torch.cuda.synchronize()
        # 400 untimed launches first (~40 ms): the shader clock and the power state need that long to settle - measured
        # 107-110 us per launch right after start-up against 97-98 us for the same launch at the end of a bench run
        probe = (dominant_kernel_probe_tape if tape else dominant_kernel_probe)(eng, reps=args.probe_reps, warm=args.probe_warm)
        emit(json.dumps({'dominant_kernel': probe, 'reps': args.probe_reps, 'warm': args.probe_warm}))
        return
    # MMD-loss rel-err vs ref, untimed: at the engine's 5th step from its seeded initial variables (at the initial variables
    # themselves D's scores are ~1e-5 and both losses are exactly 0 on both sides; a few steps in they are O(0.1) and neither
    # degenerate nor cancelling), and once more after the whole run
    rel_err, early = None, min(4, args.warmup)
    for _ in range(early):
        eng.step(real)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        rel_err = loss_rel_err(eng, arch, lr, args.loss, B, 'step %d from the seeded initial variables' % (early + 1))
    for _ in range(args.warmup - early):
        eng.step(real)
    barrier()
    want = 'eager' if args.no_graph else 'graph' if args.graph else args.launch_mode
    if tape and (want == 'graph' or (group is not None and getattr(eng, '_dp_backend', 'torch') != 'capi')):
        want = 'eager'                                   # the primitive-op engine: no hipGraph capture; under data parallelism a plan
        #                                                  needs the library-owned exchange (collectives as plan nodes)
    elif group is not None and want == 'graph':
        want = 'eager'                                   # collectives between the launches: no hipGraph
    trial = {}
    if want == 'auto':
        # untimed: a few steps each way, keep the fastest launch mode (data-parallel: eager or plan, the slowest rank's
        # time decides so that every rank makes the same choice)
        modes = ('eager', 'plan') if (group is not None or tape) else ('eager', 'graph', 'plan')
        passes = {m: [] for m in modes}
        for m in modes + modes:                          # two passes: one host hiccup must not pick the mode
            eng.launch_mode = m
            for _ in range(10):
                eng.step(real)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):                          # (25 steps under-measured the replay modes: a fixed ~1 ms start-up
                eng.step(real)                           # per timed block - the same sequence at 100 steps: plan = eager)
            torch.cuda.synchronize()
            passes[m].append((time.perf_counter() - t0) / 60)
        for m in modes:
            if group is not None:
                import torch.distributed as dist
                t = torch.tensor(passes[m], device='cuda', dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                passes[m] = [float(v) for v in t.tolist()]
            trial[m] = min(passes[m])
        # plan replay is ONE host call per step, eager issue ~90: where they tie on the GPU's clock (they do, to 0.2 %), the
        # replay is the one a busy host cannot stretch - round 5's driver line had one region of five at +11 % under eager
        # issue.  Another mode is taken only where it beats the plan by more than 1 % in BOTH passes.
        want = 'plan'
        better = [m for m in modes if m != 'plan' and all(a < 0.99 * b for a, b in zip(passes[m], passes['plan']))]
        if better:
            want = min(better, key=trial.get)
        if os.environ.get('BENCH_VERBOSE'):
            print('launch-mode trial (ms/step):', {k: [round(v * 1e3, 3) for v in vs] for k, vs in passes.items()}, file=sys.stderr)
    mode = want
    eng.launch_mode = mode
    for _ in range(3):                                   # capture / record outside the timed region
        eng.step(real)
    # the timed region: EXACTLY args.steps steps between barrier + synchronize on both sides, max over ranks; run
    # args.repeats times back to back, the line reports the median region (and every region's ms/step)
    regions = []
    for _ in range(max(1, args.repeats)):
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            eng.step(real)
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1) / args.steps
        if group is not None:
            import torch.distributed as dist
            t = torch.tensor([dt], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        regions.append((dt, ev_ms))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, ev_ms = regions[order[(len(order) - 1) // 2]]                     # the median region (lower middle for an even count)
    losses = eng.losses.cpu().numpy()
    assert np.all(np.isfinite(losses)), 'Model diverged with loss = NaN'               # graph_func.py:856
    # data parallel: how long the LAST exchange bucket (G's first layer - the only one no backward kernel is left to hide)
    # takes, and how much of it the main stream actually waits for; untimed eager steps after the timed regions
    exchange = None
    if group is not None and hasattr(eng, 'exchange_probe'):
        keep = eng.launch_mode
        eng.launch_mode, eng.exchange_probe = 'eager', {}
        took, exposed = [], []
        for _ in range(12):
            eng.step(real)
            torch.cuda.synchronize()
            pr = eng.exchange_probe
            if all(k in pr for k in ('start', 'end', 'main_ready')):
                took.append(pr['start'].elapsed_time(pr['end']))
                exposed.append(max(0.0, pr['main_ready'].elapsed_time(pr['end'])))
        if took:
            exchange = {'last_bucket_bytes': eng.exchange_probe.get('bucket_bytes'), 'last_bucket_ms': float(np.median(took[2:])),
                        'exposed_ms': float(np.median(exposed[2:])), 'buckets_per_step': sum(len(b) for b in eng._grad_buckets.values()),
                        'note': 'eager steps after the timed regions; exposed = end of the last bucket minus the point where the '
                                'main stream has only Adam left'}
        eng.launch_mode, eng.exchange_probe = keep, None

    if rank == 0:
        fg, fd = configs.flops_per_image(arch)
        flops_step = B * (3.0 * fg + 7.0 * fd)
        ms_per_step = dt / args.steps * 1e3
        achieved = flops_step / (ev_ms * 1e-3) / 1e12
        out = {
            'metric': 'images/sec/node (G+D step), CIFAR-10 32x32 B=64' if args.config == 'cifar'
                      else 'images/sec/node (G+D step), %s B=%d' % (args.config, B),
            'value': B * world * args.steps / dt, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
            'repeats': len(regions), 'ms_per_step_regions': [round(r[0] / args.steps * 1e3, 4) for r in regions],
            'ms_per_step_min': round(min(r[0] for r in regions) / args.steps * 1e3, 4),
            'ms_per_step_spread': round((max(r[0] for r in regions) - min(r[0] for r in regions)) / args.steps * 1e3, 4),
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s %dx%d %s, batch %d per GPU, %s loss, lr %g/%g, TF-Adam, one G+D step'
                                   % (args.config, h, w, ('ResNet-SN' if has_residual_blocks(arch) else 'DCGAN-SN (primitive-op engine)') if tape else 'DCGAN-SN',
                                      B, args.loss, lr[0], lr[1]),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'launch_mode': mode,
                       'dp_backend': getattr(eng, '_dp_backend', None) if group is not None else None,
                       'exchange': exchange,
                       'launch_mode_trial_ms': {k: round(v * 1e3, 4) for k, v in trial.items()} or None},
            'loss_gen': float(losses[0]), 'loss_dis': float(losses[1]),
        }
        # (algorithmic FLOPs: with Winograd kernels in the step this fraction can pass 1 - the share of SIMD cycles the MFMA
        # pipes are busy is `mfma_busy_frac`, from the committed counter pass)
        whole = {'achieved_algorithmic': achieved, 'frac_algorithmic': achieved / PEAK_FP32_MFMA_TFLOPS,
                 'scope': 'whole step: B*(3*F_G+7*F_D) = %.1f GFLOP algorithmic over the HIP-event step time %.3f ms'
                          % (flops_step / 1e9, ev_ms)}
        probe = dominant_kernel_probe_tape(eng, reps=args.probe_reps) if tape else dominant_kernel_probe(eng, reps=args.probe_reps)
        committed = committed_profile(args.config)
        if probe:
            # the roofline object is about the dominant kernel: algorithmic FLOPs of one launch over its HIP-event time,
            # measured live here.  Beside it, from the committed rocprofv3 evidence of THIS config (profiles/, written by
            # tools/collect_profiles.sh + tools/make_profiles.py): the same launch's average duration in the kernel trace
            # and the fraction that follows from it (the tracer costs this kernel ~1 %, an un-traced HIP-event run is the
            # `frac` above), and its HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.  No committed profile for
            # a config -> traffic null, never another config's number.
            share = probe.get('mfma_share', 1.0)
            out['roofline'] = {'bound': 'mfma', 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               # the Winograd kernels issue only `mfma_share` of the convolution's algorithmic FLOPs as MFMAs.
                               # `achieved` / `frac` are what the MFMA pipe actually does (<= 1 by construction);
                               # `achieved_algorithmic` / `frac_algorithmic` count the convolution's own FLOPs and can pass 1
                               'achieved': probe['tflops'] * share, 'frac': probe['tflops'] * share / PEAK_FP32_MFMA_TFLOPS,
                               'achieved_algorithmic': probe['tflops'],
                               'frac_algorithmic': probe['tflops'] / PEAK_FP32_MFMA_TFLOPS,
                               'mfma_share': share,
                               'traffic': committed.get('hbm_bytes_per_launch'),
                               'traffic_over_algorithmic': committed.get('traffic_over_algorithmic'),
                               'algorithmic_bytes_per_launch': (committed.get('algorithmic_bytes_per_launch') or {}).get('total'),
                               'traffic_source': committed.get('_file'),
                               'kernel': probe['kernel'], 'gflop_per_launch': probe['flops'] / 1e9,
                               'ms_per_launch': probe['ms'],
                               'profiled': None if not committed else {
                                   'avg_us': committed['avg_us_profiled'], 'frac': committed['frac_from_profiled_duration'] * share,
                                   'frac_algorithmic': committed['frac_from_profiled_duration'],
                                   'launches': committed['probe_launches_isolated'], 'source': committed['_file']},
                               'whole_step': whole}
        else:
            out['roofline'] = dict(whole, bound='mfma', peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', traffic=None,
                                   achieved=None, frac=None)
        busy = committed_profile('whole_step_mfma_busy', per_config=False)
        if busy and args.config == 'cifar':
            out['roofline']['whole_step']['mfma_busy_frac'] = busy.get('mfma_busy_frac_whole_step')
            out['roofline']['whole_step']['mfma_busy_source'] = busy['_file']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(arch, lr, args.loss, B, 1 if tape else args.cpu_steps)
            rel_err['after_the_timed_steps'] = loss_rel_err(eng, arch, lr, args.loss, B, 'after every step of this run (%d-odd steps on one synthetic batch)' % (args.warmup + args.repeats * args.steps))
            rel_err.update({'vs': 'oracle/restatement.py in fp64 from the engine\'s variables, same z and batch (B=%d)' % B, 'bar': 1e-4})
            # the bar, applied to BOTH points.  At step 5 the losses themselves are held to 1e-4.  After the run the criterion is
            # the one fp32 arithmetic can meet wherever the training went: every kernel mean within 1e-4, and the loss within
            # 1e-4 / conditioning (a loss that is `conditioning` of the means it is the difference of carries their error
            # divided by it)
            post = rel_err['after_the_timed_steps']
            rel_err['pass'] = bool(rel_err['loss_gen'] <= 1e-4 and rel_err['loss_dis'] <= 1e-4)
            if 'kernel_means' in post:
                cond = max(min(post['conditioning'], 1.0), 1e-12)
                post['pass'] = bool(max(post['kernel_means'].values()) <= 1e-4 and post['loss_gen'] <= 1e-4 / cond)
            else:
                post['pass'] = bool(post['loss_gen'] <= 1e-4 and post['loss_dis'] <= 1e-4)
            out['mmd_loss_rel_err'] = rel_err
            gate_failed = not (rel_err['pass'] and post['pass'])
        if world == 1 and hasattr(eng, 'plan_kernels'):
            out['config']['kernel_set'] = kernel_set_check(eng, args.config, args.loss, B)
        # the switches this line was measured under: the library's kernel selection (csrc/tuning.h) and the host side's schedule
        # (mmdgan_hip/settings.py) - only what is OFF its default; both empty = the configuration the parity tests pin
        from mmdgan_hip import ops as _ops, settings as _settings
        out['config']['switches'] = {'library': {k: v for k, (v, dflt) in _ops.tuning().items() if not dflt},
                                     'host': _settings.describe(), 'unknown': _settings.unknown()}
        emit(json.dumps(out))
        if world == 1 and not args.no_cpu_baseline and gate_failed:
            sys.exit('bench.py: mmd_loss_rel_err is above its bar (see the JSON line): %r' % (rel_err,))
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
