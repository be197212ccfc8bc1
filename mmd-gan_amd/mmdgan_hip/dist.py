"""Data-parallel replicas: one process per GPU, gradients averaged with RCCL over xGMI.

The reference is single-GPU; its dormant tower helper (graph_func.py:69-94, SynTower.average_grads)
defines the semantics kept here: every replica computes loss and gradients on its own batch (the
MMD statistic stays per-replica), gradients are AVERAGED, variables are
shared.  SN vectors and BN moving statistics need no communication: they are deterministic
functions of identical weights / per-replica statistics.

Gradients live in one flat fp32 arena per network, cut into buckets in backward order.  The ENGINES exchange per-layer
groups of MMDGAN_DP_BUCKET_MB = 8 MB (layer_buckets below; engine.py:_make_buckets) as soon as a group's lowest layer has its
gradients, underneath the backward kernels still to come: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring
all-reduce is bound by one link, and below a few MB a collective is latency - 8 MB is where the two meet for these nets
(DESIGN.md section 6 has the bucket table and the predicted exposure).  DEFAULT_BUCKET_BYTES (32 MiB) is only the default of
the stand-alone helper allreduce_sum_async for callers that exchange a whole arena after the pass.  The functions here are
device-agnostic (tests run them on CPU tensors over gloo).
"""
import torch
import torch.distributed as tdist

DEFAULT_BUCKET_BYTES = 32 << 20


def init_process_group(local_rank=None, backend='nccl', **kwargs):
    """torch.distributed.init_process_group for one replica per GPU (RANK / WORLD_SIZE / MASTER_* from the
    environment, as torch.distributed.run sets them).  RCCL's stream stays at normal priority: a high-priority
    stream does get a hardware queue of its own, but with it the same step measured 3.10 instead of 2.52 ms
    (tools/dp_probe.py, one rank) - the engine instead issues its collectives on a stream it picked itself."""
    import os
    # pin the bootstrap to `lo` only for a rendezvous that IS this host's loopback: the address the launcher / user set, or the
    # default taken here when nothing names one.  A caller that rendezvouses through init_method= / store= (possibly across
    # hosts), or a job with more ranks than this node holds, is left alone.
    elsewhere = ('init_method' in kwargs or 'store' in kwargs
                 or int(os.environ.get('WORLD_SIZE', '1')) > int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    if not elsewhere:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        pin_loopback()
    if backend != 'nccl':
        tdist.init_process_group(backend, **kwargs)
        return tdist.group.WORLD
    if local_rank is None:
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    tdist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), **kwargs)
    return tdist.group.WORLD


def pin_loopback(environ=None):
    """a ONE-NODE job (the rendezvous address is this host's loopback) bootstraps over `lo`: without an interface named, gloo
    resolves the host name and RCCL walks the other interfaces first - on a box whose name does not resolve or whose outer
    interface answers slowly that is minutes before the first collective (round 5 saw 300 s and more, twice).  Only defaults:
    GLOO_SOCKET_IFNAME / NCCL_SOCKET_IFNAME set by the user win, and a multi-node rendezvous address changes nothing.
    Returns the names it set."""
    import os
    env = os.environ if environ is None else environ
    if env.get('MASTER_ADDR', '127.0.0.1') not in ('127.0.0.1', 'localhost', '::1') or not os.path.isdir('/sys/class/net/lo'):
        return []
    done = []
    for name in ('GLOO_SOCKET_IFNAME', 'NCCL_SOCKET_IFNAME'):
        if name not in env:
            env[name] = 'lo'
            done.append(name)
    return done


def buckets(numel, bucket_bytes=DEFAULT_BUCKET_BYTES, elem_bytes=4):
    """[(start, end), ...] covering [0, numel) in chunks of at most bucket_bytes."""
    per = max(1, bucket_bytes // elem_bytes)
    return [(s, min(numel, s + per)) for s in range(0, numel, per)]


def layer_buckets(layer_ranges, target_elems):
    """the gradient arena of one net cut into exchange buckets in BACKWARD order.
    layer_ranges[i] = (start, end) of layer i's gradients in the flat arena (layers contiguous, forward order);
    returns [(lowest layer index, start, end)]: a bucket closes once it holds `target_elems` (or at layer 0) and is
    exchanged as soon as the gradients of its LOWEST layer exist - the backward pass produces them last - so every bucket
    but the final one travels underneath the backward kernels still to come.  The buckets tile [0, arena end) exactly."""
    out, hi_end, acc = [], None, 0
    for li in range(len(layer_ranges) - 1, -1, -1):
        lo, hi = layer_ranges[li]
        hi_end = hi if hi_end is None else hi_end
        acc += hi - lo
        if acc >= target_elems or li == 0:
            out.append((li, lo, hi_end))
            hi_end, acc = None, 0
    return out


def allreduce_sum_async(flat, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """start a SUM all-reduce of `flat` bucket by bucket; returns the work handles.  The caller
    divides by the world size (the Adam kernel's grad_scale does it for free)."""
    works = []
    for s, e in buckets(flat.numel(), bucket_bytes, flat.element_size()):
        works.append(tdist.all_reduce(flat[s:e], op=tdist.ReduceOp.SUM, group=group, async_op=True))
    return works


def allreduce_sum_(flat, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """SUM all-reduce of `flat`, bucket by bucket, as BLOCKING collectives: ProcessGroupNCCL runs those on the
    caller's current stream (tools/nccl_sync_probe.py) instead of its internal one, so the caller decides which
    hardware queue the exchange occupies; nothing here waits on the host."""
    for s, e in buckets(flat.numel(), bucket_bytes, flat.element_size()):
        tdist.all_reduce(flat[s:e], op=tdist.ReduceOp.SUM, group=group, async_op=False)
    return flat


def wait_all(works):
    for w in works:
        w.wait()


def average_(flat, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
    """blocking mean over replicas, in place (used by tests and by hosts without fused scaling)."""
    wait_all(allreduce_sum_async(flat, group, bucket_bytes))
    flat.div_(tdist.get_world_size(group))
    return flat


def broadcast_state(eng, group=None, src=0):
    """make every replica start from rank `src`'s weights, Adam moments, SN vectors and BN stats."""
    if hasattr(eng, 'touch'):
        eng.touch()                                      # (raw writes into the engine's tensors: GanEngine.touch)
    for net in (eng.gen, eng.dis):
        for t in (net.params, net.adam_m, net.adam_v, net.opt.step_counter):
            tdist.broadcast(t, src=src, group=group)
        for k in sorted(net.state):
            tdist.broadcast(net.state[k], src=src, group=group)
    loss = getattr(eng, '_loss', None)                   # the *_mix coin's two moving averages are training state too
    if loss is not None and getattr(loss, 'state', None) is not None:
        tdist.broadcast(loss.state, src=src, group=group)


def shard_of(n_items, rank, world):
    """contiguous shard [lo, hi) of n_items for this rank (synthetic-data / dataset sharding)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_library_comm(group, world, rank, device):
    """one RCCL communicator INSIDE the library for this process (mmdgan_comm_init): rank 0's unique id travels over
    `group`.  A communicator of the right size made by an earlier engine of this process is kept."""
    import ctypes
    from . import ops
    lib = ops.require_device()
    if lib.mmdgan_comm_size() == world:
        return
    ident = (ctypes.c_char * 128)()
    if rank == 0:
        ops.check(lib.mmdgan_comm_unique_id(ident), 'comm_unique_id')
    dev = device if tdist.get_backend(group) == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device=dev)
    tdist.broadcast(t, src=0, group=group)
    ops.check(lib.mmdgan_comm_init(bytes(t.cpu().tolist()), world, rank), 'comm_init')


def choose_dp_backend(group, device, requested=None):
    """who carries an engine's gradient exchange: 'capi' = the library's own RCCL communicator (mmdgan_allreduce_bucket:
    the collectives are launch-plan nodes like any kernel, so a data-parallel step replays from one C call, and nothing
    depends on which stream ProcessGroupNCCL picks) or 'torch' = torch.distributed on `group`.
    requested: 'capi' / 'torch' / None (then MMDGAN_DP_BACKEND, then: 'capi' under an nccl (= RCCL) group, 'torch' under any
    other backend - gloo in the tests).  'capi' is verified before it is chosen - the communicator is made and a SUM
    all-reduce of ones must give the world size - and EVERY rank takes the same decision: one rank without the library path
    means none uses it (an explicit request then raises instead of falling back)."""
    import os
    import sys
    from . import ops
    from . import settings
    explicit = requested or settings.get('MMDGAN_DP_BACKEND')
    backend = explicit
    if backend is None:
        backend = 'capi' if (group is not None and tdist.get_backend(group) == 'nccl') else 'torch'
    assert backend in ('torch', 'capi'), backend
    if backend != 'capi' or group is None:
        return backend
    world, rank = tdist.get_world_size(group), tdist.get_rank(group)
    err = None
    try:
        init_library_comm(group, world, rank, device)
        probe = torch.ones(1024, device=device)
        ops.check(ops.require_device().mmdgan_allreduce_bucket(probe.data_ptr(), probe.numel(), ops._stream()), 'allreduce_bucket')
        torch.cuda.synchronize()
        if not bool((probe == float(world)).all()):
            raise RuntimeError('self-check all-reduce returned %r, expected %d' % (probe[:2].tolist(), world))
    except Exception as e:                           # no RCCL to bind, its rendezvous failed, or it does not add up
        err = e
    flag = torch.tensor([0 if err is not None else 1], dtype=torch.int32,
                        device=device if tdist.get_backend(group) == 'nccl' else 'cpu')
    tdist.all_reduce(flag, op=tdist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        if explicit == 'capi':
            raise RuntimeError('library-owned RCCL exchange unavailable on some rank (this rank: %s)' % (err,))
        sys.stderr.write('mmdgan: library-owned RCCL exchange unavailable (this rank: %s); using torch.distributed\n' % (err,))
        return 'torch'
    return 'capi'
