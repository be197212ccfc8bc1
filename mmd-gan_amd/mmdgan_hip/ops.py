"""Thin torch-tensor wrappers over the C ABI (include/mmdgan_hip.h).

torch is plumbing here: it owns device memory and the stream; every function below launches a
hand-written HIP kernel through ctypes and fails loudly if the library or a gfx950 device is
missing.  Activations are NHWC fp32 tensors, kernels are in the reference's HWIO layout.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvGeom, check

ACT = {'linear': 0, 'relu': 1, 'lrelu': 2, 'tanh': 3}
LOSS = {'rep': 0, 'rep_mmd_g': 0, 'rmb': 1, 'rep_b': 1, 'rep_mmd_b': 1,                  # math_func.py:2644-2647
        'mmd_g': 2, 'fixed_g': 2, 'mgb': 3, 'hinge': 4, 'logistic': 5, '': 5,                  # :2602-2611
        'mmd_g_mix': 6, 'fixed_g_mix': 6, 'sgm': 7}                                            # :2613-2622
MIX_THRESHOLD = {6: 1.0, 7: 0.2}                     # default mix_threshold of _mmd_g_mix_ / _single_mmd_g_mix_ (:2195, :2230)


def is_mix_loss(loss_type):
    return LOSS.get(loss_type, -1) in MIX_THRESHOLD

_device_checked = False


def require_device():
    """the HIP path is the only path: no device or no library is an error, never a fallback."""
    global _device_checked
    lib = _lib.load()
    if not _device_checked:
        if not torch.cuda.is_available() or lib.mmdgan_device_ok() != 1:
            raise _lib.HipLibraryError('no gfx950 (MI355X) device visible: the HIP path cannot run and there is no '
                                       'CPU fallback')
        _device_checked = True
    return lib


_workspace = None


def set_workspace(nbytes=64 << 20, device=None):
    """register a torch-owned scratch buffer with the library (mmdgan_set_workspace)."""
    global _workspace
    lib = require_device()
    _workspace = torch.empty(nbytes, dtype=torch.uint8, device=device or 'cuda')
    check(lib.mmdgan_set_workspace(_workspace.data_ptr(), nbytes), 'set_workspace')
    return _workspace


import threading as _threading
_current_handles = _threading.local()


class Handle:
    """an mmdgan_handle (include/mmdgan_hip.h): one engine's workspace, prezeroed mode, launch plans and events.
    `with handle:` makes it the calling thread's current handle and restores the handle that was current before (a
    per-thread stack: constructing or stepping another engine inside the block does not strand the outer one)."""

    def __init__(self, workspace_bytes=256 << 20, device=None):
        lib = require_device()
        h = ctypes.c_void_p()
        check(lib.mmdgan_create(ctypes.byref(h)), 'create')
        self._h, self._lib = h, lib
        self.workspace = torch.empty(workspace_bytes, dtype=torch.uint8, device=device or 'cuda') if workspace_bytes else None
        if self.workspace is not None:
            with self:
                check(lib.mmdgan_set_workspace(self.workspace.data_ptr(), workspace_bytes), 'set_workspace')

    def forget_workspace_users(self):
        """call with the handle current, at a point where every stream that used the workspace has been joined (a step's
        start): the two halves have no owner again.  A launch plan recorded from here holds no ordering against streams of
        the past - a half last used on, say, a hipGraph capture's warm-up stream would otherwise put a wait for that dead
        stream (event record + wait, re-issued by every replay) into the plan: 0.07 ms per CIFAR step."""
        if self.workspace is not None:
            check(self._lib.mmdgan_set_workspace(self.workspace.data_ptr(), self.workspace.numel()), 'set_workspace')

    def __enter__(self):
        stack = getattr(_current_handles, 'stack', None)
        if stack is None:
            stack = _current_handles.stack = []
        stack.append(self)
        self._lib.mmdgan_make_current(self._h)
        return self

    def __exit__(self, *exc):
        stack = _current_handles.stack
        stack.pop()
        self._lib.mmdgan_make_current(stack[-1]._h if stack else None)     # None: the process default handle
        return False

    def __del__(self):
        try:
            self._lib.mmdgan_destroy(self._h)
        except Exception:
            pass


def wgrad_defer(on):
    """mmdgan_wgrad_defer on the current handle: the slab weight gradients leave their reduction to the prologue of the next
    weight-gradient launch of their stream (on) / issue it themselves (off; turning it off issues what is pending)"""
    check(require_device().mmdgan_wgrad_defer(1 if on else 0), 'wgrad_defer')


def wgrad_flush():
    """issue the slab reduction the last weight-gradient launch left behind (mmdgan_wgrad_flush): after this call, stream
    order behind the weight gradients' stream sees their complete dw / dbias / <G, W>"""
    check(require_device().mmdgan_wgrad_flush(), 'wgrad_flush')


def tuning():
    """the library's kernel-selection switches of this process as {name: (value, is_default)} (mmdgan_tuning_describe)"""
    from . import _lib
    lib = _lib.load()                                    # (host logic: needs no device)
    need = lib.mmdgan_tuning_describe(None, 0)
    buf = ctypes.create_string_buffer(int(need))
    lib.mmdgan_tuning_describe(buf, need)
    out = {}
    for item in buf.value.decode().split():
        name, val = item.split('=')
        out[name] = (int(val.rstrip('*')), not val.endswith('*'))
    return out


def plan_kernels(plan_id):
    """the kernel launches of a recorded plan of the CURRENT handle, in issue order: [(kernel, workgroups, threads, stream
    number)] with the kernel's demangled name cut before its parameter list (mmdgan_plan_describe)"""
    lib = require_device()
    need = lib.mmdgan_plan_describe(int(plan_id), None, 0)
    if need < 0:
        raise ValueError('no plan %r on the current handle' % (plan_id,))
    buf = ctypes.create_string_buffer(int(need))
    lib.mmdgan_plan_describe(int(plan_id), buf, need)
    out = []
    for line in buf.value.decode().splitlines():
        name, grid, block, st = line.rsplit('\t', 3)
        out.append((kernel_base_name(name), int(grid), int(block), int(st)))
    return out


def kernel_base_name(demangled):
    """'void mmdgan::wino_kernel<64, true>(mmdgan::WinoArgs)' -> 'wino_kernel<64, true>': the part before the parameter
    list (the first '(' outside template brackets), without return type and namespace"""
    demangled = demangled.replace('(anonymous namespace)::', '')
    depth, end = 0, len(demangled)
    for i, ch in enumerate(demangled):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            end = i
            break
    head = demangled[:end].strip()
    cut = head.find('<') if '<' in head else len(head)
    sp, ns = head.rfind(' ', 0, cut), head.rfind('::', 0, cut)
    start = max(sp + 1, ns + 2 if ns >= 0 else 0)
    return head[start:]


def stream_wait(waiting, signalling):
    """raw hipStream_t handles (ints): work issued later on `waiting` starts after what `signalling` holds now"""
    check(require_device().mmdgan_stream_wait(waiting, signalling), 'stream_wait')


def event_record(slot, stream):
    check(require_device().mmdgan_event_record(int(slot), stream), 'event_record')


def event_wait(slot, stream):
    check(require_device().mmdgan_event_wait(int(slot), stream), 'event_wait')


def memset_zero(t, stream=None):
    check(require_device().mmdgan_memset_zero(t.data_ptr(), t.numel() * t.element_size(), _stream() if stream is None else stream),
          'memset_zero')


def memset_zero_multi(tensors, stream=None):
    """zero several (small) tensors with one launch (mmdgan_memset_zero_multi)"""
    n = len(tensors)
    if n == 0:
        return
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (ctypes.c_size_t * n)(*[t.numel() * t.element_size() for t in tensors])
    check(require_device().mmdgan_memset_zero_multi(ptrs, sizes, n, _stream() if stream is None else stream), 'memset_zero_multi')


def copy(dst, src, stream=None):
    assert dst.numel() * dst.element_size() == src.numel() * src.element_size() and dst.is_contiguous() and src.is_contiguous()
    check(require_device().mmdgan_copy(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(),
                                       _stream() if stream is None else stream), 'copy')


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'expected a contiguous fp32 CUDA tensor'
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Stream object and resolves the
    device index through several Python layers - measured at a third of the host time of an eagerly issued
    step (150 calls) - the raw accessor is one C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def act_id(name):
    if name not in ACT:
        raise NotImplementedError('Function {} is not implemented.'.format(name))   # layer_func.py:149
    return ACT[name]


def geom(N, H, W, C, K, R, stride):
    return ConvGeom(N, H, W, C, K, R, stride)


def out_hw(H, W, stride):
    return -(-H // stride), -(-W // stride)


# ------------------------------------------------------------------------------------------------
def conv2d_fwd(x, w, stride, bias=None, scale=None, act='linear', dact_of=None, out=None, dact_batch=0,
               out_zeroed=False, wino=None, addend=None, bn_totals=None):
    """x [N,H,W,C], w [R,R,C,K] -> y [N,P,Q,K] = act(scale*conv(x,w)+bias) (layer_func.py:913-916)
    wino: the tensor wino_transform(w, ...) made from w (the library then skips its own transform)
    addend: another tensor of y's shape, added last (mmdgan_conv2d_fwd_add)"""
    lib = require_device()
    N, H, W, C = x.shape
    R, K = w.shape[0], w.shape[3]
    assert w.shape[2] == C
    P, Q = out_hw(H, W, stride)
    y = out if out is not None else torch.empty((N, P, Q, K), device=x.device, dtype=torch.float32)
    g = geom(N, H, W, C, K, R, stride)
    flags = act_id(act) | (0x100 if out_zeroed else 0) | wino_flag(wino)
    if bn_totals is not None:          # + the batch-norm totals of y (mmdgan_conv2d_fwd_stats; bn_fwd_train(have_totals=True) follows)
        assert addend is None and dact_of is None
        check(lib.mmdgan_conv2d_fwd_stats(ctypes.byref(g), _p(x), _p(w if wino is None else wino), _p(bias), _p(scale), flags, _p(y),
                                          bn_totals.data_ptr(), _stream()), 'conv2d_fwd_stats')
        return y
    if addend is not None:
        assert tuple(addend.shape) == tuple(y.shape) and addend.is_contiguous()
        check(lib.mmdgan_conv2d_fwd_add(ctypes.byref(g), _p(x), _p(w if wino is None else wino), _p(bias), _p(scale), flags,
                                        _p(dact_of), int(dact_batch), _p(addend), _p(y), _stream()), 'conv2d_fwd_add')
        return y
    check(lib.mmdgan_conv2d_fwd(ctypes.byref(g), _p(x), _p(w if wino is None else wino), _p(bias), _p(scale), flags,
                                _p(dact_of), int(dact_batch),
                                _p(y), _stream()), 'conv2d_fwd')
    return y


def conv2d_dgrad(dy, w, in_hw, stride, bias=None, scale=None, act='linear', dact_of=None, out=None, dact_batch=0,
                 wino=None, out_zeroed=False, addend=None, bn_totals=None):
    """dy [N,P,Q,K], w [R,R,C,K] -> dx [N,H,W,C]; forward form = tf.nn.conv2d_transpose (layer_func.py:926)
    out_zeroed: `out` is zero on entry - a launch with a linear epilogue may split its reduction over workgroups"""
    lib = require_device()
    N, P, Q, K = dy.shape
    R, C = w.shape[0], w.shape[2]
    assert w.shape[3] == K
    H, W = in_hw
    assert out_hw(H, W, stride) == (P, Q)
    dx = out if out is not None else torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
    g = geom(N, H, W, C, K, R, stride)
    flags = act_id(act) | (0x100 if out_zeroed else 0) | wino_flag(wino)
    if bn_totals is not None:                            # + the batch-norm totals of dx (mmdgan_conv2d_dgrad_stats)
        assert addend is None and dact_of is None
        check(lib.mmdgan_conv2d_dgrad_stats(ctypes.byref(g), _p(dy), _p(w if wino is None else wino), _p(bias), _p(scale), flags,
                                            _p(dx), bn_totals.data_ptr(), _stream()), 'conv2d_dgrad_stats')
        return dx
    if addend is not None:                               # another tensor of dx's shape, added last (mmdgan_conv2d_dgrad_add)
        assert tuple(addend.shape) == tuple(dx.shape) and addend.is_contiguous()
        check(lib.mmdgan_conv2d_dgrad_add(ctypes.byref(g), _p(dy), _p(w if wino is None else wino), _p(bias), _p(scale), flags,
                                          _p(dact_of), int(dact_batch), _p(addend), _p(dx), _stream()), 'conv2d_dgrad_add')
        return dx
    check(lib.mmdgan_conv2d_dgrad(ctypes.byref(g), _p(dy), _p(w if wino is None else wino), _p(bias), _p(scale), flags, _p(dact_of),
                                  int(dact_batch), _p(dx), _stream()), 'conv2d_dgrad')
    return dx


WINO_NONE, WINO_F23, WINO_F22S2, WINO_F43 = 0, 1, 2, 3      # include/mmdgan_hip.h: MMDGAN_WINO_*
_WINO_LEAD = {WINO_F23: (16,), WINO_F22S2: (4, 9), WINO_F43: (36,)}


def wino_eligible(N, H, W, C, K, R, stride, dgrad):
    """does conv2d_fwd (dgrad=False) / conv2d_dgrad (True) of this geometry accept the tensors wino_transform() makes by
    default - F(2x2,3x3) / F(2x2,2x2) on 4x4 stride 2?  (F(4x4,3x3) has its own query: wino_algo)"""
    g = geom(N, H, W, C, K, R, stride)
    return bool(require_device().mmdgan_wino_eligible(ctypes.byref(g), int(dgrad)))


def wino_algo(N, H, W, C, K, R, stride, dgrad):
    """WHICH Winograd algorithm the library prefers for this geometry (mmdgan_wino_algo): WINO_NONE, WINO_F23 (F(2x2,3x3)),
    WINO_F22S2 (4x4 stride 2) or WINO_F43 (F(4x4,3x3), H and W multiples of 4) - and with it the layout of the transformed
    weights a call with `wino=` must hold"""
    g = geom(N, H, W, C, K, R, stride)
    return int(require_device().mmdgan_wino_algo(ctypes.byref(g), int(dgrad)))


def wgrad_algo(N, H, W, C, K, R, stride):
    """which algorithm conv2d_wgrad takes for this geometry with a workspace registered (mmdgan_wgrad_algo): WINO_F43
    (F(4x4,3x3), csrc/conv_wino43w.hip), WINO_F23, WINO_F22S2 or WINO_NONE - what a test's rounding floor depends on"""
    g = geom(N, H, W, C, K, R, stride)
    return int(require_device().mmdgan_wgrad_algo(ctypes.byref(g)))


def wino_alloc(algo, C, K, dgrad, device):
    """the (uninitialised) transformed-weight tensor of `algo` for a [R,R,C,K] kernel: [16 | 4,9 | 36] + (K,C if dgrad else C,K)"""
    return torch.empty(_WINO_LEAD[algo] + ((K, C) if dgrad else (C, K)), device=device, dtype=torch.float32)


def wino_kind(u):
    """the algorithm a transformed-weight tensor belongs to, from its shape"""
    return WINO_F22S2 if u.dim() == 4 else (WINO_F43 if u.shape[0] == 36 else WINO_F23)


def wino_flag(wino):
    """MMDGAN_ACT_FLAG_W_WINOGRAD / _W_WINOGRAD43 for a call that passes transformed weights"""
    return 0 if wino is None else (0x400 if wino_kind(wino) == WINO_F43 else 0x200)


def wino_transform(w, dgrad, out=None, algo=None):
    """w [3,3,C,K] -> [16,C,K] (forward) or [16,K,C] (input-gradient) transformed weights, or [36,..] for algo=WINO_F43;
    w [4,4,C,K] (stride-2 layers) -> [4,9,C,K] or [4,9,K,C]"""
    lib = require_device()
    R, _, C, K = w.shape
    if algo is None:
        algo = wino_kind(out) if out is not None else (WINO_F23 if R == 3 else WINO_F22S2)
    u = out if out is not None else wino_alloc(algo, C, K, dgrad, w.device)
    assert wino_kind(u) == algo
    g = geom(1, 4, 4, C, K, R, 1 if R == 3 else 2)
    check(lib.mmdgan_wino_transform_algo(ctypes.byref(g), _p(w), int(dgrad), int(algo), _p(u), _stream()), 'wino_transform')
    return u


class WinoJob(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('u', ctypes.c_void_p), ('C', ctypes.c_int), ('K', ctypes.c_int), ('R', ctypes.c_int),
                ('stride', ctypes.c_int), ('dgrad', ctypes.c_int), ('algo', ctypes.c_int)]


class WinoTransforms:
    """mmdgan_wino_transform_multi: the transforms of many kernels as ONE launch.  jobs: [(w [R,R,C,K], u, dgrad)] -
    pointers are taken once (weights and their transformed tensors stay where they are for the life of an engine)"""

    def __init__(self, jobs):
        self.keep = list(jobs)
        self.table = (WinoJob * max(1, len(self.keep)))()
        for j, (w, u, dgrad) in zip(self.table, self.keep):
            R, _, C, K = w.shape
            j.w, j.u, j.C, j.K, j.R, j.stride, j.dgrad = w.data_ptr(), u.data_ptr(), C, K, R, 1 if R == 3 else 2, int(bool(dgrad))
            j.algo = wino_kind(u)

    def run(self, stream=None):
        if self.keep:
            check(require_device().mmdgan_wino_transform_multi(ctypes.cast(self.table, ctypes.c_void_p), len(self.keep),
                                                               _stream() if stream is None else stream), 'wino_transform_multi')


def conv2d_wgrad(x, dy, R, stride, out=None, dbias=None, w=None, dot=None):
    """x [N,H,W,C], dy [N,P,Q,K] -> dw [R,R,C,K]; dbias [K] (optional) receives the column sums of dy.
    w, dot (both or neither): the spectrally normalised kernel and a [1] tensor that receives <dw, w> - the scalar of the
    spectral-norm fix-up, which stays OUT of dw (mmdgan_conv2d_wgrad_sn; applied by AdamArena / Network.get_variable)"""
    lib = require_device()
    N, H, W, C = x.shape
    K = dy.shape[3]
    dw = out if out is not None else torch.empty((R, R, C, K), device=x.device, dtype=torch.float32)
    g = geom(N, H, W, C, K, R, stride)
    if w is not None:
        assert dot is not None and tuple(w.shape) == tuple(dw.shape)
        check(lib.mmdgan_conv2d_wgrad_sn(ctypes.byref(g), _p(x), _p(dy), _p(dw), _p(dbias), _p(w), _p(dot), _stream()),
              'conv2d_wgrad_sn')
        return dw
    if dbias is None:
        check(lib.mmdgan_conv2d_wgrad(ctypes.byref(g), _p(x), _p(dy), _p(dw), _stream()), 'conv2d_wgrad')
    else:
        check(lib.mmdgan_conv2d_wgrad_bias(ctypes.byref(g), _p(x), _p(dy), _p(dw), _p(dbias), _stream()),
              'conv2d_wgrad_bias')
    return dw


def gemm(a, b, trans_a=False, trans_b=False, bias=None, scale=None, act='linear', dact_of=None, out=None, dact_rows=0,
         out_zeroed=False):
    """C = act(scale * op(a) op(b) + bias), row-major 2-D tensors (tf.matmul, layer_func.py:911)
    out_zeroed: `out` is zero on entry (the caller zeroed it this step): the launch may split its K reduction and
    accumulate - the only way it splits while mmdgan_set_outputs_prezeroed(1) is in force"""
    lib = require_device()
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == K2, (a.shape, b.shape, trans_a, trans_b)
    c = out if out is not None else torch.empty((M, N), device=a.device, dtype=torch.float32)
    check(lib.mmdgan_gemm(int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1], _p(b), b.shape[1], _p(bias), _p(scale),
                          act_id(act) | (0x100 if out_zeroed else 0), _p(dact_of), int(dact_rows), _p(c), N, _stream()),
          'gemm')
    return c


def colsum(x2d, out=None):
    lib = require_device()
    rows, cols = x2d.shape
    o = out if out is not None else torch.empty(cols, device=x2d.device, dtype=torch.float32)
    check(lib.mmdgan_colsum(_p(x2d), rows, cols, _p(o), _stream()), 'colsum')
    return o


def dot(a, b, out=None):
    lib = require_device()
    o = out if out is not None else torch.empty(1, device=a.device, dtype=torch.float32)
    check(lib.mmdgan_dot(_p(a), _p(b), a.numel(), _p(o), _stream()), 'dot')
    return o


# ------------------------------------------------------------------------------------------------
_bn_ws = {}


def _bn_workspace(C, device):
    key = (C, device)
    if key not in _bn_ws:
        n = _lib.load().mmdgan_bn_workspace_bytes(C)
        _bn_ws[key] = torch.empty(n, device=device, dtype=torch.uint8)
    return _bn_ws[key]


def bn_fwd_train(x2d, gamma, beta, moving_mean, moving_var, act='linear', eps=1e-3, momentum=0.99, unbiased=True,
                 new_moving_mean=None, new_moving_var=None, out=None, save_mean=None, save_invstd=None, workspace=None,
                 have_totals=False):
    """x2d [rows, C].  Returns (y, save_mean, save_invstd, new_moving_mean, new_moving_var).
    have_totals: `workspace` already holds the totals of x2d (conv2d_fwd / conv2d_dgrad with bn_totals=workspace wrote x2d)
    out / save_mean / save_invstd / workspace: caller-owned buffers (an engine's per-step ones; `workspace` holds the
    per-channel totals, mmdgan_bn_workspace_bytes(C)); new_moving_* may be the moving_* tensors themselves (in place)"""
    lib = require_device()
    rows, C = x2d.shape
    y = out if out is not None else torch.empty_like(x2d)
    mean = save_mean if save_mean is not None else torch.empty(C, device=x2d.device, dtype=torch.float32)
    invstd = save_invstd if save_invstd is not None else torch.empty_like(mean)
    nmm = new_moving_mean if new_moving_mean is not None else torch.empty_like(mean)
    nmv = new_moving_var if new_moving_var is not None else torch.empty_like(mean)
    ws = workspace if workspace is not None else _bn_workspace(C, x2d.device)
    entry = lib.mmdgan_bn_fwd_apply if have_totals else lib.mmdgan_bn_fwd_train
    check(entry(_p(x2d), rows, C, _p(gamma), _p(beta), eps, momentum, int(unbiased), act_id(act), _p(y),
                _p(mean), _p(invstd), _p(moving_mean), _p(moving_var), _p(nmm), _p(nmv),
                ws.data_ptr(), _stream()), 'bn_fwd_train')
    return y, mean, invstd, nmm, nmv


def bn_fwd_infer(x2d, gamma, beta, moving_mean, moving_var, act='linear', eps=1e-3, out=None):
    lib = require_device()
    rows, C = x2d.shape
    y = out if out is not None else torch.empty_like(x2d)
    check(lib.mmdgan_bn_fwd_infer(_p(x2d), rows, C, _p(gamma), _p(beta), eps, act_id(act), _p(moving_mean),
                                  _p(moving_var), _p(y), _stream()), 'bn_fwd_infer')
    return y


def bn_bwd(x2d, y2d, dy2d, gamma, save_mean, save_invstd, act='linear', dgamma=None, dbeta=None, out=None, workspace=None,
           beta=None):
    """y2d None (act linear / relu / lrelu, beta given): the activation's sign is recomputed from x2d, bitwise the forward
    entry's decision - the output is not read back"""
    lib = require_device()
    rows, C = x2d.shape
    dx = out if out is not None else torch.empty_like(x2d)
    dgamma = dgamma if dgamma is not None else torch.empty(C, device=x2d.device, dtype=torch.float32)
    dbeta = dbeta if dbeta is not None else torch.empty(C, device=x2d.device, dtype=torch.float32)
    ws = workspace if workspace is not None else _bn_workspace(C, x2d.device)
    check(lib.mmdgan_bn_bwd(_p(x2d), _p(y2d) if y2d is not None else None, _p(dy2d), rows, C, _p(gamma),
                            _p(beta) if beta is not None else None, _p(save_mean), _p(save_invstd), act_id(act),
                            _p(dx), _p(dgamma), _p(dbeta), ws.data_ptr(), _stream()), 'bn_bwd')
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
def sn_norm(v, normalise=True, out_norm=None, out_v=None):
    lib = require_device()
    norm = out_norm if out_norm is not None else torch.empty(1, device=v.device, dtype=torch.float32)
    vn = (out_v if out_v is not None else torch.empty_like(v)) if normalise else None
    check(lib.mmdgan_sn_norm(_p(v), v.numel(), _p(norm), _p(vn), _stream()), 'sn_norm')
    return norm, vn


class SnLayer(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ('w', 'x', 'u', 'un', 'xb', 'col', 'dsigma', 'sigma', 'scale', 'xb_norm', 'norm_acc')] + \
               [('act_k', ctypes.c_float), ('form', ctypes.c_int)] + [(k, ctypes.c_int) for k in ('H', 'W', 'C', 'K', 'R', 'stride')] + \
               [('x_out', ctypes.c_void_p)]


class SnChains:
    """mmdgan_sn_power_iteration: one power-iteration step of many kernels, every stage of all chains as one launch.
    layers: dicts with the tensors w, x, u, un, xb, dsigma, sigma, scale, xb_norm, optionally x_out (col is allocated here), act_k, form
    (0 conv2d_fwd, 1 conv2d_dgrad, 2 x W, 3 x W^T) and the conv geometry H, W, C, K, R, stride (dense: C, K).
    Pointers are taken once: the tensors stay where they are for the life of an engine."""

    def __init__(self, layers, device):
        self.keep = list(layers)
        self.table = (SnLayer * max(1, len(self.keep)))()
        # the patch scratch of every convolution kernel in ONE buffer.  The half of it that receives a product (form 0: the
        # second, form 1: the first) is accumulated into by several workgroups when the product has few tiles (the rule of
        # include/mmdgan_hip.h): `zero_each_step` lists those halves - in prezeroed mode the caller zeroes them with the step's
        # other accumulation targets
        sizes = []
        for L in self.keep:
            if int(L['form']) <= 1:
                P, Q = out_hw(int(L['H']), int(L['W']), int(L['stride']))
                sizes.append((P * Q, int(L['R']) ** 2 * int(L['C'])))
        self.col_flat = torch.zeros(max(4, sum((2 * pq * r2c + 3) // 4 * 4 for pq, r2c in sizes)), device=device, dtype=torch.float32)
        self.zero_each_step = []
        col_off = 0
        self.norm_acc = torch.zeros(4 * max(1, len(self.keep)), device=device, dtype=torch.float32)
        for i, (t, L) in enumerate(zip(self.table, self.keep)):
            t.norm_acc = self.norm_acc.data_ptr() + 16 * i
            for k in ('w', 'x', 'u', 'un', 'xb', 'dsigma', 'sigma', 'scale', 'xb_norm', 'x_out'):
                setattr(t, k, L[k].data_ptr() if L.get(k) is not None else None)
            t.act_k, t.form = float(L['act_k']), int(L['form'])
            for k in ('H', 'W', 'C', 'K', 'R', 'stride'):
                setattr(t, k, int(L.get(k, 1)))
            if t.form <= 1:
                pq, r2c = sizes.pop(0)
                t.col = self.col_flat.data_ptr() + 4 * col_off
                if -(-pq // 64) * -(-r2c // 64) < 128:
                    half = col_off + (pq * r2c if t.form == 0 else 0)
                    self.zero_each_step.append(self.col_flat[half:half + pq * r2c])
                col_off += (2 * pq * r2c + 3) // 4 * 4

    def run(self, update=True, stream=None):
        if self.keep:
            check(require_device().mmdgan_sn_power_iteration(ctypes.cast(self.table, ctypes.c_void_p), len(self.keep), int(bool(update)),
                                                             _stream() if stream is None else stream), 'sn_power_iteration')


def sn_norm_scale(v, act_k, out_norm, out_scale, out_v=None):
    """||v|| -> out_norm, act_k/||v|| -> out_scale, v/(||v||+1e-10) -> out_v, one launch"""
    lib = require_device()
    check(lib.mmdgan_sn_norm_scale(_p(v), v.numel(), float(act_k), _p(out_norm), _p(out_scale), _p(out_v), _stream()),
          'sn_norm_scale')


def sn_scale(sigma, act_k, out=None):
    lib = require_device()
    o = out if out is not None else torch.empty(1, device=sigma.device, dtype=torch.float32)
    check(lib.mmdgan_sn_scale(_p(sigma), float(act_k), _p(o), _stream()), 'sn_scale')
    return o


def sn_wgrad_fixup(g, dsigma_dw, dot_gw, sigma, scale):
    lib = require_device()
    check(lib.mmdgan_sn_wgrad_fixup(_p(g), _p(dsigma_dw), _p(dot_gw), _p(sigma), _p(scale), g.numel(), _stream()),
          'sn_wgrad_fixup')
    return g


# ------------------------------------------------------------------------------------------------
_mmd_ws = {}


def mmd_loss(s_gen, s_x, loss_type='rep', rep_weights=(0.0, -1.0), lower_bound=0.25, upper_bound=4.0,
             need_grads=True, need_masks=False, need_dist=False, grads_dis_first=False):
    """fused pairwise-distance / Gaussian-kernel / rep|rmb loss (math_func.py:2505-2550); also 'mmd_g', 'mgb'
    (:2160-2193) and the score losses 'hinge', 'logistic' (:2128-2143), for which scalars[2:4] are the two means of
    loss_dis and masks / dist do not exist.
    Returns dict(scalars[8] = loss_gen, loss_dis, e_kxx, e_kxy, e_kyy, e_kxx_b, e_kyy_b, e_kxy_b;
                 grads[4,B,d] = dLg/ds_gen, dLg/ds_x, dLd/ds_gen, dLd/ds_x (grads_dis_first: dLd/ds_x, dLd/ds_gen,
                 dLg/ds_gen, dLg/ds_x), masks[3,B,B] (bool), dist[3,B,B])."""
    lib = require_device()
    if loss_type not in LOSS:
        raise NotImplementedError('Not implemented.')                            # math_func.py:2651
    if is_mix_loss(loss_type):
        raise ValueError('{}: the *_mix losses carry state and a coin, use mmd_mix_loss'.format(loss_type))
    B, d = s_gen.shape
    assert s_x.shape == s_gen.shape
    dev = s_gen.device
    key = (B, d, dev)
    if key not in _mmd_ws:
        _mmd_ws[key] = torch.zeros(lib.mmdgan_mmd_workspace_bytes(B, d), device=dev, dtype=torch.uint8)
    out = torch.empty(8, device=dev, dtype=torch.float32)
    grads = torch.empty((4, B, d), device=dev, dtype=torch.float32) if need_grads else None
    masks = torch.empty((3, B, B), device=dev, dtype=torch.uint8) if need_masks else None
    dist = torch.empty((3, B, B), device=dev, dtype=torch.float32) if need_dist else None
    check(lib.mmdgan_mmd_loss(_p(s_gen), _p(s_x), B, d, LOSS[loss_type] | (0x100 if grads_dis_first else 0), float(rep_weights[0]), float(rep_weights[1]),
                              float(lower_bound), float(upper_bound), _p(out), _p(grads),
                              masks.data_ptr() if masks is not None else None, _p(dist), _mmd_ws[key].data_ptr(),
                              _stream()), 'mmd_loss')
    return {'scalars': out, 'grads': grads, 'masks': masks.bool() if masks is not None else None, 'dist': dist}


_mix_ws = {}


def mix_workspace(B, d, device):
    key = (B, d, device)
    if key not in _mix_ws:
        _mix_ws[key] = torch.zeros(_lib.load().mmdgan_mmd_mix_workspace_bytes(B, d), device=device, dtype=torch.uint8)
    return _mix_ws[key]


def mmd_mix_loss(s_gen, s_x, uni, state, loss_type='mmd_g_mix', mix_threshold=None, loss_average_update=0.01,
                 mix_prob_update=0.01, need_grads=True, need_masks=False, grads_dis_first=False, out=None, grads=None,
                 workspace=None):
    """'mmd_g_mix' / 'fixed_g_mix' / 'sgm' (math_func.py:2195-2263).  uni [B]: this step's uniform(0,1) draw; state [2]:
    {loss_average, mix_prob}, read and then UPDATED IN PLACE (the UPDATE_OPS).  Returns dict(scalars[8] = loss_gen,
    loss_dis, e_kxx, e_kxy, e_kyy, loss_average and mix_prob as used, number of un-mixed rows; grads[4,B,d];
    masks = {mix_indices [B], mix_group_1 [2B], mix_group_2 [2B]} as bool tensors)."""
    lib = require_device()
    code = LOSS.get(loss_type, -1)
    if code not in MIX_THRESHOLD:
        raise NotImplementedError('Not implemented.')                            # math_func.py:2651
    B, d = s_gen.shape
    assert s_x.shape == s_gen.shape and uni.numel() == B and state.numel() == 2
    dev = s_gen.device
    ws = workspace if workspace is not None else mix_workspace(B, d, dev)
    out = out if out is not None else torch.empty(8, device=dev, dtype=torch.float32)
    if grads is None and need_grads:
        grads = torch.empty((4, B, d), device=dev, dtype=torch.float32)
    masks = torch.empty(5 * B, device=dev, dtype=torch.uint8) if need_masks else None
    thr = MIX_THRESHOLD[code] if mix_threshold is None else float(mix_threshold)
    check(lib.mmdgan_mmd_mix_loss(_p(s_gen), _p(s_x), B, d, code | (0x100 if grads_dis_first else 0), _p(uni), thr,
                                  float(loss_average_update), float(mix_prob_update), _p(state), _p(out), _p(grads),
                                  masks.data_ptr() if masks is not None else None, ws.data_ptr(), _stream()), 'mmd_mix_loss')
    m = None
    if masks is not None:
        m = {'mix_indices': masks[:B].bool(), 'mix_group_1': masks[B:3 * B].bool(), 'mix_group_2': masks[3 * B:].bool()}
    return {'scalars': out, 'grads': grads, 'masks': m}


class GanLossLauncher:
    """the loss node of a training step (my_sngan.py:282-289 -> GANLoss.apply): owns the loss kernel's workspace, the
    [4,B,d] score-gradient block in the order the engines back-propagate it ([dLd/ds_x ; dLd/ds_gen ; dLg/ds_gen ;
    dLg/ds_x], MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST) and - for the `*_mix` losses - the coin's two state variables
    ('mmd_g_mix/coin/gen_average', 'mmd_g_mix/coin/prob', math_func.py:2073-2078) and this step's uniform draw."""

    def __init__(self, loss_type, rep_weights, B, d, device, mix_threshold=None):
        lib = require_device()
        if loss_type not in LOSS:
            raise NotImplementedError('Not implemented.')                        # math_func.py:2651
        self.loss_type, self.code, self.B, self.d = loss_type, LOSS[loss_type], int(B), int(d)
        self.w = (float(rep_weights[0]), float(rep_weights[1]))
        self.mix = self.code in MIX_THRESHOLD
        self.grads = torch.zeros(4, B, d, device=device)
        if self.mix:
            self.ws = torch.zeros(lib.mmdgan_mmd_mix_workspace_bytes(B, d), device=device, dtype=torch.uint8)
            self.state = torch.zeros(2, device=device)                           # zeros_initializer, :1995, :2027
            self.uni = torch.zeros(B, device=device)
            self.mix_threshold = MIX_THRESHOLD[self.code] if mix_threshold is None else float(mix_threshold)
        else:
            self.ws = torch.zeros(max(lib.mmdgan_mmd_workspace_bytes(B, d), 64), device=device, dtype=torch.uint8)
            self.state = self.uni = None

    def draw(self, generator=None, uni=None):
        """this step's tf.random_uniform([B]) (math_func.py:2079), or the caller's"""
        if not self.mix:
            return
        if uni is not None:
            self.uni.copy_(torch.as_tensor(uni, dtype=torch.float32).reshape(-1))
        else:
            self.uni.uniform_(0.0, 1.0, generator=generator)

    def launch(self, scores, losses_out):
        """scores [2B, d]: rows [:B] = s_x (real), [B:] = s_gen, as D produced them from [real ; fake]"""
        lib, B, d = require_device(), self.B, self.d
        s_x, s_gen = scores[:B], scores[B:]
        if self.mix:
            check(lib.mmdgan_mmd_mix_loss(s_gen.data_ptr(), s_x.data_ptr(), B, d, self.code | 0x100, self.uni.data_ptr(),
                                          self.mix_threshold, 0.01, 0.01, self.state.data_ptr(), losses_out.data_ptr(),
                                          self.grads.data_ptr(), None, self.ws.data_ptr(), _stream()), 'mmd_mix_loss')
        else:
            check(lib.mmdgan_mmd_loss(s_gen.data_ptr(), s_x.data_ptr(), B, d, self.code | 0x100, self.w[0], self.w[1], 0.25,
                                      4.0, losses_out.data_ptr(), self.grads.data_ptr(), None, None, self.ws.data_ptr(),
                                      _stream()), 'mmd_loss')

    def state_dict(self):
        return {} if not self.mix else {'mmd_g_mix/coin/gen_average': float(self.state[0].item()),
                                        'mmd_g_mix/coin/prob': float(self.state[1].item())}

    def load_state_dict(self, sd):
        if self.mix and 'mmd_g_mix/coin/gen_average' in sd:
            self.state.copy_(torch.tensor([float(sd['mmd_g_mix/coin/gen_average']), float(sd['mmd_g_mix/coin/prob'])]))


# ------------------------------------------------------------------------------------------------
class AdamGroup:
    """one flat-arena group for mmdgan_adam_multi: params / grads / m / v are lists of tensors."""

    def __init__(self, params, grads, ms, vs):
        dev = params[0].device
        ptrs = []
        for p, g, m, v in zip(params, grads, ms, vs):
            ptrs += [_p(p), _p(g), _p(m), _p(v)]
        self.keep = (params, grads, ms, vs)
        self.ptrs = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self.sizes = torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev)
        self.n = len(params)
        self.max_size = max(p.numel() for p in params)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)     # device-side t (graph-capturable)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)

    def step(self, lr, step=None, beta1=0.5, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """step=None: use and advance the device-side counter; else use the given host step."""
        lib = require_device()
        check(lib.mmdgan_adam_multi(self.ptrs.data_ptr(), self.sizes.data_ptr(), self.n, self.max_size, float(lr),
                                    float(beta1), float(beta2), float(eps), int(step or 0),
                                    self.step_counter.data_ptr() if step is None else None, self.lr_t.data_ptr(),
                                    float(grad_scale), _stream()), 'adam_multi')


class AdamSegment(ctypes.Structure):
    _fields_ = [('off', ctypes.c_long), ('n', ctypes.c_long), ('dsigma', ctypes.c_void_p), ('dot', ctypes.c_void_p),
                ('sigma', ctypes.c_void_p), ('scale', ctypes.c_void_p)]


class AdamArena:
    """mmdgan_adam_segments: TF-Adam over ONE flat arena (params / grads / m / v share element offsets) cut into segments.
    segments: [(off, n, sn)] with sn = None or dict(dsigma=, dot=, sigma=, scale=) of device tensors - the spectral-norm
    fix-up of that segment's gradient is applied as the gradient is read (the arena keeps the raw gradient)."""

    def __init__(self, params, grads, m, v, segments):
        dev = params.device
        self.keep = (params, grads, m, v, segments)
        self.params, self.grads, self.m, self.v = params, grads, m, v
        segs = (AdamSegment * len(segments))()
        blocks = []
        for i, (off, n, sn) in enumerate(segments):
            segs[i].off, segs[i].n = int(off), int(n)
            if sn is not None:
                assert sn['dsigma'].numel() == n and sn['dsigma'].data_ptr() % 16 == 0
                segs[i].dsigma, segs[i].dot = sn['dsigma'].data_ptr(), sn['dot'].data_ptr()
                segs[i].sigma, segs[i].scale = sn['sigma'].data_ptr(), sn['scale'].data_ptr()
            blocks += [(i, b) for b in range((int(n) + 1023) // 1024)]
        raw = bytes(segs)
        self.segs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.blocks = torch.tensor(blocks, dtype=torch.int32, device=dev).contiguous()
        self.n_segs, self.n_blocks = len(segments), len(blocks)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)     # device-side t (graph-capturable)
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)

    fold_fixup = True        # False: the caller applied sn_wgrad_fixup itself (data-parallel replicas, before their all-reduce)
    prepared = False         # prepare() has run for the coming step()

    def prepare(self, lr, step=None, beta1=0.5, beta2=0.999):
        """the step count and the bias-corrected learning rate of the coming step() - no gradient needed, so a step can issue
        it early, beside its forward pass, instead of in front of the update"""
        check(require_device().mmdgan_adam_prepare(float(lr), float(beta1), float(beta2), int(step or 0),
                                                   self.step_counter.data_ptr() if step is None else None,
                                                   self.lr_t.data_ptr(), _stream()), 'adam_prepare')
        self.prepared = True

    def step(self, lr, step=None, beta1=0.5, beta2=0.999, eps=1e-8, grad_scale=1.0):
        lib = require_device()
        pre, self.prepared = self.prepared, False
        check(lib.mmdgan_adam_segments(_p(self.params), _p(self.grads), _p(self.m), _p(self.v), self.segs.data_ptr(),
                                       self.n_segs, self.blocks.data_ptr(), self.n_blocks, float(lr), float(beta1),
                                       float(beta2), float(eps), -1 if pre else int(step or 0),
                                       None if (pre or step is not None) else self.step_counter.data_ptr(), self.lr_t.data_ptr(),
                                       float(grad_scale), int(self.fold_fixup), _stream()), 'adam_segments')


class AdamPrepareJob(ctypes.Structure):
    _fields_ = [('lr', ctypes.c_float), ('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('step', ctypes.c_int),
                ('step_counter', ctypes.c_void_p), ('lr_t_scratch', ctypes.c_void_p)]


def adam_prepare_multi(jobs, beta1=0.5, beta2=0.999):
    """AdamArena.prepare() of several arenas - jobs: [(arena, lr)] - as one launch (mmdgan_adam_prepare_multi)"""
    table = (AdamPrepareJob * len(jobs))()
    for t, (arena, lr) in zip(table, jobs):
        t.lr, t.beta1, t.beta2, t.step = float(lr), float(beta1), float(beta2), 0
        t.step_counter, t.lr_t_scratch = arena.step_counter.data_ptr(), arena.lr_t.data_ptr()
    check(require_device().mmdgan_adam_prepare_multi(ctypes.cast(table, ctypes.c_void_p), len(jobs), _stream()),
          'adam_prepare_multi')
    for arena, _ in jobs:
        arena.prepared = True


def nchw_to_nhwc(x):
    lib = require_device()
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_nchw_to_nhwc(_p(x), _p(y), N, C, H, W, _stream()), 'nchw_to_nhwc')
    return y


def resample_down(x, factor=2, scale=None, out=None, accumulate=False):
    """window sums of an NHWC tensor: scale = 1/factor^2 (default) is ImageScaling 'avg' (layer_func.py:1155-1159),
    scale = 1 the gradient of 'unpool'."""
    lib = require_device()
    n, h, w, c = x.shape
    assert h % factor == 0 and w % factor == 0, 'resample_down: {}x{} is not a multiple of {}'.format(h, w, factor)
    if out is None:
        assert not accumulate
        out = torch.empty((n, h // factor, w // factor, c), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_resample_down(_p(x), _p(out), n, h // factor, w // factor, c, factor,
                                   1.0 / (factor * factor) if scale is None else float(scale), int(accumulate),
                                   _stream()), 'resample_down')
    return out


def resample_up(x, factor=2, scale=1.0, out=None, accumulate=False):
    """every pixel of an NHWC tensor repeated factor x factor: scale = 1 is ImageScaling 'unpool'
    (layer_func.py:1160-1163), scale = 1/factor^2 the gradient of 'avg'."""
    lib = require_device()
    n, h, w, c = x.shape
    if out is None:
        assert not accumulate
        out = torch.empty((n, h * factor, w * factor, c), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_resample_up(_p(x), _p(out), n, h, w, c, factor, float(scale), int(accumulate), _stream()),
          'resample_up')
    return out


def periodic_shuffle(x, factor, to_big, out=None):
    """ImageScaling 'ps' (layer_func.py:197-244): to_big: [N,H,W,f*f*C] -> [N,H*f,W*f,C] (tf.depth_to_space), else the
    inverse (tf.space_to_depth); channel order as the reference's NCHW ops define it"""
    lib = require_device()
    n, h, w, c = x.shape
    f = int(factor)
    if to_big:
        assert c % (f * f) == 0, 'periodic_shuffle: {} channels are not a multiple of {}'.format(c, f * f)
        H, W, C = h, w, c // (f * f)
        shape = (n, h * f, w * f, C)
    else:
        assert h % f == 0 and w % f == 0
        H, W, C = h // f, w // f, c
        shape = (n, H, W, c * f * f)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    check(lib.mmdgan_periodic_shuffle(_p(x), _p(out), n, H, W, C, f, int(bool(to_big)), _stream()), 'periodic_shuffle')
    return out


def bilinear_resize(x, size, out=None):
    """ImageScaling 'bil' (layer_func.py:1128-1137): tf.image.resize_bilinear(align_corners=True) of an NHWC tensor"""
    lib = require_device()
    n, h, w, c = x.shape
    oh, ow = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_bilinear_resize(_p(x), _p(out), n, h, w, c, oh, ow, 0, _stream()), 'bilinear_resize')
    return out


def bilinear_resize_grad(dy, in_hw, out=None):
    """gradient of bilinear_resize w.r.t. its input: dy [N,OH,OW,C] -> dx [N,H,W,C]; `out` must be zero when
    mmdgan_set_outputs_prezeroed(1) is in force"""
    lib = require_device()
    n, oh, ow, c = dy.shape
    h, w = int(in_hw[0]), int(in_hw[1])
    if out is None:
        out = torch.zeros((n, h, w, c), device=dy.device, dtype=torch.float32)
    check(lib.mmdgan_bilinear_resize(_p(dy), _p(out), n, h, w, c, oh, ow, 1, _stream()), 'bilinear_resize_grad')
    return out


def bicubic_resize(x, size, out=None):
    """ImageScaling 'bic' (layer_func.py:1138-1147): tf.image.resize_bicubic(align_corners=True) of an NHWC tensor"""
    lib = require_device()
    n, h, w, c = x.shape
    oh, ow = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_bicubic_resize(_p(x), _p(out), n, h, w, c, oh, ow, 0, _stream()), 'bicubic_resize')
    return out


def bicubic_resize_grad(dy, in_hw, out=None):
    """gradient of bicubic_resize w.r.t. its input: dy [N,OH,OW,C] -> dx [N,H,W,C]; `out` must be zero when
    mmdgan_set_outputs_prezeroed(1) is in force"""
    lib = require_device()
    n, oh, ow, c = dy.shape
    h, w = int(in_hw[0]), int(in_hw[1])
    if out is None:
        out = torch.zeros((n, h, w, c), device=dy.device, dtype=torch.float32)
    check(lib.mmdgan_bicubic_resize(_p(dy), _p(out), n, h, w, c, oh, ow, 1, _stream()), 'bicubic_resize_grad')
    return out


def max_pool(x, factor=2, dy=None, out=None):
    """ImageScaling 'max' (layer_func.py:1149-1153).  dy=None: window maxima of x [N,H,W,C]; with dy [N,H/f,W/f,C]: the
    gradient w.r.t. x (dy at each window's first maximum, zero elsewhere)"""
    lib = require_device()
    n, h, w, c = x.shape
    f = int(factor)
    assert h % f == 0 and w % f == 0, 'max_pool: {}x{} is not a multiple of {}'.format(h, w, f)
    if out is None:
        out = torch.empty((n, h // f, w // f, c) if dy is None else (n, h, w, c), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_max_pool(_p(x), _p(dy), _p(out), n, h // f, w // f, c, f, _stream()), 'max_pool')
    return out


def compose_scaled_conv(w3, mode, out=None):
    """w3 [3,3,C,K] -> the 4x4 stride-2 kernel of avgpool/2 o conv (mode 'avg': [4,4,C,K], use with conv2d_fwd) or of
    conv o unpool x2 (mode 'unpool': [4,4,K,C], use with the forward form of conv2d_dgrad)"""
    lib = require_device()
    _, _, C, K = w3.shape
    m = {'avg': 0, 'unpool': 1}[mode]
    if out is None:
        out = torch.empty((4, 4, C, K) if m == 0 else (4, 4, K, C), device=w3.device, dtype=torch.float32)
    check(lib.mmdgan_compose_scaled_conv(_p(w3), _p(out), C, K, m, 0, _stream()), 'compose_scaled_conv')
    return out


def compose_scaled_conv_grad(d4, mode, out=None):
    """the adjoint: gradient w.r.t. the composed 4x4 kernel -> gradient w.r.t. the 3x3 kernel [3,3,C,K]"""
    lib = require_device()
    m = {'avg': 0, 'unpool': 1}[mode]
    C, K = (d4.shape[2], d4.shape[3]) if m == 0 else (d4.shape[3], d4.shape[2])
    if out is None:
        out = torch.empty((3, 3, C, K), device=d4.device, dtype=torch.float32)
    check(lib.mmdgan_compose_scaled_conv(_p(d4), _p(out), C, K, m, 1, _stream()), 'compose_scaled_conv_grad')
    return out


def act_fwd(x, act, out=None):
    lib = require_device()
    if out is None:
        out = torch.empty_like(x)
    check(lib.mmdgan_act_fwd(_p(x), _p(out), x.numel(), act_id(act), _stream()), 'act_fwd')
    return out


def strided_slice(x, off, step, out_hw, out=None, adjoint_hw=None):
    """x [N,H,W,C] -> [N,P,Q,C] = x[:, off + p*step, off + q*step]; adjoint_hw=(H, W): x is [N,P,Q,C] and the result the
    [N,H,W,C] tensor with x at those positions and zeros elsewhere (every element written)"""
    lib = require_device()
    if adjoint_hw is None:
        n, h, w, c = x.shape
        p, q = out_hw
        out = out if out is not None else torch.empty((n, p, q, c), device=x.device, dtype=torch.float32)
        check(lib.mmdgan_strided_slice(_p(x), _p(out), n, h, w, c, int(off), int(step), p, q, 0, _stream()), 'strided_slice')
    else:
        n, p, q, c = x.shape
        h, w = adjoint_hw
        out = out if out is not None else torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
        check(lib.mmdgan_strided_slice(_p(x), _p(out), n, h, w, c, int(off), int(step), p, q, 1, _stream()), 'strided_slice')
    return out


def space_batch(x, d, hw=None, out=None):
    """hw=None: x [N,H,W,C] -> the d*d phase images [N*d*d, ceil(H/d), ceil(W/d), C] (zeros beyond the image);
    hw=(H, W): the reverse, x [N*d*d, ceil(H/d), ceil(W/d), C] -> [N,H,W,C]"""
    lib = require_device()
    if hw is None:
        n, h, w, c = x.shape
        out = out if out is not None else torch.empty((n * d * d, -(-h // d), -(-w // d), c), device=x.device, dtype=torch.float32)
        check(lib.mmdgan_space_batch(_p(x), _p(out), n, h, w, c, int(d), 1, _stream()), 'space_batch')
    else:
        h, w = hw
        n, c = x.shape[0] // (d * d), x.shape[3]
        out = out if out is not None else torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
        check(lib.mmdgan_space_batch(_p(x), _p(out), n, h, w, c, int(d), 0, _stream()), 'space_batch')
    return out


def act_bwd(dy, y, act, out=None, accumulate=False):
    """dx (+)= dy * act'(.), the derivative taken from the activation's output y"""
    lib = require_device()
    if out is None:
        assert not accumulate
        out = torch.empty_like(dy)
    check(lib.mmdgan_act_bwd(_p(dy), _p(y), _p(out), dy.numel(), act_id(act), int(accumulate), _stream()), 'act_bwd')
    return out


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    lib = require_device()
    assert a.numel() == b.numel()
    if out is None:
        out = torch.empty_like(a)
    check(lib.mmdgan_axpby(_p(a), float(alpha), _p(b), float(beta), _p(out), a.numel(), _stream()), 'axpby')
    return out


def u8_records_to_nhwc(records, channels, height, width, chw=True, out=None):
    """uint8 device tensor of N records (any shape with N*C*H*W bytes) -> fp32 NHWC in [-1,1]
    (input_func.py:797-801, 839-842)."""
    lib = require_device()
    assert records.dtype == torch.uint8 and records.is_cuda and records.is_contiguous()
    n = records.numel() // (channels * height * width)
    assert n * channels * height * width == records.numel(), 'u8_records_to_nhwc: size is not a multiple of C*H*W'
    if out is None:
        out = torch.empty((n, height, width, channels), device=records.device, dtype=torch.float32)
    check(lib.mmdgan_u8_records_to_nhwc(records.data_ptr(), int(chw), _p(out), n, channels, height, width, _stream()),
          'u8_records_to_nhwc')
    return out


def nhwc_to_nchw(x):
    lib = require_device()
    N, H, W, C = x.shape
    y = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    check(lib.mmdgan_nhwc_to_nchw(_p(x), _p(y), N, C, H, W, _stream()), 'nhwc_to_nchw')
    return y
