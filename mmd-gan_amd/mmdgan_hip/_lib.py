"""ctypes loader for mmd-gan_amd/lib/libmmdgan_hip.so (include/mmdgan_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libmmdgan_hip.so')

c_fp = ctypes.c_void_p      # device pointers travel as integers (tensor.data_ptr())


class ConvGeom(ctypes.Structure):
    _fields_ = [('N', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('C', ctypes.c_int),
                ('K', ctypes.c_int), ('R', ctypes.c_int), ('stride', ctypes.c_int)]


# name -> (restype, argtypes); must list every symbol include/mmdgan_hip.h declares
_I, _L, _F, _P = ctypes.c_int, ctypes.c_long, ctypes.c_float, c_fp
_G = ctypes.POINTER(ConvGeom)
SIGNATURES = {
    'mmdgan_last_error': (ctypes.c_char_p, []),
    'mmdgan_version': (_I, []),
    'mmdgan_device_ok': (_I, []),
    'mmdgan_tuning_describe': (_L, [ctypes.c_char_p, ctypes.c_size_t]),
    'mmdgan_create': (_I, [ctypes.POINTER(ctypes.c_void_p)]),
    'mmdgan_destroy': (_I, [_P]),
    'mmdgan_make_current': (_I, [_P]),
    'mmdgan_set_workspace': (_I, [_P, ctypes.c_size_t]),
    'mmdgan_set_outputs_prezeroed': (_I, [_I]),
    'mmdgan_wgrad_defer': (_I, [_I]),
    'mmdgan_wgrad_flush': (_I, []),
    'mmdgan_plan_begin': (_I, []),
    'mmdgan_plan_mark': (_I, []),
    'mmdgan_plan_end': (_I, [ctypes.POINTER(_I)]),
    'mmdgan_plan_abort': (_I, []),
    'mmdgan_plan_segments': (_I, [_I]),
    'mmdgan_plan_nodes': (_L, [_I]),
    'mmdgan_plan_describe': (_L, [_I, ctypes.c_char_p, ctypes.c_size_t]),
    'mmdgan_plan_replay': (_I, [_I, _I]),
    'mmdgan_plan_destroy': (_I, [_I]),
    'mmdgan_stream_wait': (_I, [_P, _P]),
    'mmdgan_event_record': (_I, [_I, _P]),
    'mmdgan_event_wait': (_I, [_I, _P]),
    'mmdgan_memset_zero': (_I, [_P, ctypes.c_size_t, _P]),
    'mmdgan_memset_zero_multi': (_I, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), _I, _P]),
    'mmdgan_copy': (_I, [_P, _P, ctypes.c_size_t, _P]),
    'mmdgan_comm_unique_id': (_I, [_P]),
    'mmdgan_comm_init': (_I, [_P, _I, _I]),
    'mmdgan_comm_size': (_I, []),
    'mmdgan_comm_destroy': (_I, []),
    'mmdgan_allreduce_bucket': (_I, [_P, ctypes.c_size_t, _P]),
    'mmdgan_conv2d_fwd': (_I, [_G, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    'mmdgan_conv2d_dgrad': (_I, [_G, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    'mmdgan_conv2d_fwd_add': (_I, [_G, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P]),
    'mmdgan_conv2d_dgrad_add': (_I, [_G, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P]),
    'mmdgan_conv2d_wgrad': (_I, [_G, _P, _P, _P, _P]),
    'mmdgan_conv2d_wgrad_bias': (_I, [_G, _P, _P, _P, _P, _P]),
    'mmdgan_conv2d_wgrad_sn': (_I, [_G, _P, _P, _P, _P, _P, _P, _P]),
    'mmdgan_wino_eligible': (_I, [_G, _I]),
    'mmdgan_wino_weight_bytes': (ctypes.c_size_t, [_G]),
    'mmdgan_wino_transform': (_I, [_G, _P, _I, _P, _P]),
    'mmdgan_wino_transform_multi': (_I, [_P, _I, _P]),
    'mmdgan_wino_algo': (_I, [_G, _I]),
    'mmdgan_wgrad_algo': (_I, [_G]),
    'mmdgan_wino_algo_weight_bytes': (ctypes.c_size_t, [_G, _I]),
    'mmdgan_wino_transform_algo': (_I, [_G, _P, _I, _I, _P, _P]),
    'mmdgan_gemm': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    'mmdgan_colsum': (_I, [_P, _L, _I, _P, _P]),
    'mmdgan_dot': (_I, [_P, _P, _L, _P, _P]),
    'mmdgan_bn_workspace_bytes': (ctypes.c_size_t, [_I]),
    'mmdgan_bn_fwd_train': (_I, [_P, _L, _I, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'mmdgan_bn_fwd_apply': (_I, [_P, _L, _I, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'mmdgan_conv2d_fwd_stats': (_I, [_G, _P, _P, _P, _P, _I, _P, _P, _P]),
    'mmdgan_conv2d_dgrad_stats': (_I, [_G, _P, _P, _P, _P, _I, _P, _P, _P]),
    'mmdgan_bn_fwd_infer': (_I, [_P, _L, _I, _P, _P, _F, _I, _P, _P, _P, _P]),
    'mmdgan_bn_bwd': (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    'mmdgan_sn_norm': (_I, [_P, _L, _P, _P, _P]),
    'mmdgan_sn_norm_scale': (_I, [_P, _L, _F, _P, _P, _P, _P]),
    'mmdgan_sn_scale': (_I, [_P, _F, _P, _P]),
    'mmdgan_sn_wgrad_fixup': (_I, [_P, _P, _P, _P, _P, _L, _P]),
    'mmdgan_sn_power_iteration': (_I, [_P, _I, _I, _P]),
    'mmdgan_mmd_workspace_bytes': (ctypes.c_size_t, [_I, _I]),
    'mmdgan_mmd_loss': (_I, [_P, _P, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    'mmdgan_mmd_mix_workspace_bytes': (ctypes.c_size_t, [_I, _I]),
    'mmdgan_mmd_mix_loss': (_I, [_P, _P, _I, _I, _I, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    'mmdgan_adam_multi': (_I, [_P, _P, _I, _L, _F, _F, _F, _F, _I, _P, _P, _F, _P]),
    'mmdgan_adam_segments': (_I, [_P, _P, _P, _P, _P, _I, _P, _L, _F, _F, _F, _F, _I, _P, _P, _F, _I, _P]),
    'mmdgan_adam_prepare': (_I, [_F, _F, _F, _I, _P, _P, _P]),
    'mmdgan_adam_prepare_multi': (_I, [_P, _I, _P]),
    'mmdgan_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'mmdgan_nhwc_to_nchw': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'mmdgan_resample_down': (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    'mmdgan_resample_up': (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    'mmdgan_periodic_shuffle': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mmdgan_bilinear_resize': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'mmdgan_bicubic_resize': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'mmdgan_max_pool': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mmdgan_compose_scaled_conv': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'mmdgan_strided_slice': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'mmdgan_space_batch': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mmdgan_act_fwd': (_I, [_P, _P, _L, _I, _P]),
    'mmdgan_act_bwd': (_I, [_P, _P, _P, _L, _I, _I, _P]),
    'mmdgan_axpby': (_I, [_P, _F, _P, _F, _P, _L, _P]),
    'mmdgan_u8_records_to_nhwc': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
}

ABI_VERSION = 610           # include/mmdgan_hip.h: MMDGAN_VERSION this binding was written against

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """load the library or raise loudly - there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            'libmmdgan_hip.so is missing at %s - build it with `python mmd-gan_amd/build_ext.py` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; the library links the system one.  Whichever is loaded FIRST becomes the
    # process's HIP runtime for both (same SONAME) - and torch finds no device when that is not its own copy.  The engines
    # use torch for device memory and streams, so torch's runtime has to be the one: import it before the library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass                                     # a torch-free caller (a C / ctypes host): the system runtime is the only one
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.mmdgan_version()
    if got != ABI_VERSION:              # a stale .so against a newer binding (or the reverse): shifted arguments, no diagnostics
        raise HipLibraryError('libmmdgan_hip.so at %s reports ABI version %d, this binding needs %d - rebuild it with '
                              '`python mmd-gan_amd/build_ext.py --force`' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().mmdgan_last_error().decode()
        if rc == -1:
            raise ValueError('%s: %s' % (what, msg))
        raise HipLibraryError('%s failed (rc=%d): %s' % (what, rc, msg))
