"""Every environment switch of the host side, in ONE place (the library's own are csrc/tuning.h).

The engines read them here, once, when they are constructed; `describe()` lists the ones that are off their default, and
bench.py puts that list (and the library's, ops.tuning()) into its JSON line - the schedule a number was measured under is
part of the number.  The defaults are what `python bench.py` runs and what tests/test_production_gpu.py pins; everything
else is a way to reproduce an A/B that DESIGN.md quotes, or a debugging aid.

  MMDGAN_LAUNCH_MODE   eager | graph | plan    how a step reaches the GPU when the caller does not say (engine.py)
  MMDGAN_SIDE_WGRAD    0   weight / bias gradients on the main stream instead of a stream of their own (single-stream
                           kernel statistics: tools/collect_profiles.sh)
  MMDGAN_SN_STREAMS    n   power-iteration streams (2)
  MMDGAN_SN_FUSED      0   one chain of launches per normalised kernel instead of the eight phase launches per net
  MMDGAN_EARLY_D_ADAM  0   D's Adam at the tail of the step instead of beside G's backward pass
  MMDGAN_GEN_TAIL_MAIN n   parameter gradients of G's first n layers behind their input-gradients on the main stream (2)
  MMDGAN_QUEUE_OPT     0   the round-3 placement of the step's cross-stream dependencies
  MMDGAN_DP_BACKEND    capi | torch   who carries the gradient exchange (dist.choose_dp_backend)
  MMDGAN_DP_BUCKET_MB  x   exchange bucket size (8)
  MMDGAN_DP_FORCE      1   run the exchange with a one-rank group too (plumbing tests on a 1-GPU box)
  MMDGAN_TAPE_STREAMS  0   the primitive-op engine on one stream
  MMDGAN_TAPE_COMPOSE  0   a block's scaling op and its 3x3 conv as two launches instead of one 4x4 stride-2 launch
  MMDGAN_TAPE_JOINT    0   D's two backward passes separately instead of one 3B-row pass
  MMDGAN_TAPE_FUSE_ADD 0   branch sums / gradient fan-ins as axpby passes instead of conv epilogues
  MMDGAN_TAPE_FUSE_ACT 0   an activation that is the only reader of a convolution's output as its own pass instead of on the epilogue
  MMDGAN_BN_RESIGN     0   batch-norm backward reads the activated output back instead of recomputing its sign from the input
  MMDGAN_STEP_AHEAD    auto | 1 | 0   D's power iterations and Winograd weight transform at the tail of the step before
                           (engine.py: _ahead_tail; round 5) - auto: for images of 64 x 64 and larger, where it measured faster
  MMDGAN_WGRAD_DEFER   0   every slab weight gradient followed by its own reduction launch instead of leaving it to the next
                           weight-gradient launch's prologue (mmdgan_wgrad_defer; round 5)
"""
import os

_DEFAULTS = {
    'MMDGAN_LAUNCH_MODE': None, 'MMDGAN_SIDE_WGRAD': '1', 'MMDGAN_SN_STREAMS': '2', 'MMDGAN_SN_FUSED': '1',
    'MMDGAN_EARLY_D_ADAM': '1', 'MMDGAN_GEN_TAIL_MAIN': '2', 'MMDGAN_QUEUE_OPT': '1', 'MMDGAN_DP_BACKEND': None,
    'MMDGAN_DP_BUCKET_MB': '8', 'MMDGAN_DP_FORCE': '0', 'MMDGAN_TAPE_STREAMS': '1', 'MMDGAN_TAPE_COMPOSE': '1',
    'MMDGAN_TAPE_JOINT': '1', 'MMDGAN_TAPE_FUSE_ADD': '1', 'MMDGAN_BN_RESIGN': '1', 'MMDGAN_WGRAD_DEFER': '1', 'MMDGAN_TAPE_FUSE_ACT': '1', 'MMDGAN_STEP_AHEAD': 'auto',
}


def get(name):
    """the switch's value as a string (None where there is no default and it is unset)"""
    assert name in _DEFAULTS, name
    return os.environ.get(name, _DEFAULTS[name])


def on(name):
    return get(name) not in (None, '0', '')


def describe():
    """{name: value} of the host-side switches that are set to something else than their default"""
    return {k: os.environ[k] for k in sorted(_DEFAULTS) if k in os.environ and os.environ[k] != _DEFAULTS[k]}


def unknown():
    """MMDGAN_* variables in the environment that nothing reads (a typo, or a switch of an earlier round)"""
    lib = {'MMDGAN_FORCE_DIRECT', 'MMDGAN_THIN_VALU', 'MMDGAN_WINO', 'MMDGAN_WINO_MIN_TILES', 'MMDGAN_WINO_KSPLIT_BELOW',
           'MMDGAN_WINO_WGRAD', 'MMDGAN_WINO_WGRAD_SLAB', 'MMDGAN_WINO2', 'MMDGAN_WINO2_KSPLIT', 'MMDGAN_WINO2_KSPLIT_BELOW',
           'MMDGAN_WINO2_WGRAD', 'MMDGAN_WINO2_WGRAD_MIN_TILES', 'MMDGAN_WINO43', 'MMDGAN_WINO43_MIN_TILES', 'MMDGAN_WINO43_KSPLIT_BELOW', 'MMDGAN_WINO43_WGRAD', 'MMDGAN_WINO43_WGRAD_MIN_TILES', 'MMDGAN_WINO43_WGRAD_CUS', 'MMDGAN_WGRAD_CUS', 'MMDGAN_GEMM_SKINNY', 'MMDGAN_GEMM_PANEL', 'MMDGAN_MMD_D16'}
    return sorted(k for k in os.environ if k.startswith('MMDGAN_') and k not in _DEFAULTS and k not in lib)


_warned = False


def warn_unknown():
    """one line on stderr, once per process, when the environment holds MMDGAN_* switches nothing reads - a switch of an
    earlier round (MMDGAN_HIP_GRAPH, MMDGAN_TILE, ...) would otherwise be ignored in silence (the engines call this when they
    are constructed)"""
    global _warned
    names = unknown()
    if names and not _warned:
        import sys
        _warned = True
        sys.stderr.write('mmdgan_hip: ignoring unknown environment switch(es) %s - see mmdgan_hip/settings.py and '
                         'csrc/tuning.h for the ones that exist\n' % ', '.join(names))
    return names
