"""shared helpers for the parity tests (oracle = checker, never the thing measured)."""
import ast
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# Tolerances (BASELINE.json north_star): fp32 MMD loss and conv activations within 1e-4
# relative; index masks bit-exact.  "relative" for a tensor = max|a-b| / max|b|.
RTOL = 1e-4


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.max(np.abs(b))
    if scale == 0.0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)


def elementwise_err(a, b, floor_frac=1e-2):
    """the ELEMENT-WISE companion of rel_err: max over elements of |a-b| / (|b| + floor_frac * max|b|).  rel_err is
    norm-wise (max|a-b| / max|b|) and leaves small-magnitude entries unconstrained; here every entry above floor_frac
    (1 %) of the tensor's scale is held to the relative bar ON ITS OWN, and the entries below it to an absolute
    floor_frac * bar * scale (1e-6 of the scale at the 1e-4 bar).  The floor is what fp32 allows: a sum of hundreds of
    products of either sign carries an absolute error of ~1e-7 of the SCALE of its terms whatever the size of the
    result (measured on the Winograd kernels: 1.2e-7 of max|y| at the zero crossings), so an entry at 1e-4 of the scale
    has three significant digits in ANY fp32 implementation, the reference's included."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor_frac * np.max(np.abs(b)))))


def sign_flips(got_nhwc, ref_nchw):
    """how many activation outputs have a different SIGN in an fp32 evaluation (engine buffer, NHWC or [N,F]) than in
    the fp64 oracle's (NCHW or [N,F]): the lrelu / relu derivative masks that differ between the two.  One such flip
    moves ONE element of the back-propagated gradient by 0.9 (lrelu) or 1.0 (relu) of itself - against a tensor of N
    elements that is ~1/sqrt(N) of its L2 norm times the element's relative size, i.e. 2e-3 ... 1e-2 for the 1e5 ... 1e6
    elements of a D layer (tools/flip_probe.py: with NO flip above a layer its gradients agree to 3e-6, with one flip
    to 6e-3).  Pre-activations land within fp32 resolution (~1e-6 of their scale) of zero about once per million."""
    got, ref = np.asarray(got_nhwc), np.asarray(ref_nchw)
    if ref.ndim == 4:
        ref = ref.transpose(0, 2, 3, 1)
    return int(((got.reshape(ref.shape) > 0) != (ref > 0)).sum())


def golden(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def designs_of(fx):
    return ast.literal_eval(str(fx['designs_repr']))


def l2_err(got, ref, gscale=0.0):
    """||got - ref|| / (||ref|| + 1e-6 * gscale): the L2 distance of a gradient tensor from its reference, relative"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-6 * gscale))


def engine_masks(eng):
    """the relu / lrelu sign decisions of an engine's last training step in the oracle's layouts (NCHW / reference
    feature order): {'gen': [...], 'dis': [...]}, one bool tensor per relu / lrelu in evaluation order - what
    OracleGan.grads(masks=...) takes to evaluate fp32 arithmetic under the SAME decisions (oracle/restatement.py:_act)"""
    import torch

    def to_ref(y, perm):
        y = y.detach().cpu()
        if y.dim() == 4:
            y = y.permute(0, 3, 1, 2)
        elif perm is not None:                           # features of a dense layer that feeds an image reshape: NHWC order
            t = torch.empty_like(y)
            t[:, torch.as_tensor(perm)] = y
            y = t
        return (y > 0).contiguous()
    out = {'gen': [], 'dis': []}
    if hasattr(eng, 'buf'):                              # the hand-scheduled engine: one activation per layer
        for name, net in (('gen', eng.gen), ('dis', eng.dis)):
            for s in net.specs:
                if s.act in ('relu', 'lrelu'):
                    out[name].append(to_ref(eng.buf[s.scope + '#y'], s.col_perm))
        return out
    for name, net, vals in (('gen', eng.gen, eng._last_vals[0]), ('dis', eng.dis, eng._last_vals[1])):   # primitive-op engine
        produced_by = {p['out']: p for p in net.prims}
        for p in net.prims:
            if p['kind'] in ('bn', 'act') and p['act'] in ('relu', 'lrelu'):
                perm = None
                if p['kind'] == 'bn':
                    perm = net._bn_perm.get(p['prefix'])
                else:
                    src = produced_by.get(p['ins'][0])
                    if src is not None and src['kind'] == 'dense':
                        perm = src['k'].col_perm
                out[name].append(to_ref(vals[p['out']], perm))
    return out


GRAD_BAR = float(os.environ.get("TEST_GRAD_BAR", "1e-4"))          # north_star's 1e-4 (rounds 1-3: 5e-4)
GRAD_MAXABS_BAR = float(os.environ.get("TEST_GRAD_MAXABS_BAR", "1e-3"))

# every time a test falls back to the comparison "under the engine's own sign decisions" (a relu / lrelu output within fp32
# resolution of zero decided the other way in this run) it is counted here; tests/conftest.py prints the tally at the end
# of the session, so a log shows how often the fallback fired and for which test
KNIFE_EDGE_RETRIES = []


class KnifeEdgeRetry(UserWarning):
    pass


FLOOR_CLAUSE_USES = []          # (test, tensor) every time assert_grads_within_fp32_floor had to take its fp32-floor clause


def note_knife_edge_retry(what, allowed=True):
    """allowed=False: the test's fixture was built with an activation margin (oracle/make_golden.py: ACT_MARGIN_MIN = 1e-5 of
    the layer's scale, a hundred times fp32 resolution), so a differing sign decision there is a fault, not luck - fail
    unless TEST_ALLOW_KNIFE_EDGE=1."""
    import warnings
    KNIFE_EDGE_RETRIES.append(str(what))
    if not allowed and os.environ.get('TEST_ALLOW_KNIFE_EDGE', '0') != '1':
        raise AssertionError('knife-edge fallback taken on a fixture with an activation margin: %s' % (what,))
    warnings.warn('%s: knife-edge activation decided the other way in this run; compared under the engine\'s sign decisions'
                  % (what,), KnifeEdgeRetry)


KNIFE_EDGE_MARGIN = 1e-5        # |pre-activation| / the layer's largest: fp32 sums land within ~1e-7 of the scale of their terms
KNIFE_EDGE_MAX = 8              # decisions per step that may differ in a tiny net ...
KNIFE_EDGE_PER_ELEMENT = 2e-6   # ... plus this share of the activations a step evaluates: a pre-activation of spread s = scale / 5
#                                 lies within d of zero with probability ~0.8 d / s = 4 d / scale; at fp32's d ~ 1e-6 of the scale
#                                 that is 4 per million elements (a CIFAR step at batch 64 evaluates 34 M: ~140 expected).  The
#                                 count is a plausibility bound only - what makes a differing decision legitimate is its MARGIN
FLOOR_CLAUSE_MAX = 3            # tensors of one audited step that may still need the fp32-floor clause


def knife_edge_budget(elements):
    return KNIFE_EDGE_MAX + int(KNIFE_EDGE_PER_ELEMENT * elements)


def with_audit(masks_per_step):
    """the engine's sign decisions of each step with a fresh 'audit' list attached (oracle/restatement.py:_act fills it)"""
    return [dict(m, audit=[]) for m in masks_per_step]


def assert_knife_edges_only(audited, what=''):
    """what makes the comparison "under the engine's sign decisions" legitimate: the decisions forced on the fp64 evaluation
    differ from the ones it would have taken itself in a handful of elements (knife_edge_budget of the activations
    evaluated), each with a pre-activation within fp32 resolution of zero.  A kernel fault that moved a sign anywhere else
    fails here.  Returns (differing decisions, the largest relative |pre-activation| among them)."""
    total, worst = 0, 0.0
    for step, m in enumerate(audited):
        n = sum(a[0] for a in m['audit'])
        elements = sum(a[2] for a in m['audit'] if len(a) > 2)
        assert n <= knife_edge_budget(elements), (what, 'step %d: %d sign decisions of %d differ from the fp64 evaluation (budget %d)'
                                                  % (step, n, elements, knife_edge_budget(elements)))
        total += n
        worst = max([worst] + [a[1] for a in m['audit']])
    assert worst <= KNIFE_EDGE_MARGIN, (what, 'a differing sign decision at %.2e of its layer\'s scale: not a knife edge' % worst)
    return total, worst


def max_err(got, ref, gscale=0.0):
    """max|got - ref| / (max|ref| + 1e-6 * gscale): the ENTRY-wise companion of l2_err - a localised fault (one wrong border
    row, one bad tile, one phase of a composed convolution) that an L2 norm over a whole kernel dilutes shows up here"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6 * gscale))


AUDITED_STEPS = []              # one line per step whose gradients were held against the AUDITED fp64 evaluation (AuditedStep)


def assert_grads_within_fp32_floor(grads, ref64, floor32, skip=(), what=''):
    """THE rule for gradients of a step against the fp64 oracle (one rule, every step test): each tensor within 1e-4 in L2
    (BASELINE.json's bar) AND every entry within 1e-3 of the tensor's largest, against the oracle's fp64 evaluation.
    A tensor above those bars is then held, to the SAME bars, against the fp64 evaluation under the engine's own relu /
    lrelu sign decisions - a reference only after its AUDIT (AuditedStep.audited64: the forced decisions differ from the
    fp64 evaluation's own in a few elements, every one within 1e-5 of its layer's scale of zero; a kernel that produced a
    wrong sign anywhere else fails there, before any gradient is looked at).  Only what that cannot explain may take the
    last clause: within twice what an fp32 evaluation of the oracle itself loses against the audited fp64 one (same two
    measures), plus the bars - at most FLOOR_CLAUSE_MAX tensors of a step, each one on record in the session's tally.
    grads / ref64: name -> array (ref64 None: the audited evaluation is the reference from the start - the production-batch
    tests, where one fp64 backward pass is affordable and two are not); floor32: an AuditedStep (helpers.fp32_floor), or -
    the free-running fixture tests, which audit on their own - a callable / dict giving the fp32 oracle's gradients.
    skip: variables whose gradient is ZERO analytically (the last D bias: the loss sees score differences only) - what any
    implementation holds there is rounding noise, so they are held to 1e-4 of their net's gradient scale."""
    cache = {}
    audited = getattr(floor32, 'audited64', None)
    direct = ref64 is None
    if direct:
        assert audited is not None, 'no reference'
        ref64 = audited()

    def f32():
        if 'v' not in cache:
            cache['v'] = floor32() if callable(floor32) else floor32
        return cache['v']
    floor_uses = 0
    for net in ('gen', 'dis'):
        names = [n for n in grads if n.startswith(net)]
        if not names:
            continue
        gscale = max(float(np.abs(np.asarray(ref64[n])).max()) for n in names)
        for n in names:
            if n in skip:
                assert float(np.abs(np.asarray(grads[n])).max()) <= 1e-4 * gscale, (what, n, 'analytically zero gradient', gscale)
                continue
            r = np.asarray(ref64[n], np.float64)
            err, emax = l2_err(grads[n], r, gscale), max_err(grads[n], r, gscale)
            if err <= GRAD_BAR and emax <= GRAD_MAXABS_BAR:
                continue
            if audited is not None and not direct:
                r = np.asarray(audited()[n], np.float64)                 # (audit asserted inside, once)
                floor32.explained.append(n)
                err, emax = l2_err(grads[n], r, gscale), max_err(grads[n], r, gscale)
                if err <= GRAD_BAR and emax <= GRAD_MAXABS_BAR:
                    continue
            f = np.asarray(f32()[n], np.float64)
            fl, flmax = l2_err(f, r, gscale), max_err(f, r, gscale)
            FLOOR_CLAUSE_USES.append('%s %s: L2 %.2e (fp32 oracle %.2e), max-abs %.2e (%.2e)%s'
                                     % (what, n, err, fl, emax, flmax, ' vs the audited fp64 evaluation' if audited is not None else ''))
            floor_uses += 1
            assert err <= 2.0 * fl + GRAD_BAR, (what, n, 'L2', err, fl)
            assert emax <= 2.0 * flmax + GRAD_MAXABS_BAR, (what, n, 'max-abs', emax, flmax)
    if audited is not None:
        assert floor_uses <= FLOOR_CLAUSE_MAX, (what, '%d tensors needed the fp32-floor clause after the audit' % floor_uses)
        floor32.floor_uses = floor_uses
        if 'audit' in floor32.__dict__:
            AUDITED_STEPS.append(floor32.describe(what))


class AuditedStep:
    """the two lazy references of a teacher-forced step beyond the oracle's plain fp64 evaluation, both under the relu /
    lrelu sign decisions the ENGINE's kernels took (engine_masks):
      audited64()  the fp64 oracle's gradients under those decisions, AFTER the audit that makes them a reference: the
                   decisions differ from the fp64 evaluation's own in at most knife_edge_budget(elements) elements and every
                   differing one has |pre-activation| <= KNIFE_EDGE_MARGIN of its layer's largest (assert_knife_edges_only).
                   Two evaluations of the same algebra decide differently only there; a fault that flips a sign far from
                   zero is caught by the audit itself.
      f32() / ()   the fp32 oracle's gradients under the same decisions: what fp32 ARITHMETIC loses (the floor clause)."""

    def __init__(self, arch, loss_type, lr, prev_vars, z, real, eng, uni=None, mix_state=None, **kw):
        self.args = (arch, loss_type, lr, prev_vars, z, real, uni, mix_state, kw)
        self.eng = eng                                   # (its decisions are read on first use: call before the engine steps again)
        self.explained, self.floor_uses, self._c = [], 0, {}

    @property
    def masks(self):
        if 'masks' not in self._c:
            self._c['masks'] = engine_masks(self.eng)
        return self._c['masks']

    def _run(self, dtype, masks):
        arch, loss_type, lr, prev_vars, z, real, uni, mix_state, kw = self.args
        o = _R().OracleGan(arch, loss_type, lr, dtype=dtype, params=prev_vars, **kw)
        if mix_state is not None:
            o.mix_state = mix_state
        r = o.grads(_torch().tensor(z, dtype=dtype), _torch().tensor(real, dtype=dtype), uni=uni, masks=masks)
        out = {n: g.numpy() for n, g in r[4].items()}
        out.update({n: g.numpy() for n, g in r[5].items()})
        return out, r

    def audited64(self):
        if 'a64' not in self._c:
            m = dict(self.masks, audit=[])
            self._c['a64'], self._c['r64'] = self._run(_torch().float64, m)
            self.elements = sum(a[2] for a in m['audit'])
            self.per_activation = [(a[0], a[1]) for a in m['audit']]
            if os.environ.get('TEST_AUDIT_REPORT_ONLY') == '1':          # debugging aid: the numbers instead of the verdict
                self.audit = (sum(a[0] for a in m['audit']), max([0.0] + [a[1] for a in m['audit']]))
            else:
                self.audit = assert_knife_edges_only([m], 'audit of the engine\'s sign decisions')
        return self._c['a64']

    def f32(self):
        if 'f32' not in self._c:
            self._c['f32'], _ = self._run(_torch().float32, self.masks)
        return self._c['f32']

    __call__ = f32

    def describe(self, what=''):
        return ('%s: %d of %d sign decisions differ from the fp64 evaluation (budget %d), worst |pre-activation| %.1e of its '
                'layer scale; %s tensor(s) held against the audited evaluation, %d took the fp32-floor clause'
                % (what, self.audit[0], self.elements, knife_edge_budget(self.elements), self.audit[1],
                   len(self.explained) or 'all', self.floor_uses))


def fp32_floor(arch, loss_type, lr, prev_vars, z, real, eng, uni=None, mix_state=None, **kw):
    """the references of helpers.assert_grads_within_fp32_floor for a teacher-forced step: the oracle from the same variables
    on the same batch, its relu / lrelu sign decisions forced to the ones the engine's kernels took - in fp64 and audited
    (the reference for tensors a differing knife-edge decision moved) and in fp32 (the floor: the loss of fp32 ARITHMETIC,
    not the luck of which evaluation met an element within rounding of zero).  Call it right after the engine's step."""
    return AuditedStep(arch, loss_type, lr, prev_vars, z, real, eng, uni=uni, mix_state=mix_state, **kw)



def oracle_trajectory(fx, arch, sn_mode, dtype, at_step=None, masks_per_step=None, want='grads'):
    """the restatement free-running from the fixture's initial state on the fixture's inputs.  want='grads': the gradients
    of step `at_step` (default: the last); want='final': every variable after the last step.  masks_per_step: the relu /
    lrelu sign decisions an engine took at each step (engine_masks) - the run then takes the same ones (_act)"""
    init = {k[len('init/'):]: v for k, v in fx.items() if k.startswith('init/')}
    gan = _R().OracleGan(arch, str(fx['loss_type']), tuple(fx['lr']), dtype=dtype, params=init, sn_mode=sn_mode)
    if 'adam_t' in fx:
        gan.set_adam_state({k[len('adam_m/'):]: v for k, v in fx.items() if k.startswith('adam_m/')},
                           {k[len('adam_v/'):]: v for k, v in fx.items() if k.startswith('adam_v/')}, int(fx['adam_t']))
    n_steps = fx['z'].shape[0]
    last = n_steps - 1 if at_step is None else at_step
    for step in range(n_steps if want == 'final' else last + 1):
        z, real = _torch().tensor(fx['z'][step], dtype=dtype), _torch().tensor(fx['real'][step], dtype=dtype)
        masks = masks_per_step[step] if masks_per_step is not None else None
        if want == 'grads' and step == last:
            r = gan.grads(z, real, masks=masks)
            out = {n: g.numpy() for n, g in r[4].items()}
            out.update({n: g.numpy() for n, g in r[5].items()})
            return out
        gan.step(z, real, masks=masks)
    return {n: v.numpy() for n, v in gan.params.items()}


def fp32_oracle_trajectory_grads(fx, arch, sn_mode, at_step=None, masks_per_step=None):
    """what an fp32 evaluation of the fixture's trajectory loses against the fp64 one: the fp32 side of
    assert_grads_within_fp32_floor for the free-running fixture tests"""
    return oracle_trajectory(fx, arch, sn_mode, _torch().float32, at_step, masks_per_step)


def _R():
    from oracle import restatement
    return restatement


def _torch():
    import torch
    return torch
