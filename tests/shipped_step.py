"""One teacher-forced G+D step of a shipped architecture (configs.py = BASELINE.json's configs) at its OWN per-GPU batch
against the fp64 oracle - generated images, D scores, losses, EVERY gradient - plus the list of kernels the recorded
step consists of.

Used two ways:
  * imported by the gpu tests (tests/test_production_gpu.py), and
  * run as a script in a SUBPROCESS whose environment holds no MMDGAN_* variable, i.e. under the library's production
    kernel selection (tests/conftest.py lowers the Winograd thresholds for the small parity cases; the library caches
    such switches on first use, so only a fresh process sees the defaults bench.py runs with):
        python tests/shipped_step.py <config> <loss> <batch> <launch mode>      -> one JSON line on stdout

The reference runs one fixed graph per config (graph_func.py:851-854); the build's counterpart is the recorded launch
plan, whose kernel list (mmdgan_plan_describe) the tests and bench.py compare with tests/golden/production_kernels.json.
The oracle is the checker only.
"""
import collections
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'mmd-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

KERNELS_FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'production_kernels.json')


def case_key(config, loss, B):
    return '%s/%s/B%d' % (config, loss, B)


def kernel_multiset(plan_kernels):
    """{'kernel name (workgroups x threads)': launches per step}: what a step IS, launch geometry included - the same
    kernel template at another grid is another tile / split choice"""
    c = collections.Counter('%s [%dx%d]' % (k, g, b) for k, g, b, _ in plan_kernels)
    return dict(sorted(c.items()))


def make_engine(config, loss, B, mode, seed=5):
    import configs
    from mmdgan_hip.engine import GanEngine
    from mmdgan_hip.tape import TapeEngine, needs_tape_engine
    arch, lr = configs.CONFIGS[config]()
    cls = TapeEngine if needs_tape_engine(arch) else GanEngine
    return cls(arch, loss, tuple(lr), batch_size=B, seed=seed, launch_mode=mode), arch, tuple(lr)


def engine_views(eng, B):
    """(fake images NCHW, scores [2B, d]) of the engine's last step, either engine"""
    import torch  # noqa: F401
    if hasattr(eng, 'buf'):
        fake = eng.buf['dis_in'][B:]
        scores = eng.buf[eng.dis.specs[-1].scope + '#y']
    else:
        fake = eng._dis_in[B:]
        scores = eng._last_vals[1][eng.dis.out_val]
    return np.transpose(fake.cpu().numpy(), (0, 3, 1, 2)), scores.cpu().numpy().reshape(2 * B, -1)


def run(config, loss, B, mode='plan', warm=3, check_grads=True, seed=5):
    """`warm` free-running steps of the engine alone (the spectral-norm start vectors are not normalised and D's scores are
    ~1e-14 at the initial variables - SURVEY A.5 #1 - so the first steps' gradients are rounding noise), then ONE step from
    the state reached, teacher-forced: the fp64 oracle starts from the engine's variables and sees the same z / batch.
    In 'plan' mode the first warm step records the plan and the checked step is a replay.  Raises AssertionError on a
    parity failure; returns the measured errors and the step's kernel list."""
    import torch
    from helpers import FLOOR_CLAUSE_USES, RTOL, assert_grads_within_fp32_floor, fp32_floor, knife_edge_budget, l2_err, max_err
    from oracle import restatement as R
    eng, arch, lr = make_engine(config, loss, B, mode, seed)
    c, h, w = arch['input'][0]
    rs = np.random.RandomState(7)

    def batch():
        z = rs.randn(B, arch['code'][0][0]).astype(np.float32)
        real = rs.uniform(-1, 1, (B, c, h, w)).astype(np.float32)
        return z, real

    def nhwc(a):
        return torch.as_tensor(np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))).cuda()
    for _ in range(warm):
        z, real = batch()
        eng.step(nhwc(real), torch.as_tensor(z).cuda())
    prev_vars = eng.get_variables()
    ora = R.OracleGan(arch, loss, lr, dtype=torch.float64, params=prev_vars)
    z, real = batch()
    zt, rt = torch.tensor(z, dtype=torch.float64), torch.tensor(real, dtype=torch.float64)
    # the forward quantities against the oracle's own fp64 evaluation; the gradients (below) against its fp64 evaluation under
    # the engine's sign decisions, audited - one fp64 backward pass at the production batch, not two
    with torch.no_grad():
        lg, ld, stats, upd, (gen, s_x, s_gen) = ora.forward_losses(zt, rt)
    eng.step(nhwc(real), torch.as_tensor(z).cuda())
    fake, scores = engine_views(eng, B)
    out = {'config': config, 'loss': loss, 'B': B, 'mode': mode, 'engine': type(eng).__name__}

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / np.abs(b).max())
    out['err_images'] = rel(fake, gen.detach().numpy())
    out['err_scores'] = max(rel(scores[:B], s_x.detach().numpy()), rel(scores[B:], s_gen.detach().numpy()))
    assert out['err_images'] <= RTOL and out['err_scores'] <= RTOL, out
    losses = eng.losses.cpu().numpy().astype(np.float64)
    escale = float(max(losses[2:5]))
    out['loss_gen'], out['loss_dis'] = [float(losses[0]), float(lg)], [float(losses[1]), float(ld)]
    out['loss_rel_err'] = [abs(losses[0] - float(lg)) / abs(float(lg)), abs(losses[1] - float(ld)) / abs(float(ld))]
    assert abs(losses[0] - float(lg)) <= RTOL * abs(float(lg)) + 4e-7 * escale, out          # (the floor every loss test uses:
    assert abs(losses[1] - float(ld)) <= RTOL * abs(float(ld)) + 4e-7 * escale, out          #  tests/test_ops_gpu.py:44)
    if check_grads:
        grads = eng.get_variables(grad=True)
        floors = fp32_floor(arch, loss, lr, prev_vars, z, real, eng)
        ref_g = floors.audited64()                       # raises unless the engine's decisions differ at knife edges only
        out['audit'] = {'flips': floors.audit[0], 'worst_margin': floors.audit[1], 'activations': floors.elements,
                        'budget': knife_edge_budget(floors.elements),
                        'per_activation': [[n, m] for n, m in floors.per_activation if n]}
        assert sorted(grads) == sorted(ref_g)
        zero = set()
        for net in ('gen', 'dis'):                       # analytically zero gradients (biases in front of a batch norm / behind
            gscale = max(float(np.abs(ref_g[n]).max()) for n in grads if n.startswith(net))   # which only score differences matter)
            zero |= {n for n in grads if n.startswith(net) and np.abs(ref_g[n]).max() <= 1e-9 * gscale}
        assert len(zero) <= 10, zero
        gs = {net: max(float(np.abs(ref_g[n]).max()) for n in grads if n.startswith(net)) for net in ('gen', 'dis')}
        if os.environ.get('SHIPPED_STEP_REPORT'):        # debugging aid: every tensor's error instead of the first failure
            out['per_tensor'] = {n: [l2_err(grads[n], ref_g[n], gs[n[:3]]), max_err(grads[n], ref_g[n], gs[n[:3]])]
                                 for n in grads if n not in zero}
            return out
        n_floor = len(FLOOR_CLAUSE_USES)
        assert_grads_within_fp32_floor(grads, None, floors, skip=zero, what=(config, loss, B, mode))
        out['floor_clause'] = FLOOR_CLAUSE_USES[n_floor:]
        out['audited'] = floors.describe('%s/%s/B%d %s' % (config, loss, B, mode))
        out['grad_err_l2_max'] = max(l2_err(grads[n], ref_g[n], gs[n[:3]]) for n in grads if n not in zero)
        out['grad_err_maxabs_max'] = max(max_err(grads[n], ref_g[n], gs[n[:3]]) for n in grads if n not in zero)
        out['grad_tensors'] = len(grads) - len(zero)
    if mode == 'plan':
        ks = eng.plan_kernels()
        out['kernels'] = kernel_multiset(ks)
        out['launches'] = len(ks)
    out['env'] = sorted(k for k in os.environ if k.startswith('MMDGAN_'))
    return out


def expected_kernels(config, loss, B):
    """the committed kernel list of this case (None: never recorded)"""
    if not os.path.exists(KERNELS_FIXTURE):
        return None
    with open(KERNELS_FIXTURE) as f:
        return json.load(f).get(case_key(config, loss, B))


if __name__ == '__main__':
    cfg, loss_name, batch_size, launch = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    res = run(cfg, loss_name, batch_size, launch, check_grads='--no-grads' not in sys.argv)
    print(json.dumps(res))
