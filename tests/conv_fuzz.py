"""Random convolution geometries through the library's own dispatch: the adjointness identity
<conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)> (three kernels, one bilinear form), the caller-transformed-weights
route against the library's own, and every image's independence of its batch neighbours.

A debugging aid and a test body: tests/test_ops_gpu.py runs a fixed-seed sample in its own process (under tests/conftest.py's
thresholds) and in a subprocess without any MMDGAN_* variable (the production kernel selection);
    python tests/conv_fuzz.py <seed> <cases>      prints one JSON line {"cases": n, "bad": [...]}
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'mmd-gan_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CHANNELS = [3, 8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512]
SIZES = [4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64]
KERNELS = [(1, 1), (3, 1), (3, 1), (3, 1), (4, 2), (4, 2), (4, 2), (3, 2), (5, 1), (2, 2), (5, 2)]


def sample(rs, budget=24e6):
    """one geometry (N, H, W, C, K, R, stride) whose largest tensor stays under `budget` elements"""
    while True:
        R, s = KERNELS[rs.randint(len(KERNELS))]
        C, K = CHANNELS[rs.randint(len(CHANNELS))], CHANNELS[rs.randint(len(CHANNELS))]
        H = SIZES[rs.randint(len(SIZES))]
        W = H if rs.rand() < 0.7 else SIZES[rs.randint(len(SIZES))]
        N = int(rs.choice([1, 2, 3, 5, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 200]))
        if R > min(H, W) + 1:
            continue
        if N * H * W * max(C, K) <= budget and R * R * C * K <= 8e6:
            return N, H, W, C, K, R, s


def sweep_cases():
    """the deterministic part: every (C, K) pair of the Winograd kernels' channel classes at three sizes each with MANY work
    items per persistent workgroup - short reductions (32 / 64 channels = one- and two-stage items) included: the shape class
    of the F(2x2,2x2) output-offset bug of round 4"""
    out = []
    for (R, s) in ((3, 1), (4, 2)):
        for C in (32, 64, 128, 256):
            for K in (32, 64, 128, 256):
                for (N, H) in ((48, 64), (192, 16), (24, 32)):
                    out.append((N, H, H, C, K, R, s))
    return out


def check(ops, torch, case, seed):
    N, H, W, C, K, R, s = case
    g = torch.Generator(device='cuda').manual_seed(seed)
    P, Q = -(-H // s), -(-W // s)
    x = torch.empty(N, H, W, C, device='cuda').uniform_(-1, 1, generator=g)
    w = torch.randn(R, R, C, K, device='cuda', generator=g) / float(np.sqrt(R * R * C))
    dy = torch.randn(N, P, Q, K, device='cuda', generator=g)
    y = ops.conv2d_fwd(x, w, s)
    dx = ops.conv2d_dgrad(dy, w, (H, W), s)
    dw = ops.conv2d_wgrad(x, dy, R, s)

    def dot(a, b):
        return float((a.double() * b.double()).sum())
    form, scale = dot(y, dy), float(y.double().norm() * dy.double().norm()) + 1e-30
    errs = {'dgrad': abs(dot(x, dx) - form) / scale, 'wgrad': abs(dot(w, dw) - form) / scale}
    if ops.wgrad_algo(N, H, W, C, K, R, s) == ops.WINO_F43:
        # the F(4x4,3x3) weight gradient (csrc/conv_wino43w.hip) rounds at 1-4e-6 of the scale (constants up to 8 in its
        # transforms, up to 2048 products per accumulator: 2.8e-6 measured on CelebA's 16x16x256 layer at 384 rows): bar 5e-6
        errs['wgrad'] /= 5.0
    if (R, s) in ((3, 1), (4, 2)):                       # the route the engines take: weights transformed by the caller
        for dgrad, ref, name in ((False, y, 'fwd_wino'), (True, dx, 'dgrad_wino')):
            algos = [None] if ops.wino_eligible(N, H, W, C, K, R, s, dgrad) else []          # F(2x2,3x3) / F(2x2,2x2) ...
            if ops.wino_algo(N, H, W, C, K, R, s, dgrad) == ops.WINO_F43:                    # ... and F(4x4,3x3) where it applies
                algos.append(ops.WINO_F43)
            for algo in algos:
                u = ops.wino_transform(w, dgrad, algo=algo)
                got = ops.conv2d_dgrad(dy, w, (H, W), s, wino=u) if dgrad else ops.conv2d_fwd(x, w, s, wino=u)
                # two routes to the same convolution: within 1e-5 of the tensor's scale, 3e-5 where one of them is F(4x4,3x3) (its
                # own rounding is 3-5e-6 of the scale, up to 1e-5 at 512-channel reductions: tools/wino43_gate.py) - on the bar's scale
                f43 = algo == ops.WINO_F43 or ops.wino_algo(N, H, W, C, K, R, s, dgrad) == ops.WINO_F43
                e = float((got - ref).abs().max() / (ref.abs().max() + 1e-30)) / (30.0 if f43 else 10.0)
                errs[name] = max(errs.get(name, 0.0), e)
    # the fused epilogues against the same launch with a linear epilogue, finished in torch: scale, bias, activation forward;
    # scale and activation derivative backward, the operand holding 2/3 of the images where the batch allows (the 3B-row
    # wrap of the discriminator's joint backward pass: the last third reuses the last third of the operand's images)
    b = torch.randn(K, device='cuda', generator=g) * 0.1
    sc = torch.tensor([0.61], device='cuda')
    yl = ops.conv2d_fwd(x, w, s, bias=b, scale=sc, act='lrelu')
    ref = 0.61 * y + b
    ref = torch.where(ref > 0, ref, 0.1 * ref)
    errs['epilogue_fwd'] = float((yl - ref).abs().max() / (ref.abs().max() + 1e-30)) / 10.0
    rows = 2 * N // 3 if (N % 3 == 0 and N >= 3) else N
    yp = torch.randn(rows, H, W, C, device='cuda', generator=g)
    full = torch.cat([yp, yp[rows - (N - rows):]], 0) if rows < N else yp
    dl = ops.conv2d_dgrad(dy, w, (H, W), s, scale=sc, act='lrelu', dact_of=yp, dact_batch=rows if rows < N else 0)
    ref = 0.61 * dx * torch.where(full > 0, torch.ones_like(full), torch.full_like(full, 0.1))
    errs['epilogue_bwd'] = float((dl - ref).abs().max() / (ref.abs().max() + 1e-30)) / 10.0
    if N > 1:                                            # the last image alone
        alone = ops.conv2d_fwd(x[N - 1:].contiguous(), w, s)
        errs['batch'] = float((alone[0] - y[N - 1]).abs().max() / (y[N - 1].abs().max() + 1e-30)) / 100.0   # bar 1e-4
    return errs


def run(seed, cases, bar=1e-6, sweep=True):
    import torch
    from mmdgan_hip import ops
    ops.require_device()
    ops.set_workspace(128 << 20)
    rs = np.random.RandomState(seed)
    bad = []
    todo = [sample(rs) for _ in range(cases)] + (sweep_cases() if sweep else [])
    for i, case in enumerate(todo):
        errs = check(ops, torch, case, seed * 100003 + i)
        worst = max(errs.values())
        if not worst <= bar:
            bad.append({'case': list(case), 'errs': errs})
    ops.require_device().mmdgan_set_workspace(None, 0)
    return {'seed': seed, 'cases': len(todo), 'bad': bad, 'env': sorted(k for k in os.environ if k.startswith('MMDGAN_'))}


if __name__ == '__main__':
    print(json.dumps(run(int(sys.argv[1]), int(sys.argv[2]))))
