#!/usr/bin/env python3
"""Record the kernel list of one training step of every BASELINE.json config under the library's production kernel
selection (no MMDGAN_* variable set) -> gpurun_out/production_kernels.json; copy it to tests/golden/ when the kernel
selection was changed on purpose.  tests/test_production_gpu.py and bench.py compare against the committed copy.
    python tools/record_production_kernels.py            (on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import shipped_step  # noqa: E402

CASES = [('cifar', 'rep', 64), ('stl', 'rmb', 64), ('celeba', 'rep', 128), ('lsun_resnet', 'rep', 32)]


def main():
    env = {k: v for k, v in os.environ.items() if not k.startswith('MMDGAN_')}
    out = {}
    for config, loss, B in CASES:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'shipped_step.py'), config, loss, str(B), 'plan', '--no-grads'],
                           env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit('%s failed:\n%s' % (config, r.stderr[-3000:]))
        res = json.loads(r.stdout.strip().splitlines()[-1])
        out[shipped_step.case_key(config, loss, B)] = res['kernels']
        print(config, res['launches'], 'launches', file=sys.stderr)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'production_kernels.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')


if __name__ == '__main__':
    main()
