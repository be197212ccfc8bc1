#!/usr/bin/env python3
"""A/B of environment switches on the bench step: runs `python bench.py` once per variant and prints ms/step.
    python tools/ab_env.py [--config cifar] [--modes eager,plan] VARIANT [VARIANT ...]
a VARIANT is 'NAME=VALUE[,NAME=VALUE...]' or 'base' (no switch set)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    config, modes, steps = 'cifar', ['eager', 'plan'], '200'
    while args and args[0].startswith('--'):
        k, v = args[0], args[1]
        args = args[2:]
        if k == '--config':
            config = v
        elif k == '--modes':
            modes = v.split(',')
        elif k == '--steps':
            steps = v
    for rnd in range(2):                                 # two rounds, interleaved: drift between boxes / clocks shows up
        for var in args:
            env = {k: v for k, v in os.environ.items() if not k.startswith('MMDGAN_')}
            if var != 'base':
                for kv in var.split(','):
                    k, v = kv.split('=', 1)
                    env[k] = v
            for mode in modes:
                r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--no-cpu-baseline', '--steps', steps,
                                    '--repeats', '3', '--launch-mode', mode], env=env, capture_output=True, text=True)
                try:
                    out = json.loads(r.stdout.strip().splitlines()[-1])
                    print('%-50s %-6s round %d  %.4f ms/step  regions %s' % (var, mode, rnd, out['ms_per_step'], out['ms_per_step_regions']), flush=True)
                except Exception:
                    print('%-50s %-6s FAILED rc=%d %s' % (var, mode, r.returncode, r.stderr[-500:]), flush=True)


if __name__ == '__main__':
    main()
