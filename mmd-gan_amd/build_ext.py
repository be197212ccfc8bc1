#!/usr/bin/env python3
"""Build libmmdgan_hip.so (gfx950) in-tree: mmd-gan_amd/lib/libmmdgan_hip.so.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their source (or a header)
is newer.  Usage: python mmd-gan_amd/build_ext.py [--force]
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'lib', 'libmmdgan_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-result']


def _newer(src, deps, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, '..', 'include', 'mmdgan_hip.h')]
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        if force or _newer(s, hdrs, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        for (s, o), r in ex.map(cc, jobs):
            if verbose and (r.returncode != 0 or r.stderr.strip()):
                sys.stderr.write(r.stderr)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s' % s)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError('link failed')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
