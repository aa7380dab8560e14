# -*- coding: utf-8 -*-
''' Build libpysonic_amd.so in-tree:  python -m pysonic_amd.build
    hipcc cross-compiles for gfx950 without a GPU. '''
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(PKG, 'csrc', f) for f in ('sonic_lib.hip', 'mech_lib.hip', 'full_lib.hip')]
OUT_DIR = os.path.join(PKG, '_lib')
OUT = os.path.join(OUT_DIR, 'libpysonic_amd.so')
DEPS = SRCS + [os.path.join(PKG, 'csrc', f) for f in ('sonic_integrator.hpp', 'sonic_models.hpp',
                                                        'mech_core.hpp', 'full_core.hpp', 'lib_common.hpp')] \
    + [os.path.join(os.path.dirname(PKG), 'include', 'pysonic_amd.h')]


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError('hipcc not found: the native library can only be built with ROCm')


def up_to_date():
    if not os.path.isfile(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS if os.path.isfile(d))


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [find_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-o', OUT] + SRCS
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'hipcc failed:\n{res.stdout}\n{res.stderr}')
    if verbose:
        print(res.stderr)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
