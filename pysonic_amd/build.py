# -*- coding: utf-8 -*-
''' Build libpysonic_amd.so in-tree:  python -m pysonic_amd.build
    hipcc cross-compiles for gfx950 without a GPU. One object per translation unit (compiled in
    parallel), then one link. '''
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OUT_DIR = os.path.join(PKG, '_lib')
OUT = os.path.join(OUT_DIR, 'libpysonic_amd.so')
STAMP = os.path.join(OUT_DIR, 'source_hash.txt')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


class HipccNotFound(RuntimeError):
    ''' No ROCm toolchain on this machine (the only build failure a caller may tolerate when a
        previously built library is present). '''


class CompileError(RuntimeError):
    ''' hipcc ran and failed: a stale library must NOT be used in its place. '''


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def deps():
    ''' everything the library is made of: every file under csrc/ and the public header '''
    return sorted(glob.glob(os.path.join(CSRC, '*'))) + \
        sorted(glob.glob(os.path.join(os.path.dirname(PKG), 'include', '*.h')))


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.isfile(cand):
            return cand
    raise HipccNotFound('hipcc not found: the native library can only be built with ROCm')


def source_hash():
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for d in deps():
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def up_to_date():
    ''' the library on disk was built from exactly these sources (content hash, not mtimes: the
        tree is copied to the GPU box) '''
    if not (os.path.isfile(OUT) and os.path.isfile(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == source_hash()


def _compile(hipcc, src, obj, verbose):
    cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise CompileError(f'hipcc failed on {os.path.basename(src)}:\n{res.stdout}\n{res.stderr}')
    return res.stderr


def object_hash(src):
    ''' what an object file is made of: the flags, its translation unit and every header (any of them may be
        included). Content, not mtimes: the tree -- objects included -- is copied to the GPU box, where an object
        older than its edited source can carry a newer mtime. '''
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for d in [src] + [d for d in deps() if not d.endswith('.hip')]:
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _object_current(src, obj):
    try:
        with open(obj + '.hash') as fh:
            return os.path.isfile(obj) and fh.read().strip() == object_hash(src)
    except OSError:
        return False


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    hipcc = find_hipcc()
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sources()
    objs = [os.path.join(OUT_DIR, os.path.basename(s)[:-4] + '.o') for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if force or not _object_current(s, o)]

    def one(so):
        src, obj = so
        if os.path.isfile(obj + '.hash'):
            os.remove(obj + '.hash')
        log = _compile(hipcc, src, obj, verbose)
        with open(obj + '.hash', 'w') as fh:
            fh.write(object_hash(src) + '\n')
        return log
    with ThreadPoolExecutor(max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        logs = list(pool.map(one, todo))
    if verbose:
        print('\n'.join(logs))
    res = subprocess.run([hipcc] + FLAGS + ['-shared', '-o', OUT] + objs, capture_output=True, text=True)
    if res.returncode != 0:
        raise CompileError(f'link failed:\n{res.stdout}\n{res.stderr}')
    with open(STAMP, 'w') as fh:
        fh.write(source_hash() + '\n')
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
