# -*- coding: utf-8 -*-
''' Import helper for the golden-vector generators (THIS CONTAINER ONLY).

    The upstream reference at /root/reference is pure Python but imports four cosmetic
    packages that are absent from this image (colorlog, lockfile, boltons, tkinter) and uses
    two removed APIs (matplotlib>=3.9 plt.register_cmap, numpy>=1.24 np.float). This module
    writes inert stand-ins for those into a scratch directory OUTSIDE the repo, patches the two
    APIs, and puts /root/reference on sys.path. None of this touches numerics.

    Nothing under tests/ (other than the make_golden_*.py generators), oracle/ or the product
    imports this file; /root/reference does not exist on the GPU box.
'''
import os
import sys
import tempfile

REFERENCE_ROOT = '/root/reference'

_SHIMS = {
    'colorlog.py': '''
import logging
class ColoredFormatter(logging.Formatter):
    def __init__(self, fmt=None, datefmt=None, reset=True, log_colors=None, style='%', **kw):
        super().__init__((fmt or '%(message)s').replace('%(log_color)s', ''), datefmt, style)
StreamHandler = logging.StreamHandler
getLogger = logging.getLogger
''',
    'lockfile.py': '''
class FileLock:
    def __init__(self, path): self.path = path
    def acquire(self, *a, **k): pass
    def release(self): pass
''',
    'boltons/__init__.py': '',
    'boltons/strutils.py': '''
def cardinalize(word, n): return word if n == 1 else word + 's'
''',
    'tkinter/__init__.py': '''
class Tk:
    def withdraw(self): pass
''',
    'tkinter/filedialog.py': '''
def _no(*a, **k): raise RuntimeError('no GUI')
askopenfilenames = askdirectory = asksaveasfilename = _no
''',
}


def setup():
    ''' Make `import PySONIC` (the reference) work in this process. '''
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f'{REFERENCE_ROOT} not present: golden vectors can only be '
                           'regenerated in the build container')
    sys.dont_write_bytecode = True
    shimdir = os.path.join(tempfile.gettempdir(), 'pysonic_ref_shims')
    for rel, src in _SHIMS.items():
        path = os.path.join(shimdir, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as fh:
            fh.write(src)
    for p in (REFERENCE_ROOT, shimdir):
        if p not in sys.path:
            sys.path.insert(0, p)
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import numpy as np
    if not hasattr(plt, 'register_cmap'):
        def register_cmap(name=None, cmap=None, **kw):
            try:
                matplotlib.colormaps.register(cmap, name=name or cmap.name)
            except ValueError:
                pass
        plt.register_cmap = register_cmap
    if not hasattr(np, 'float'):
        np.float = float
    import warnings
    warnings.filterwarnings('ignore')
