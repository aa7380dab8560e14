# -*- coding: utf-8 -*-
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
LOOKUPS = os.path.join(ROOT, 'pysonic_amd', 'lookups')
NEURONS = ['RS', 'FS', 'LTS', 'RE', 'TC', 'STN']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def load_tables(name):
    ''' (A, Q, keys, tables[ntab, nA, nQ]) of the shipped 2-D lookup of a neuron. '''
    d = np.load(os.path.join(LOOKUPS, f'tables_{name}_32nm_500kHz.npz'))
    keys = [str(k) for k in d['keys']]
    return d['A'], d['Q'], keys, np.array([d[f'tab_{k}'] for k in keys])


def load_golden(fname):
    return np.load(os.path.join(GOLDEN, fname), allow_pickle=False)


@pytest.fixture(scope='session')
def native():
    ''' The ctypes binding; building the library if hipcc is around and it is stale. '''
    from pysonic_amd import build as nbuild
    try:
        nbuild.build()
    except nbuild.HipccNotFound:
        # a box without ROCm's compiler may use the library that travelled with the tree; a
        # compile ERROR (nbuild.CompileError) is never papered over with a stale binary
        if not os.path.isfile(nbuild.OUT):
            raise
    from pysonic_amd import _native
    _native.load()
    return _native


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b))**2)))
