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


@pytest.fixture(scope='session')
def _session_cache(tmp_path_factory):
    return tmp_path_factory.mktemp('pysonic_amd_cache')


@pytest.fixture(autouse=True)
def _isolated_caches(tmp_path, _session_cache, monkeypatch):
    ''' No test reads or writes the user's ~/.cache/pysonic_amd: the titration log is a fresh file per test (a
        logged threshold would otherwise answer for the kernels after the first run on a box), generated lookups
        are shared within the session only. '''
    import pysonic_amd.core.nbls as nbls_mod
    monkeypatch.setattr(nbls_mod, 'TITRATION_LOG', str(tmp_path / 'astim_titrations.log'))
    monkeypatch.setattr(nbls_mod, 'GENERATED_LOOKUP_DIR', str(_session_cache))
    monkeypatch.setenv('PYSONIC_AMD_TITRATIONS', str(tmp_path / 'astim_titrations.log'))
    monkeypatch.setenv('PYSONIC_AMD_CACHE', str(_session_cache))


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
