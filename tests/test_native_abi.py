# -*- coding: utf-8 -*-
''' The C-ABI library: loads, exports every symbol include/pysonic_amd.h declares, and its pure
    host entry points agree with the oracle. No GPU compute calls here. '''
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O


def declared_symbols():
    with open(os.path.join(ROOT, 'include', 'pysonic_amd.h')) as fh:
        src = fh.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b((?:sonic|mech|full|hybrid)_[a-z_0-9]+)\s*\(', src)))


def test_exports_match_header(native):
    syms = declared_symbols()
    assert len(syms) >= 22 and 'full_batch_run' in syms and 'hybrid_batch_run' in syms
    lib = ctypes.CDLL(native.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/pysonic_amd.h but not exported'
        assert s in native.SIGNATURES, f'{s} has no ctypes prototype in pysonic_amd/_native.py'
    assert native.load().sonic_abi_version() == native.ABI_VERSION == 6


def test_neuron_dimensions(native):
    lib = native.load()
    for name in ('RS', 'FS', 'LTS', 'RE', 'TC', 'STN'):
        nid = native.NEURON_IDS[name]
        ns = lib.sonic_neuron_nstates(nid)
        if ns < 0:
            continue   # not on the device yet
        assert ns == len(O.STATES[name])
        assert lib.sonic_neuron_ntables(nid) == 1 + len(O.RATES[name])
    assert lib.sonic_neuron_nstates(99) < 0


def test_count_rows_matches_reference_grid(native):
    ''' row counts of EventDrivenSolver for CW / PW / zero-offset protocols '''
    cfgs = [(0.1, 0.05, 100., 1.0), (0.1, 0.05, 100., 0.5), (0.1, 0., 100., 0.5),
            (0.1, 0., 100., 1.0), (0.1, 0.05, 1000., 0.3), (0.02, 0.01, 100., 1.0),
            (1.0, 0.1, 10., 0.33)]
    tstop, dt, ev_t, ev_off = [], [], [], [0]
    expected = []
    for tstim, toffset, PRF, DC in cfgs:
        ev, ts = O.pulsed_events(tstim, toffset, PRF, DC)
        tstop.append(ts); dt.append(5e-5)
        ev_t += [e[0] for e in ev]; ev_off.append(len(ev_t))
        n, tnow = 1, 0.
        for te in [e[0] for e in ev] + [ts]:
            n += O.get_nsamples(tnow, te, 5e-5)
            tnow = te
        expected.append(n)
    got = native.count_rows(tstop, dt, ev_t, ev_off)
    assert list(got) == expected
    assert expected[:4] == [3003, 3003, 2003, 2005]    # SURVEY 8(a) A1 + golden shapes
    with pytest.raises(ValueError):
        native.count_rows([0.1], [5e-5], [0.2], [0, 1])    # event after tstop


def test_default_opts_and_errors(native):
    o = native.default_opts()
    assert (o.rtol, o.atol, o.write_traces, o.chunks) == (0., 0., 1, 0)      # tolerances 0: the kernel's own
    with pytest.raises(TypeError):
        native.default_opts(bogus=1)
    lib = native.load()
    if lib.sonic_device_count() == 0:
        # no silent fallback: creating a model without a GPU fails loudly
        with pytest.raises(native.NativeLibraryError):
            native.SonicModel('RS', np.zeros(7), np.zeros((9, 2, 2)), [0., 1.], [0., 1.])
