# -*- coding: utf-8 -*-
''' method='full' of the REFERENCE at the shape of BASELINE config 5 (pulsed protocol, runs much longer
    than the CW goldens of make_golden_mech.py / make_golden_full_hi.py), RS, a = 32 nm, f = 500 kHz:

      * pw100 / pw600: PulsedProtocol(60 us, 10 us, PRF 50 kHz, DC 0.5) at 100 and 600 kPa: three ON/OFF
        switches of EventDrivenSolver on the detailed system (PySONIC/core/nbls.py:331-354,
        solvers.py:408-415, 445-480), default and rtol = 1e-12 odeint tolerances;
      * cw200: 100 kPa CW, 200 us + 10 us: 9x the longest CW golden so far;
      * stiff neurons (usage: ... stiff): STN at 500 kPa and SUseg at 120 kPa, 4 us + 1 us — the
        configurations on which an explicit pair runs out of its step budget
        while the reference's LSODA switches to BDF.

    Logger at WARNING as for the other detailed-model goldens (no 'log' events).
    Rows are decimated (kept every `dec`-th row plus the row indices around every event).

      * stiff2 (round 3): TC and RE at 600 kPa, 4 us + 1 us.

    Output: tests/golden/golden_full_pw.npz, golden_full_stiff.npz, golden_full_stiff2.npz   (build container only)
'''
import os
import sys
import time
import logging
import numpy as np
import scipy.integrate

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    atol = np.array([1e-12, 1e-21, 1e-34] + [1e-15] * (len(y0) - 3))
    return _odeint(f, y0, t, rtol=1e-12, atol=atol, mxstep=10000000, **kw)


def run(nbls, drive, pp, tags=('default', 'tight')):
    res = {}
    for tag in tags:
        solvers.odeint = _odeint if tag == 'default' else tight_odeint
        t0 = time.perf_counter()
        data, _ = nbls.simulate(drive, pp, 1., 'full')
        res[tag] = data
        print('   ', tag, data.shape, f'{time.perf_counter() - t0:.1f} s', flush=True)
    solvers.odeint = _odeint
    return res


def pack(out, key, res, dec):
    ''' every dec-th row; all rows within 3 of a change of stimstate (the event rows) '''
    data = next(iter(res.values()))
    n = data.shape[0]
    keep = np.zeros(n, bool)
    keep[::dec] = True
    keep[-1] = True
    st = data['stimstate'].values
    for i in np.flatnonzero(np.diff(st) != 0):
        keep[max(i - 3, 0):i + 5] = True
    idx = np.flatnonzero(keep)
    out[f'{key}_rows'] = idx
    out[f'{key}_nrows'] = n
    out[f'{key}_columns'] = np.array(list(data.columns))
    for tag, d in res.items():
        assert d.shape[0] == n
        out[f'{key}_{tag}'] = d.values[idx]


def pulsed():
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    out = {}
    pp = PulsedProtocol(60e-6, 10e-6, 50e3, 0.5)
    for key, A in (('pw100', 100e3), ('pw600', 600e3)):
        print(key, flush=True)
        pack(out, key, run(nbls, AcousticDrive(500e3, A), pp), 4)
        out[f'{key}_cfg'] = np.array([500e3, A, 60e-6, 10e-6, 50e3, 0.5])
    print('cw200', flush=True)
    pack(out, 'cw200', run(nbls, AcousticDrive(500e3, 100e3), PulsedProtocol(200e-6, 10e-6)), 10)
    out['cw200_cfg'] = np.array([500e3, 100e3, 200e-6, 10e-6, 100., 1.])
    np.savez_compressed(os.path.join(HERE, 'golden_full_pw.npz'), **out)


def stiff():
    out = {}
    for name, A in (('STN', 500e3), ('SUseg', 120e3)):
        print(name, A, flush=True)
        nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        pack(out, name, run(nbls, AcousticDrive(500e3, A), PulsedProtocol(4e-6, 1e-6)), 1)
        out[f'{name}_cfg'] = np.array([500e3, A, 4e-6, 1e-6, 100., 1.])
    np.savez_compressed(os.path.join(HERE, 'golden_full_stiff.npz'), **out)


def stiff2():
    ''' round 3: the configurations the row-cooperative kernel gives up as stiff besides STN -- TC and RE at
        600 kPa (the O / C pair of iH, the T-type gates under the full swing of Vm) -- 4 us + 1 us '''
    out = {}
    for name, A in (('TC', 600e3), ('RE', 600e3)):
        print(name, A, flush=True)
        nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        pack(out, name, run(nbls, AcousticDrive(500e3, A), PulsedProtocol(4e-6, 1e-6)), 1)
        out[f'{name}_cfg'] = np.array([500e3, A, 4e-6, 1e-6, 100., 1.])
    np.savez_compressed(os.path.join(HERE, 'golden_full_stiff2.npz'), **out)


if __name__ == '__main__':
    logger.setLevel(logging.WARNING)
    for w in (sys.argv[1:] or ['pulsed', 'stiff', 'stiff2']):
        {'pulsed': pulsed, 'stiff': stiff, 'stiff2': stiff2}[w]()
