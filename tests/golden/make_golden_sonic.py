# -*- coding: utf-8 -*-
''' Golden `sonic` simulations captured from the REFERENCE (NeuronalBilayerSonophore.simulate,
    PySONIC/core/nbls.py:513-536 -> __simSonic 389-437), using the 2-D lookup tables produced by
    make_golden_tables.py injected through getLookup2D (the shipped .pkl are LFS stubs).

    For every config two runs are stored:
      * `default`: scipy odeint defaults (rtol = atol ~ 1.49e-8) = what a user of the reference gets
        -> full output array (all columns), spike rows from detectSpikes, firing rate
      * `tight`: odeint rtol=1e-12, atol=1e-15 -> the converged solution (Qm and states)

    Usage (build container only): python tests/golden/make_golden_sonic.py RS [FS ...]
    Output: tests/golden/golden_sonic_<neuron>.npz
'''
import os
import sys
import logging
import numpy as np
import scipy.integrate

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol,  # noqa: E402
                          EffectiveVariablesLookup)
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.postpro import detectSpikes  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

FREQ = 500e3
A_RADIUS = 32e-9

# (A [Pa], tstim, toffset, PRF, DC)
CONFIGS_FULL = [
    (100e3, 100e-3, 50e-3, 100., 1.0),     # BASELINE config 1 (run_astim.py defaults)
    (100e3, 100e-3, 50e-3, 100., 0.5),
    (100e3, 100e-3, 0., 100., 0.5),        # activation-map cell (actmap.py:32)
    (50e3, 100e-3, 50e-3, 100., 1.0),
    (300e3, 100e-3, 50e-3, 100., 1.0),
    (600e3, 100e-3, 50e-3, 100., 1.0),     # upper bound of the A grid
    (200e3, 100e-3, 50e-3, 1000., 0.3),    # 200 events
    (30e3, 100e-3, 0., 10., 0.05),         # single short pulse
    (10e3, 100e-3, 0., 100., 0.05),        # activation-map corner
    (600e3, 100e-3, 0., 100., 1.0),        # activation-map corner, CW with toffset = 0
    (150e3, 100e-3, 0., 100., 0.95),
    (80e3, 20e-3, 10e-3, 100., 1.0),
]
CONFIGS_SHORT = [CONFIGS_FULL[i] for i in (0, 1, 2, 6)]
# neurons with a 0.5 us output step (SWnode, MRGnode): shorter protocols; the last one exceeds
# MAX_NSAMPLES_EFFECTIVE rows, so the reference resamples it (stored every 10th row)
CONFIGS_FAST = [
    (100e3, 10e-3, 5e-3, 100., 1.0),
    (60e3, 10e-3, 5e-3, 1000., 0.5),      # (at 300 kPa the reference's own charge leaves the lookup range)
    (80e3, 40e-3, 20e-3, 100., 0.5),
]
FAST_NEURONS = ('SWnode', 'MRGnode', 'SUseg')

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    return _odeint(f, y0, t, rtol=1e-12, atol=1e-15, mxstep=100000, **kw)


def main(names):
    logger.setLevel(logging.WARNING)
    for name in names:
        d = np.load(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pysonic_amd', 'lookups',
                             f'tables_{name}_32nm_500kHz.npz'))
        keys = [str(k) for k in d['keys']]
        lkp = EffectiveVariablesLookup(
            {'A': d['A'], 'Q': d['Q']}, {k: d[f'tab_{k}'] for k in keys})
        pneuron = getPointNeuron(name)
        nbls = NeuronalBilayerSonophore(A_RADIUS, pneuron)
        nbls.getLookup2D = lambda f, fs: lkp
        configs = CONFIGS_FULL if name == 'RS' else (CONFIGS_FAST if name in FAST_NEURONS else CONFIGS_SHORT)
        out = {'configs': np.array(configs), 'y0': None}
        for i, (A, tstim, toffset, PRF, DC) in enumerate(configs):
            drive = AcousticDrive(FREQ, A)
            pp = PulsedProtocol(tstim, toffset, PRF, DC)
            solvers.odeint = _odeint
            data, meta = nbls.simulate(drive, pp)
            ispikes, props = detectSpikes(data)
            cols = list(data.columns)
            dec = 10 if data.shape[0] > 50000 else 1
            out[f'c{i}_dec'] = dec
            out[f'c{i}_nrows'] = data.shape[0]
            out[f'c{i}_default'] = data.values[::dec]
            out[f'c{i}_spikes'] = np.asarray(ispikes, dtype=np.int64)
            out[f'c{i}_widths'] = np.asarray(props.get('widths', []), dtype=float)
            out[f'c{i}_prominences'] = np.asarray(props.get('prominences', []), dtype=float)
            out[f'c{i}_tcomp'] = meta['tcomp']
            solvers.odeint = tight_odeint
            data_t, _ = nbls.simulate(drive, pp)
            solvers.odeint = _odeint
            ist = cols.index('Vm')
            out[f'c{i}_tight'] = data_t.values[::dec, 2:ist]      # Qm + states
            print(name, i, configs[i], data.shape, 'nspikes', len(ispikes),
                  'rms(default-tight) Qm = %.3e' % np.sqrt(np.mean(
                      (data['Qm'].values - data_t['Qm'].values)**2)), flush=True)
        out['columns'] = np.array(cols)
        out['y0'] = data.values[0, 2:ist]
        np.savez_compressed(os.path.join(HERE, f'golden_sonic_{name}.npz'), **out)


if __name__ == '__main__':
    main(sys.argv[1:] or ['RS'])
