# -*- coding: utf-8 -*-
''' `sonic` goldens at a SECOND ultrasound frequency (the f axis of BASELINE config 4).

    The reference has no lookups here at all (LFS stubs), and generating one more full table with
    its computeEffVars costs ~10 core-minutes; instead the (A, Q) tables were generated ON THE
    DEVICE (tools/r02_probe.py tables: RS 32 nm 100 kHz, LTS 32 nm 2 MHz -- the same
    mech_batch_run that tests/test_gpu_mech.py and test_gpu_axes.py hold to 1e-6 of the
    reference's computeEffVars), committed as tests/golden/devtables_<neuron>_32nm_<f>kHz.npz,
    and the REFERENCE's NeuronalBilayerSonophore.simulate (PySONIC/core/nbls.py:513-536) is fed
    with exactly those tables through getLookup2D. Default and rtol = 1e-12 runs, as in
    make_golden_sonic.py.

    Output: tests/golden/golden_sonic_freq.npz      (build container only)
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

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    return _odeint(f, y0, t, rtol=1e-12, atol=1e-15, mxstep=100000, **kw)


# (A [Pa], tstim, toffset, PRF, DC)
CASES = {
    ('RS', 100e3): [(100e3, 100e-3, 50e-3, 100., 1.0), (40e3, 100e-3, 50e-3, 100., 0.5),
                    (300e3, 100e-3, 0., 100., 0.3), (600e3, 50e-3, 10e-3, 1000., 0.2)],
    ('LTS', 2e6): [(150e3, 100e-3, 50e-3, 100., 1.0), (80e3, 100e-3, 50e-3, 10., 0.5),
                   (400e3, 100e-3, 0., 100., 0.3), (600e3, 50e-3, 10e-3, 100., 1.0)],
}


def main():
    logger.setLevel(logging.ERROR)
    out = {}
    for (name, f), configs in CASES.items():
        d = np.load(os.path.join(HERE, f'devtables_{name}_32nm_{f * 1e-3:.0f}kHz.npz'))
        keys = [str(k) for k in d['keys']]
        lkp = EffectiveVariablesLookup({'A': d['A'], 'Q': d['Q']}, {k: d[f'tab_{k}'] for k in keys})
        nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        nbls.getLookup2D = lambda f_, fs_, _l=lkp: _l
        out[f'{name}_f'] = f
        out[f'{name}_configs'] = np.array(configs)
        for i, (A, tstim, toffset, PRF, DC) in enumerate(configs):
            drive, pp = AcousticDrive(f, A), PulsedProtocol(tstim, toffset, PRF, DC)
            solvers.odeint = _odeint
            data, _ = nbls.simulate(drive, pp)
            ispikes, _ = detectSpikes(data)
            cols = list(data.columns)
            ist = cols.index('Vm')
            solvers.odeint = tight_odeint
            data_t, _ = nbls.simulate(drive, pp)
            solvers.odeint = _odeint
            out[f'{name}_c{i}_default'] = data.values
            out[f'{name}_c{i}_tight'] = data_t.values[:, 2:ist]
            out[f'{name}_c{i}_spikes'] = np.asarray(ispikes, dtype=np.int64)
            out[f'{name}_columns'] = np.array(cols)
            print(name, f, configs[i], data.shape, 'nspikes', len(ispikes), 'rms(default-tight) Qm = %.3e' %
                  np.sqrt(np.mean((data['Qm'].values - data_t['Qm'].values)**2)), flush=True)
    np.savez_compressed(os.path.join(HERE, 'golden_sonic_freq.npz'), **out)


if __name__ == '__main__':
    main()
