# -*- coding: utf-8 -*-
''' Golden `hybrid` simulations captured from the REFERENCE (NeuronalBilayerSonophore.simulate with
    method='hybrid': nbls.py:356-387 -> HybridSolver, solvers.py:483-633): default run and a run
    with tightened tolerances (odeint rtol=1e-12 for the dense cycles, dop853 rtol=1e-11 for the
    sparse phases). Build container only.
    usage: make_golden_hybrid.py [neuron = RS]      Output: tests/golden/golden_hybrid_<neuron>.npz '''
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
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

logger.setLevel(logging.ERROR)
_odeint, _ode = scipy.integrate.odeint, scipy.integrate.ode


class TightOde(_ode):
    def set_integrator(self, name, **kw):
        kw.update(rtol=1e-11, atol=1e-14)
        return super().set_integrator(name, **kw)


# (A [Pa], tstim, toffset, PRF, DC): intervals of HYBRID_UPDATE_INTERVAL = 0.5 ms
CONFIGS = [(100e3, 1.2e-3, 0.4e-3, 100., 1.0),
           (300e3, 1.0e-3, 0.2e-3, 2e3, 0.5)]
NAME = sys.argv[1] if len(sys.argv) > 1 else 'RS'
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(NAME))
DECIM = 16
res = {'configs': np.array(CONFIGS), 'decimation': np.array(DECIM)}
for i, (A, tstim, toffset, PRF, DC) in enumerate(CONFIGS):
    drive, pp = AcousticDrive(500e3, A), PulsedProtocol(tstim, toffset, PRF, DC)
    solvers.odeint, solvers.ode = _odeint, _ode
    data, meta = nbls.simulate(drive, pp, method='hybrid')
    res[f'c{i}_columns'] = np.array(list(data.columns))
    res[f'c{i}_nrows'] = np.array(data.shape[0])
    res[f'c{i}_t_first_last'] = np.array([data['t'].values[0], data['t'].values[-1]])
    assert np.array_equal(data['t'].values, np.linspace(data['t'].values[0], data['t'].values[-1], data.shape[0]))
    res[f'c{i}_stimstate'] = data['stimstate'].values.astype(np.int8)
    res[f'c{i}_default'] = data.values[::DECIM]
    solvers.odeint = lambda f, y0, t, **k: _odeint(
        f, y0, t, rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34] + [1e-15] * (len(y0) - 3)),
        mxstep=1000000, **k)
    solvers.ode = TightOde
    data_t, _ = nbls.simulate(drive, pp, method='hybrid')
    res[f'c{i}_tight'] = data_t.values[::DECIM]
    print(i, data.shape, data_t.shape, 'tcomp', meta['tcomp'], flush=True)
    np.savez_compressed(os.path.join(HERE, f'golden_hybrid_{NAME}.npz'), **res)
