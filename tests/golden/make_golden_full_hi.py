# -*- coding: utf-8 -*-
''' method='full' of the REFERENCE at the top of the amplitude range (RS, a = 32 nm, f = 500 kHz,
    A = 600 kPa, 8 us + 2 us; NeuronalBilayerSonophore.simulate, PySONIC/core/nbls.py:331-354), default
    and rtol = 1e-12 runs like make_golden_mech.py::full: the configuration that takes the most steps in
    BASELINE config 5 (~8500 per simulated microsecond for an explicit 5(4) pair against ~2500 at 100 kPa),
    and the one on which the detailed-model kernels differ most from one another.

    Output: tests/golden/golden_full_RS_600kPa.npz   (build container only; rows decimated by 2)
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
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    atol = np.array([1e-12, 1e-21, 1e-34] + [1e-15] * (len(y0) - 3))
    return _odeint(f, y0, t, rtol=1e-12, atol=atol, mxstep=1000000, **kw)


if __name__ == '__main__':
    logger.setLevel(logging.ERROR)
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    drive, pp = AcousticDrive(500e3, 600e3), PulsedProtocol(8e-6, 2e-6)
    out = {'A': 600e3, 'tstim': 8e-6, 'toffset': 2e-6, 'dec': 2}
    for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
        solvers.odeint = ode
        data, meta = nbls.simulate(drive, pp, 1., 'full')
        out[tag] = data.values[::2]
        out['nrows'] = data.shape[0]
        out['columns'] = np.array(list(data.columns))
        print('full 600 kPa', tag, data.shape, meta['tcomp'], flush=True)
    solvers.odeint = _odeint
    np.savez_compressed(os.path.join(HERE, 'golden_full_RS_600kPa.npz'), **out)
