# -*- coding: utf-8 -*-
''' Golden vectors for the mechanical path, captured from the REFERENCE:
      * BilayerSonophore.simCycles (PySONIC/core/bls.py:749-789) via PeriodicSolver
        (PySONIC/core/solvers.py:224-365): last-cycle Z / ng traces, total row count
        (-> number of cycles) for 23 (A, Q) pairs including A = 0 (11-cycle quirk)
      * NeuronalBilayerSonophore.computeEffVars (PySONIC/core/nbls.py:153-222): effective
        variables for the same pairs, default odeint tolerances AND rtol=1e-12 ("tight")
      * y0 / Qm0 / Qbounds / rate-function samples of the six BASELINE neurons
        (pins A7 of SURVEY.md section 8)
      * NeuronalBilayerSonophore.simulate(method='full') (nbls.py:331-354) for RS, 20 us + 4 us

    Output: tests/golden/golden_mech.npz, tests/golden/golden_full_RS.npz, golden_neurons.npz
    (build container only)
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
from PySONIC.core import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol)  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    # absolute tolerances scaled to the variables: U (m/s), Z (m), ng (mol), then Qm and states
    atol = np.array([1e-12, 1e-21, 1e-34] + [1e-15] * (len(y0) - 3))
    return _odeint(f, y0, t, rtol=1e-12, atol=atol, mxstep=1000000, **kw)


NEURONS = ['RS', 'FS', 'LTS', 'RE', 'TC', 'STN']


def neurons():
    out = {}
    Vsamples = np.linspace(-150., 60., 43)
    for name in NEURONS:
        pn = getPointNeuron(name)
        out[f'{name}_states'] = np.array(pn.statesNames())
        out[f'{name}_rates'] = np.array(list(pn.effRates().keys()))
        out[f'{name}_y0'] = np.array([pn.steadyStates()[k](pn.Vm0) for k in pn.statesNames()])
        out[f'{name}_Qm0'] = pn.Qm0
        out[f'{name}_Vm0'] = pn.Vm0
        out[f'{name}_Qbounds'] = pn.Qbounds
        out[f'{name}_ratevals'] = np.array(
            [[float(f(V)) for V in Vsamples] for f in pn.effRates().values()])
        # iNet and true derivatives at a few (Vm, states) points
        rng = np.random.default_rng(1234)
        pts, inet, ders = [], [], []
        for _ in range(8):
            Vm = rng.uniform(-120, 40)
            x = {k: rng.uniform(0.05, 0.95) for k in pn.statesNames()}
            if 'Cai' in x:
                x['Cai'] = rng.uniform(1e-8, 1e-6)
            pts.append([Vm] + list(x.values()))
            inet.append(float(pn.iNet(Vm, x)))
            ders.append([float(pn.derStates()[k](Vm, x)) for k in pn.statesNames()])
        out[f'{name}_pts'] = np.array(pts)
        out[f'{name}_iNet'] = np.array(inet)
        out[f'{name}_ders'] = np.array(ders)
    out['Vsamples'] = Vsamples
    np.savez_compressed(os.path.join(HERE, 'golden_neurons.npz'), **out)
    print('neurons done', flush=True)


def mech():
    pn = getPointNeuron('RS')
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    f = 500e3
    pairs = [(A, Q) for A in (0., 1e3, 20e3, 100e3, 300e3, 600e3)
             for Q in (-107e-5, -71.9e-5, 0., 50e-5)][:-1]   # 23 pairs
    out = {'pairs': np.array(pairs), 'f': f}
    keys = None
    for i, (A, Q) in enumerate(pairs):
        drive = AcousticDrive(f, A)
        for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
            solvers.odeint = ode
            data = nbls.simCycles(drive, Q)
            effvars, _ = nbls.computeEffVars(drive, 1., Q)
            if keys is None:
                keys = list(effvars[0].keys())
            out[f'p{i}_{tag}_nrows'] = len(data)
            out[f'p{i}_{tag}_Z'] = data['Z'].values[-1000:]
            out[f'p{i}_{tag}_ng'] = data['ng'].values[-1000:]
            out[f'p{i}_{tag}_eff'] = np.array([effvars[0][k] for k in keys])
            if tag == 'default' and i < 3:
                out[f'p{i}_t'] = data['t'].values
        solvers.odeint = _odeint
        print('mech', i, A, Q, out[f'p{i}_default_nrows'], out[f'p{i}_tight_nrows'],
              np.abs(out[f'p{i}_default_eff'] / out[f'p{i}_tight_eff'] - 1).max(), flush=True)
    out['keys'] = np.array(keys)
    # fs < 1 (spatial averaging) on one pair
    solvers.odeint = _odeint
    effvars, _ = nbls.computeEffVars(AcousticDrive(f, 100e3), np.array([0.5, 0.75, 1.0]), -71.9e-5)
    out['fs_vals'] = np.array([0.5, 0.75, 1.0])
    out['fs_eff'] = np.array([[e[k] for k in keys] for e in effvars])
    np.savez_compressed(os.path.join(HERE, 'golden_mech.npz'), **out)
    print('mech done', flush=True)


def full():
    pn = getPointNeuron('RS')
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    drive = AcousticDrive(500e3, 100e3)
    pp = PulsedProtocol(20e-6, 4e-6)
    out = {}
    for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
        solvers.odeint = ode
        data, meta = nbls.simulate(drive, pp, 1., 'full')
        out[f'{tag}'] = data.values
        out['columns'] = np.array(list(data.columns))
        print('full', tag, data.shape, meta['tcomp'], flush=True)
    solvers.odeint = _odeint
    np.savez_compressed(os.path.join(HERE, 'golden_full_RS.npz'), **out)


if __name__ == '__main__':
    logger.setLevel(logging.ERROR)
    which = sys.argv[1:] or ['neurons', 'mech', 'full']
    for w in which:
        {'neurons': neurons, 'mech': mech, 'full': full}[w]()
