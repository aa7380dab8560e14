# -*- coding: utf-8 -*-
''' Golden vectors for one more point neuron (usage: make_golden_neuron.py HHseg [SWnode ...]), captured
    from the REFERENCE (PySONIC/neurons/*.py): definition samples (states, rate functions, resting
    state, iNet and true derivatives at random points -- as golden_neurons.npz holds for the six
    BASELINE neurons) and NeuronalBilayerSonophore.computeEffVars for a few (A, Q) cells at default
    and tight odeint tolerances, and NeuronalBilayerSonophore.simulate(method='full') for 4 us + 1 us
    (nbls.py:331-354) at both tolerances.

    Output: tests/golden/golden_<name>.npz (build container only)
'''
import os
import sys
import numpy as np
import scipy.integrate

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol  # noqa: E402
import logging  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    atol = np.array([1e-12, 1e-21, 1e-34] + [1e-15] * (len(y0) - 3))
    return _odeint(f, y0, t, rtol=1e-12, atol=atol, mxstep=1000000, **kw)


def main(name):
    pn = getPointNeuron(name)
    out = {}
    Vsamples = np.linspace(-150., 60., 43)
    out['Vsamples'] = Vsamples
    out[f'{name}_states'] = np.array(pn.statesNames())
    out[f'{name}_rates'] = np.array(list(pn.effRates().keys()))
    out[f'{name}_y0'] = np.array([pn.steadyStates()[k](pn.Vm0) for k in pn.statesNames()])
    out[f'{name}_Qm0'] = pn.Qm0
    out[f'{name}_Vm0'] = pn.Vm0
    out[f'{name}_Qbounds'] = pn.Qbounds
    out[f'{name}_ratevals'] = np.array([[float(f(V)) for V in Vsamples] for f in pn.effRates().values()])
    rng = np.random.default_rng(1234)
    pts, inet, ders = [], [], []
    for _ in range(8):
        Vm = rng.uniform(-120, 40)
        x = {k: rng.uniform(0.05, 0.95) for k in pn.statesNames()}
        pts.append([Vm] + list(x.values()))
        inet.append(float(pn.iNet(Vm, x)))
        ders.append([float(pn.derStates()[k](Vm, x)) for k in pn.statesNames()])
    out[f'{name}_pts'] = np.array(pts)
    out[f'{name}_iNet'] = np.array(inet)
    out[f'{name}_ders'] = np.array(ders)

    nbls = NeuronalBilayerSonophore(32e-9, pn)
    Qlo, Qhi = pn.Qbounds
    pairs = [(100e3, pn.Qm0), (300e3, 0.), (50e3, 0.6 * Qhi), (600e3, 0.9 * Qlo), (0., 0.5 * Qlo)]
    keys = ['V'] + list(pn.effRates().keys())
    out['pairs'] = np.array(pairs)
    out['keys'] = np.array(keys)
    out['f'] = 500e3
    for i, (A, Q) in enumerate(pairs):
        for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
            solvers.odeint = ode
            res = nbls.computeEffVars(AcousticDrive(500e3, A), 1., Q)
            effs = res[0] if isinstance(res, tuple) else res
            out[f'p{i}_{tag}_eff'] = np.array([effs[0][k] for k in keys])
        solvers.odeint = _odeint
        print(i, out[f'p{i}_tight_eff'][:3], flush=True)
    # detailed model, 4 us of stimulus + 1 us. Logger at WARNING as for golden_full_RS.npz: at INFO the
    # reference inserts 100 'log' events (nbls.py:345-346) that split the integration into 100 more
    # segments with their own np.linspace grids, which moves the linearly resampled rows by ~5e-5
    # of the deflection range
    logger.setLevel(logging.WARNING)
    drive, pp = AcousticDrive(500e3, 120e3), PulsedProtocol(4e-6, 1e-6)
    for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
        solvers.odeint = ode
        data, _ = nbls.simulate(drive, pp, method='full')
        out[f'full_{tag}'] = data.values
        out['full_columns'] = np.array(list(data.columns))
    # the same with the logger at INFO (what scripts/run_astim.py and Batch.run(mpi=True) set): rows of
    # the 100-log-event variant, tight tolerances
    logger.setLevel(logging.INFO)
    solvers.odeint = tight_odeint
    data, _ = nbls.simulate(drive, pp, method='full')
    out['full_loginfo_tight'] = data.values
    logger.setLevel(logging.WARNING)
    solvers.odeint = _odeint
    np.savez_compressed(os.path.join(HERE, f'golden_{name}.npz'), **out)


if __name__ == '__main__':
    for n in sys.argv[1:]:
        main(n)
