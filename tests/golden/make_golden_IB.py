# -*- coding: utf-8 -*-
''' Golden vectors for the cortical intrinsically bursting neuron, captured from the REFERENCE
    (PySONIC/neurons/cortical.py:307-400): definition samples (states, rate functions, resting
    state, iNet and true derivatives at random points -- as golden_neurons.npz holds for the six
    BASELINE neurons) and NeuronalBilayerSonophore.computeEffVars for a few (A, Q) cells at default
    and tight odeint tolerances.

    Output: tests/golden/golden_IB.npz (build container only)
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
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive  # noqa: E402
import PySONIC.core.solvers as solvers  # noqa: E402

_odeint = scipy.integrate.odeint


def tight_odeint(f, y0, t, **kw):
    return _odeint(f, y0, t, rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34]), mxstep=1000000, **kw)


def main():
    name = 'IB'
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
    pairs = [(100e3, -71.4e-5), (300e3, 0.), (50e3, 30e-5), (600e3, -100e-5), (0., -50e-5)]
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
    np.savez_compressed(os.path.join(HERE, 'golden_IB.npz'), **out)


if __name__ == '__main__':
    main()
