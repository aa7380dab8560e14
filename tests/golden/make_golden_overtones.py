# -*- coding: utf-8 -*-
''' Golden vectors for computeEffVars with charge overtones, captured from the REFERENCE
    (NeuronalBilayerSonophore.computeEffVars with Qm_overtones, PySONIC/core/nbls.py:153-222;
    BilayerSonophore.simCycles with a charge profile, PySONIC/core/bls.py:749-789):
    RS, a = 32 nm, one and two overtones, default odeint tolerances and rtol = 1e-12 ("tight").

    Output: tests/golden/golden_overtones.npz (build container only)
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
    atol = np.array([1e-12, 1e-21, 1e-34])
    return _odeint(f, y0, t, rtol=1e-12, atol=atol, mxstep=1000000, **kw)


CASES = [  # f, A, Qm0, [(A_Q, phi_Q), ...], fs
    (500e3, 100e3, -71.9e-5, [(20e-5, 0.5)], [1.0]),
    (500e3, 100e3, 0., [(50e-5, 2.0)], [1.0]),
    (500e3, 300e3, -50e-5, [(25e-5, np.pi)], [1.0, 0.5]),
    (500e3, 50e3, 20e-5, [(10e-5, 4.0), (5e-5, 1.0)], [1.0]),
    (4e6, 200e3, -71.9e-5, [(30e-5, 1.5)], [0.75]),
    (500e3, 0., -71.9e-5, [(20e-5, 0.3)], [1.0]),
]


def main():
    pn = getPointNeuron('RS')
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    out = {'ncases': len(CASES), 'keys': np.array(['V'] + list(pn.effRates().keys()))}
    for i, (f, A, Q0, ov, fs) in enumerate(CASES):
        drive = AcousticDrive(f, A)
        for tag, ode in (('default', _odeint), ('tight', tight_odeint)):
            solvers.odeint = ode
            res = nbls.computeEffVars(drive, np.array(fs), Q0, Qm_overtones=ov)
            effs = res[0] if isinstance(res, tuple) else res
            cols = list(effs[0].keys())
            out[f'c{i}_{tag}'] = np.array([[e[k] for k in cols] for e in effs])
            out[f'c{i}_cols'] = np.array(cols)
        solvers.odeint = _odeint
        out[f'c{i}_in'] = np.array([f, A, Q0])
        out[f'c{i}_ov'] = np.array(ov)
        out[f'c{i}_fs'] = np.array(fs)
        print(i, cols[:4], out[f'c{i}_tight'][0][:4], flush=True)
    np.savez_compressed(os.path.join(HERE, 'golden_overtones.npz'), **out)


if __name__ == '__main__':
    main()
