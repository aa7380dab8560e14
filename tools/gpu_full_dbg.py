import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
np.set_printoptions(linewidth=250, precision=4)
for name in ['STN']:
    pn = getPointNeuron(name)
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    for A in (80e3,):
        cfgs = [(AcousticDrive(500e3, A), PulsedProtocol(1e-6, 1e-6))]
        Aa, tstop, _, ev_t, ev_x, ev_off = nbls._packConfigs(cfgs)
        tr, ro, st, ns, ms = N.full_batch_run(name, pn.device_params(), nbls.device_params(), [500e3], Aa, [1.], tstop, ev_t, ev_x, ev_off,
                                              nbls.initialConditionsSonic(), N.full_default_opts(rtol=1e-6, max_steps=2000000))
        print(name, A, 'status', st, 'nsteps', ns)
        print(pn.statesNames())
        bad = np.where(np.isnan(tr[:, 2]))[0]
        i = bad[0] if bad.size else len(tr)
        print('first nan row', i, 'of', len(tr)); print(tr[max(0,i-2):i+1]); print('DBG y, k1, F(y), [t h en err..]'); print(tr[-4:])
        print('y0', nbls.initialConditionsSonic())
        print('params', pn.device_params())
