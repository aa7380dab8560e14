''' Helpers of the development scripts, built on the package's own host API (the scripts of tools/ do not
    use the oracle: that is test infrastructure). '''
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import PulsedProtocol, getPointNeuron  # noqa: E402


def pulsed_events(tstim, toffset, PRF=100., DC=1.):
    ''' sorted (t, x) stimulus events and the stopping time of a pulsed protocol '''
    pp = PulsedProtocol(float(tstim), float(toffset), float(PRF), float(DC))
    return sorted(pp.stimEvents(), key=lambda e: e[0]), pp.tstop


def neuron_Qm0(name):
    return getPointNeuron(name).Qm0


def steady_states(name):
    pn = getPointNeuron(name)
    return pn.getSteadyStates(pn.Vm0)


def get_nsamples(t0, tend, dt):
    ''' ODESolver.getNSamples (solvers.py:77-87) '''
    return max(int(np.round((tend - t0) / dt)), 2)
