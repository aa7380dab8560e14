''' Development (GPU box): generated lookup for (64 nm, 1 MHz) and a sonic run on it. '''
import sys, os, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pysonic_amd.core.nbls as nbls_mod
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
nbls_mod.LOOKUP_DIR = tempfile.mkdtemp()
nbls = NeuronalBilayerSonophore(64e-9, getPointNeuron('RS'))
lkp = nbls.getLookup2D(1e6, 1.)
for k in lkp.outputs:
    v = lkp[k]
    print(k, 'finite', np.isfinite(v).all(), 'min %.3g max %.3g' % (np.nanmin(v), np.nanmax(v)))
for A in (50e3, 100e3, 200e3, 300e3):
    for env in ({}, {'PYSONIC_AMD_QUAD': '0'}):
        os.environ.pop('PYSONIC_AMD_QUAD', None); os.environ.update(env)
        rows, met, st, ms = nbls.runSonicBatch(1e6, 1., [(AcousticDrive(1e6, A), PulsedProtocol(20e-3, 5e-3))])
        print(A, env, 'status', st, 'steps', met[0, 0], 'rej', met[0, 1], 'Qmin/max', met[0, 3], met[0, 4], 'nan rows', np.isnan(rows[0][:, 2]).sum())
