''' Development (GPU box): hybrid / full kernels, one short configuration: with and without the shadow
    copies in the idle lanes (PYSONIC_AMD_SHADOW), 1 or 4 identical configurations. Short runs only. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
cfg = (AcousticDrive(500e3, 100e3), PulsedProtocol(60e-6, 20e-6), 1.)
for shadow in ('1', '0'):
    os.environ['PYSONIC_AMD_SHADOW'] = shadow
    for n in (1, 4):
        for method in ('runHybridBatch', 'runFullBatch'):
            t0 = time.perf_counter()
            res = getattr(nbls, method)([cfg] * n)
            print(f'shadow={shadow} n={n} {method}: kernel {res[-1]:.0f} ms', flush=True)
